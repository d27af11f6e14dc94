"""ADD / ADD-S / R-T error / IoU / prediction-CSV helpers against outputs of the reference's utils/metrics.py
(tests/golden/g7_metrics.npz)."""
import os

import numpy as np

from oryon_amd import evaluation as ev

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_metrics_match_reference():
    g = np.load(os.path.join(GOLD, "g7_metrics.npz"))
    n = g["pred"].shape[0]
    add = np.array([ev.compute_add(g["pcd"], g["pred"][i], g["gt"][i]) for i in range(n)])
    adds = np.array([ev.compute_adds(g["pcd"], g["pred"][i], g["gt"][i]) for i in range(n)])
    assert np.array_equal(add, g["add"])                       # same float16 arithmetic -> identical
    np.testing.assert_allclose(adds, g["adds"], rtol=1e-6, atol=1e-9)
    theta, shift = ev.compute_RT_distances(g["pred"], g["gt"])
    np.testing.assert_allclose(theta, g["theta"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(shift, g["shift"], rtol=1e-12)
    t1, s1 = ev.compute_RT_distances(g["pred"][0], g["gt"][0])
    assert t1.shape == (1,) and abs(t1[0] - g["theta"][0]) < 1e-9
    np.testing.assert_allclose(ev.mask_iou(g["mask1"], g["mask2"]), g["iou"], atol=1e-7)
    assert ev.compute_RT_distances(None, g["gt"]) == -1


def test_csv_round_trip(tmp_path):
    P = np.eye(4)
    P[:3, :] = np.arange(12, dtype=np.float64).reshape(3, 4) * 0.125
    line = ev.format_pred_line("1 2 3", "1 9 3", np.float32(0.5), np.float32(0.75), P)
    assert line.count(",") == 4 and line.endswith("\n")
    f = tmp_path / "pred.csv"
    f.write_text(line + line)
    rows = ev.read_pred_csv(str(f))
    assert len(rows) == 2 and rows[0]["id_q"] == "1 9 3" and rows[1]["iou_q"] == 0.75
    assert np.array_equal(rows[0]["pose"], P)
    assert ev.add_accuracy(np.array([0.01, 0.05]), np.array([0.2, 0.2])) == 0.5
