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


def _g9_objects(g):
    info = __import__("json").loads(str(g["info_json"]))
    objs = {}
    for name in g["model_names"].tolist():
        syms = ev.format_sym_set(ev.get_symmetry_transformations(info[name], max_sym_disc_step=0.05))
        objs[name] = {"pts": g[f"pts_{name}"], "diameter": info[name]["diameter"], "syms": syms}
    return objs, info


def test_symmetry_sets_match_bop_toolkit():
    g = np.load(os.path.join(GOLD, "g9_bop_metrics.npz"))
    objs, _ = _g9_objects(g)
    assert [objs[k]["syms"].shape[0] for k in ("box", "brick", "can")] == [1, 2, 126]
    for name, o in objs.items():
        np.testing.assert_allclose(o["syms"], g[f"syms_{name}"], rtol=0, atol=1e-15)
    for i, name in enumerate(g["model_names"].tolist()):
        assert abs(ev.extent_diameter(objs[name]["pts"]) / 1000.0 - g["add_diam"][i]) < 1e-15


def test_mssd_mspd_match_reference():
    """my_mssd / my_mspd of the reference on float16-rounded poses (tests/golden/g9_bop_metrics.npz), numpy restatement."""
    g = np.load(os.path.join(GOLD, "g9_bop_metrics.npz"))
    objs, _ = _g9_objects(g)
    cls = g["cls"].tolist()
    for i in range(len(cls)):
        if i in g["failures"].tolist():
            continue
        pq = ev.Evaluator.effective_pose(g["rel"][i] @ g["anchor"][i], g["rel"][i])
        o = objs[cls[i]]
        assert abs(ev.mssd_error(pq, g["gt"][i], o["pts"], o["syms"]) - g["mssd_raw"][i]) <= 1e-9 * max(1.0, g["mssd_raw"][i])
        assert abs(ev.mspd_error(pq, g["gt"][i], g["K"], o["pts"], o["syms"]) - g["mspd_raw"][i]) <= 1e-9 * max(1.0, g["mspd_raw"][i])


def _check_against_reference_evaluator(E, g, rt_atol=1e-9):
    """rt_atol: absolute slack on the R (degrees) / T (centimetres) error lists - the device kernel takes fp32 poses, and the angle of a
    nearly exact pose is arccos of a value within 1e-8 of one."""
    for k in g.files:
        if k.startswith("metric_"):
            atol = rt_atol if k in ("metric_R error", "metric_T error") else 1e-9
            np.testing.assert_allclose(np.asarray(E.metrics[k[7:]], dtype=np.float64), g[k], rtol=1e-6, atol=atol, err_msg=k)
        elif k.startswith("count_"):
            assert E.counts[k[6:]] == g[k].tolist(), k
    means = E.get_means()
    for name, val in zip(g["mean_names"].tolist(), g["mean_values"].tolist()):
        assert abs(means[name] - val) < max(1e-6, rt_atol if name in ("R error", "T error") else 0.0), name
    assert E.get_latex_str() == str(g["latex"])


def test_evaluator_matches_reference_evaluator():
    """The accumulator end to end on the host path: failures, zero / failed poses, IoU bookkeeping, recalls, LaTeX row - equal to what
    the reference's Evaluator(compute_vsd=False) registered for the same batch."""
    g = np.load(os.path.join(GOLD, "g9_bop_metrics.npz"))
    objs, _ = _g9_objects(g)
    n = len(g["cls"])
    status = [2 if i in g["failures"].tolist() else 0 for i in range(n)]
    E = ev.Evaluator("g9")
    ev.evaluate_batch(E, pred_pose_rel=g["rel"], anchor_pose=g["anchor"], gt_pose=g["gt"], K=np.tile(g["K"], (n, 1, 1)), status=status,
                      cls_ids=g["cls"].tolist(), instance_ids=[f"inst{i}" for i in range(n)], objects=objs,
                      iou_a=[np.float32(0.5 + 0.04 * i) for i in range(n)], iou_q=[np.float32(0.9 - 0.05 * i) for i in range(n)])
    _check_against_reference_evaluator(E, g)
    assert sum(E.counts["Missing segm"]) == 1 and sum(E.counts["Failed pose"]) == 1 and sum(E.counts["Zero pose"]) == 1
    assert len(E.test_summary()) == 3
