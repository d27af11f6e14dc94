"""The driver's bench.py contract, checked on the GPU box: one JSON line, BASELINE.json's metric, the roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], check=True, capture_output=True, text=True, timeout=900,
                         cwd=ROOT).stdout
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_default_line_carries_the_contract_fields():
    rec = _run("--steps", "2", "--warmup", "1", "--batch", "8")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert rec["metric"] == base["metric"] and rec["unit"] == "pairs/s" and rec["n_gpus"] == 1
    assert rec["steps"] == 2 and rec["warmup"] == 1 and rec["higher_is_better"] is True and rec["scaling"] == "weak"
    assert rec["vs_baseline"] is None and rec["data"] == "synthetic" and rec["dtype"] == "f32"
    assert rec["value"] > 0 and abs(rec["value"] - 8 * 1e3 / rec["ms_per_step"]) < 1e-6 * rec["value"]
    assert "workload" in rec["config"] and "model" not in rec["config"] and rec["config"]["pairs_ok"] == 8
    roof = rec["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s") and roof["peak"] > 0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0 < roof["frac"] < 1 and 0 < roof["hbm_frac"] < 1
    assert "traffic" in roof                                   # null away from the profiled cfg2 workload
    cpu = rec["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["unit"] == "pairs/s" and cpu["sample"]


def test_other_matcher_modes_report_their_own_kernel():
    for mode, kern in (("screened16", "match_f16_screen_kernel"), ("exact", "match_f32_regb_kernel")):
        rec = _run("--steps", "1", "--warmup", "1", "--batch", "4", "--no-cpu-baseline", "--match-mode", mode)
        assert kern in rec["roofline"]["kernel"] and rec["config"]["pairs_ok"] == 4 and "cpu_baseline" not in rec
