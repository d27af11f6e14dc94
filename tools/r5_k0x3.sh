#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
ORYON_K0V4=1 python tools/r5_k0x3.py v4 | tail -1
ORYON_K0V4=0 python tools/r5_k0x3.py v3 | tail -1
python - <<'PY'
import json
a = json.load(open("gpurun_out/r5_k0x3_v4.json")); b = json.load(open("gpurun_out/r5_k0x3_v3.json"))
print("identical:", {k: a["fp"][k] == b["fp"][k] for k in a["fp"]}, "| v4 %.3f ms  v3 %.3f ms" % (a["ms"], b["ms"]))
import numpy as np
print("max rel diff losq", max(abs(x - y) / max(y, 1e-30) for x, y in zip(a["fp"]["losq"], b["fp"]["losq"])))
PY
python -m pytest tests/test_gpu_native_engine.py tests/test_gpu_default_route_vs_oracle.py -x -q -m gpu 2>&1 | tail -3
echo "== hard step pipelined v4 / v3 / v4 / v3"
for v in 1 0 1 0; do ORYON_K0V4=$v ENG_HARD=1 python tools/engine_timeline.py 40 2>&1 | grep "ms/step"; done
} 2>&1 | tee gpurun_out/r5_k0x3.log
