#!/bin/bash
# round 5: K0v4 against K0v3 - fingerprints (bit-identical outputs) and stand-alone timing, then the K0 / default-route tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
ORYON_K0V4=1 python tools/r5_k0.py v4 ${1:-64} | tail -1
ORYON_K0V4=0 python tools/r5_k0.py v3 ${1:-64} | tail -1
python - <<'PY'
import json
a = json.load(open("gpurun_out/r5_k0_v4.json")); b = json.load(open("gpurun_out/r5_k0_v3.json"))
for k in ("q", "a"):
    print(k, "identical:", {x: (a[k][x] == b[k][x]) for x in a[k]})
print("v4 query %.3f ms anchor %.3f ms | v3 query %.3f ms anchor %.3f ms" % (a["query_ms"], a["anchor_ms"], b["query_ms"], b["anchor_ms"]))
PY
if [ -z "$SKIP_TESTS" ]; then
python -m pytest tests/test_gpu_matcher.py -x -q -m gpu -k "k0 or mx6" 2>&1 | tail -3
fi
echo "== pipelined step v4 / v3 / v4 / v3"
for v in 1 0 1 0; do ORYON_K0V4=$v python tools/engine_timeline.py 40 2>&1 | grep "ms/step"; done
} 2>&1 | tee gpurun_out/r5_k0.log
