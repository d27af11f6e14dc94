"""Development aid (round 6, DESIGN.md "stream placement"): bench.main() at B = 16 with a one-rank process group created in different ways
before it.  python tools/pg_probe4.py plain|warm+eager|warm+lazy|warm+gloo|eager-nowarm [bench flags, e.g. --stream-roles 2301]"""
import os, sys, json, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
from oryon_amd.dist import warm_engine_streams
mode = sys.argv[1]
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29513")
if mode == "warm+eager":
    warm_engine_streams(0); dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
elif mode == "warm+lazy":
    warm_engine_streams(0); dist.init_process_group("nccl", rank=0, world_size=1)
elif mode == "warm+gloo":
    warm_engine_streams(0); dist.init_process_group("gloo", rank=0, world_size=1)
elif mode == "eager-nowarm":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
sys.argv = ["bench.py", "--steps", "20", "--warmup", "3", "--reps", "3", "--batch", "16", "--no-cpu-baseline", "--no-stage-sets", "--stream-roles", "0"] + sys.argv[2:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        bench.main()
    except SystemExit:
        pass
    except Exception as e:
        print("EXC", repr(e), file=sys.stderr)
L = [l for l in buf.getvalue().splitlines() if l.startswith("{")]
r = json.loads(L[-1])
print(mode, sys.argv[13:], "pairs/s", round(r["value"]), "ms_per_step", round(r["ms_per_step"], 4), "grouped", r["multi_gpu"] is not None)
