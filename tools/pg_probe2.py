"""Development aid: does the mere existence of an RCCL communicator slow the pipelined step?  One process, B pairs per step:
ms per step (a) before any process group, (b) after init_process_group("nccl", world 1, device_id=...), (c) after a first collective,
(d) after destroy_process_group.   python tools/pg_probe2.py [batch] [lazy]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from bench import make_inputs, build_solver
from oryon_amd.engine import MatchPoseEngine, MatchPoseConfig
from oryon_amd.dist import warm_engine_streams

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lazy = len(sys.argv) > 2 and sys.argv[2] == "lazy"
dev = torch.device("cuda", 0)
warm_engine_streams(0)

key = torch.arange(B, dtype=torch.int64, device=dev)
d = None
first = len(sys.argv) > 2 and sys.argv[2] == "first"
if first:          # the order of bench.py: stream pool, process group, THEN inputs / engine
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
d = make_inputs(B, 224, 256, first=0, dev=dev)
d["cam"] = d["cam"].reshape(B, 9).to(torch.float32).contiguous()
eng = MatchPoseEngine(build_solver(dev), MatchPoseConfig(dist_th=0.25, n_corrs=500, src_sampling=5000, seed=1, match_mode="screened"),
                      overlap_registration=True, overlap_gather=True, native=True, result_views=True)
eng.native_geometry["screen"] = 1
if os.environ.get("PG_TIMING"):
    eng.native_timing = True

def run(n):
    prev = None
    for _ in range(n):
        out = eng.run(d["feat_a"], d["feat_q"], d["mask_a"], d["mask_q"], d["depth_a"], d["depth_q"], d["cam"], d["cam"], key, inputs_resident=True)
        if prev is not None:
            eng.finish(prev)
        prev = out
    eng.finish(prev)

def measure(tag):
    run(5)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(20); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"{tag:55s} {sorted(ts)[2]:.3f} ms per step (B={B})", flush=True)

if first:
    measure("process group created before the engine")
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    measure("... after dist.barrier()")
    el = torch.tensor([1.0], dtype=torch.float64, device=dev); dist.all_reduce(el, op=dist.ReduceOp.MAX); el.item()
    measure("... after all_reduce(MAX, float64) + item()")
    from oryon_amd.dist import gather_poses
    out = eng.run(d["feat_a"], d["feat_q"], d["mask_a"], d["mask_q"], d["depth_a"], d["depth_q"], d["cam"], d["cam"], key, inputs_resident=True)
    eng.finish(out); gather_poses(out["pose"], out["status"], B); torch.cuda.synchronize()
    measure("... after one gather_poses")
    eng.native_timing = True
    measure("... with native_timing (HIP events per section)")
    dist.destroy_process_group()
    measure("after destroy_process_group")
    sys.exit(0)
measure("no process group")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
if lazy:
    dist.init_process_group("nccl", rank=0, world_size=1)
    measure("after init_process_group (lazy: no communicator yet)")
else:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    measure("after init_process_group (device_id: communicator)")
x = torch.zeros(4, device=dev)
dist.all_reduce(x); torch.cuda.synchronize()
measure("after the first collective")
dist.destroy_process_group()
measure("after destroy_process_group")
