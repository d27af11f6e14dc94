"""Per-sample façade (`get_pointdsc_pose`, the drop-in for utils/pointdsc/init.py:10-29) at the reference's default sizes (500 correspondences,
12 x 128 encoder): wall time per call, GPU time of the registration alone, and where the host time goes.   usage (GPU box): python tools/time_facade.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oryon_amd.pointdsc import get_pointdsc_pose
dev = torch.device("cuda", 0)
solver = bench.build_solver(dev)
g = torch.Generator().manual_seed(3)
n = 500
src = torch.rand((n, 3), generator=g) - 0.5
R = torch.linalg.qr(torch.randn((3, 3), generator=g))[0]
tgt = src @ R.T + 0.1 + 0.002 * torch.randn((n, 3), generator=g)
tgt[::3] = torch.rand((len(tgt[::3]), 3), generator=g)          # a third outliers
for _ in range(20):
    T = get_pointdsc_pose(solver, src, tgt, "cuda:0")
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    T = get_pointdsc_pose(solver, src, tgt, "cuda:0")
t1 = time.perf_counter()
print(f"get_pointdsc_pose: {(t1 - t0) / N * 1e3:.3f} ms per pair (CPU tensors in, CPU pose out)")
# registration alone, inputs resident
sp = torch.zeros((1, 512, 3), device=dev); tp = torch.zeros((1, 512, 3), device=dev)
sp[0, :n] = src.to(dev); tp[0, :n] = tgt.to(dev)
nn_ = torch.full((1,), n, dtype=torch.int32, device=dev)
for _ in range(10):
    solver.register(sp, tp, nn_)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(N):
    solver.register(sp, tp, nn_)
e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"register(B=1) back to back: GPU {e0.elapsed_time(e1) / N:.3f} ms, wall {(t1 - t0) / N * 1e3:.3f} ms per call")
t0 = time.perf_counter()
for _ in range(N):
    solver.register(sp, tp, nn_)
    torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"register(B=1) + synchronize: {(t1 - t0) / N * 1e3:.3f} ms per call")
