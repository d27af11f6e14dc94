"""Do two PointDSC registrations (64 pairs each) overlap when launched on two HIP streams?   (GPU box)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_solver
dev = torch.device("cuda", 0)
solver = build_solver(dev)
B, n_cap = 64, 512
g = torch.Generator(device=dev).manual_seed(0)
def batch():
    src = torch.rand(B, n_cap, 3, generator=g, device=dev) * 0.3
    R = torch.linalg.qr(torch.randn(3, 3, generator=g, device=dev))[0]
    tgt = src @ R.T + 0.05
    n = torch.full((B,), 500, dtype=torch.int32, device=dev)
    return src, tgt, n
b0, b1 = batch(), batch()
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
def run(conc, reps=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        if conc:
            with torch.cuda.stream(s0): solver.register(*b0, ws_slot=0)
            with torch.cuda.stream(s1): solver.register(*b1, ws_slot=1)
        else:
            with torch.cuda.stream(s0):
                solver.register(*b0, ws_slot=0); solver.register(*b1, ws_slot=1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
run(False, 3); run(True, 3)
print(f"two registrations of {B} pairs: sequential {run(False):.2f} ms, concurrent on two streams {run(True):.2f} ms")
