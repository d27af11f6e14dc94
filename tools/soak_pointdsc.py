"""Run-to-run determinism of the registration: the same 64 pairs registered N times, every pose compared bit for bit with the first call's
(and the encoder features with the first call's).  usage (GPU box): python tools/soak_pointdsc.py [calls]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_solver
dev = torch.device("cuda", 0)
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
solver = build_solver(dev)
g = torch.Generator(device=dev).manual_seed(0)
B = 64
src = torch.rand(B, 512, 3, generator=g, device=dev)
tgt = src + 0.01 * torch.randn(B, 512, 3, generator=g, device=dev)
n = torch.full((B,), 500, dtype=torch.int32, device=dev)
status = torch.zeros(B, dtype=torch.int32, device=dev)
T0 = solver.register(src, tgt, n, status)[0].clone()
f0, c0 = [x.clone() for x in solver.encode(src, tgt, n)]
bad_T = bad_f = 0
for i in range(calls):
    T = solver.register(src, tgt, n, status)[0]
    if not torch.equal(T, T0):
        bad_T += 1
        d = (T - T0).abs()
        print(f"  call {i}: pose max diff {float(d.max()):.3e}, pairs {sorted(set((d.reshape(B, -1) > 0).any(1).nonzero().flatten().tolist()))[:8]}")
    if i % 10 == 0:
        f, c = solver.encode(src, tgt, n)
        if not (torch.equal(f[:, :500], f0[:, :500]) and torch.equal(c[:, :500], c0[:, :500])):
            bad_f += 1
            d = (f[:, :500] - f0[:, :500]).abs()
            print(f"  call {i}: encoder features differ, max {float(d.max()):.3e}, pairs {sorted(set((d.reshape(B, -1) > 0).any(1).nonzero().flatten().tolist()))[:8]}")
print(f"{calls} calls: poses differing from the first call: {bad_T}; encoder checks differing: {bad_f}")
# two registrations in flight on two streams (the engine's alternating registration streams), separate workspace slots
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
bad = 0
for i in range(calls // 2):
    with torch.cuda.stream(sa):
        Ta = solver.register(src, tgt, n, status, ws_slot=0)[0]
    with torch.cuda.stream(sb):
        Tb = solver.register(src, tgt, n, status, ws_slot=1)[0]
    torch.cuda.synchronize()
    for nm, T in (("a", Ta), ("b", Tb)):
        if not torch.equal(T, T0):
            bad += 1
            if bad <= 4:
                d = (T - T0).abs()
                print(f"  concurrent call {i}{nm}: pose max diff {float(d.max()):.3e}, pairs {sorted(set((d.reshape(B, -1) > 0).any(1).nonzero().flatten().tolist()))[:8]}")
print(f"{calls // 2} concurrent pairs of calls: poses differing from the serial result: {bad}")
