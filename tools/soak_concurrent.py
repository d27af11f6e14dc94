"""Two registrations on two streams: does a kernel's result depend on what runs beside it?

Stream A repeats one stage of the registration (refine / hypotheses / the whole register call) on fixed inputs and compares every result bit
for bit with the serial one; stream B runs `noise` at the same time: another solver instance's encoder / whole registration, torch GEMMs
(fp32 / fp16 / bf16, i.e. hipBLASLt), or device copies.  With the library built WITH packed fp32 VALU ops (drop $(NOPK) from
oryon_amd/csrc/Makefile) `refine` under `encode` noise differs in ~25 % of the launches (lanes 48-63 of one wave get a wrong v_pk_*_f32
result; fp16x3 / int8 MFMA kernels beside it are what it takes - fp32-MFMA kernels, hipBLASLt GEMMs and copies do not do it); as shipped
(no packed fp32 ops) every count is 0.  DESIGN.md "Concurrency and the packed-fp32 finding".

usage (GPU box): python tools/soak_concurrent.py [refine|hypotheses|register] [encode|register|gemm|gemm16|gemmbf16|copy] [iterations]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_solver

which = sys.argv[1] if len(sys.argv) > 1 else "refine"
noise_kind = sys.argv[2] if len(sys.argv) > 2 else "encode"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda", 0)
m, noise = build_solver(dev), build_solver(dev)
g = torch.Generator(device=dev).manual_seed(0)
B = 64
src = torch.rand(B, 512, 3, generator=g, device=dev)
tgt = src + 0.01 * torch.randn(B, 512, 3, generator=g, device=dev)
n = torch.full((B,), 500, dtype=torch.int32, device=dev)
status = torch.zeros(B, dtype=torch.int32, device=dev)
feat, conf = [x.clone() for x in m.encode(src, tgt, n)]
seeds, ns = m.pick_seeds_batched(src, conf, n)
sT, fit, best = m.hypotheses(src, tgt, feat, n, seeds, ns)
sT = sT.clone()
T0 = sT[torch.arange(B, device=dev), best.long()].contiguous()
T = m.refine(src, tgt, n, T0).clone()
Treg = m.register(src, tgt, n, status)[0].clone()
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
gx, gw = torch.randn(8192, 4096, device=dev), torch.randn(4096, 4096, device=dev)
gd = torch.randn(64 << 20, device=dev)
gc = torch.empty_like(gd)
lo = {"gemm16": torch.float16, "gemmbf16": torch.bfloat16}
if noise_kind in lo:
    gx, gw = gx.to(lo[noise_kind]), gw.to(lo[noise_kind])
run = {"refine": lambda: m.refine(src, tgt, n, T0), "hypotheses": lambda: m.hypotheses(src, tgt, feat, n, seeds, ns)[0],
       "register": lambda: m.register(src, tgt, n, status)[0]}[which]
want = {"refine": T, "hypotheses": sT, "register": Treg}[which]
reps = 20 if which != "register" else 3
bad = 0
for it in range(iters):
    with torch.cuda.stream(sb):
        if noise_kind.startswith("gemm"):
            for _ in range(6 if noise_kind == "gemm" else 12):
                gy = gx @ gw
        elif noise_kind == "copy":
            for _ in range(40):
                gc.copy_(gd)
        elif noise_kind == "encode":
            for _ in range(2):
                noise.encode(src, tgt, n)
        else:
            for _ in range(2):
                noise.register(src, tgt, n, status)
    with torch.cuda.stream(sa):
        outs = [run() for _ in range(reps)]
    torch.cuda.synchronize()
    for o in outs:
        if not torch.equal(o, want):
            bad += 1
            if bad <= 5:
                d = (o - want).abs().reshape(B, -1).amax(1)
                print(f"  iteration {it}: pairs {d.nonzero().flatten().tolist()[:8]} differ, max {float(d.max()):.3e}")
print(f"{which} beside {noise_kind}: {bad} of {iters * reps} launches differ from the serial result")
