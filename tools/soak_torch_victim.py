"""Are torch's own fp32 elementwise kernels (compiled with packed fp32 ops) bit-stable beside this library's fp16x3 MFMA kernels?  Stream A repeats a
few torch fp32 expressions on fixed inputs and compares every result bit for bit with the serial one; stream B runs the PointDSC encoder.
INTEGRATION.md "Streams: a caution ..." quotes the outcome.   usage (GPU box): python tools/soak_torch_victim.py [iterations]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_solver
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
noise = build_solver(dev)
g = torch.Generator(device=dev).manual_seed(0)
B = 64
src = torch.rand(B, 512, 3, generator=g, device=dev)
tgt = src + 0.01 * torch.randn(B, 512, 3, generator=g, device=dev)
n = torch.full((B,), 500, dtype=torch.int32, device=dev)
x = torch.randn(1 << 22, generator=g, device=dev); y = torch.randn(1 << 22, generator=g, device=dev); z = torch.randn(1 << 22, generator=g, device=dev)
m = torch.randn(4096, 1024, generator=g, device=dev)
exprs = {
    "addcmul(z, x, y)": lambda: torch.addcmul(z, x, y),
    "x * y + z": lambda: x * y + z,
    "layer_norm(m)": lambda: torch.nn.functional.layer_norm(m, (1024,)),
    "softmax(m)": lambda: torch.softmax(m, dim=-1),
    "gelu(x)": lambda: torch.nn.functional.gelu(x),
}
want = {k: f().clone() for k, f in exprs.items()}
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
bad = {k: 0 for k in exprs}
runs = 0
for it in range(iters):
    with torch.cuda.stream(sb):
        for _ in range(2):
            noise.encode(src, tgt, n)
    with torch.cuda.stream(sa):
        outs = [(k, f()) for _ in range(6) for k, f in exprs.items()]
    torch.cuda.synchronize()
    runs += 6
    for k, o in outs:
        if not torch.equal(o, want[k]):
            bad[k] += 1
print(f"torch fp32 kernels beside the PointDSC encoder, {runs} runs each: differing from the serial result:", bad)
