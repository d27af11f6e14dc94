"""How many query rows lie within a given cosine margin of an anchor's best match on bench.py's `hard_descriptors` fields?  (Sizing of a
second-level screen: the fp16 margin is 2.2e-3, an fp16x3 / fp32-grade one ~6e-5, the mx6 / int8 ones ~0.13.)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
B, H, C = 4, 224, 256
inp = bench.make_inputs(B, H, C, 0, dev)
gen = torch.Generator(device=dev).manual_seed(77)
yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device=dev), torch.linspace(0, 1, H, device=dev), indexing="ij")
coef = torch.stack([torch.ones_like(xx), xx, yy, xx * yy, torch.sin(3 * xx), torch.cos(3 * yy), torch.sin(7 * yy), torch.cos(5 * xx)])
basis = torch.randn((B, C, coef.shape[0]), generator=gen, device=dev)
fq = torch.einsum("bck,khw->bchw", basis, coef)
fq.add_(0.02 * torch.randn(fq.shape, generator=gen, device=dev))
fa = fq + 0.01 * torch.randn(fq.shape, generator=gen, device=dev)
for b in range(2):
    q = fq[b].reshape(C, -1)[:, inp["mask_q"][b].reshape(-1) == 1].double()
    a = fa[b].reshape(C, -1)[:, inp["mask_a"][b].reshape(-1) == 1][:, ::25].double()
    q = q / q.norm(dim=0, keepdim=True); a = a / a.norm(dim=0, keepdim=True)
    S = a.T @ q                                   # [n_a, n_q]
    m = S.max(dim=1, keepdim=True).values
    top2 = S.topk(2, dim=1).values
    print(f"pair {b}: {a.shape[1]} anchors x {q.shape[1]} queries; median gap best - second = {float((top2[:,0]-top2[:,1]).median()):.2e}")
    for margin in (6e-5, 2.4e-4, 2.2e-3, 0.036, 0.13):
        cnt = (S >= m - margin).sum(dim=1).double()
        print(f"   margin {margin:8.1e}: candidates per anchor median {float(cnt.median()):8.0f}  p90 {float(cnt.quantile(0.9)):8.0f}  max {float(cnt.max()):8.0f}")
