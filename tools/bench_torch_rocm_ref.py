"""BASELINE configs[1] names its comparison: "HIP matcher + PointDSC vs PyTorch-ROCm ref".  This is that reference leg: the oracle's
torch restatement of the reference path (normalise + chunked matmul + amin / argmin matcher, torch lift, torch PointDSC with the
reference's host SVD) run per pair on the SAME MI355X through PyTorch-ROCm - the reference's own per-sample loop shape
(pipeline.py:313-355), on the cfg2 workload.  usage (GPU box): python tools/bench_torch_rocm_ref.py [pairs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs  # noqa: E402
from oracle import oryon_oracle as orc  # noqa: E402

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, C = 224, 256
inp = make_inputs(N, H, C, 0, dev)
P = {k: v.to(dev) for k, v in orc.analytic_pointdsc_params(12, 128).items()}
cfg = dict(num_layers=12, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1, inlier_threshold=0.1)
torch.manual_seed(1)


def one_pair(i):
    fa, fq, ma, mq = inp["feat_a"][i], inp["feat_q"][i], inp["mask_a"][i], inp["mask_q"][i]
    roi1 = orc.roi_from_mask(ma)
    if roi1.shape[0] > 5000:
        roi1 = roi1[orc.sample_select(roi1.shape[0], 5000).to(dev)]
    pre = orc.match_presample(fa, fq, ma, mq, 0.25, roi1_override=roi1, form="gemm")
    keep = torch.nonzero(pre["valid"]).squeeze(1)
    pairs = torch.cat((pre["roi1"][keep], pre["roi2"][pre["argmin"]][keep]), dim=1)
    pairs = pairs[orc.sample_select(pairs.shape[0], 500).to(dev)]
    cam = inp["cam"][i].reshape(9).to(torch.float32)
    pa, pq, ok = orc.lift_pair(inp["depth_a"][i], inp["depth_q"][i], cam, cam, pairs, (H, H), (H, H), (H, H))
    return orc.pointdsc_forward(pa[ok], pq[ok], P, cfg)


t_match = 0.0
for it in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    errs = []
    for i in range(N):
        T = one_pair(i)
        errs.append(float((T.cpu().double() - inp["pose_gt"][i]).abs().max()))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"pass {it}: {N} pairs in {dt:.2f} s -> {N / dt:.2f} pairs/s ({dt / N * 1e3:.1f} ms per pair), max |T - T_gt| {max(errs):.2e}")
# the matcher part alone
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    roi1 = orc.roi_from_mask(inp["mask_a"][i])
    roi1 = roi1[orc.sample_select(roi1.shape[0], 5000).to(dev)]
    orc.match_presample(inp["feat_a"][i], inp["feat_q"][i], inp["mask_a"][i], inp["mask_q"][i], 0.25, roi1_override=roi1, form="gemm")
torch.cuda.synchronize()
print(f"matcher alone (gather + normalise + chunked fp32 matmul + amin/argmin): {(time.perf_counter() - t0) / N * 1e3:.1f} ms per pair")
