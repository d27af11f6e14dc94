"""What the NCHW gather of K0 must fetch from HBM at cfg2, whatever the kernel: the ROI rows of a channel plane are 4-byte values at
the ROI's pixel positions, and DRAM delivers whole sectors.  For the synthetic cfg2 pairs (oryon_amd.synth.make_pair) this counts,
per plane, the distinct 32- / 64- / 128-byte pieces the ROI touches (= the traffic with PERFECT sharing between tiles through L2) and
the sum over 64-row tiles (= no sharing at all), relative to the algorithmic 4 bytes per ROI pixel.  CPU only.
    python tools/r5_roi_floor.py [pairs]
Round 5 result (8 pairs): query ROI (D_q > 0: the projected points leave single-pixel holes) 1.26x at 64 B (1.13 - 1.38 per pair),
anchor ROI (5000 random of the 12544 mask pixels) 2.86x; with cfg2's 2.58 GB + 0.33 GB of algorithmic reads that is 4.2 GB fetched
(rocprofv3 FETCH_SIZE x 2: 4.36 GB measured in round 4) - K0 moves >= 5.3 GB per step against 3.99 GB algorithmic, i.e. its
algorithmic-bytes roofline fraction cannot exceed 0.75 x (achievable HBM rate / 8 TB/s) = 0.59 at the 6.3 TB/s copy rate."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oryon_amd.synth import make_pair

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = 224
rng = np.random.default_rng(0)
acc = {("query", g): [] for g in (32, 64, 128)}
acc.update({("anchor", g): [] for g in (32, 64, 128)})
tile = {k: [] for k in acc}
rows = {"query": 0, "anchor": 0}
for i in range(P):
    p = make_pair(i, H, H, 8)
    rq = np.nonzero(p["mask_q"].numpy().reshape(-1) == 1)[0]
    ra = np.nonzero(p["mask_a"].numpy().reshape(-1) == 1)[0]
    ra = np.sort(rng.choice(ra, min(5000, len(ra)), replace=False))
    for name, roi in (("query", rq), ("anchor", ra)):
        rows[name] += len(roi)
        for g in (32, 64, 128):
            px = g // 4
            acc[(name, g)].append(len(np.unique(roi // px)) * g / (len(roi) * 4.0))
            tile[(name, g)].append(sum(len(np.unique(roi[t:t + 64] // px)) for t in range(0, len(roi), 64)) * g / (len(roi) * 4.0))
print(f"{P} cfg2 pairs: {rows['query'] / P:.0f} query rows, {rows['anchor'] / P:.0f} anchor rows per pair")
for name in ("query", "anchor"):
    for g in (32, 64, 128):
        a, t = np.array(acc[(name, g)]), np.array(tile[(name, g)])
        print(f"  {name:6s} {g:3d}-byte pieces: fetched / algorithmic = {a.mean():.3f} (min {a.min():.3f}, max {a.max():.3f}) with perfect L2 sharing, "
              f"{t.mean():.3f} without any")
q64, a64 = np.mean(acc[("query", 64)]), np.mean(acc[("anchor", 64)])
rq_, ra_ = rows["query"] / P * 64, 5000 * 64
alg_r = (rq_ + ra_) * 1024 / 1e9
fet = (rq_ * q64 + ra_ * a64) * 1024 / 1e9
wr = (rq_ * 260 + ra_ * (260 + 1024)) / 1e9
print(f"cfg2 step (64 pairs): algorithmic reads {alg_r:.2f} GB, fetched at 64-byte granularity {fet:.2f} GB, writes {wr:.2f} GB -> "
      f"{fet + wr:.2f} GB moved for {alg_r + wr:.2f} GB algorithmic: floor {1e3 * (fet + wr) / 6.3e3:.2f} ms at 6.3 TB/s, "
      f"largest possible algorithmic-bytes fraction of 8 TB/s: {(alg_r + wr) / (fet + wr) * 6.3 / 8:.2f}")
