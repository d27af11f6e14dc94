"""Round 5: stand-alone timing + output fingerprints of the K0 MX-fp6 passes at cfg2 size (64 pairs, 224 x 224, C = 256, NCHW):
queries (mx6 rows + norms) and anchors (+ fp32 unit rows).  Run once per variant (ORYON_K0V4=1|0, development library) and compare
the fingerprints:   python tools/r5_k0.py <label> [B]"""
import json, os, sys
import _devlib  # noqa: F401
import torch
from oryon_amd import ops
from oryon_amd.synth import make_pair

label = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H, C = 224, 256
dev = "cuda"
pairs = [make_pair(i, H, H, C, device=dev) for i in range(B)]
st = lambda k: torch.stack([p[k] for p in pairs])
feat_a, feat_q, mask_a, mask_q = st("feat_a"), st("feat_q"), st("mask_a"), st("mask_q")
del pairs
roi_a, na = ops.roi_compact(mask_a)
roi_q, nq = ops.roi_compact(mask_q)
ops.roi_subsample_(roi_a, na, 5000, seed=1)
cap_a, cap_q = ops.round_up(5000, 256), ops.round_up(H * H, 256)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def fingerprint(rows, err, norm, hat, n):
    fp = {"err": err.cpu().tolist()}
    acc6 = acc32 = accn = 0
    for m in range(rows.shape[0]):
        k = int(n[m]); kf = (k + 255) // 256 * 256
        acc6 += int(rows[m, :kf].contiguous().view(torch.int64).sum().item())
        accn += int(norm[m, :kf].contiguous().view(torch.int32).to(torch.int64).sum().item())
        if hat is not None:
            acc32 += int(hat[m, :k].contiguous().view(torch.int32).to(torch.int64).sum().item())
    fp.update(rows=acc6 & (2**63 - 1), norm=accn, f32=acc32)
    return fp


q = ops.gather_mx6(feat_q, roi_q, nq, cap_q, 256)
a = ops.gather_mx6(feat_a, roi_a, na, cap_a, 256, want_f32=True)
torch.cuda.synchronize()
out = {"label": label, "B": B, "q": fingerprint(q[0], q[1], q[2], None, nq), "a": fingerprint(a[0], a[1], a[2], a[3], na)}
del q, a
tq = timeit(lambda: ops.gather_mx6(feat_q, roi_q, nq, cap_q, 256))
ta = timeit(lambda: ops.gather_mx6(feat_a, roi_a, na, cap_a, 256, want_f32=True))
rows_q, rows_a = float(nq.sum()), float(na.sum())
alg_q = rows_q * (4 * C + 256 + 4) / 1e9
alg_a = rows_a * (4 * C + 256 + 4 + 4 * 256) / 1e9
out.update(query_ms=tq[0], query_ms_min=tq[1], anchor_ms=ta[0], anchor_ms_min=ta[1], rows_q=rows_q, rows_a=rows_a,
           alg_gb_q=alg_q, alg_gb_a=alg_a, query_tbs=alg_q / tq[0], anchor_tbs=alg_a / ta[0],
           k0_step_ms=tq[0] + ta[0], k0_frac_of_8tbs=(alg_q + alg_a) / (tq[0] + ta[0]) / 8.0)
# where the anchor pass spends its time: without the fp32 rows; with the whole (unsampled, contiguous) mask region as ROI
ta_nof32 = timeit(lambda: ops.gather_mx6(feat_a, roi_a, na, cap_a, 256))
roi_f, nf = ops.roi_compact(mask_a)
cap_f = ops.round_up(int(nf.max()), 256)
ta_full = timeit(lambda: ops.gather_mx6(feat_a, roi_f, nf, cap_f, 256, want_f32=True))
out.update(anchor_nof32_ms=ta_nof32[0], anchor_fullroi_ms=ta_full[0], rows_fullroi=float(nf.sum()))
print(json.dumps({k: v for k, v in out.items() if k not in ("q", "a")}))
os.makedirs(os.path.join(_devlib.ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(_devlib.ROOT, "gpurun_out", f"r5_k0_{label}.json"), "w"), indent=1)
