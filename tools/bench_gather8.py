"""Time K0v3 (gather8.hip) against the round-1 K0 at a BASELINE geometry, and the K1s8 matcher on both operand sets.
usage (GPU box): python tools/bench_gather8.py [H C B]      env ORYON_GATHER8_LPR=1|2, ORYON_RESCORE_LANES=1|2|4"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oryon_amd import ops
from oryon_amd.synth import make_pair

H, C, B = (int(x) for x in (sys.argv[1:4] + ["224", "256", "64"][len(sys.argv) - 1:]))
dev = "cuda"
pairs = [make_pair(i, H, H, C, device=dev) for i in range(B)]
st = lambda k: torch.stack([p[k] for p in pairs])
feat_a, feat_q, mask_a, mask_q = st("feat_a"), st("feat_q"), st("mask_a"), st("mask_q")
del pairs
roi_a, na = ops.roi_compact(mask_a)
roi_q, nq = ops.roi_compact(mask_q)
ops.roi_subsample_(roi_a, na, 5000, seed=1)
cap_a, cap_q = ops.round_up(min(5000, H * H), 256), ops.round_up(H * H, 256)
cp = 256 if C <= 256 else 512


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


rows_q, rows_a = float(nq.sum()), float(na.sum())
t_old_q = timeit(lambda: ops.gather_normalise_q8(feat_q, roi_q, nq, cap_q, cp))
t_old_a = timeit(lambda: ops.gather_normalise_q8(feat_a, roi_a, na, cap_a, cp))
t_new_q = timeit(lambda: ops.gather_q8(feat_q, roi_q, nq, cap_q, cp))
t_new_a = timeit(lambda: ops.gather_q8(feat_a, roi_a, na, cap_a, cp, want_f32=True))
gb = lambda rows, out_b: rows * (4 * C + out_b) / 1e9
print(f"K0 round 1  query {t_old_q:.3f} ms ({gb(rows_q, 5 * cp) / t_old_q:.2f} TB/s)  anchor {t_old_a:.3f} ms")
print(f"K0v3 NCHW   query {t_new_q:.3f} ms ({gb(rows_q, cp + 4.25) / t_new_q:.2f} TB/s; reads alone {rows_q * 4 * C / 1e9 / t_new_q:.2f} TB/s)  "
      f"anchor(+f32) {t_new_a:.3f} ms ({gb(rows_a, 5 * cp) / t_new_a:.2f} TB/s)")
fq_cl, fa_cl = feat_q.contiguous(memory_format=torch.channels_last), feat_a.contiguous(memory_format=torch.channels_last)
t_cl_q = timeit(lambda: ops.gather_q8(fq_cl, roi_q, nq, cap_q, cp))
t_cl_a = timeit(lambda: ops.gather_q8(fa_cl, roi_a, na, cap_a, cp, want_f32=True))
print(f"K0v3 NHWC   query {t_cl_q:.3f} ms ({gb(rows_q, cp + 4.25) / t_cl_q:.2f} TB/s)  anchor(+f32) {t_cl_a:.3f} ms")

# matcher on both operand sets
a_hat, _, a8, a_sc, _ = ops.gather_normalise_q8(feat_a, roi_a, na, cap_a, cp)
q_hat, _, q8, q_sc, q_eps = ops.gather_normalise_q8(feat_q, roi_q, nq, cap_q, cp)
t_m_old = timeit(lambda: ops.match_screened8(a_hat, q_hat, a8, q8, a_sc, q_sc, q_eps, na, nq, 0.25, C), 5)
r_old = ops.match_screened8(a_hat, q_hat, a8, q8, a_sc, q_sc, q_eps, na, nq, 0.25, C)
del q_hat
a8n, a_scn, _, _, a_hatn = ops.gather_q8(feat_a, roi_a, na, cap_a, cp, want_f32=True)
for name, fq in (("NCHW", feat_q), ("NHWC", fq_cl)):
    q8n, q_scn, q_epsn, q_norm, _ = ops.gather_q8(fq, roi_q, nq, cap_q, cp)
    und = torch.zeros(B, dtype=torch.int32, device=dev)
    t_m_new = timeit(lambda: ops.match_screened8_raw(a_hatn, a8n, a_scn, fq, roi_q, q_norm, q8n, q_scn, q_epsn, na, nq, 0.25, und), 5)
    r_new = ops.match_screened8_raw(a_hatn, a8n, a_scn, fq, roi_q, q_norm, q8n, q_scn, q_epsn, na, nq, 0.25, und)
    same = all(torch.equal(r_old[2][b, :int(na[b])], r_new[2][b, :int(na[b])]) and
               torch.equal(r_old[1][b, :int(na[b])][r_old[2][b, :int(na[b])].bool()], r_new[1][b, :int(na[b])][r_old[2][b, :int(na[b])].bool()]) and
               torch.equal(r_old[0][b, :int(na[b])][r_old[2][b, :int(na[b])].bool()], r_new[0][b, :int(na[b])][r_old[2][b, :int(na[b])].bool()])
               for b in range(B))
    print(f"matcher K1s8: round-1 operands {t_m_old:.3f} ms | K0v3 {name} operands (raw re-scoring) {t_m_new:.3f} ms | identical on valid rows: {same} "
          f"| undecided anchors {int(und.sum())}")
