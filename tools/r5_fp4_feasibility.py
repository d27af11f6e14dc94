"""VERDICT r04 item 3, evaluated before building anything: would MX-fp4 (e2m1, per-32-channel E8M0 exponents) screening operands settle
the anchors of the cfg2 workload?  For synthetic cfg2 pairs (oryon_amd.synth.make_pair, 224 x 224, C = 256) this script quantises the
unit rows exactly as K0 would (block exponent puts the block maximum into (3, 6] code units, round-to-nearest-even onto
{0, .5, 1, 1.5, 2, 3, 4, 6}), measures every row's 2-norm quantisation error (the quantity the MX-fp6 screen's bound is built from:
|s - a^.q^| <= |ea| + |eq| + |ea||eq|), runs the screen in fp32 on the dequantised rows and classifies every anchor the way
match_decide_lite_kernel does:   INVALID  m1 + d < 1 - 2 thr      VALID + decided  m1 - d > 1 - 2 thr and m1 - m2 > 2 d
(m1 / m2: best / runner-up SLICE maximum).  The same is done for e2m3 (fp6) as the control.  CPU, a few seconds per pair.
usage: python tools/r5_fp4_feasibility.py [pairs]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oryon_amd.synth import make_pair

FP4 = np.array([0, .5, 1, 1.5, 2, 3, 4, 6], dtype=np.float32)
FP6 = np.array(sorted({(m / 8.0) for m in range(8)} | {(1 + m / 8.0) * 2.0 ** e for e in range(3) for m in range(8)}), dtype=np.float32)


def quantise(rows, grid):
    """rows [n, C] unit rows -> dequantised rows (block-scaled, nearest grid value, ties to the even code = lower index parity ignored)."""
    n, C = rows.shape
    blk = rows.reshape(n, C // 32, 32)
    bm = np.abs(blk).max(axis=2, keepdims=True)
    top = grid[-1]
    e = np.where(bm > 0, np.ceil(np.log2(np.maximum(bm, 1e-30) / top)), -40.0)          # bm / 2^e in (top / 2, top]
    sc = np.exp2(e).astype(np.float32)
    x = np.abs(blk) / sc
    idx = np.abs(x[..., None] - grid[None, None, None, :]).argmin(axis=3)
    deq = np.sign(blk) * grid[idx] * sc
    return deq.reshape(n, C).astype(np.float32)


def classify(a_hat, q_hat, grid, thr=0.25):
    aq, qq = quantise(a_hat, grid), quantise(q_hat, grid)
    ea = np.linalg.norm(a_hat - aq, axis=1).max()
    eq = np.linalg.norm(q_hat - qq, axis=1).max()
    d = ea + eq + ea * eq + 1.2e-4
    S = torch.from_numpy(aq) @ torch.from_numpy(qq).T                                     # [n_a, n_q]
    nq = S.shape[1] // 16 * 16
    sl = S[:, :nq].reshape(S.shape[0], nq // 16, 16).amax(dim=2)                          # slice maxima (16 consecutive rows: a proxy)
    top2 = sl.topk(2, dim=1).values
    m1, m2 = top2[:, 0].numpy(), top2[:, 1].numpy()
    cut = 1 - 2 * thr
    invalid = m1 + d < cut
    decided = (m1 - d > cut) & (m1 - m2 > 2 * d)
    uncertain = ~invalid & ~(m1 - d > cut)
    return dict(delta=float(d), ea=float(ea), eq=float(eq), invalid=float(invalid.mean()), valid_decided=float(decided.mean()),
                validity_uncertain=float(uncertain.mean()), settled=float((invalid | decided).mean()))


if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    rng = np.random.default_rng(0)
    for i in range(P):
        p = make_pair(i, 224, 224, 256)
        fa, fq = p["feat_a"].numpy().reshape(256, -1).T, p["feat_q"].numpy().reshape(256, -1).T
        ra = np.nonzero(p["mask_a"].numpy().reshape(-1) == 1)[0]
        rq = np.nonzero(p["mask_q"].numpy().reshape(-1) == 1)[0]
        ra = np.sort(rng.choice(ra, 5000, replace=False))
        a = fa[ra]; q = fq[rq]
        a_hat = a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-8)
        q_hat = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-8)
        for name, grid in (("fp6 e2m3", FP6), ("fp4 e2m1", FP4)):
            r = classify(a_hat.astype(np.float32), q_hat.astype(np.float32), grid)
            print(f"pair {i} {name}: row error {r['ea']:.4f}/{r['eq']:.4f} delta {r['delta']:.3f} | invalid {r['invalid']:.3f} "
                  f"valid+decided {r['valid_decided']:.3f} validity uncertain {r['validity_uncertain']:.3f} | settled {r['settled']:.3f}")
