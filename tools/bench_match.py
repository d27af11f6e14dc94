#!/usr/bin/env python3
"""Micro-benchmark of the K0/K1 matcher stages on one GPU (development aid; bench.py is the contract)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oryon_amd import ops  # noqa: E402
from oryon_amd.synth import make_pair  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--C", type=int, default=256)
    ap.add_argument("--H", type=int, default=224)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = "cuda"
    t0 = time.time()
    pairs = [make_pair(i, a.H, a.H, a.C, device=dev) for i in range(a.B)]
    feat_a = torch.stack([p["feat_a"] for p in pairs])
    feat_q = torch.stack([p["feat_q"] for p in pairs])
    mask_a = torch.stack([p["mask_a"] for p in pairs])
    mask_q = torch.stack([p["mask_q"] for p in pairs])
    del pairs
    torch.cuda.synchronize()
    print(f"generated {a.B} pairs in {time.time() - t0:.1f}s")

    def stage_times():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        ev[0].record()
        roi_a, na = ops.roi_compact(mask_a)
        roi_q, nq = ops.roi_compact(mask_q)
        ev[1].record()
        ops.roi_subsample_(roi_a, na, 5000, seed=1)
        ev[2].record()
        a_hat = ops.gather_normalise(feat_a, roi_a, na, 5120)
        q_hat = ops.gather_normalise(feat_q, roi_q, nq, ops.round_up(a.H * a.H, 256))
        ev[3].record()
        md, am, va = ops.match(a_hat, q_hat, na, nq, 0.25)
        ev[4].record()
        corrs, nv, _, st = ops.select_corrs(roi_a, roi_q, na, nq, am, va, a.H, 500, 1)
        ev[5].record()
        torch.cuda.synchronize()
        return [ev[i].elapsed_time(ev[i + 1]) for i in range(5)], na, nq, nv, st

    for it in range(a.iters):
        ts, na, nq, nv, st = stage_times()
        flops = 2.0 * (na.double() * nq.double()).sum().item() * a.C
        print(f"iter {it}: compact {ts[0]:.3f} ms  subsample {ts[1]:.3f}  gather {ts[2]:.3f}  match {ts[3]:.3f}  select {ts[4]:.3f}"
              f" | match {flops / ts[3] / 1e9:.1f} TF/s  ({a.B / (sum(ts) / 1e3):.1f} pairs/s matcher-only)")
    print("n_a", na[:4].tolist(), "n_q", nq[:4].tolist(), "n_valid", nv[:4].tolist(), "status", st[:4].tolist())


if __name__ == "__main__":
    main()
