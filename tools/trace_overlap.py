#!/usr/bin/env python3
"""Steady-state concurrency picture from a rocprofv3 --kernel-trace rocpd database: per HW queue, the kernels of the last few steps
grouped into runs (consecutive launches on one queue with gaps < 50 us), and for the registration kernels the average duration and
the average idle gap in front of each launch (dependent-launch latency + waiting for workgroup slots beside the screening kernel).

    python tools/trace_overlap.py <results.db> [n_steps]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
scol = "stream_id" if "stream_id" in cols else None
sel = f"select {name_col}, start, end, {qcol or 0}, {scol or 0} from kernels order by start"
rows = db.execute(sel).fetchall()
screens = [i for i, r in enumerate(rows) if "screen" in r[0] and "i8" in r[0]]
if len(screens) < nsteps + 3:
    raise SystemExit("not enough steps in the trace")
i0, i1 = screens[-nsteps - 2], screens[-2]
t0 = rows[i0][1]
win = [r for r in rows if rows[i0][1] <= r[1] < rows[i1][1]]
print(f"window: {nsteps} steps, {(rows[i1][1] - t0) / 1e6 / nsteps:.3f} ms/step, {len(win)} launches; columns: {cols}")


def cls(n):
    if "pdsc" in n or "kabsch" in n:
        return "R"
    if "roi_" in n or "gather_q8" in n:
        return "G"
    return "M"


byq = defaultdict(list)
for r in win:
    byq[(r[3], r[4])].append(r)
for q, rs in sorted(byq.items()):
    runs = []
    for r in rs:
        c = cls(r[0])
        if runs and runs[-1][0] == c and r[1] - runs[-1][2] < 200e3:
            runs[-1][2] = max(runs[-1][2], r[2]); runs[-1][3] += r[2] - r[1]; runs[-1][4] += 1
        else:
            runs.append([c, r[1], r[2], r[2] - r[1], 1])
    print(f"queue {q}: " + "  ".join(f"{c}[{(s - t0) / 1e6:.2f}-{(e - t0) / 1e6:.2f} busy {b / 1e6:.2f} n={n}]" for c, s, e, b, n in runs))
# registration kernels: duration and gap in front
stat = defaultdict(lambda: [0, 0.0, 0.0])
for q, rs in byq.items():
    prev_end = None
    for r in rs:
        if cls(r[0]) == "R":
            s = stat[r[0][:50]]
            s[0] += 1; s[1] += r[2] - r[1]
            if prev_end is not None and r[1] - prev_end < 2e6:
                s[2] += max(0, r[1] - prev_end)
        prev_end = r[2]
print("registration kernels: calls, avg duration us, avg gap in front us")
tot_d = tot_g = 0.0
for n, (k, d, g) in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:52s} {k:5d} {d / k / 1e3:8.1f} {g / k / 1e3:8.1f}")
    tot_d += d; tot_g += g
print(f"  per step: kernel time {tot_d / nsteps / 1e6:.3f} ms, gaps {tot_g / nsteps / 1e6:.3f} ms")
mstat = defaultdict(lambda: [0, 0.0])
for r in win:
    if cls(r[0]) != "R":
        mstat[r[0][:50]][0] += 1; mstat[r[0][:50]][1] += r[2] - r[1]
print("other kernels: calls, avg us")
for n, (k, d) in sorted(mstat.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {n:52s} {k:5d} {d / k / 1e3:8.1f}")
