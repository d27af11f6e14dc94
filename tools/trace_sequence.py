#!/usr/bin/env python3
"""Launch sequence of ONE step from a rocprofv3 --kernel-trace rocpd database of a SERIAL engine run (ENG_SERIAL=1
tools/engine_timeline.py): kernel, start offset, duration, idle gap in front - where the registration's 1.4 ms go.

    python tools/trace_sequence.py <results.db> [filter-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "pdsc_center_kernel" in r[0]]
if len(marks) < 3:
    raise SystemExit("not enough steps")
i0, i1 = marks[-2], marks[-1]
# the step = from the previous roi_compact before center to the next one
seq = rows[i0:i1]
t0 = seq[0][1]
prev = None
tot_busy = tot_gap = 0.0
for n, s, e in seq:
    if flt and flt not in n:
        prev = e
        continue
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {n[:90]}")
    if "pdsc" in n or "kabsch" in n:
        tot_busy += (e - s) / 1e3
        tot_gap += max(gap, 0.0)
    prev = e
print(f"registration kernels: busy {tot_busy:.1f} us, gaps in front {tot_gap:.1f} us")
