#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (`*_results.db`) into a per-kernel stats table
(the same content as `--stats` CSV output): calls, total / average / min / max duration, share.
Optionally also dumps PMC counter sums per kernel.

    python tools/rocpd_summary.py gpurun_out/prof/bench_results.db [--exclude REGEX] [--after REGEX] > profiles/r01_bench_kernel_stats.md

--between REGEX: only dispatches between the first and the last dispatch matching REGEX (e.g. the screen kernel: whole steps of the
pipeline, nothing of the set-up); the header then says how many matching dispatches span the window.
--after REGEX: only dispatches that START after the last dispatch matching REGEX has ended (round 5: bench.py's input generation is
torch RNG kernels and ~2000 copies; `--after distribution_elementwise` leaves the engine's steps, so that the per-step copy / fill
counts can be read off)."""
import re
import sqlite3
import sys


def main(path, exclude=None, after=None, between=None):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    where = ""
    if after:
        ends = [e for n, e in c.execute(f"select {name_col}, end from kernels") if re.search(after, n)]
        if ends:
            where = f" where start > {max(ends)}"
    n_between = 0
    if between:
        marks = [(st_, e) for n, st_, e in c.execute(f"select {name_col}, start, end from kernels") if re.search(between, n)]
        if len(marks) >= 2:
            n_between = len(marks)
            where = f" where start >= {min(m[0] for m in marks)} and start < {max(m[0] for m in marks)}"
    rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels{where} group by {name_col} order by 3 desc").fetchall()
    dropped = 0.0
    if exclude:
        dropped = sum(r[2] for r in rows if re.search(exclude, r[0])) / 1e6
        rows = [r for r in rows if not re.search(exclude, r[0])]
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 kernel-trace summary of `{path.split('/')[-1]}`\n")
    if n_between:
        print(f"(only dispatches from the first to the last one matching /{between}/: {n_between - 1} whole periods of that kernel, i.e. steps of the pipeline)\n")
    elif where:
        print(f"(only dispatches after the last one matching /{after}/ - the run's set-up - had ended)\n")
    if exclude:
        print(f"(kernels matching /{exclude}/ left out: {dropped:.1f} ms in total - one-off library auto-tuning launches of the warm-up pass)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % of GPU time |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, k, tot, avg, mn, mx in rows:
        short = n if len(n) < 90 else n[:87] + "..."
        print(f"| `{short}` | {k} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100.0 * tot / total:.2f} |")
    try:
        pm = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                       "group by 1, 2 order by 1, 2").fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        print("\n## PMC counters (sum over dispatches)\n")
        print("| kernel | counter | sum | dispatches | per dispatch |")
        print("|---|---|---:|---:|---:|")
        for n, cn, v, k in pm:
            short = n if len(n) < 70 else n[:67] + "..."
            print(f"| `{short}` | {cn} | {v:.6g} | {k} | {v / k:.6g} |")


if __name__ == "__main__":
    opt = dict(zip(sys.argv[2::2], sys.argv[3::2]))
    main(sys.argv[1], opt.get("--exclude"), opt.get("--after"), opt.get("--between"))
