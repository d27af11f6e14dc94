"""One steady-state step of a rocprofv3 --kernel-trace CSV as a timeline: tools/timeline_step.py <kernel_trace.csv> [anchor-kernel-substring]
(dev aid: which launches of the registration stream really run under the matcher's kernels)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "screen_v2"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchors = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
nwin = int(sys.argv[3]) if len(sys.argv) > 3 else 1
i0, i1 = anchors[-2 - nwin], anchors[-2]
t0 = int(rows[i0]["Start_Timestamp"])
print(f"step window: {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us, {i1 - i0} launches")
qs = {}
busy_end = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = qs.setdefault(r["Queue_Id"], len(qs))
    name = r["Kernel_Name"].replace("oryon::", "").replace("void ", "")[:46]
    print(f"{s / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q{q}  {name}")
