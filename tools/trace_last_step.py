"""Aggregates the LAST replayed step of a rocprofv3 --kernel-trace CSV (bench.py under hipGraph replay): per kernel name the
total time, launches, average duration and average workgroup count.  usage: trace_last_step.py <kernel_trace.csv> [top]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 50
# a step = the launches between two occurrences of the step's first kernel; take the last complete period
names = [r["Kernel_Name"] for r in rows]
# period detection: find the smallest p such that the last 2p names repeat
n = len(names)
per = None
for p in range(200, n // 2):
    if names[n - p:] == names[n - 2 * p:n - p]:
        per = p
        break
last = rows[n - per:] if per else rows[-640:]
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for r in last:
    g = 1
    for ax in "XYZ":
        g *= int(r[f"Grid_Size_{ax}"]) // max(1, int(r[f"Workgroup_Size_{ax}"]))
    a = agg[r["Kernel_Name"][:100]]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a[2] += g
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values())
print(f"launches per step {len(last)}, sum of kernel time {tot:.0f} us, span {span:.0f} us")
print(f"{'total us':>9} {'n':>4} {'avg us':>7} {'avg WGs':>7}  kernel")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{a[1]:9.1f} {a[0]:4d} {a[1] / a[0]:7.1f} {a[2] // a[0]:7d}  {name}")
