"""Launch sequence of the LAST replayed step of a rocprofv3 --kernel-trace CSV: index, duration (us), workgroups, kernel, start (us from the step's first launch), idle gap in front when > 2 us.
usage: dump_step.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]; n = len(names)
per = next(p for p in range(200, n // 2) if names[n - p:] == names[n - 2 * p:n - p])
t0 = int(rows[n - per]["Start_Timestamp"])
end = t0
for i, r in enumerate(rows[n - per:]):
    gap = (int(r["Start_Timestamp"]) - end) / 1e3        # idle time in front of this launch (negative: it overlaps an earlier one)
    end = max(end, int(r["End_Timestamp"]))
    g = 1
    for ax in "XYZ":
        g *= int(r[f"Grid_Size_{ax}"]) // max(1, int(r[f"Workgroup_Size_{ax}"]))
    nm = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "")).replace("at::native::", "").replace("vectorized_elementwise_kernel", "vec_elt")
    print(f"{i:3d} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:6.1f} {g:6d} {nm[:70]:70s} @{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f}" + (f"  gap {gap:.1f}" if gap > 2.0 else ""))
