"""Timeline of the LAST replayed step of a rocprofv3 --kernel-trace CSV (bench.py under hipGraph replay): busy time, idle gaps
between consecutive kernels, time by workgroup-count class and by kernel family.
usage: timeline_step.py <kernel_trace.csv> [families.json [source note]]   (the JSON is what bench.py's roofline object quotes as
elementwise_ms_per_step: profiles/step_families.json)"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
n = len(names)
per = None
for p in range(200, n // 2):
    if names[n - p:] == names[n - 2 * p:n - p]:
        per = p
        break
last = rows[n - per:]
prev = rows[n - 2 * per:n - per]
S = [int(r["Start_Timestamp"]) for r in last]
E = [int(r["End_Timestamp"]) for r in last]
span = (E[-1] - S[0]) / 1e3
period = (S[0] - int(prev[0]["Start_Timestamp"])) / 1e3
busy, cur_s, cur_e = 0, S[0], E[0]
gaps = []
for s, e in zip(S[1:], E[1:]):
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e) / 1e3)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"launches {per}, step period {period:.0f} us, span {span:.0f} us, union busy {busy / 1e3:.0f} us, sum of durations {sum(e - s for s, e in zip(S, E)) / 1e3:.0f} us")
med = sorted(gaps)[len(gaps) // 2] if gaps else 0.0
print(f"idle gaps: {len(gaps)} totalling {sum(gaps):.0f} us (median {med:.2f} us, > 5 us: {sum(1 for g in gaps if g > 5)})")


def fam(nm):
    nm = re.sub(r"^void ", "", nm)
    b = nm.split("<")[0].split("(")[0]
    if b.startswith("conv3_q4"): return "conv3_q4"
    if b.startswith("conv3_wgrad_q4"): return "wgrad_q4"
    if b.startswith("conv1x1"): return "1x1"
    if b.startswith("conv7"): return "7^3"
    if b.startswith("conv3_mfma") or b.startswith("conv3_wgrad_mfma"): return "conv3_mfma"
    if b.startswith("conv") : return "vector convs"
    if b.startswith("vil") or b.startswith("mlstm"): return "ViL"
    if b.startswith("at::") or b.startswith("__amd"): return "ATen"
    return "norm / elementwise"


byf = collections.defaultdict(lambda: [0, 0.0])
bywg = collections.defaultdict(lambda: [0, 0.0])
for r in last:
    g = 1
    for ax in "XYZ":
        g *= int(r[f"Grid_Size_{ax}"]) // max(1, int(r[f"Workgroup_Size_{ax}"]))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    f = byf[fam(r["Kernel_Name"])]
    f[0] += 1; f[1] += d
    cls = "<64" if g < 64 else "<256" if g < 256 else "<1024" if g < 1024 else ">=1024"
    w = bywg[cls]
    w[0] += 1; w[1] += d
for k, v in sorted(byf.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:22s} {v[0]:4d} launches {v[1]:8.0f} us")
for k in ("<64", "<256", "<1024", ">=1024"):
    v = bywg[k]
    print(f"  workgroups {k:7s} {v[0]:4d} launches {v[1]:8.0f} us")

if len(sys.argv) > 2:
    import json
    ne = byf["norm / elementwise"]
    json.dump({"launches": per, "step_period_us": period, "sum_of_durations_us": sum(e - s for s, e in zip(S, E)) / 1e3,
               "norm_elementwise_launches": ne[0], "norm_elementwise_ms": ne[1] / 1e3,
               "families": {k: {"launches": v[0], "ms": v[1] / 1e3} for k, v in byf.items()},
               "workgroup_classes": {k: {"launches": v[0], "ms": v[1] / 1e3} for k, v in bywg.items()},
               "source": sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 --kernel-trace of bench.py (last replayed step), tools/timeline_step.py"},
              open(sys.argv[2], "w"), indent=1)
