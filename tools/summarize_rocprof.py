#!/usr/bin/env python3
"""Turns a rocprofv3 --kernel-trace --stats run (gpurun_out/<dir>/**/*_kernel_stats.csv) into the summary committed under
profiles/: usage  python tools/summarize_rocprof.py gpurun_out/prof_x profiles/r1_name.md --steps 9 --title "..." """
import argparse
import csv
import glob
import os

ap = argparse.ArgumentParser()
ap.add_argument("src")
ap.add_argument("dst")
ap.add_argument("--steps", type=int, required=True, help="fwd+bwd steps executed under the profiler (incl. warm-up)")
ap.add_argument("--title", default="")
ap.add_argument("--cmd", default="")
a = ap.parse_args()
f = glob.glob(os.path.join(a.src, "**", "*_kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
native = sum(float(r["TotalDurationNs"]) for r in rows if not (r["Name"].startswith("void at::") or r["Name"].startswith("__amd") or "at::native" in r["Name"]))
with open(a.dst, "w") as o:
    o.write(f"# {a.title}\n\n")
    if a.cmd:
        o.write(f"Command: `{a.cmd}`\n\n")
    o.write(f"Source: rocprofv3 --kernel-trace --stats ({os.path.basename(f)}), {a.steps} fwd+bwd steps under the profiler.\n\n")
    o.write(f"* GPU kernel time per step: **{tot / a.steps / 1e6:.3f} ms** ({len(rows)} distinct kernels)\n")
    o.write(f"* share spent in this repo's HIP kernels: {100 * native / tot:.1f} % (rest: ATen fills/adds/copies/reductions used as glue)\n\n")
    o.write("| kernel | launches/step | avg us | ms/step | % |\n|---|---:|---:|---:|---:|\n")
    for r in rows:
        t = float(r["TotalDurationNs"])
        if t / tot < 0.002:
            continue
        name = r["Name"].replace("|", "\\|")
        if len(name) > 120:
            name = name[:117] + "..."
        o.write(f"| `{name}` | {int(r['Calls']) / a.steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {t / a.steps / 1e6:.3f} | {100 * t / tot:.2f} |\n")
print(open(a.dst).read()[:3000])
