#!/usr/bin/env python3
"""FETCH_SIZE (KiB) per launch of tools/fetch_calib/fetch_calib.hip divided by the bytes each kernel is known to read:
usage  summarize.py <rocprofv3 output dir> <out.json>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

BYTES = 1 << 30
f = glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True)[0]
acc = defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"].split("(")[0]].append((float(r["Counter_Value"]) * 1024.0, (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3))
out = {"bytes_read_per_kernel": BYTES, "note": "FETCH_SIZE x 1024 / bytes actually read (cold caches); factor = what FETCH_SIZE must be multiplied with",
       "kernels": {}}
for k, v in acc.items():
    if k.startswith("flush_write"):
        continue
    raw = sorted(x[0] for x in v)[len(v) // 2]
    us = sorted(x[1] for x in v)[len(v) // 2]
    out["kernels"][k] = {"launches": len(v), "fetch_size_bytes_reported": raw, "reported_over_actual": raw / BYTES, "factor": BYTES / raw,
                         "median_us": us, "GBps": BYTES / us / 1e3}
    print(f"{k:12s} reported/actual {raw / BYTES:6.3f}  -> factor {BYTES / raw:5.2f}   {us:8.1f} us  {BYTES / us / 1e3:7.1f} GB/s")
json.dump(out, open(sys.argv[2], "w"), indent=1)
