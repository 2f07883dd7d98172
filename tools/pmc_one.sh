#!/bin/bash
# One counter pass over a small program:  bash tools/pmc_one.sh <kernel substring> "<counters>" -- python3 prog.py args
KSUB=$1; CNT=$2; shift 3
export TMPDIR=/tmp
rm -rf /tmp/pp1; timeout -k 5 150 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d /tmp/pp1 -- "$@" > /tmp/pp1.log 2>&1
python3 - "$KSUB" <<'P'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pp1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:40s} {sum(v) / len(v):16.1f}  (n={len(v)})")
P
