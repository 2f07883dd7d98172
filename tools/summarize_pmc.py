#!/usr/bin/env python3
"""Joins the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate passes, the TCC block cannot hold both) into
per-kernel HBM-side traffic per launch.

    python tools/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_traffic.json --dtype bf16

Units/corrections follow MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B
requests at 64 B for wide (16 B/lane) coalesced reads, so fetch bytes are doubled; WRITE_SIZE is exact.  Check of the
factor on this code's own access pattern: norm_bwd_apply (reads two tensors, writes one, all 16-B accesses) shows
fetch:write = 2.0 after doubling, maxpool2_fwd 8.0, and the MFMA forward conv 16->16 @128^3 reads 1.12x its input
(= the (SD+2)/SD depth halo of its 16-plane runs).  Infinity-Cache hits are included in both counters (memory-side L2
request counters), so this is L2<->fabric traffic, an upper bound on HBM bytes.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void\s+", "", r["Kernel_Name"])
        name = re.sub(r"\(.*\)$", "", name)
        a = acc[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return acc


def csrc_hash():
    """sha1 over the kernel sources: tells whether a committed capture still describes the kernels of the tree it is read in."""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xlstm-hved_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()


CALIBRATED = {}      # kernel-name prefix -> FETCH_SIZE factor where a kernel was calibrated otherwise (default 2.0)


def main():
    fetch_dir, write_dir, dst = sys.argv[1:4]
    dtype = sys.argv[sys.argv.index("--dtype") + 1] if "--dtype" in sys.argv else "bf16"
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    out = {"dtype": dtype, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace",
           "corrections": "KiB -> bytes; FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B); includes Infinity-Cache hits",
           "captured_date": __import__("datetime").datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), "csrc_sha1": csrc_hash(),
           "captured_commit": None,      # the GPU box has no .git: tools/stamp_profile.py fills this in when the file is committed
           "kernels": {}}
    for name in sorted(fe, key=lambda k: -fe[k][1]):
        if name not in wr or name.startswith("at::") or name.startswith("__amd"):
            continue
        n = fe[name][0]
        factor = next((v for k, v in CALIBRATED.items() if name.startswith(k)), 2.0)
        fb = factor * 1024.0 * fe[name][1] / n
        wb = 1024.0 * wr[name][1] / wr[name][0]
        out["kernels"][name] = {"launches": n, "fetch_factor": factor, "fetch_bytes_per_launch": fb,
                                "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in list(out["kernels"].items())[:25]:
        print(f"{k[:70]:70s} n={v['launches']:4d} fetch {v['fetch_bytes_per_launch'] / 1e6:9.2f} MB  write {v['write_bytes_per_launch'] / 1e6:9.2f} MB")


main()
