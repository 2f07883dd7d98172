#!/usr/bin/env python3
"""Per-kernel SQ counters from rocprofv3 --pmc passes (one directory per pass; the SQ block holds 8 counters per pass):

    python tools/summarize_sq.py <out.json> <pass_dir> [<pass_dir> ...] [--source "<command>"]

Per kernel name (launch averages): every counter found, plus derived figures
  mfma_busy_frac      SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs * GRBM_GUI_ACTIVE / XCDs)    matrix-core busy share of the launch
  issue_busy_frac     SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES                            a wave-cycle in which the wave issued
  wait_any_frac       SQ_WAIT_ANY / SQ_WAVE_CYCLES                                   parked on s_waitcnt / barrier
  wait_inst_frac      SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                              waiting for an issue slot
  insts_per_wave      SQ_INSTS_VALU.. (all categories summed) / SQ_WAVES
  lds_bank_conflict_frac  SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS                  LDS bank-conflict cycles per LDS-active cycle
(MI355X_MICROARCH.md, SQ row: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES.)  GRBM_GUI_ACTIVE is summed over the
8 XCDs by rocprofv3; 256 CUs x 4 SIMDs = 1024 SIMDs."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def main():
    args = sys.argv[1:]
    source = None
    if "--source" in args:
        i = args.index("--source")
        source = args[i + 1]
        del args[i:i + 2]
    dst, dirs = args[0], args[1:]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    dur = defaultdict(lambda: [0, 0.0])
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                name = re.sub(r"\(.*\)$", "", re.sub(r"^void\s+", "", r["Kernel_Name"]))
                if name.startswith("at::") or name.startswith("__amd") or name.startswith("Cijk") or name.startswith("rocblas"):
                    continue
                a = acc[name][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                key = (name, r["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    dur[name][0] += 1
                    dur[name][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = {"source": source or "rocprofv3 --pmc <SQ counters> --kernel-trace (one pass per counter group)",
           "note": "per-launch averages; GRBM_GUI_ACTIVE summed over the 8 XCDs; 1024 SIMDs", "kernels": {}}
    for name, cs in acc.items():
        k = {c: v[1] / v[0] for c, v in cs.items()}
        k["launches"] = max(v[0] for v in cs.values())
        k["avg_us_under_pmc"] = dur[name][1] / max(dur[name][0], 1) / 1e3
        if "SQ_VALU_MFMA_BUSY_CYCLES" in k and k.get("GRBM_GUI_ACTIVE", 0) > 0:
            k["mfma_busy_frac"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * k["GRBM_GUI_ACTIVE"] / 8.0)
        wc = k.get("SQ_WAVE_CYCLES", 0)
        if wc > 0:
            for src, dstk in (("SQ_ACTIVE_INST_ANY", "issue_busy_frac"), ("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_frac")):
                if src in k:
                    k[dstk] = k[src] / wc
        if k.get("SQ_ACTIVE_INST_LDS", 0) > 0 and "SQ_LDS_BANK_CONFLICT" in k:
            k["lds_bank_conflict_frac"] = k["SQ_LDS_BANK_CONFLICT"] / k["SQ_ACTIVE_INST_LDS"]     # conflict cycles per LDS-active cycle
        insts = [v for c, v in k.items() if c.startswith("SQ_INSTS_")]
        if insts and k.get("SQ_WAVES", 0) > 0:
            k["insts_per_wave"] = sum(insts) / k["SQ_WAVES"]
        out["kernels"][name] = k
    out["kernels"] = dict(sorted(out["kernels"].items(), key=lambda kv: -kv[1]["avg_us_under_pmc"] * kv[1]["launches"]))
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    for name, k in list(out["kernels"].items())[:30]:
        print(f"{name[:64]:64s} n={k['launches']:4d} {k['avg_us_under_pmc']:8.1f} us  mfma {k.get('mfma_busy_frac', float('nan')):.3f}  "
              f"issue {k.get('issue_busy_frac', float('nan')):.2f}  wait {k.get('wait_any_frac', float('nan')):.2f}  inst/wave {k.get('insts_per_wave', float('nan')):.0f}")


if __name__ == "__main__":
    main()
