#!/bin/bash
# Counter passes (one rocprofv3 --pmc run per group) over a small program:  bash tools/pmc_passes.sh <outdir> <kernel substring> -- python3 prog.py args
# prints per counter the mean over the dispatches whose kernel name contains the substring
OUT=$1; KSUB=$2; shift 3
export TMPDIR=/tmp
mkdir -p $OUT
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed: $GROUP"
  echo "pass $i done"
done <<'G'
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
TCC_HIT_sum TCC_MISS_sum
TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum
TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES
SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
G
python3 - "$OUT" "$KSUB" <<'P'
import csv, glob, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:45s} {sum(v) / len(v):16.1f}   (n={len(v)})")
P
