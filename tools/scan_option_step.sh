#!/bin/bash
# In-step average duration of one kernel (rocprofv3 kernel stats of the default bench command) for a list of xh_set_option settings:
#   bash tools/scan_option_step.sh TAG KERNEL_SUBSTRING "12=256" "12=1024" ...
TAG=$1; KER=$2; shift; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
BENCH="bench.py --steps 20 --warmup 5 --inner 1 --no-cpu --no-roofline --no-modes --no-trainstep --no-config3"
for cfg in "$@"; do
  OUT=/tmp/scanopt_${TAG}_$$_$(echo $cfg | tr -c 'A-Za-z0-9' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $BENCH --xh-option $cfg > $OUT.log 2>&1 < /dev/null
  f=$(find $OUT -name "*kernel_stats.csv" 2>/dev/null | head -1)
  echo "$cfg: $(grep -h "$KER" ${f:-/dev/null} | cut -d, -f1-5 | head -3 | tr '\n' ' ')  step: $(grep -o '"ms_per_step": [0-9.]*' $OUT.log | head -1)" | tee -a gpurun_out/$TAG/scan.txt
done
