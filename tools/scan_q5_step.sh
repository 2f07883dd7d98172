#!/bin/bash
# In-step average duration of conv3_wgrad_q5_multi_kernel (rocprofv3 kernel stats of the default bench command) for a list of
# launch-plan settings given as "NAME=VALUE,NAME=VALUE" words:  bash tools/scan_q5_step.sh TAG "XH_Q5_UQ=3" "XH_Q5_F3=2.4" ...
TAG=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
BENCH="bench.py --steps 20 --warmup 5 --inner 1 --no-cpu --no-roofline --no-modes --no-trainstep --no-config3"
for cfg in "$@"; do
  OUT=/tmp/scan_${TAG}_$$_$(echo $cfg | tr -c 'A-Za-z0-9' '_')
  envs=$(echo $cfg | tr ',' ' ')
  env $envs timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $BENCH > $OUT.log 2>&1 < /dev/null
  f=$(find $OUT -name "*kernel_stats.csv" 2>/dev/null | head -1)
  echo "$cfg: $(grep -h 'q5_multi' ${f:-/dev/null} | cut -d, -f1-5)  step: $(grep -o '"ms_per_step": [0-9.]*' $OUT.log | head -1)" | tee -a gpurun_out/$TAG/scan.txt
done
