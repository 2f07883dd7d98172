#!/bin/bash
# Kernel trace of the default bench command, summarised (no PMC passes):  bash tools/profile_step.sh <tag>
TAG=${1:-r05x}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=/tmp/prof_$TAG
mkdir -p $OUT gpurun_out/profiles_out
BENCH="bench.py --steps 20 --warmup 5 --inner 1 --no-cpu --no-roofline --no-modes --no-trainstep --no-config3 ${XH_BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $BENCH > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
python3 tools/summarize_rocprof.py $OUT/trace gpurun_out/profiles_out/${TAG}_bench_graph_bf16_kernel_stats.md --steps 28 --title "Round 6 ($TAG): bench.py bf16, hipGraph replay, 128^3 patch" --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 $BENCH  (2 eager + 1 capture + 25 replayed steps)" >> $OUT/trace.log 2>&1 || echo "summarize_rocprof failed"
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) gpurun_out/profiles_out/${TAG}_bench_graph_bf16_kernel_stats.csv
python3 tools/timeline_step.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) gpurun_out/profiles_out/step_families.json "profiles/${TAG}_timeline.txt: rocprofv3 --kernel-trace -- python3 $BENCH, last replayed step (tools/timeline_step.py)" > gpurun_out/profiles_out/${TAG}_timeline.txt 2>&1 || true
python3 tools/dump_step.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > gpurun_out/profiles_out/${TAG}_launches.txt 2>&1 || true
head -30 gpurun_out/profiles_out/${TAG}_timeline.txt
