#!/bin/bash
# Profiles of one round, run on the GPU box:  bash tools/profile_round.sh r03a
# Output lands in gpurun_out/profiles_out/ (merged back by gpurun); copy what is to be judged into profiles/ afterwards.
#   1. rocprofv3 --kernel-trace --stats of the default bench command (hipGraph replay)  -> profiles/<tag>_bench_graph_bf16_kernel_stats.{csv,md}
#   2. --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs)                              -> profiles/<tag>_pmc_traffic.json (+ pmc_traffic.json)
#   3. --pmc SQ passes (MFMA busy, issue / wait cycles, instruction counts)              -> profiles/<tag>_pmc_sq.json (+ pmc_mfma_busy.json)
# The program goes directly after `--` (no env / bash -c hop: the profiler's library has initialised the GPU before).
set -e
TAG=${1:-r03}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=/tmp/prof_${TAG}_$$   # raw traces stay on the box (tens of MB); only summaries travel back; a directory of its own per run (boxes are reused)
mkdir -p $OUT gpurun_out/profiles_out
BENCH="bench.py --steps 20 --warmup 5 --inner 1 --no-cpu --no-roofline --no-modes --no-trainstep --no-config3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $BENCH > $OUT/trace.log 2>&1
echo "trace done"
# steps under the profiler: 2 eager warm-up + 1 capture + (warmup + steps) replays
python3 tools/summarize_rocprof.py $OUT/trace gpurun_out/profiles_out/${TAG}_bench_graph_bf16_kernel_stats.md --steps 28 --title "Round 6 ($TAG): bench.py bf16, hipGraph replay, 128^3 patch" --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 $BENCH  (2 eager + 1 capture + 25 replayed steps)" >> $OUT/trace.log 2>&1 || echo "summarize_rocprof failed"
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) gpurun_out/profiles_out/${TAG}_bench_graph_bf16_kernel_stats.csv
python3 tools/timeline_step.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) gpurun_out/profiles_out/step_families.json "profiles/${TAG}_timeline.txt: rocprofv3 --kernel-trace -- python3 $BENCH, last replayed step (tools/timeline_step.py)" > gpurun_out/profiles_out/${TAG}_timeline.txt 2>&1 || true
python3 tools/dump_step.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > gpurun_out/profiles_out/${TAG}_launches.txt 2>&1 || true
if [ "${PMC:-1}" = "0" ]; then echo "trace only"; exit 0; fi
PB="bench.py --steps 4 --warmup 1 --inner 1 --no-cpu --no-roofline --no-modes --no-trainstep --no-config3"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $PB > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $PB > $OUT/write.log 2>&1
echo "write done"
python3 tools/summarize_pmc.py $OUT/fetch $OUT/write gpurun_out/profiles_out/${TAG}_pmc_traffic.json --dtype bf16 > $OUT/pmc_traffic.log 2>&1 && cp gpurun_out/profiles_out/${TAG}_pmc_traffic.json gpurun_out/profiles_out/pmc_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq1 -- python3 $PB > $OUT/sq1.log 2>&1
echo "sq1 done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq2 -- python3 $PB > $OUT/sq2.log 2>&1
echo "sq2 done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/sq3 -- python3 $PB > $OUT/sq3.log 2>&1 || echo "sq3 failed"
echo "sq3 done"
python3 tools/summarize_sq.py gpurun_out/profiles_out/${TAG}_pmc_sq.json $OUT/sq1 $OUT/sq2 $OUT/sq3 --source "rocprofv3 --pmc <group> --kernel-trace -- python3 $PB (three passes: MFMA busy / wave-cycle split / instruction counts)" > $OUT/sq_summary.log 2>&1 && cp gpurun_out/profiles_out/${TAG}_pmc_sq.json gpurun_out/profiles_out/pmc_mfma_busy.json
tail -32 $OUT/sq_summary.log
