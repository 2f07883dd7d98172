#!/bin/bash
# Tracked MFMA-bound evidence: the Discriminator's implicit GEMMs and the 7^3 gate convs inside the training step.
#   bash tools/profile_trainstep.sh r04     (on the GPU box; results under gpurun_out/profiles_out/, copy into profiles/)
#   1. rocprofv3 --kernel-trace --stats of bench.py WITH its training-step leg  -> <tag>_trainstep_graph_bf16_kernel_stats.{csv,md}
#   2. --pmc passes of the same command: MFMA busy / wave-cycle split / LDS bank conflicts -> <tag>_trainstep_pmc_sq.json
# The program goes directly after `--`.
set -e
TAG=${1:-r04}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=/tmp/prof_ts_${TAG}_$$   # raw traces stay on the box (tens of MB); only summaries travel back; a directory of its own per run (boxes are reused)
mkdir -p $OUT gpurun_out/profiles_out
BENCH="bench.py --steps 10 --warmup 3 --inner 1 --no-cpu --no-roofline --no-modes --no-config3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $BENCH > $OUT/trace.log 2>&1
echo "trace done"
python3 tools/summarize_rocprof.py $OUT/trace gpurun_out/profiles_out/${TAG}_trainstep_graph_bf16_kernel_stats.md --steps 35 --title "Round 6 ($TAG): bench.py with the training-step leg (train.py:208-296, ks=4 Discriminator), bf16, hipGraph replay, 128^3" --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 $BENCH  (generator-only step: 2 eager + 1 capture + 16 replays; training step: 2 eager + 1 capture + 33 replays -- per-launch averages of the dconv_* / dwgrad_* kernels are training-step figures)" >> $OUT/trace.log 2>&1 || echo "summarize_rocprof failed"
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) gpurun_out/profiles_out/${TAG}_trainstep_graph_bf16_kernel_stats.csv
# the training step is the last graph the command replays: its launch list and family timeline
python3 tools/dump_step.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > gpurun_out/profiles_out/${TAG}_trainstep_launches.txt 2>&1 || true
python3 tools/timeline_step.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) gpurun_out/profiles_out/trainstep_families.json "profiles/${TAG}_trainstep_timeline.txt: rocprofv3 --kernel-trace -- python3 $BENCH, last replayed training step (tools/timeline_step.py)" > gpurun_out/profiles_out/${TAG}_trainstep_timeline.txt 2>&1 || true
if [ "${PMC:-1}" = "0" ]; then echo "trace only"; exit 0; fi
PB="bench.py --steps 3 --warmup 1 --inner 1 --no-cpu --no-roofline --no-modes --no-config3"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq1 -- python3 $PB > $OUT/sq1.log 2>&1
echo "sq1 done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq2 -- python3 $PB > $OUT/sq2.log 2>&1
echo "sq2 done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/sq3 -- python3 $PB > $OUT/sq3.log 2>&1 || echo "sq3 failed"
echo "sq3 done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/sq4 -- python3 $PB > $OUT/sq4.log 2>&1 || echo "sq4 failed"
echo "sq4 done"
python3 tools/summarize_sq.py gpurun_out/profiles_out/${TAG}_trainstep_pmc_sq.json $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/sq4 --source "rocprofv3 --pmc <group> --kernel-trace -- python3 $PB (four passes: MFMA busy / wave-cycle split / instruction counts / LDS conflicts)" > $OUT/sq_summary.log 2>&1 || echo "summarize_sq failed"
tail -40 $OUT/sq_summary.log
