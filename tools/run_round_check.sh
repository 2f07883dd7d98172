# Full -m gpu suite + the default bench line, as the driver runs them:  bash tools/run_round_check.sh <tag>
TAG=${1:-r6x}
mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_all.txt 2>&1; tail -5 gpurun_out/$TAG/pytest_all.txt
timeout -k 10 500 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; python tools/show_bench.py gpurun_out/$TAG/bench.json
