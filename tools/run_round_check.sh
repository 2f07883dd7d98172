mkdir -p gpurun_out/r5h
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5h/pytest_all.txt 2>&1; tail -5 gpurun_out/r5h/pytest_all.txt
timeout -k 10 400 python bench.py > gpurun_out/r5h/bench.json 2> gpurun_out/r5h/bench.err; python tools/show_bench.py gpurun_out/r5h/bench.json
