mkdir -p gpurun_out/r5d
timeout -k 10 200 python tools/microbench_wgrad_q5.py > gpurun_out/r5d/mb_q5.txt 2>&1; cat gpurun_out/r5d/mb_q5.txt | tail -12
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5d/pytest_all.txt 2>&1; tail -5 gpurun_out/r5d/pytest_all.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-trainstep > gpurun_out/r5d/bench.json 2> gpurun_out/r5d/bench.err; python - <<'P'
import json
j=json.load(open('gpurun_out/r5d/bench.json'))
print('ms', j['ms_per_step'], {k:v['ms_per_step'] for k,v in j['modes'].items()}, 'config3', j.get('config3',{}).get('ms_per_step'))
r=j['roofline']; print(r['kernel'], r['frac'])
for k,v in r['other_conv_kernels'].items(): print(k, round(v['avg_launch_us'],1), round(v['frac_of_hbm_peak'],3), v['launches_per_step'])
P
