"""ATen glue launches of one eager fwd+bwd step of bench.py's workload (fills, adds, copies) with the Python frame that issued
each: what is left to fold into the HIP kernels.   python tools/trace_aten.py [size]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import xlstm_hved_amd as X
from bench import bench_loss
ops = X.ops
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
torch.manual_seed(1)
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS); m.apply(X.init_weights); m = m.to(dev).train()
x = torch.rand(1, 4, S, S, S).to(dev, torch.bfloat16)
grads = X.parallel.FlatGrads(list(m.parameters()))
ops.set_wgrad_defer(True)
seed = torch.full((), 1.0, dtype=torch.float32, device=dev)


def step():
    grads.zero()
    seg, (mu, lv), rec = m(x, [14], recon=True)
    loss = bench_loss(seg, mu, lv, rec[0])
    loss.backward(seed if loss.dim() == 0 else seed.view(loss.shape))
    ops.join_wgrad_stream()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.count)
for e in rows:
    print(f"{e.count:3d} x {e.key:24s} cpu {e.cpu_time_total:8.1f} us  dev {getattr(e, 'device_time_total', 0.0):8.1f} us  {str(e.input_shapes)[:100]}")
