"""Counts the ATen (non-library) GPU launches of one fwd+bwd step by ATen op and nearest python frame of this repo."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import xlstm_hved_amd as X
torch.manual_seed(1)
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS); m.apply(X.init_weights); m = m.cuda().train()
grads = X.parallel.FlatGrads(list(m.parameters()))
X.ops.set_wgrad_defer(True)
x = torch.rand(1, 4, 64, 64, 64, device="cuda").bfloat16()
def step():
    grads.zero()
    seg, (mu, lv), rec = m(x, [14], recon=True)
    from xlstm_hved_amd.losses import sum_of_means
    loss = sum_of_means([seg, rec[0] if isinstance(rec, (list, tuple)) else rec] + [t for ab in zip(mu, lv) for t in ab])
    loss.backward()
    X.ops.join_wgrad_stream()
step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.kernels:
        # only leaf aten ops that launched kernels themselves
        if any(c.name.startswith("aten::") and c.kernels for c in ev.cpu_children):
            continue
        frame = next((s for s in ev.stack if "xlstm-hved_amd" in s or "trace_aten" in s), "(autograd engine)")
        frame = frame.split("/")[-1]
        cnt[(ev.name, frame)] += len(ev.kernels)
tot = sum(cnt.values())
print("aten-launched kernels per step:", tot)
for k, v in cnt.most_common(60):
    print(v, k)
