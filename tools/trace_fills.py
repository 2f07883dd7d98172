"""Where do the ATen fill / copy / add launches of one fwd+bwd step come from?  Wraps the Python entry points (torch.zeros,
zeros_like, ones, full, Tensor.zero_/fill_/copy_/clone/contiguous/add_/add, torch.cat/stack) and counts calls on CUDA tensors by
the nearest frame inside this repo."""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
cnt = collections.Counter()
ON = [False]

def where():
    fr = [f for f in traceback.extract_stack()[:-2] if "xlstm-hved_amd" in f.filename or f.filename.endswith("trace_fills.py")]
    return " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr[-3:])) if fr else "(outside)"

def wrap_fn(mod, name):
    orig = getattr(mod, name)
    def w(*a, **k):
        r = orig(*a, **k)
        if ON[0] and isinstance(r, torch.Tensor) and r.is_cuda:
            cnt[(name, where())] += 1
        return r
    setattr(mod, name, w)

def wrap_method(name):
    orig = getattr(torch.Tensor, name)
    def w(self, *a, **k):
        if ON[0] and self.is_cuda:
            if name != "contiguous" or not self.is_contiguous():
                cnt[("Tensor." + name, where())] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, w)

for n in ("zeros", "zeros_like", "ones", "ones_like", "full", "cat", "stack", "empty_like"):
    wrap_fn(torch, n)
for n in ("zero_", "fill_", "copy_", "clone", "contiguous", "add_", "add", "mul_", "float", "to"):
    wrap_method(n)
torch.manual_seed(1)
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS); m.apply(X.init_weights); m = m.cuda().train()
grads = X.parallel.FlatGrads(list(m.parameters()))
X.ops.set_wgrad_defer(True)
x = torch.rand(1, 4, 64, 64, 64, device="cuda").bfloat16()
from xlstm_hved_amd.losses import sum_of_means
if "--trainstep" in sys.argv:                       # the whole training step (train_step.TrainStep) instead of one fwd+bwd
    from xlstm_hved_amd.train_step import TrainStep
    d = X.Discriminator(in_channels=7, ks=4, strides=[1, 2, 2, 2]); d.apply(X.init_weights); d = d.cuda()
    ts = TrainStep(m, d, storage=torch.bfloat16)
    mask = (torch.rand(1, 3, 64, 64, 64, device="cuda") > 0.7).float()
    def step():
        ts.compute(x, mask, [6])
else:
    def step():
        grads.zero()
        seg, (mu, lv), rec = m(x, [14], recon=True)
        loss = sum_of_means([seg, rec[0]] + [t for ab in zip(mu, lv) for t in ab])
        loss.backward()
        X.ops.join_wgrad_stream()
step(); step(); torch.cuda.synchronize()
ON[0] = True
step(); torch.cuda.synchronize()
ON[0] = False
for k, v in cnt.most_common(60):
    print(v, k)
