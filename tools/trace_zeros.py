import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from xlstm_hved_amd import ops
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS); m.apply(X.init_weights); m = m.cuda().train()
grads = X.parallel.FlatGrads(list(m.parameters()))
x = torch.rand(1, 4, 64, 64, 64, device="cuda").bfloat16()
def step():
    grads.zero()
    seg, (mu, lv), rec = m(x, [14], recon=True)
    (seg.float().mean() + rec[0].float().mean() + sum(a.float().mean() + b.float().mean() for a, b in zip(mu, lv))).backward()
step()
log = collections.Counter()
oz = torch.zeros
def z(*a, **k):
    if k.get("dtype") == torch.float64:
        st = traceback.extract_stack(limit=5)
        log[" < ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in reversed(st[:-1]))] += 1
    return oz(*a, **k)
torch.zeros = z
ozero = torch.Tensor.zero_
def zz(self):
    if self.dtype == torch.float64:
        st = traceback.extract_stack(limit=5)
        log["zero_ " + " < ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in reversed(st[:-1]))] += 1
    return ozero(self)
torch.Tensor.zero_ = zz
step(); torch.cuda.synchronize()
print("arena used doubles:", [v[1] for v in ops._ARENA.values()])
for k, v in log.most_common(): print(v, k)
