"""Lists, for one fwd+bwd step at 128^3, every call of selected ops with tensor shapes and the autograd Function that made it."""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from xlstm_hved_amd import ops
names = sys.argv[1:] or ["moments"]
log = collections.Counter()
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        t = next(x for x in a if torch.is_tensor(x))
        st = traceback.extract_stack(limit=6)
        who = " < ".join(f"{s.name}:{s.lineno}" for s in reversed(st[:-1]) if "functional" in s.filename or "blocks" in s.filename or "model" in s.filename)
        log[(name, tuple(t.shape), who)] += 1
        return orig(*a, **k)
    setattr(ops, name, f)
for n in names: wrap(n)
torch.manual_seed(1)
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS); m.apply(X.init_weights); m = m.cuda().train()
x = torch.rand(1, 4, 128, 128, 128, device="cuda").bfloat16()
seg, (mu, lv), rec = m(x, [14], recon=True)
loss = seg.float().mean() + rec[0].float().mean() + sum(a.float().mean() + b.float().mean() for a, b in zip(mu, lv))
loss.backward(); torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: -kv[0][1][-1] * kv[0][1][1]):
    print(v, k)
