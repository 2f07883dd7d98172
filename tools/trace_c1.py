import os, sys
sys.path.insert(0, "/root/repo")
import torch
import xlstm_hved_amd as X
from xlstm_hved_amd import ops
torch.manual_seed(1)
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS); m.apply(X.init_weights); m = m.cuda().train()
grads = X.parallel.FlatGrads(list(m.parameters()))
x = torch.rand(1, 4, 128, 128, 128, device="cuda").bfloat16()
from xlstm_hved_amd.losses import sum_of_means
orig = ops.conv3d
seen = []
def hook(xa, xb, weights, biases, **kw):
    r = orig(xa, xb, weights, biases, **kw)
    if kw.get("k") == 1 and xa.shape[2] <= 32:
        cin = xa.shape[1] + (xb.shape[1] if xb is not None else 0)
        seen.append((ops.last_conv_kernel(), cin, kw.get("cout"), kw.get("groups", 1), tuple(xa.shape[2:]), kw.get("epi", 0), kw.get("pre") is not None or kw.get("in_stats") is not None, kw.get("transposed", False)))
    return r
ops.conv3d = hook
import xlstm_hved_amd.functional as Fn
seg, (mu, lv), rec = m(x, [14], recon=True)
loss = sum_of_means([seg, rec[0]] + [t for ab in zip(mu, lv) for t in ab])
loss.backward()
torch.cuda.synchronize()
for s in seen: print(s)
