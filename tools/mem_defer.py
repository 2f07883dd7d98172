"""Peak device memory of one 1x4x128^3 bf16 forward+backward with and without deferred weight gradients.  python tools/mem_defer.py"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xlstm_hved_amd as X
from xlstm_hved_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(1)
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS); m.apply(X.init_weights); m = m.to(dev).train()
x = torch.rand(1, 4, 128, 128, 128).to(dev, torch.bfloat16)
grads = X.parallel.FlatGrads(list(m.parameters()))
for defer in (False, True, False, True):
    ops.set_wgrad_defer(defer)
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    seg, (mu, lv), rec = m(x, [14], recon=True)
    loss = seg.float().mean() + rec[0].float().mean()
    loss.backward(); ops.join_wgrad_stream(); torch.cuda.synchronize()
    print(f"defer={defer}: peak above baseline {(torch.cuda.max_memory_allocated() - base) / 2**20:.0f} MiB")
    del seg, mu, lv, rec, loss
