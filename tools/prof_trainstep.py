"""Runs N eager iterations of the whole training step (train_step.TrainStep: train.py:208-296) at 128^3 in bf16, for
`rocprofv3 --kernel-trace --stats -- python3 tools/prof_trainstep.py [N]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from xlstm_hved_amd.train_step import TrainStep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.manual_seed(1)
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS); m.apply(X.init_weights); m = m.cuda().train()
torch.manual_seed(2)
d = X.Discriminator(in_channels=7, ks=4, strides=[1, 2, 2, 2]); d.apply(X.init_weights); d = d.cuda()
x = torch.rand(1, 4, 128, 128, 128, device="cuda").bfloat16()
mask = (torch.rand(1, 3, 128, 128, 128, device="cuda") > 0.7).float()
ts = TrainStep(m, d, storage=torch.bfloat16)
if "--graph" in sys.argv:                      # the captured step replayed with a new subset each time (what bench.py times)
    ts.capture(x.float(), mask)
    for i in range(n):
        ts.replay(x.float(), mask, [[3], [6], [12]][i % 3], update=False)
else:
    for _ in range(n):
        ts.compute(x, mask, [6])
torch.cuda.synchronize()
print("steps", n)
