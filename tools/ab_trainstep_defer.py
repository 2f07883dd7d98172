"""Generator gradients of one fp32 TrainStep.compute at the test size with and without the deferred weight-gradient batch, and the
run-to-run noise of each (fp32 atomics): plain vs deferred must sit at the noise level (1e-6 of the largest gradient).
usage (GPU box): python tools/ab_trainstep_defer.py"""
import sys, torch
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import xlstm_hved_amd as X
from xlstm_hved_amd.train_step import TrainStep
from gpu_common import load
import test_gpu_trainstep as T
x, mask, eps = T._inputs()
w = load("weights_seed1")
def run(defer):
    m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS); m.load_state_dict(w, strict=True); m = m.cuda().train()
    ts = TrainStep(m, T._disc().cuda(), alpha=T.ALPHA, beta=T.BETA, storage=torch.float32, shared_encoder=True, defer_wgrads=defer)
    ts.compute(x.cuda(), mask.cuda(), [6], eps_lists=[[e.cuda() for e in el] for el in eps])
    torch.cuda.synchronize()
    return ts.grads.flat.clone()
a = run(False); b = run(False); c = run(True); d = run(True)
s = a.abs().max().item()
print("scale", s)
print("plain vs plain", ((a - b).abs().max() / s).item(), ((a - b).norm() / a.norm()).item())
print("plain vs defer", ((a - c).abs().max() / s).item(), ((a - c).norm() / a.norm()).item())
print("defer vs defer", ((c - d).abs().max() / s).item(), ((c - d).norm() / c.norm()).item())
