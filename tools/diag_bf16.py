"""Diagnostic: run XLSTM_HVED in fp32 and bf16 storage on the GPU and print per-stage relative L2 deviation."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import xlstm_hved_amd as X
from gpu_common import load

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS)
m.load_state_dict(load("weights_seed1"))
m = m.cuda().eval()
torch.manual_seed(5)
x = torch.rand(1, 4, S, S, S, device="cuda")
caps = {}
def hook(name):
    def f(mod, inp, out):
        outs = out if isinstance(out, (tuple, list)) else [out]
        caps.setdefault(cur[0], {})[name] = [o.float().clone() for o in outs if torch.is_tensor(o)]
    return f
cur = ["f32"]
for name, mod in m.named_modules():
    if name and name.count(".") <= 2 and not name.startswith("rdecoder") and not name.startswith("decoders"):
        mod.register_forward_hook(hook(name))
with torch.no_grad():
    cur[0] = "f32"; o32 = m(x, [14], recon=True, valid=True)
    cur[0] = "bf16"; o16 = m(x.bfloat16(), [14], recon=True, valid=True)
def l2(a, b): return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
for name in caps["f32"]:
    if name in caps["bf16"]:
        errs = [l2(a, b) for a, b in zip(caps["bf16"][name], caps["f32"][name])]
        print(f"{name:55s} " + " ".join(f"{e:.3e}" for e in errs) + "  absmax " + " ".join(f"{b.abs().max().item():.2f}" for b in caps["f32"][name]))
print("seg", l2(o16[0].float(), o32[0]), "max", (o16[0].float() - o32[0]).abs().max().item())
for i in range(4):
    print("mu", i, l2(o16[1][0][i].float(), o32[1][0][i]), "lv", l2(o16[1][1][i].float(), o32[1][1][i]))
print("rec", l2(o16[2][0].float(), o32[2][0]))
