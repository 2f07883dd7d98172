"""Diagnostic: run XLSTM_HVED in fp32, bf16 and fp16 storage on the GPU; per-stage relative L2 deviation and value ranges
(absmax / smallest per-channel std) to find range problems of a 16-bit format."""
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
        caps.setdefault(cur[0], {})[name] = [o.float().clone() for o in outs if torch.is_tensor(o) and o.dim() == 5]
    return f
cur = ["f32"]
for name, mod in m.named_modules():
    if name and name.count(".") <= 3 and not name.startswith("rdecoder") and not name.startswith("decoders"):
        mod.register_forward_hook(hook(name))
outs = {}
with torch.no_grad():
    for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("f16", torch.float16)):
        cur[0] = tag
        outs[tag] = m(x.to(dt), [14], recon=True, valid=True)
def l2(a, b): return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
for name in caps["f32"]:
    for k, b in enumerate(caps["f32"][name]):
        e16 = l2(caps["bf16"][name][k], b); eh = l2(caps["f16"][name][k], b)
        std = b.flatten(2).std(-1)
        print(f"{name:50s} bf16 {e16:.2e} f16 {eh:.2e}  absmax {b.abs().max().item():.3g} min ch-std {std.min().item():.3g} max ch-std {std.max().item():.3g}"
              + ("   <<<" if eh > e16 else ""))
o32 = outs["f32"]
for tag in ("bf16", "f16"):
    o = outs[tag]
    print(tag, "seg", l2(o[0].float(), o32[0]), "rec", l2(o[2][0].float(), o32[2][0]))
    for i in range(4):
        print("  mu", i, l2(o[1][0][i].float(), o32[1][0][i]), "lv", l2(o[1][1][i].float(), o32[1][1][i]),
              "absmax mu", o32[1][0][i].abs().max().item(), "lv", o32[1][1][i].abs().max().item())
