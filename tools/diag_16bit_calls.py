"""Diagnostic: record the output of every functional stage call (in call order) in fp32 / bf16 / fp16 storage."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import xlstm_hved_amd as X
from gpu_common import load
Fn = X.functional
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS)
m.load_state_dict(load("weights_seed1"))
m = m.cuda().eval()
torch.manual_seed(5)
x = torch.rand(1, 4, S, S, S, device="cuda")
rec = {}
cur = [None]
def wrap_fn(name):
    orig = getattr(Fn, name)
    def f(*a, **k):
        out = orig(*a, **k)
        o = out[0] if isinstance(out, tuple) else out
        rec[cur[0]].append((name, tuple(o.shape), o.float().clone(), X.ops.last_conv_kernel()))
        return out
    setattr(Fn, name, f)
for n_ in ("in_lrelu_conv", "conv"):
    wrap_fn(n_)
for cls in ("MaxPool2", "Gate", "PoE", "Upsample", "ConvInLrelu", "ViL", "DuSE", "SkipReturnAttention", "GateCat", "ChannelPool2"):
    c = getattr(Fn, cls)
    orig = c.apply
    def mk(orig, cls):
        def f(*a, **k):
            out = orig(*a, **k)
            o = out[0] if isinstance(out, tuple) else out
            rec[cur[0]].append((cls, tuple(o.shape), o.float().clone(), ""))
            return out
        return f
    c.apply = mk(orig, cls)
with torch.no_grad():
    for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("f16", torch.float16)):
        cur[0] = tag; rec[tag] = []
        m(x.to(dt), [14], recon=True, valid=True)
def l2(a, b): return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
for i, (name, shape, t, kern) in enumerate(rec["f32"]):
    eb, eh = l2(rec["bf16"][i][2], t), l2(rec["f16"][i][2], t)
    print(f"{i:3d} {name:20s} {str(shape):26s} bf16 {eb:.2e} f16 {eh:.2e} absmax {t.abs().max().item():.3g} {'<<<' if eh > eb else ''}  {rec['f16'][i][3]}")
