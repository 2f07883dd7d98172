import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import xlstm_hved_amd as X
from gpu_common import load
Fn = X.functional
S = 64
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS)
m.load_state_dict(load("weights_seed1"))
m = m.cuda().eval()
torch.manual_seed(5)
x = torch.rand(1, 4, S, S, S, device="cuda")
def l2(a, b): return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30)).item()
with torch.no_grad():
    iw = [b[0].weight for b in m.init_blocks]; ib = [b[0].bias for b in m.init_blocks]
    w, b = m._stream_weights(0, "SingleConv1")
    X0, st0 = Fn.conv(x, iw, ib, groups=4, out_stats=True)
    y32 = Fn.in_lrelu_conv(X0, None, w, b, 1, 4)
    ref = F.conv3d(F.leaky_relu(F.instance_norm(X0, eps=1e-5), 0.01), torch.cat(w, 0), torch.cat(b, 0), padding=1, groups=4)
    print("fp32 kernel vs torch", l2(y32, ref))
    print("init weights", [float(t.flatten()[0]) for t in iw][:4], "per-channel mean/std of X0:",
          (X0.flatten(2).mean(-1) / X0.flatten(2).std(-1)).flatten().tolist())
    for dt in (torch.bfloat16, torch.float16):
        Xr = X0.to(dt)
        refr = F.conv3d(F.leaky_relu(F.instance_norm(Xr.float(), eps=1e-5), 0.01), torch.cat(w, 0), torch.cat(b, 0), padding=1, groups=4)
        print(dt, "torch fp32 on rounded input vs fp32:", l2(refr, ref))
        y = Fn.in_lrelu_conv(Xr, None, w, b, 1, 4)
        print(dt, "  HIP mfma, own moments: vs rounded-input torch", l2(y, refr), "vs fp32", l2(y, ref), X.ops.last_conv_kernel())
        xs, st = Fn.conv(x.to(dt), iw, ib, groups=4, out_stats=True)
        y = Fn.in_lrelu_conv(xs, None, w, b, 1, 4, in_stats=st)
        print(dt, "  HIP mfma, producer stats:", l2(y, ref), " X0 err", l2(xs, X0))
        X.ops.set_mfma(False)
        y = Fn.in_lrelu_conv(Xr, None, w, b, 1, 4)
        print(dt, "  HIP vector kernel:", l2(y, refr), l2(y, ref), X.ops.last_conv_kernel())
        X.ops.set_mfma(True)
        # per-channel error of the mfma result
        y = Fn.in_lrelu_conv(Xr, None, w, b, 1, 4)
        pc = ((y.float() - refr).flatten(2).norm(dim=-1) / refr.flatten(2).norm(dim=-1)).flatten().tolist()
        print(dt, "  per-channel:", " ".join(f"{v:.1e}" for v in pc))
