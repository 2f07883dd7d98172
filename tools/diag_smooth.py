import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import xlstm_hved_amd as X
from gpu_common import load
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
def smooth_input(n, S, seed):
    g = torch.Generator().manual_seed(seed)
    lo = torch.rand(n, 4, S // 8, S // 8, S // 8, generator=g)
    x = torch.nn.functional.interpolate(lo, size=(S, S, S), mode="trilinear", align_corners=False)
    zz, yy, xx = torch.meshgrid(*[torch.linspace(-1, 1, S)] * 3, indexing="ij")
    brain = ((zz ** 2 + yy ** 2 / 0.8 + xx ** 2 / 0.7) < 0.8).float()
    blob = torch.exp(-((zz - 0.2) ** 2 + (yy + 0.1) ** 2 + (xx - 0.3) ** 2) / 0.05)
    x = (0.6 * x + 0.4 * blob) * brain
    return x
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS)
m.load_state_dict(load("weights_seed1"))
m = m.cuda().eval()
def l2(a, b): return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()
for kind in ("white", "smooth"):
    x = torch.rand(1, 4, S, S, S, generator=torch.Generator().manual_seed(5)) if kind == "white" else smooth_input(1, S, 5)
    x = x.cuda()
    with torch.no_grad():
        s32, (mu32, _), r32 = m(x, [14], recon=True, valid=True)
        s16, (mu16, _), r16 = m(x.bfloat16(), [14], recon=True, valid=True)
    tgt = (s32 > 0.5).float(); p = (s16.float() > 0.5).float()
    dice = ((2 * (p * tgt).sum((2, 3, 4)) + 1e-6) / ((p + tgt).sum((2, 3, 4)) + 1e-6)).flatten().tolist()
    frac = tgt.mean((0, 2, 3, 4)).tolist()
    near = ((s32 - 0.5).abs() < 0.05).float().mean().item()
    print(f"{kind} {S}^3: seg L2 {l2(s16, s32):.3e} max {(s16.float()-s32).abs().max().item():.3f} rec L2 {l2(r16[0], r32[0]):.3e} "
          f"mu L2 {[round(l2(a,b),4) for a,b in zip(mu16,mu32)]} dice(bf16 vs fp32 mask) {[round(d,5) for d in dice]} fg frac {[round(f,3) for f in frac]} near0.5 {near:.3f}")
