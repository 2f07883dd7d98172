"""Times the discriminator's implicit-GEMM convolutions (csrc/dconv.hip) at the shapes of a 128^3 patch: forward, data
gradient, weight gradient per layer (TFLOP/s of useful work), the whole Discriminator forward + backward, in hipGraph replays."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from xlstm_hved_amd import disc as D

def bench(fn, n=5, reps=3):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3

dt = torch.bfloat16 if "--fp16" not in sys.argv else torch.float16
S = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
KS = 3 if "--k3" in sys.argv else 4                     # train.py:146 builds ks=4
NB = 2 if "--n2" in sys.argv else 1                     # the D update of the training step runs fake + real as one batch of 2
layers, sp = [], S
for cin, cout, s in [(8, 64, 1), (64, 128, 2), (128, 256, 2), (256, 512, 2), (512, 1, 1)]:
    layers.append((cin, cout, s, sp))
    sp = (sp + 2 - KS) // s + 1
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
totfl = 0.0
for cin, cout, s, sp in layers:
    so = (sp + 2 - KS) // s + 1
    x = torch.randn(NB, sp, sp, sp, cin, device="cuda").to(dt)
    w = torch.randn(cout, 7 if cin == 8 else cin, KS, KS, KS, device="cuda") * 0.05
    wp = D._pack(w, 2 if cin == 8 else 0, cout, cin, dt)
    red = torch.zeros(NB, cout, 2, dtype=torch.float64, device="cuda")
    lib = X._lib.load()
    lib.xh_set_option(14, int(os.environ.get("XH_DCFG", "0")))
    lib.xh_set_option(15, int(os.environ.get("XH_DBIG", "1024")))
    lib.xh_set_option(5, 1)
    t_f1 = bench(lambda: D._conv(x, wp, None, 0, s, NB, (sp,) * 3, (so,) * 3, cin, cout, red=red if cout > 1 and cin > 8 else None, ks=KS))
    lib.xh_set_option(5, 2)
    t_f = bench(lambda: D._conv(x, wp, None, 0, s, NB, (sp,) * 3, (so,) * 3, cin, cout, red=red if cout > 1 and cin > 8 else None, ks=KS))
    cop = max(32, cout)
    dy = torch.randn(NB, so, so, so, cop, device="cuda").to(dt)
    wpt = D._pack(w, 1, cop, cin, dt)
    t_d = bench(lambda: D._conv(dy, wpt, None, 1, s, NB, (so,) * 3, (sp,) * 3, cop, cin, ks=KS))
    co8 = max(8, cout) if cout >= 8 else 32
    dy8 = torch.randn(NB, so, so, so, co8, device="cuda").to(dt)
    t_w = bench(lambda: D._wgrad(x, dy8, s, NB, (sp,) * 3, (so,) * 3, cin, co8, ks=KS))
    fl = 2.0 * NB * cout * (7 if cin == 8 else cin) * KS ** 3 * so ** 3
    totfl += fl
    print(f"{cin:3d}->{cout:3d} s{s} @{sp}^3: fwd[K32] {t_f1:7.1f} us fwd {t_f:8.1f} us ({fl / t_f / 1e6:6.1f} TF/s)  dgrad {t_d:8.1f} us ({fl / t_d / 1e6:6.1f})  wgrad {t_w:8.1f} us ({fl / t_w / 1e6:6.1f})   [{fl / 1e9:.1f} GFLOP]")
    tot["fwd"] += t_f; tot["dgrad"] += t_d; tot["wgrad"] += t_w
print("ks=%d batch %d: %.1f GFLOP per pass; sum of conv launches: fwd %.2f ms (%.0f TF/s), dgrad %.2f ms, wgrad %.2f ms" % (
    KS, NB, totfl / 1e9, tot["fwd"] / 1e3, totfl / tot["fwd"] / 1e6, tot["dgrad"] / 1e3, tot["wgrad"] / 1e3))
m = X.Discriminator(in_channels=7, ks=KS, strides=[1, 2, 2, 2]); m.apply(X.init_weights); m = m.cuda()
fg = X.parallel.FlatGrads(m.parameters())
xin = torch.randn(NB, 7, S, S, S, device="cuda").to(dt).requires_grad_(True)
def fb():
    fg.zero()
    y = m(xin)
    y.float().mean().backward()
print("Discriminator forward: %.2f ms" % (bench(lambda: m(xin.detach()), n=3) / 1e3))
print("Discriminator forward + backward (incl. input gradient): %.2f ms" % (bench(fb, n=3) / 1e3))
