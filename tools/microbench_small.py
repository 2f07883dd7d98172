"""Times the small streaming kernels of the step (depthwise k3 forward / weight gradient, k1 forward / weight gradient)
at the network's shapes while sweeping their launch-plan tunables (xh_set_option keys 6..8).  Calls are captured into a
hipGraph (20 per replay), so the numbers are GPU time per call."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from tools.microbench_conv import bench
ops = X.ops; L = X._lib


def opt(k, v):
    L.load().xh_set_option(k, v)


def dw_case(C, S, dt):
    x = torch.randn(1, C, S, S, S, device="cuda").to(dt)
    dy = torch.randn(1, C, S, S, S, device="cuda").to(dt)
    w = [torch.randn(C, 1, 3, 3, 3, device="cuda") * 0.1]
    sc = torch.rand(1, C, device="cuda") + 0.5; sh = torch.randn(1, C, device="cuda")
    red = torch.zeros(1, C, 2, dtype=torch.float64, device="cuda")
    dw = [torch.zeros_like(w[0])]
    fwd = lambda: ops.conv3d(x, None, w, None, k=3, cout=C, groups=C, pre=(sc, sh, 0.01), epi=2, red=red)
    wg = lambda: ops.conv3d_wgrad(x, None, dy, dw, None, k=3, groups=C, pre=(sc, sh, 0.01))
    return fwd, wg, 2 * C * S ** 3 * x.element_size()


def c1_case(cin, cout, S, dt):
    x = torch.randn(1, cin, S, S, S, device="cuda").to(dt)
    dy = torch.randn(1, cout, S, S, S, device="cuda").to(dt)
    w = [torch.randn(cout, cin, 1, 1, 1, device="cuda") * 0.1]
    b = [torch.randn(cout, device="cuda")]
    dw = [torch.zeros_like(w[0])]; db = [torch.zeros_like(b[0])]
    red = torch.zeros(1, cout, 2, dtype=torch.float64, device="cuda")
    fwd = lambda: ops.conv3d(x, None, w, b, k=1, cout=cout, epi=2, red=red)
    wg = lambda: ops.conv3d_wgrad(x, None, dy, dw, db, k=1)
    return fwd, wg, (cin + cout) * S ** 3 * x.element_size()


if __name__ == "__main__" and "--reducers" not in sys.argv and "--gates" not in sys.argv:
    dt = torch.bfloat16
    print("== depthwise k3 (forward, wgrad): min planes per segment (0 = built-in rule)")
    for C, S in [(4, 128), (8, 64), (16, 32)]:
        fwd, wg, nb = dw_case(C, S, dt)
        for minsd in (0, 8, 4, 2):
            opt(6, minsd)
            tf, tw = bench(fwd), bench(wg)
            print(f"dw3 C={C:3d} @{S:3d} minsd={minsd}: fwd {tf:6.1f} us ({nb / tf / 1e3:6.0f} GB/s)  wgrad {tw:6.1f} us")
    opt(6, 0)
    print("== k1 forward; weight gradient: workgroup target")
    for cin, cout, S in [(4, 4, 128), (16, 4, 128), (8, 8, 64), (16, 16, 32), (32, 32, 16), (64, 64, 8), (128, 64, 8)]:
        fwd, wg, nb = c1_case(cin, cout, S, dt)
        line = f"1x1 {cin}->{cout} @{S}: fwd {bench(fwd):6.1f} us ({ops.last_conv_kernel()}); wgrad"
        for tgt in (128, 256, 384, 512, 768):
            opt(8, tgt)
            line += f"  wg{tgt} {bench(wg):6.1f}"
        print(line + f" us   ({nb / 1e6:.1f} MB)")
    opt(8, 0)
    print("== k3 weight gradient on narrow volumes: vector vs MFMA kernel")
    for cin, cout, g, S in [(64, 128, 4, 16), (128, 64, 4, 16), (128, 256, 4, 8), (256, 128, 4, 8), (32, 32, 1, 16)]:
        x = torch.randn(1, cin, S, S, S, device="cuda").to(dt)
        dy = torch.randn(1, cout, S, S, S, device="cuda").to(dt)
        dws = [torch.zeros(cout // g, cin // g, 3, 3, 3, device="cuda") for _ in range(g)]
        dbs = [torch.zeros(cout // g, device="cuda") for _ in range(g)]
        sc = torch.rand(1, cin, device="cuda") + 0.5; sh = torch.randn(1, cin, device="cuda")
        wg = lambda: ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, groups=g, pre=(sc, sh, 0.01))
        res = []
        for mfma in (False, True):
            ops.set_mfma(mfma)
            res.append((bench(wg), ops.last_conv_kernel()))
        ops.set_mfma(True)
        print(f"k3 wgrad {cin}->{cout} g{g} @{S}: vector {res[0][0]:6.1f} us ({res[0][1]}), MFMA {res[1][0]:6.1f} us ({res[1][1]})")
    sys.stdout.flush()


def reducers():
    """moments / act_bwd_reduce / pair_sums at the step's shapes vs the reducing kernels' workgroup target (key 9)."""
    dt = torch.bfloat16
    for C, S in [(4, 128), (16, 128), (8, 64), (32, 64), (16, 32)]:
        x = torch.randn(1, C, S, S, S, device="cuda").to(dt)
        dy = torch.randn(1, C, S, S, S, device="cuda").to(dt)
        sc = torch.rand(1, C, device="cuda") + 0.5; sh = torch.randn(1, C, device="cuda")
        red = torch.zeros(1, C, 2, dtype=torch.float64, device="cuda")
        line = f"C={C:3d} @{S:3d} ({x.numel() * 2 / 1e6:5.1f} MB/tensor):"
        for tgt in (2048, 1024, 512, 256):
            opt(9, tgt)
            tm = bench(lambda: ops.moments(x, red))
            ta = bench(lambda: ops.act_bwd_reduce(dy, x, sc, sh, 0.01))
            line += f"  wg{tgt}: moments {tm:5.1f} act_bwd_reduce {ta:5.1f}"
        print(line)
    opt(9, 0)


if __name__ == "__main__" and "--reducers" in sys.argv:
    reducers()


def gates():
    """gate_bwd fused (dx and ds in one lane-per-run kernel) vs split (row kernel for dx + channel-sum kernel for ds)."""
    import ctypes as C
    dt = torch.bfloat16
    lib = L.load()
    for Cc, S in [(4, 128), (16, 128), (8, 64), (16, 32)]:
        x = torch.randn(1, Cc, S, S, S, device="cuda").to(dt)
        dy = torch.randn(1, Cc, S, S, S, device="cuda").to(dt)
        s = torch.rand(1, 1, S, S, S, device="cuda").to(dt)
        dx = torch.empty_like(x); ds = torch.empty_like(s)
        st = lambda: torch.cuda.current_stream().cuda_stream
        dhw = S ** 3

        def fused():
            lib.xh_gate_bwd(st(), L.XH_BF16, x.data_ptr(), Cc * dhw, s.data_ptr(), dhw, dy.data_ptr(), Cc * dhw, dx.data_ptr(), Cc * dhw,
                            ds.data_ptr(), dhw, 1, Cc, dhw, 0, 0)

        def split():
            lib.xh_gate_fwd(st(), L.XH_BF16, dy.data_ptr(), Cc * dhw, s.data_ptr(), dhw, dx.data_ptr(), Cc * dhw, 1, Cc, dhw)
            lib.xh_gate_bwd(st(), L.XH_BF16, x.data_ptr(), Cc * dhw, s.data_ptr(), dhw, dy.data_ptr(), Cc * dhw, None, 0,
                            ds.data_ptr(), dhw, 1, Cc, dhw, 0, 0)
        mb = (3 * Cc + 2) * dhw * 2 / 1e6
        tf, ts = bench(fused), bench(split)
        print(f"gate_bwd C={Cc:2d} @{S:3d} ({mb:6.1f} MB): fused {tf:6.1f} us ({mb / tf * 1e-3:5.2f} TB/s)   split {ts:6.1f} us")


if __name__ == "__main__" and "--gates" in sys.argv:
    gates()


def appliers():
    """in_bwd_apply / affine_act (the two-pass norm's streaming halves) at the step's big shapes vs the row kernels' workgroup
    target (key 27)."""
    dt = torch.bfloat16
    for C, S in [(8, 128), (4, 128), (16, 64), (8, 64)]:
        x = torch.randn(1, C, S, S, S, device="cuda").to(dt)
        dy = torch.randn(1, C, S, S, S, device="cuda").to(dt)
        out = torch.empty_like(x)
        sc = torch.rand(1, C, device="cuda") + 0.5; sh = torch.randn(1, C, device="cuda")
        red = torch.randn(1, C, 2, dtype=torch.float64, device="cuda")
        mean = torch.randn(1, C, device="cuda"); rstd = torch.rand(1, C, device="cuda") + 0.5
        mb = x.numel() * 2 / 1e6
        line = f"C={C:3d} @{S:3d} ({mb:5.1f} MB/tensor):"
        for tgt in (1024, 2048, 4096, 8192, 16384):
            opt(27, tgt)
            ta = bench(lambda: ops.in_bwd_apply(dy, x, red, mean, rstd, have_g=False, sc=sc, sh=sh, out=out))
            tb = bench(lambda: ops.affine_act(x, sc, sh, ops.ACT_LRELU, out=out)) if hasattr(ops, "affine_act") else 0.0
            line += f"  wg{tgt}: apply {ta:5.1f} ({3 * mb / ta / 1e3:4.2f} TB/s) affine {tb:5.1f}"
        print(line)
    opt(27, 2048)


if __name__ == "__main__" and "--appliers" in sys.argv:
    appliers()
