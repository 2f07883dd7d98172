"""Times the k3 conv forward (vector vs MFMA kernel) at the 128^3 / 64^3 shapes of the network.
Calls are captured into a hipGraph (20 per replay) so the numbers are GPU time, not Python launch overhead."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
ops = X.ops; L = X._lib

def bench(fn, n=20, reps=5):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3

if __name__ == "__main__":
    abl = "--abl" in sys.argv
    shapes = [(16, 16, 4, 128), (12, 4, 1, 128), (4, 4, 1, 128), (16, 32, 4, 64), (24, 8, 1, 64), (8, 8, 1, 64),
              (16, 16, 1, 32), (48, 16, 1, 32), (32, 32, 4, 32)]
    for (cin, cout, g, S) in shapes[2:3] if abl else shapes:
        x = torch.randn(1, cin, S, S, S, device="cuda").bfloat16()
        ws = [torch.randn(cout // g, cin // g, 3, 3, 3, device="cuda") * 0.1 for _ in range(g)]
        bs = [torch.randn(cout // g, device="cuda") for _ in range(g)]
        sc = torch.rand(1, cin, device="cuda") + 0.5; sh = torch.randn(1, cin, device="cuda")
        red = torch.zeros(1, cout, 2, dtype=torch.float64, device="cuda")
        call = lambda: ops.conv3d(x, None, ws, bs, k=3, cout=cout, groups=g, pre=(sc, sh, 0.01), epi=2, red=red)
        nbytes = (cin + cout) * S ** 3 * 2
        flops = 2 * cout * S ** 3 * 27 * cin / g
        if abl:
            for mask, name in [(0, "full"), (2, "no loads"), (4, "no mfma loop"), (8, "no stores"), (14, "nothing"), (30, "nothing-noLDSstore"), (62, "nothing-nostore-nopad"), (2048, "no atomics"), (2048 + 62, "nothing, no atomics")]:
                L.load().xh_set_option(1, mask)
                print(f"{cin}->{cout} g{g} @{S}: {name:16s} {bench(call):7.1f} us")
            L.load().xh_set_option(1, 0)
        else:
            res = {}
            for mfma in (False, True):
                ops.set_mfma(mfma)
                res[mfma] = bench(call)
            ops.set_mfma(True)
            extra = ""
            if "--occ" in sys.argv:
                for occ in (0, 1):
                    for wgs in (512, 1024, 2048, 4096):
                        L.load().xh_set_option(3, wgs); L.load().xh_set_option(4, occ)
                        extra += f", occ{occ}/wg{wgs}: {bench(call):.1f}"
                L.load().xh_set_option(3, 512); L.load().xh_set_option(4, 0)
            if "--th4" in sys.argv:
                L.load().xh_set_option(1, 64)
                extra = f", 4-row tiles: {bench(call):.1f} us"
                L.load().xh_set_option(1, 0)
            print(f"{cin}->{cout} g{g} @{S}^3: vector {res[False]:.1f} us, mfma {res[True]:.1f} us "
                  f"({nbytes / res[True] / 1e3:.0f} GB/s algorithmic, {flops / res[True] / 1e6:.1f} TFLOP/s useful){extra}")

if "--wgrad" in sys.argv:
    caps = [(0, "cp<=16")] + ([(128, "cp<=8"), (256, "cp=4")] if "--cp" in sys.argv else [])
    for (cin, cout, g, S) in [(16, 16, 4, 128), (12, 4, 1, 128), (4, 4, 1, 128), (16, 32, 4, 64), (24, 8, 1, 64), (8, 8, 1, 64),
                              (16, 16, 1, 32), (48, 16, 1, 32), (32, 32, 4, 32)]:
        x = torch.randn(1, cin, S, S, S, device="cuda").bfloat16()
        dy = torch.randn(1, cout, S, S, S, device="cuda").bfloat16()
        sc = torch.rand(1, cin, device="cuda") + 0.5; sh = torch.randn(1, cin, device="cuda")
        dws = [torch.zeros(cout // g, cin // g, 3, 3, 3, device="cuda") for _ in range(g)]
        dbs = [torch.zeros(cout // g, device="cuda") for _ in range(g)]
        call = lambda: ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, groups=g, pre=(sc, sh, 0.01))
        res = {}
        for mfma in (False, True):
            ops.set_mfma(mfma)
            res[mfma] = bench(call)
        ops.set_mfma(True)
        flops = 2 * cout * S ** 3 * 27 * cin / g
        extra = ""
        for mask, nm in caps[1:]:
            L.load().xh_set_option(1, mask)
            extra += f", {nm}: {bench(call):.1f} us"
        L.load().xh_set_option(1, 0)
        print(f"wgrad {cin}->{cout} g{g} @{S}^3: vector {res[False]:.1f} us, mfma {res[True]:.1f} us ({flops / res[True] / 1e6:.1f} TFLOP/s useful){extra}")
