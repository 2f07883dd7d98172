"""64 <- 128 stride-2 ks = 4 data gradient of the discriminator @127^3 with / without the mask + bias + channel-sum epilogue of the
training step, source-block kernel vs gather kernel (xh_set_option(14, 16384)); operands rotated (XH_ROT sets) so they come from HBM."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import xlstm_hved_amd as X
from xlstm_hved_amd import disc as D
from microbench_disc import bench
ROT = int(os.environ.get("XH_ROT", "3"))
dt = torch.bfloat16
KS, s, cin, cout, sp = 4, 2, 64, 128, 127
so = (sp + 2 - KS) // s + 1
w = torch.randn(cout, cin, KS, KS, KS, device="cuda") * 0.05
wpt = D._pack(w, 1, cout, cin, dt)
sets = [(torch.randn(1, so, so, so, cout, device="cuda").to(dt), torch.randn(1, sp, sp, sp, cin, device="cuda").to(dt)) for _ in range(ROT)]
bias = torch.randn(cin, device="cuda")
red = torch.zeros(1, cin, 2, dtype=torch.float64, device="cuda")
tick = [0]
lib = X._lib.load()
for opt, nm in ((0, "source-block"), (16384, "gather")):
    lib.xh_set_option(14, opt)
    def plain():
        tick[0] += 1
        dy, _ = sets[tick[0] % ROT]
        D._conv(dy, wpt, None, 1, s, 1, (so,) * 3, (sp,) * 3, cout, cin, ks=KS)
    def epi():
        tick[0] += 1
        dy, m = sets[tick[0] % ROT]
        D._conv(dy, wpt, bias, 1, s, 1, (so,) * 3, (sp,) * 3, cout, cin, ks=KS, mask=m, red=red)
    print(f"{nm}: plain {bench(plain):.1f} us, with mask + bias + sums {bench(epi):.1f} us", flush=True)
lib.xh_set_option(14, 0)
