import sys, torch, collections
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch.nn.functional as F
import xlstm_hved_amd as X
from xlstm_hved_amd import functional as Fn, ops
torch.manual_seed(19)
sp = (32, 64, 128)
n, cin, cout = 1, 4, 4
x = (torch.randn((n, cin) + sp) * 1.5 + 0.3).cuda()
w = (torch.randn(cout, cin, 3, 3, 3) * 0.1).cuda(); b = torch.randn(cout).cuda()
ref = F.conv3d(F.leaky_relu(F.instance_norm(x), 0.01), w, b, padding=1)
ops.set_fp32_mfma(True)
for rep in range(2):
    y, st = Fn.in_lrelu_conv(x, None, [w], [b], 1, 1, out_stats=True)
    torch.cuda.synchronize()
    bad = ((y - ref).abs() > 1e-2).nonzero().cpu()
    h = collections.Counter((int(c), int(d) % 2, int(hh) % 8, int(ww) % 32) for _, c, d, hh, ww in bad.tolist())
    print(len(bad), sorted(h.items(), key=lambda kv: -kv[1])[:12])
    print(y[0, :, 3, 13, 32:48].cpu())
X.ops.set_fp32_mfma(False)
