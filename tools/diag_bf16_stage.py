import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
Fn = X.functional
torch.manual_seed(0)
def l2(a, b): return ((a.float() - b.float()).norm() / b.float().norm()).item()
for cin, cout, sp in [(4, 4, 32), (16, 16, 16)]:
    x = (torch.randn(1, cin, sp, sp, sp) * 1.0 + torch.randn(1, cin, 1, 1, 1)).cuda()
    w = (torch.randn(cout, cin, 3, 3, 3) * (2 / (cin * 27)) ** 0.5).cuda()
    b = torch.randn(cout).cuda()
    y32 = Fn.in_lrelu_conv(x, None, [w], [b], 1, 1)
    xr = x.bfloat16()
    y32r = Fn.in_lrelu_conv(xr.float(), None, [w], [b], 1, 1)
    y16 = Fn.in_lrelu_conv(xr, None, [w], [b], 1, 1)
    ref = torch.nn.functional.conv3d(torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(x), 0.01), w, b, padding=1)
    print(f"cin {cin} cout {cout} sp {sp}: fp32 kernel vs torch {l2(y32, ref):.2e}; input-rounded fp32 vs fp32 {l2(y32r, y32):.2e}; "
          f"bf16 kernel vs input-rounded fp32 {l2(y16, y32r):.2e}; bf16 vs fp32 {l2(y16, y32):.2e}; "
          f"centered: {l2(y16.float()-y32.mean((2,3,4),keepdim=True), y32-y32.mean((2,3,4),keepdim=True)):.2e}  mean/std {(y32.mean((2,3,4)).abs()/y32.std((2,3,4))).mean().item():.2f}")
