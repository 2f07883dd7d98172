"""Scan of the full-row weight gradient's launch-plan cost factors (XH_Q5_F64 / F2 / F4 in the environment of a child process each):
the step's batch of 128^3 + 64^3 problems, time per batch."""
import sys, os, subprocess, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import xlstm_hved_amd as X
    from tools.microbench_conv import bench
    ops = X.ops
    allp = [(8, 8, 2, 128), (16, 16, 4, 128), (4, 4, 1, 128), (4, 4, 1, 128), (16, 16, 4, 128), (16, 16, 4, 128), (12, 4, 1, 128), (12, 4, 1, 128),
            (8, 8, 8, 64), (20, 40, 5, 64), (20, 20, 5, 64), (16, 16, 2, 64), (8, 8, 1, 64), (8, 8, 1, 64), (24, 8, 1, 64), (24, 8, 1, 64)]
    data = []
    for cin, cout, g, S in allp:
        x = torch.randn(1, cin, S, S, S, device="cuda").bfloat16(); dy = torch.randn(1, cout, S, S, S, device="cuda").bfloat16()
        nw = g if g <= 4 else 1
        data.append((x, dy, (torch.rand(1, cin, device="cuda") + 0.5, torch.randn(1, cin, device="cuda"), 0.01), g,
                     [torch.zeros(cout // nw, cin // g, 3, 3, 3, device="cuda") for _ in range(nw)],
                     [torch.zeros(cout // nw, device="cuda") for _ in range(nw)]))
    def batch():
        ops.set_wgrad_defer(True)
        for x, dy, pre, g, dws, dbs in data:
            ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, groups=g, pre=pre, side=True)
        ops.join_wgrad_stream()
        ops.set_wgrad_defer(False)
    ts = sorted(bench(batch, n=8, reps=8) for _ in range(3))
    print(f"{ts[0]:.1f} {ts[1]:.1f} {ts[2]:.1f}", flush=True)
    sys.exit(0)
for f64, f2, f4 in itertools.product((0.55, 0.7, 0.85), (1.2, 1.45, 1.7), (1.5, 1.9, 2.3)):
    env = dict(os.environ, XH_Q5_F64=str(f64), XH_Q5_F2=str(f2), XH_Q5_F4=str(f4))
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    print(f"f64 {f64} f2 {f2} f4 {f4}: {r.stdout.strip()} {r.stderr.strip()[-200:] if r.returncode else ''}", flush=True)
