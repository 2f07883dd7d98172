"""Call log of one eager fwd+bwd step of bench.py's workload: every xlstm_hved_amd.ops function in launch order with the
shapes of its tensor arguments (and, for the convs, the kernel instance that ran) -- the key to tools/dump_step.py's launch list.
usage: python tools/trace_ops.py [size] > gpurun_out/ops_trace.txt"""
import os, sys, types
if len(sys.argv) > 1 and not sys.argv[1].isdigit():
    print(__doc__)
    raise SystemExit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from bench import bench_loss
ops = X.ops

if len(sys.argv) > 1 and not sys.argv[1].isdigit():
    print(__doc__)
    raise SystemExit(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
LOG = []
ON = [False]


def shp(a):
    if isinstance(a, torch.Tensor):
        return "x".join(map(str, a.shape)) + ("" if a.dtype in (torch.bfloat16, torch.float16) else f":{str(a.dtype)[6:]}")
    if isinstance(a, (list, tuple)):
        return "[" + ",".join(shp(t) for t in a) + "]"
    if a is None:
        return "-"
    return repr(a) if isinstance(a, (int, float, bool, str)) else type(a).__name__


def wrap(name, fn):
    def w(*args, **kw):
        r = fn(*args, **kw)
        if ON[0]:
            line = f"{name}({', '.join(shp(a) for a in args)}" + "".join(f", {k}={shp(v)}" for k, v in kw.items()) + ")"
            if name.startswith("conv"):
                line += f"  -> {ops.last_conv_kernel()}"
            LOG.append(line)
        return r
    return w


SKIP = {"new_like", "zeros_f32", "fan_block", "last_conv_kernel", "red_arena", "red_arena_reset", "join_wgrad_stream"}
for k, v in list(vars(ops).items()):
    if isinstance(v, types.FunctionType) and not k.startswith("_") and k not in SKIP and v.__module__ == ops.__name__:
        setattr(ops, k, wrap(k, v))

torch.manual_seed(1)
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS); m.apply(X.init_weights); m = m.cuda().train()
MODE = os.environ.get("XH_TRACE_MODE", "bf16")          # bf16 | fp16 | fp32 | fp32_mfma
x = torch.rand(1, 4, S, S, S, device="cuda").to({"bf16": torch.bfloat16, "fp16": torch.float16}.get(MODE, torch.float32))
ops.set_fp32_mfma(MODE == "fp32_mfma")
grads = X.parallel.FlatGrads(list(m.parameters()))
ops.set_wgrad_defer(True)
for it in range(2):
    ON[0] = it == 1
    grads.zero()
    seg, (mu, lv), rec = m(x, [14], recon=True)
    if ON[0]:
        LOG.append("---- backward ----")
    bench_loss(seg, mu, lv, rec[0]).backward()
    ops.join_wgrad_stream()
torch.cuda.synchronize()
print("\n".join(LOG))
