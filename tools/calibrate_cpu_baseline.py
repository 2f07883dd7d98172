#!/usr/bin/env python3
"""before the CPU restatement (oracle/xlstm_hved_oracle.py) is trusted as the `cpu_baseline` of bench.py on the GPU
box -- where the reference sources do not exist -- its wall time must be within +-10 % of the REAL reference on the same host
and thread count.  Build container only (imports /root/reference through tools/ref_shim.py).

    python tools/calibrate_cpu_baseline.py [--sizes 64 128] [--threads 8] [--out profiles/cpu_baseline_calibration.json]

Same weights (seeded init_weights), same input, same loss, train mode, subset [14], recon=True, fp32, forward +
backward; best of `--reps` runs each, alternating reference / oracle so that both see the same machine state."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_shim  # noqa: E402
import xlstm_hved_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[64, 128])
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "cpu_baseline_calibration.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    ns = ref_shim.load_reference()
    model = ref_shim.build_reference_model(ns, seed=1).train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    R = ns.RA_HVED
    res = {"threads": a.threads, "torch": torch.__version__, "host_cpus": os.cpu_count(), "cases": []}
    for s in a.sizes:
        g = torch.Generator().manual_seed(1)
        x = torch.rand(1, 4, s, s, s, generator=g)
        eps = [torch.randn(1, 2 ** l, s >> (l + 1), s >> (l + 1), s >> (l + 1), generator=g) for l in range(4)]

        def run_ref():
            it = iter(eps)
            orig = R.reparametrize
            R.reparametrize = lambda mu, logvar, valid=False: mu if valid else next(it) * torch.exp(0.5 * logvar) + mu   # RA_HVED.py:741-747 with injected noise
            try:
                model.zero_grad(set_to_none=True)
                t0 = time.perf_counter()
                seg, (mu, lv), rec = model(x, [14], recon=True)
                loss = seg.mean() + rec[0].mean()
                for m_, l_ in zip(mu, lv):
                    loss = loss + m_.mean() + l_.mean()
                loss.backward()
                return time.perf_counter() - t0
            finally:
                R.reparametrize = orig

        def run_oracle():
            sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd0.items()}
            t0 = time.perf_counter()
            prob, _, mu, lv, rec = O.xlstm_hved_forward(sd, x, 14, eps_list=eps, training=True, reference_cost=True)
            O.bench_loss(prob, mu, lv, rec).backward()
            return time.perf_counter() - t0
        run_ref() if s <= 64 else None          # thread-pool / allocator warm-up
        tr, to = [], []
        for _ in range(a.reps):
            tr.append(run_ref())
            to.append(run_oracle())
        case = {"size": s, "reference_s": min(tr), "oracle_s": min(to), "oracle_over_reference": min(to) / min(tr),
                "reference_voxels_per_s": s ** 3 / min(tr), "oracle_voxels_per_s": s ** 3 / min(to)}
        print(case, flush=True)
        res["cases"].append(case)
    res["within_10_percent"] = all(abs(c["oracle_over_reference"] - 1.0) <= 0.10 for c in res["cases"])
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", a.out, "within 10 %:", res["within_10_percent"])


if __name__ == "__main__":
    main()
