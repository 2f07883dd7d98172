"""Per-stage deviation of the HIP forward from the fp64 oracle (taps) -- locates where an error enters."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from gpu_common import load
import xlstm_hved_amd as X
import xlstm_hved_oracle as O
g = load("net32_subsets_eval")
w = load("weights_seed1")
x = g["x2"][:1]
for train in (False, True):
    m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS); m.load_state_dict(w, strict=True); m = m.cuda().train(train)
    got = {}
    L = 4
    for lvl in range(L):
        m.conv_blocks[lvl].register_forward_hook(lambda mod, i, o, lvl=lvl: got.__setitem__(f"feat.{lvl}", o))
        m.VU_blocks[lvl].register_forward_hook(lambda mod, i, o, lvl=lvl: got.__setitem__(f"vu.{lvl}", o))
    for i in range(len(m.skr_att)):
        m.skr_att[i].register_forward_hook(lambda mod, i_, o, i=i: got.__setitem__(f"skr_att.{L - i}", o))
    for j in range(3):
        m.srdecoder.dusfe_decoders[j].register_forward_hook(lambda mod, i_, o, j=j: got.__setitem__(f"dec.{j}", o))
    with torch.no_grad():
        seg, (mu, lv), rec = m(x.cuda(), [14], recon=True, valid=True)
    sd = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in w.items()}
    taps = {}
    prob, _, omu, olv, orec = O.xlstm_hved_forward(sd, x.double(), 14, eps_list=None, training=train, taps=taps)
    def rel(a, b):
        a = a.detach().double().cpu(); b = b.double()
        return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
    print(f"--- training={train}")
    for lvl in range(L):
        print(f"mu{lvl} {rel(mu[lvl], omu[lvl]):.2e}  feat.{lvl} {rel(got[f'feat.{lvl}'], taps[f'feat.{lvl}']):.2e}", end="")
        if f"skr_att.{lvl}" in got and f"skr_att.{lvl}" in taps:
            print(f"  skr_att.{lvl} {rel(got[f'skr_att.{lvl}'], taps[f'skr_att.{lvl}']):.2e}", end="")
        print()
    for j in range(3):
        r, s_ = got[f"dec.{j}"]
        print(f"dec.{j} recon {rel(r, taps[f'dec.{j}'][0]):.2e} seg {rel(s_, taps[f'dec.{j}'][1]):.2e}")
    print(f"seg {rel(seg, prob):.2e} rec {rel(rec[0], orec):.2e}")
