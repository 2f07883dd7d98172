import sys, os, ctypes as C
sys.path.insert(0, "/root/repo")
import torch
import xlstm_hved_amd as X
from xlstm_hved_amd import ops
L = X._lib
torch.manual_seed(1)
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS); m.apply(X.init_weights); m = m.cuda().train()
x = torch.rand(1, 4, 128, 128, 128, device="cuda").bfloat16()
for _ in range(2):
    seg, (mu, lv), rec = m(x, [14], recon=True)
    (seg.float().mean() + rec[0].float().mean()).backward()
ops.join_wgrad_stream()
torch.cuda.synchronize()
ops._pack_arrays_refresh()
ents, darr, parr = ops._PACK_STATE["arrays"]
lib = L.load()
nbytes = int(lib.xh_conv3d_prepack_table_bytes())
host = (C.c_char * nbytes)()
assert lib.xh_conv3d_prepack_table(len(ents), darr, parr, C.cast(host, C.c_void_p)) == 0
W = int(sys.argv[1])
class PackJob(C.Structure):
    _fields_ = [("w", C.c_void_p * W), ("ws", C.c_void_p), ("kind", C.c_int)] + [(k, C.c_int) for k in
        "f16 groups n_wptr transposed Cin_g Cout_g ntile cin_stride cin_off cin_blk cout_set nm nch cpr cinp ci4 dw nelem".split()]
head = (C.c_int * 2).from_buffer(host)
n = head[0]
off = 4 * (2 + 257 + 1)
print("jobs", n, "blocks", head[1], "sizeof", C.sizeof(PackJob))
tot = {0: 0, 1: 0}
for i in range(n):
    j = PackJob.from_buffer(host, off + i * C.sizeof(PackJob))
    b = (j.nelem + 2047) // 2048
    tot[j.kind] += b
    print(i, "kind", j.kind, "g", j.groups, "cin_g", j.Cin_g, "cout_g", j.Cout_g, "T" if j.transposed else "F", "nelem", j.nelem, "blocks", b, "nm", j.nm, "cinp", j.cinp, "ci4", j.ci4)
print(tot)
