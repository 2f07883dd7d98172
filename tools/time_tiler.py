"""Times the sliding-window tiler leg of bench.py alone, and one replay of its captured window forward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
import xlstm_hved_amd as X
torch.manual_seed(1)
dev = torch.device("cuda:0")
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS); m.apply(X.init_weights); m = m.to(dev)
for _ in range(3):
    print(B.tiler_leg(m, dev)["ms_per_volume"])
g = next(iter(m.__dict__["_xh_window_graphs"].values()))[1]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    g.replay()
torch.cuda.synchronize(); print("window replay ms", (time.perf_counter() - t0) / 50 * 1e3)
for fold in (False, True):
    X.functional.set_init_fold(fold)
    X.inference.clear_window_graphs(m)
    print("fold", fold, B.tiler_leg(m, dev)["ms_per_volume"])
