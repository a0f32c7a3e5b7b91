"""Probe: 10-sweep merge of a nuScenes-sized frame: pn_merge_sweeps (inputs resident in HBM) vs the numpy
restatement of the reference's loader on the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np, torch
from oracle import loading_oracle as lo
from pillarnet_lts_amd import _lib, loading
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
S, n = 10, 30000
raws = [np.concatenate([rs.randn(n, 3) * [30, 30, 2], rs.rand(n, 2) * 100], 1).astype(np.float32) for _ in range(S)]
mats = [None] + [np.eye(4) + rs.randn(4, 4) * 0.01 for _ in range(S - 1)]
lags = [0.05 * i for i in range(S)]
t0 = time.time()
for _ in range(5):
    want = lo.combined(raws, mats, lags)
cpu_ms = (time.time() - t0) / 5 * 1e3
# device-resident timing of the three launches alone
raw = torch.cat([torch.from_numpy(r) for r in raws]).to(dev)
starts = np.arange(S + 1, dtype=np.int32) * n
T = torch.from_numpy(np.stack([np.eye(4) if m is None else m for m in mats]).reshape(S, 16)).to(dev)
has = torch.tensor([0] + [1] * (S - 1), dtype=torch.uint8, device=dev)
lag = torch.tensor(lags, dtype=torch.float32, device=dev)
filt = torch.tensor([0] + [1] * (S - 1), dtype=torch.uint8, device=dev)
out = torch.empty((S * n, 5), device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
L = _lib.lib(); ws = torch.empty(L.pn_merge_sweeps_workspace_bytes(S * n), dtype=torch.uint8, device=dev)
def call():
    L.pn_merge_sweeps(_lib.ptr(raw), 5, 4, starts.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), S, _lib.ptr(T), _lib.ptr(has),
                      _lib.ptr(lag), _lib.ptr(filt), 1.0, _lib.ptr(out), _lib.ptr(cnt), _lib.ptr(ws), ws.numel(), _lib.current_stream(dev))
for _ in range(3): call()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): call()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
nb = S * n * (20 * 2 + 8) + int(cnt.item()) * 20
print(f"merge 10 x 30000 points: GPU {us:.1f} us ({nb / us * 1e-3:.0f} GB/s of {nb / 1e6:.1f} MB algorithmic), "
      f"numpy on the host {cpu_ms:.1f} ms, kept {int(cnt.item())} of {S * n}")
