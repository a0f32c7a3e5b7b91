"""fp32 rulebook convolution (training forward / data gradient) at the backbone's level shapes, batch 4:
PN_SPCONV_F32_V1=1 in the environment selects the per-wave kernel for comparison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scripts.gg_bench import grid_rules, dev
from pillarnet_lts_amd import _lib
L = _lib.lib()
torch.manual_seed(0)
for (H, cin, cout) in ((344, 256, 256), (666, 128, 128), (1000, 64, 64), (344, 128, 256), (344, 256, 128), (666, 64, 128),
                       (666, 128, 64), (1000, 32, 64)):
    nbr, rows = grid_rules(H, H, 1)
    x = torch.randn((H * H, cin), device=dev)
    w = torch.randn((9, cout, cin), device=dev) * 0.05
    scale, shift = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    res = torch.randn((rows, cout), device=dev)
    out = torch.empty((rows, cout), device=dev)
    count = torch.tensor([rows], dtype=torch.int32, device=dev)

    def call():
        rc = L.pn_spconv_forward(0, _lib.ptr(x), cin, x.shape[0], _lib.ptr(nbr), nbr.shape[1], _lib.ptr(count), rows,
                                 _lib.ptr(w), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(res), 1, _lib.ptr(out), cout,
                                 _lib.current_stream(dev))
        _lib.check(rc, "pn_spconv_forward")
    for _ in range(2):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    # fp32 reference on a slice of the rows
    n = 4096
    ref = torch.zeros((n, cout), device=dev, dtype=torch.float64)
    for k in range(9):
        idx = nbr[k, :n].long(); ok = idx >= 0
        ref[ok] += x[idx[ok]].double() @ w[k].double().t()
    ref = torch.relu(ref * scale.double() + shift.double() + res[:n].double())
    err = (out[:n].double() - ref).abs().max().item()
    print(f"fp32 conv {H}^2 {cin}->{cout}: {ms:.3f} ms  {2.0 * rows * 9 * cin * cout / ms / 1e9:.1f} TF/s   max err vs fp64 {err:.2e} (max {ref.abs().max().item():.1f})", flush=True)
