"""Runs the fp32 rulebook convolution on one layer shape a few times (for rocprofv3 --pmc passes): spconv_f32_one.py [H cin cout]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scripts.gg_bench import grid_rules, dev
from pillarnet_lts_amd import _lib
L = _lib.lib()
H, cin, cout = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (344, 256, 256)
torch.manual_seed(0)
nbr, rows = grid_rules(H, H, 1)
x = torch.randn((H * H, cin), device=dev)
w = torch.randn((9, cout, cin), device=dev) * 0.05
scale, shift = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
out = torch.empty((rows, cout), device=dev)
count = torch.tensor([rows], dtype=torch.int32, device=dev)
for _ in range(4):
    _lib.check(L.pn_spconv_forward(0, _lib.ptr(x), cin, x.shape[0], _lib.ptr(nbr), nbr.shape[1], _lib.ptr(count), rows, _lib.ptr(w),
                                   _lib.ptr(scale), _lib.ptr(shift), None, 1, _lib.ptr(out), cout, _lib.current_stream(dev)), "conv")
torch.cuda.synchronize()
print("done", float(out.abs().max()))
