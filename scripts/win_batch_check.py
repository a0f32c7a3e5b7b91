"""Dense 3x3 conv over a batch of identical frames: every frame's output must be bit-identical (tile boundaries
fall differently in each frame), and equal to the per-tap kernel's within rounding."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pillarnet_lts_amd import _lib, dense_hip
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gather_gemm_gpu import call, reference, _mk
dev = torch.device("cuda:0")
for (B, H, W, cin, cout, stride) in [(3, 40, 40, 256, 256, 1), (3, 40, 40, 512, 256, 1), (3, 80, 80, 128, 128, 1),
                                     (3, 40, 40, 256, 64, 1), (3, 40, 40, 64, 64, 1), (3, 80, 80, 128, 256, 2),
                                     (3, 37, 41, 256, 256, 1), (2, 180, 180, 256, 256, 1)]:
    nbr, Ho, Wo = dense_hip.grid_rules(B, H, W, stride, dev)
    rows = B * Ho * Wo
    x1, w, scale, shift = _mk(cin, cout, H * W, dev, 3)
    x = x1.repeat(B, 1).contiguous()
    for it in range(3):
        got = call(x, nbr, w, scale, shift, rows, relu=False, fp32_out=True).view(B, Ho * Wo, cout)
        same = all(torch.equal(got[0], got[b]) for b in range(1, B))
        want = reference(x, nbr, w, scale, shift, None, False, rows).view(B, Ho * Wo, cout)
        err = (got - want).abs().max().item()
        bad = (got[0] != got[1]).any(dim=1).nonzero().view(-1)
        print(f"B{B} {H}x{W} {cin}->{cout} s{stride} it{it}: frames identical {same}  max err vs fp32 {err:.2e}  "
              f"differing pixels {bad.numel()} first {bad[:6].tolist()}", flush=True)
