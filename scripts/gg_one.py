"""Runs ONE gather-GEMM shape repeatedly (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scripts.gg_bench import grid_rules, run, dev  # noqa
H, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nbr, rows = grid_rules(H, H, 1)
x = torch.randn((H * H, cin), device=dev).bfloat16()
run(f"dense {H}^2 {cin}->{cout}", x, nbr, rows, None, cin, cout, iters=int(sys.argv[4]) if len(sys.argv) > 4 else 10)
