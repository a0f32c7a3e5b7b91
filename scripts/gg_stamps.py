"""Reads the s_memtime stamps of the per-tap gather-GEMM (PN_GG_DIAG=3, PN_GG_NO_WIN=1) on one layer shape.
Needs the probe build of the library (scripts/build_probe_lib.sh, PN_LIB_PATH=.../scripts/ab/stamps.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PN_GG_DIAG"] = "3"
os.environ["PN_GG_NO_WIN"] = "1"
import numpy as np, torch
from scripts.gg_bench import grid_rules, run, dev
from pillarnet_lts_amd import _lib
H, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nbr, rows = grid_rules(H, H, 1)
x = torch.randn((H * H, cin), device=dev).bfloat16()
run(f"dense {H}^2 {cin}->{cout}", x, nbr, rows, None, cin, cout, iters=5)
torch.cuda.synchronize()
WG, ST, PER = 4, 40, 6
buf = (ctypes.c_ulonglong * (WG * ST * PER))()
L = ctypes.CDLL(_lib.lib()._name)
rc = L.pn_gg_debug_stamps(buf, WG * ST * PER)
t = np.frombuffer(buf, dtype=np.uint64).reshape(WG, ST, PER).astype(np.int64)
nst = 9 * cin // 64
for wg in range(2):
    a = t[wg, 1:nst - 1]
    print(f"wg {wg}: per-stage cycles  issue {np.mean(a[:,1]-a[:,0]):.0f}  compute {np.mean(a[:,2]-a[:,1]):.0f}  "
          f"vmcnt-wait {np.mean(a[:,3]-a[:,2]):.0f}  barrier {np.mean(a[:,4]-a[:,3]):.0f}  "
          f"stage {np.mean(t[wg,2:nst-1,0]-t[wg,1:nst-2,0]):.0f}")
