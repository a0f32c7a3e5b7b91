"""s_memtime stamps of the window gather-GEMM (PN_GG_DIAG=4) on one dense layer shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PN_GG_DIAG"] = "4"
import numpy as np, torch
from scripts.gg_bench import grid_rules, run, dev
from pillarnet_lts_amd import _lib
H, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nbr, rows = grid_rules(H, H, 1)
x = torch.randn((H * H, cin), device=dev).bfloat16()
run(f"dense {H}^2 {cin}->{cout}", x, nbr, rows, None, cin, cout, iters=5)
torch.cuda.synchronize()
ST = 40
buf = (ctypes.c_ulonglong * (4 + ST * 4 + 8))()
L = ctypes.CDLL(_lib.lib()._name)
L.pn_gg_win_debug_stamps(buf, 4 + ST * 4 + 8)
t = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
k, st, x = t[:4], t[4:4 + ST * 4].reshape(ST, 4), t[4 + ST * 4:]
nst = min(ST, 9 * cin // 64)
print(f"prologue {k[1]-k[0]}  main loop {k[2]-k[1]}  epilogue {k[3]-k[2]}  total {k[3]-k[0]} cycles")
a = st[2:nst - 2]
print(f"per stage: compute {np.mean(a[:,1]-a[:,0]):.0f}  barrier wait {np.mean(a[:,2]-a[:,1]):.0f}  "
      f"stage {np.mean(st[3:nst-2,0]-st[2:nst-3,0]):.0f}  loader 0 landed (ready for barrier) at +{np.mean(a[:,3]-a[:,0]):.0f}")
print(f"prologue: loads issued +{x[0]-k[0]}  nbr in LDS/atomics +{x[1]-k[0]}  after barrier +{x[2]-k[0]}  row table +{x[3]-k[0]}  loop start +{k[1]-k[0]}")
print(f"epilogue: parked +{x[4]-k[2]}  after sync +{x[5]-k[2]}  stores issued +{k[3]-k[2]}")
