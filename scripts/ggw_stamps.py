"""s_memtime stamps of the window gather-GEMM (PN_GG_DIAG=4) on one dense layer shape.  Needs the probe build of the
library (scripts/build_probe_lib.sh, PN_LIB_PATH=.../scripts/ab/stamps.so): the shipped kernels carry no stamp code."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PN_GG_DIAG"] = str(4 + 16 * int(sys.argv[4]) if len(sys.argv) > 4 else 4)      # argv[4]: which row tile to stamp
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

# every workgroup's start / end tick: how the launch's workgroups overlap (ticks of different XCDs may be offset, so
# spans are taken per XCD: workgroup i runs on XCD i % 8)
nb = int(os.environ.get("GGW_BLOCKS", "0"))
if nb:
    sp = (ctypes.c_ulonglong * (2 * nb))()
    L.pn_gg_win_debug_spans(sp, 2 * nb)
    s = np.frombuffer(sp, dtype=np.uint64).astype(np.int64).reshape(nb, 2)
    dur = s[:, 1] - s[:, 0]
    print(f"workgroups {nb}: duration min {dur.min()}  median {int(np.median(dur))}  max {dur.max()}")
    for xcd in range(8):
        q = s[xcd::8]
        st = np.sort(q[:, 0] - q[:, 0].min())
        print(f"  xcd {xcd}: {len(q)} workgroups, span {q[:,1].max() - q[:,0].min()} ticks; start offsets: median {int(np.median(st))}  "
              f"90% {int(st[int(0.9 * (len(st) - 1))])}  last {st[-1]}")
    if os.environ.get("GGW_DETAIL"):
        gx = int(os.environ["GGW_DETAIL"])          # grid x (row tiles)
        print("  deciles:", [int(np.percentile(dur, p)) for p in range(0, 101, 10)])
        print("  mean by i % 8:", [int(dur[k::8].mean()) for k in range(8)])
        print("  mean by i % 32:", [int(dur[k::32].mean()) for k in range(32)])
        ny = nb // gx
        for y in range(ny):
            d = dur[y * gx:(y + 1) * gx]
            print(f"  column tile {y}: mean {int(d.mean())}; by row-tile eighths:", [int(c.mean()) for c in np.array_split(d, 8)])
        slow = np.argsort(-dur)[:24]
        print("  slowest:", [(int(i % gx), int(i // gx), int(dur[i])) for i in slow])
        # start order: who starts late (ticks are comparable only inside an XCD; print raw starts relative to the global minimum modulo 2^20 to spot groups)
        st = s[:, 0]
        print("  start tick spread (max - min):", int(st.max() - st.min()))

if os.environ.get("GGW_WAVES"):
    wb = (ctypes.c_ulonglong * (16 * 4 * 3 + 16 * 4))()
    L.pn_gg_win_debug_waves(wb, 16 * 4 * 3 + 16 * 4)
    wall = np.frombuffer(wb, dtype=np.uint64).astype(np.int64)
    w, ksteps = wall[:192].reshape(16, 4, 3), wall[192:].reshape(16, 4)
    print("per wave, stages 8..11 of the per-tap loop: arrival at / departure from the stage barrier relative to the earliest stage start")
    for st in range(4):
        starts = w[:, st, 2]
        t0 = starts[starts > 0].min() if (starts > 0).any() else w[:, st, 0][w[:, st, 0] > 0].min()
        print(f"  stage {8 + st}: " + "  ".join(f"w{i}:{w[i, st, 0] - t0}/{w[i, st, 1] - t0}" for i in range(16) if w[i, st, 0] > 0))
    if os.environ.get("GGW_KSTEPS"):     # library built with -DGGW_KSTAMPS
        starts = w[:, 2, 2]
        t0 = starts[starts > 0].min()
        print("  stage 10, k-step starts (fragments landed) per compute wave, then arrival at the barrier:")
        for i in range(8):
            print(f"    w{i}: stage start {w[i, 2, 2] - t0}  k-steps {[int(v - t0) for v in ksteps[i]]}  barrier {w[i, 2, 0] - t0}")
