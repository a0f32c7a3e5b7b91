#!/bin/bash
# Builds scripts/ab/stamps.so: the library with the s_memtime stamp code of the window gather-GEMM compiled in
# (-DGGW_STAMPS, in gather_gemm_win.hip and gather_gemm.hip; add KSTAMPS=1 for the per-k-step stamps).  Run here (CPU, hipcc cross-compiles), then on the GPU:
#   PN_LIB_PATH=$GRAFT_REPO_ROOT/scripts/ab/stamps.so python scripts/ggw_stamps.py 180 256 256 [row tile]
# scripts/ab/ is git-ignored; delete the copy afterwards (every gpurun call uploads the tree).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/pillarnet-lts_amd/csrc
python3 build.py > /dev/null
mkdir -p $ROOT/scripts/ab
DEFS="-DGGW_STAMPS"
[ -n "$KSTAMPS" ] && DEFS="$DEFS -DGGW_KSTAMPS"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops \
  -Wall -Wno-unused-function -I$ROOT/include -I. $DEFS $EXTRA_DEFS -c gather_gemm_win.hip -o /tmp/ggw_probe.o 2>&1 | grep -v "recognized feature" || true
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops \
  -Wall -Wno-unused-function -I$ROOT/include -I. $DEFS $EXTRA_DEFS -c gather_gemm.hip -o /tmp/gg_probe.o 2>&1 | grep -v "recognized feature" || true
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/*.o | grep -v -e gather_gemm_win.o -e /gather_gemm.o) /tmp/ggw_probe.o /tmp/gg_probe.o -o $ROOT/scripts/ab/stamps.so
echo $ROOT/scripts/ab/stamps.so
