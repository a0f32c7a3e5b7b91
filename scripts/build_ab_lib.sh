#!/bin/bash
# Builds scripts/ab/<name>.so: the library with extra -D flags on chosen kernels' sources, for A/B runs inside one GPU job
#   scripts/build_ab_lib.sh prio "-DGGW_PRIO=1" gather_gemm_win     (here, on the CPU; hipcc cross-compiles)
#   PN_LIB_PATH=$GRAFT_REPO_ROOT/scripts/ab/prio.so python scripts/gg_bench.py     (on the GPU)
# scripts/ab/ is git-ignored; delete the copies afterwards (every gpurun call uploads the tree).
set -e
NAME=$1; DEFS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/pillarnet-lts_amd/csrc
python3 build.py > /dev/null
mkdir -p $ROOT/scripts/ab
SKIP=""
OBJS=""
for src in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops \
    -Wall -Wno-unused-function -I$ROOT/include -I. $DEFS -c $src.hip -o /tmp/ab_${NAME}_$src.o 2>&1 | grep -v "recognized feature" || true
  SKIP="$SKIP -e /$src.o"
  OBJS="$OBJS /tmp/ab_${NAME}_$src.o"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/*.o | grep -v $SKIP) $OBJS -o $ROOT/scripts/ab/$NAME.so
echo $ROOT/scripts/ab/$NAME.so
