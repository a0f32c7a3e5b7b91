#!/bin/bash
# needs the probe build of the library: scripts/build_probe_lib.sh, then PN_LIB_PATH=$GRAFT_REPO_ROOT/scripts/ab/stamps.so
# (the shipped kernels carry no probe code)
# window-kernel probes on the dense 180^2 256->256 layer: config x diag mode
for cfg in "" A L; do
  for diag in 0 1 2; do
    echo "cfg=${cfg:-default} diag=$diag: $(PN_GG_WIN_CFG=$cfg PN_GG_DIAG=$diag python scripts/gg_one.py dense 180 256 256 20 2>&1 | grep dense)"
  done
done
echo "classic: $(PN_GG_NO_WIN=1 python scripts/gg_one.py dense 180 256 256 20 2>&1 | grep dense)"
