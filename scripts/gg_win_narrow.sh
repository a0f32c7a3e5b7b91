#!/bin/bash
# narrow-layer (64 channels) tile configurations of the window kernel on the real level-2 rulebook
for cfg in "" H I; do
  echo "cfg=${cfg:-default}: $(PN_GG_WIN_CFG=$cfg python scripts/gg_bench.py 2>&1 | grep 'L2 subm')"
done
echo "classic: $(PN_GG_NO_WIN=1 python scripts/gg_bench.py 2>&1 | grep 'L2 subm')"
