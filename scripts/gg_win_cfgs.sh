#!/bin/bash
# window-kernel tile configurations on the 256-wide layer shapes
for cfg in "" D T; do
  for shape in "180 256 256" "180 512 256" "90 256 256"; do
    echo "cfg=${cfg:-default}: $(PN_GG_WIN_CFG=$cfg python scripts/gg_one.py dense $shape 20 2>&1 | grep dense)"
  done
  echo "cfg=${cfg:-default}: $(PN_GG_WIN_CFG=$cfg python scripts/gg_one.py l4 20 2>&1 | grep sparse)"
done
for shape in "180 256 256" "180 512 256" "90 256 256"; do
  echo "classic: $(PN_GG_NO_WIN=1 python scripts/gg_one.py dense $shape 20 2>&1 | grep dense)"
done
echo "classic: $(PN_GG_NO_WIN=1 python scripts/gg_one.py l4 20 2>&1 | grep sparse)"
