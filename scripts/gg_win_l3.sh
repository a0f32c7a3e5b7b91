#!/bin/bash
for cfg in "" U; do
  echo "cfg=${cfg:-default}: $(PN_GG_WIN_CFG=$cfg python scripts/gg_bench.py 2>&1 | grep 'L3 subm')"
done
