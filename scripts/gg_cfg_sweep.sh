#!/bin/bash
# tile-config sweep of the wide layers (tuning aid)
for f in "" G; do
  echo "== PN_GG_FORCE=$f"
  PN_GG_FORCE=$f timeout -k 10 200 python scripts/gg_bench.py 2>&1 | grep -E "L4|dense 180\^2 256->256|dense 180\^2 512|dense 90|s2"
done
