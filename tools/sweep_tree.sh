#!/bin/bash
# Sweep of the tree walk's scheduling knobs: tools/sweep_tree.sh <scene> [spp]
set -u
S=${1:-mesh1m.yaml}; SPP=${2:-}
for PB in 4 8 16 32; do for RP in 35 50 65 80; do
  echo "== prim_batch $PB ready $RP"
  CRT_PRIM_BATCH=$PB CRT_READY_PERCENT=$RP timeout -k 10 200 python3 tools/quick_perf.py $S 3 $SPP 2>&1 | grep -E "rep2" || exit 1
done; done
