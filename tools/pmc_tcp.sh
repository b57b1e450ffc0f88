#!/bin/bash
# Vector-memory pipeline counters of one scene render (run on the GPU box from the repo root):
# tools/pmc_tcp.sh <tag> <scene> [spp]
set -u
TAG=${1:-tcp}
SCENE=${2:-obj.yaml}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE" \
  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
  "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py $SCENE 1 ${3:-} > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_fused" in row["Kernel_Name"]:
            tot[(row["Kernel_Name"][:40], row["Counter_Name"])] += float(row["Counter_Value"])
for k in sorted(tot): print(k[0], k[1], tot[k])
PY
