#!/bin/bash
# Collect PMC counters for one scene render, one rocprofv3 run per counter group.
# usage: tools/pmc_passes.sh <tag> [scene]      (run on the GPU box from the repo root)
set -u
TAG=${1:-pmc}
SCENE=${2:-cornell_box.yaml}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE GRBM_COUNT" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py $SCENE 1 ${3:-} > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if any(k in row["Kernel_Name"] for k in ("render_fused", "trace_", "wf_")):
            tot[(row["Kernel_Name"][:40], row["Counter_Name"])] += float(row["Counter_Value"])
for k in sorted(tot): print(k[0], k[1], tot[k])
PY
