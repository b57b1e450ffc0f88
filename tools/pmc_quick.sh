#!/bin/bash
# One PMC pass (instruction counts) for a scene: tools/pmc_quick.sh [scene]
SCENE=${1:-cornell_box.yaml}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py $SCENE 1 > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py $SCENE 1 > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_fused" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
for k in sorted(tot): print(k, f"{tot[k]:.4g}")
PY
