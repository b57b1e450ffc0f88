#!/bin/bash
# One bench.py JSON line per BASELINE config plus the rocprofv3 kernel statistics of the same
# command (run on the GPU box from the repo root):  tools/bench_all.sh <tag>
# Results: gpurun_out/<tag>_bench_<scene>.json, gpurun_out/<tag>_stats_<scene>.csv
set -u
TAG=${1:-r3}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {   # scene steps warmup extra...
  local S=$1 K=$2 Wm=$3; shift 3
  local N=${S%.yaml}
  timeout -k 10 400 python3 $GRAFT_REPO_ROOT/bench.py --scene $S --steps $K --warmup $Wm "$@" > $OUT/${TAG}_bench_$N.json 2> $OUT/${TAG}_bench_$N.err || { echo "bench $S failed"; tail -5 $OUT/${TAG}_bench_$N.err; return 1; }
  rm -rf $OUT/${TAG}_prof_$N
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_$N -- python3 $GRAFT_REPO_ROOT/bench.py --scene $S --steps $K --warmup $Wm "$@" --cpu-spp 0 > $OUT/${TAG}_prof_$N.json 2> $OUT/${TAG}_prof_$N.err || { echo "rocprof $S failed"; tail -5 $OUT/${TAG}_prof_$N.err; return 1; }
  find $OUT/${TAG}_prof_$N -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_stats_$N.csv \;
  rm -rf $OUT/${TAG}_prof_$N
  cat $OUT/${TAG}_bench_$N.json
}
run cornell_box.yaml 20 3 --no-configs &&
run orthographic.yaml 20 3 --cpu-spp 0 &&
run obj.yaml 20 3 --cpu-spp 0 &&
run materials.yaml 20 3 --cpu-spp 0 &&
run materials_hdr.yaml 20 3 --cpu-spp 0 &&
run mesh1m.yaml 3 1 --cpu-spp 0
