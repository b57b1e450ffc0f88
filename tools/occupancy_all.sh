#!/bin/bash
# Loop occupancy (CRT_PROFILE build) and section clocks (CRT_PROFILE_CLOCK build) of every BASELINE config:
#   make -C cudaraytracer_amd/csrc profile   (here), then on the GPU box:  tools/occupancy_all.sh <tag>
# Result: gpurun_out/<tag>_loop_occupancy_and_sections.txt
set -u
TAG=${1:-r3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_loop_occupancy_and_sections.txt
: > $OUT
for S in cornell_box.yaml orthographic.yaml obj.yaml materials.yaml materials_hdr.yaml "mesh1m.yaml 8"; do
  CRT_LIB=$GRAFT_REPO_ROOT/cudaraytracer_amd/libcrt_hip_prof.so timeout -k 10 200 python3 tools/prof_counters.py $S >> $OUT 2>&1 || { echo "counters $S failed"; tail -3 $OUT; exit 1; }
  CRT_LIB=$GRAFT_REPO_ROOT/cudaraytracer_amd/libcrt_hip_clk.so timeout -k 10 200 python3 tools/prof_counters.py $S >> $OUT 2>&1 || { echo "clocks $S failed"; tail -3 $OUT; exit 1; }
done
cat $OUT
