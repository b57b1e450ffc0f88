#!/bin/bash
# Device time per frame of every scene (run on the GPU box from the repo root): tools/scene_times.sh <out file>
set -u
OUT=${1:-gpurun_out/scene_times.txt}
: > $OUT
for S in cornell_box.yaml orthographic.yaml materials.yaml obj.yaml normal_map_without.yaml hdr_env.yaml features_tree.yaml nested_wrappers.yaml; do
  timeout -k 10 120 python3 tools/quick_perf.py $S 4 2>&1 | grep -E "rep[1-3]" >> $OUT || exit 1
  timeout -k 10 60 python3 -c "
import sys; sys.path.insert(0, '.')
import cudaraytracer_amd as crt
sc = crt.load_scene('scenes/$S', base_dir='scenes'); sc.upload(); print('$S:', sc.describe())" >> $OUT 2>&1 || exit 1
done
timeout -k 10 300 python3 tools/quick_perf.py mesh1m.yaml 3 8 2>&1 | grep -E "rep[1-2]|upload|load" >> $OUT || exit 1
cat $OUT
