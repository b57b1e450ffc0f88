#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in - wf4 wf8; do for RP in 2 12 25 50; do for PB in 8 16; do
  if [ "$L" = "-" ]; then unset CRT_LIB; else export CRT_LIB=$PWD/cudaraytracer_amd/libcrt_hip_$L.so; fi
  echo -n "lib=$L ready=$RP prim_batch=$PB: "
  CRT_WAVEFRONT=1 CRT_READY_PERCENT=$RP CRT_PRIM_BATCH=$PB timeout -k 10 200 python3 tools/quick_perf.py mesh1m.yaml 2 8 2>&1 | grep rep1 | cut -c1-60
done; done; done
