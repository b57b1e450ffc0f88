#!/bin/bash
# A/B device times: tools/ab_perf.sh "<lib suffixes, '-' = product>" "<scene[:spp]> ..."   (GPU box, repo root)
set -u
LIBS=${1:--}
SCENES=${2:-"cornell_box.yaml obj.yaml mesh1m.yaml:8"}
for S in $SCENES; do
  NAME=${S%%:*}; SPP=""; [[ $S == *:* ]] && SPP=${S##*:}
  for L in $LIBS; do
    if [ "$L" = "-" ]; then unset CRT_LIB; else export CRT_LIB=$PWD/cudaraytracer_amd/libcrt_hip_$L.so; fi
    echo "== $NAME lib=$L"
    timeout -k 10 200 python3 tools/quick_perf.py $NAME ${REPS:-13} $SPP 2>&1 | grep -E "summary|Error|error" || exit 1
  done
done
