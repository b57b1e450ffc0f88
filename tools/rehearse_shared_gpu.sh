#!/bin/bash
# The N-rank control flow of bench.py on a ONE-GPU machine: N processes share the GPU and talk over
# gloo (CRT_BENCH_SHARED_GPU=1; RCCL takes one rank per device), every mode, and rank 0 compares the
# image the ranks produced together with the frame rendered alone (--check).  Not a measurement.
#   tools/rehearse_shared_gpu.sh <tag> [ranks ...]
set -u
TAG=${1:-r3}; shift
RANKS=${*:-2 3 8}
OUT=$GRAFT_REPO_ROOT/gpurun_out
PORT=29520
for N in $RANKS; do
  for MODE in "--sharding tiles" "--sharding samples" "--sharding auto" "--scaling weak"; do
    M=$(echo $MODE | tr -d ' -')
    PORT=$((PORT + 1))
    CRT_BENCH_TRACE_AFTER=90 CRT_BENCH_SHARED_GPU=1 timeout -k 10 240 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
      bench.py --gpus $N --steps 7 --warmup 1 --check $MODE > $OUT/${TAG}_shared_${N}_$M.json 2> $OUT/${TAG}_shared_${N}_$M.err \
      || { echo "rehearsal $N ranks $MODE failed"; grep "bench.py:" $OUT/${TAG}_shared_${N}_$M.err; FAILED=1; continue; }
    python3 -c "
import json,sys
d=json.loads(open('$OUT/${TAG}_shared_${N}_$M.json').read().strip().splitlines()[-1])
print($N, '$M', d['n_gpus'], d['ms_per_step'], d['config'].get('check_against_single_render'), d['config'].get('sharding_choice',{}).get('chosen'))
"
  done
done
[ -z "${FAILED:-}" ] && echo rehearsal ok
