#!/bin/bash
# The N > 1 code paths of bench.py with ONE rank under torch.distributed.run (RCCL loaded, the
# collectives and the assemble kernel run): tools/rehearse_dist.sh <tag>
set -u
TAG=${1:-r3}
OUT=$GRAFT_REPO_ROOT/gpurun_out
for MODE in "--sharding tiles" "--sharding samples" "--scaling weak"; do
  N=$(echo $MODE | tr -d ' -')
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 20 --warmup 3 $MODE > $OUT/${TAG}_dist1_$N.json 2> $OUT/${TAG}_dist1_$N.err || { echo "rehearsal $MODE failed"; tail -8 $OUT/${TAG}_dist1_$N.err; exit 1; }
  cat $OUT/${TAG}_dist1_$N.json
done
