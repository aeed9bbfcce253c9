#!/bin/bash
# CPU baseline (oracle, 1 clip fwd+loss+bwd+Adam) against the host thread count on the GPU box.
# usage: tools/cpu_threads.sh <outfile>
here=$(cd $(dirname $0)/.. && pwd)
echo "# python bench.py --cpu-baseline-only --cpu-threads N on $(nproc) host threads ($(date -u +%F))" > $1
for n in 8 16 32 64 128 0; do
  HIP_VISIBLE_DEVICES= CUDA_VISIBLE_DEVICES= timeout 300 python $here/bench.py --cpu-baseline-only --cpu-threads $n 2>/dev/null | tail -1 >> $1
done
cat $1
