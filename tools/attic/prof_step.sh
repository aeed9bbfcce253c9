#!/bin/bash
# rocprofv3 kernel-trace summary of the training step for one dtype.   usage: tools/prof_step.sh <outfile> [dtype] [extra bench args]
out=$1; dt=${2:-f32}; shift; shift
here=$(cd $(dirname $0)/.. && pwd); mkdir -p $(dirname $out); out=$(cd $(dirname $out) && pwd)/$(basename $out)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_one
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_one -o bench -- python $here/bench.py --steps 5 --warmup 2 --no-cpu-baseline ${NOGRAPH---no-graph} --no-frozen --dtype $dt --also none "$@" > /tmp/prof_one.log 2>&1
db=$(find /tmp/prof_one -name '*.db' | head -1)
python $here/tools/rocprof_summary.py $db 2>/dev/null | sed "1s|.*|# rocprofv3 --kernel-trace --stats summary of \`rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${NOGRAPH---no-graph} --no-frozen --dtype $dt --also none $*\` (7 steps incl. warm-up, MI355X; stream lanes overlap kernels, so per-kernel durations include contention)|" > $out
