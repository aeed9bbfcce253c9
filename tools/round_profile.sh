#!/bin/bash
# Round-end measurement on the GPU box: GPU test suite, smoke, the default bench line, rocprofv3 kernel-trace
# summaries of the f32 and bf16 step.   usage: tools/round_profile.sh <outdir under gpurun_out/>
out=$1; mkdir -p $out; out=$(cd $out && pwd); here=$(cd $(dirname $0)/.. && pwd)
cd $here
python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.json
OFFSTD=0.01 python tools/bench_dcn.py 2>&1 | head -8 > $out/bench_dcn_offstd0.txt
cd /tmp; export TMPDIR=/tmp
for dt in f32 bf16; do
  rm -rf /tmp/prof_$dt
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$dt -o bench -- python $here/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-frozen --dtype $dt --also none > $out/prof_$dt.log 2>&1
  db=$(find /tmp/prof_$dt -name '*.db' | head -1)
  python $here/tools/rocprof_summary.py $db 2>/dev/null | sed "1s|.*|# rocprofv3 --kernel-trace --stats summary of \`rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-frozen --dtype $dt --also none\` (7 steps incl. warm-up, MI355X; stream lanes overlap kernels, so per-kernel durations include contention)|" > $out/kernel_stats_$dt.txt
  head -8 $out/kernel_stats_$dt.txt
done
