#!/bin/bash
# config 5 (W64 fp16) bench line + rocprofv3 kernel summary only (tools/config_records.sh does all three configs).  usage: config5_profile.sh <outdir> <tag>
out=$1; tag=${2:-r05}; mkdir -p $out; out=$(cd $out && pwd); here=$(cd $(dirname $0)/../.. && pwd)
cd /tmp; export TMPDIR=/tmp
python $here/bench.py --also none --no-frozen --no-cpu-baseline --width 64 --dtype f16 > $out/${tag}_bench_config5_w64_f16.json 2> $out/${tag}_bench_config5.err
tail -c 600 $out/${tag}_bench_config5_w64_f16.json; echo
rm -rf /tmp/prof_c5
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o bench -- python $here/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-frozen --also none --width 64 --dtype f16 > $out/${tag}_prof_config5.log 2>&1
db=$(find /tmp/prof_c5 -name '*.db' | head -1)
python $here/tools/rocprof_summary.py $db 2>/dev/null | sed "1s|.*|# rocprofv3 --kernel-trace --stats summary of \`rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-frozen --also none --width 64 --dtype f16\` (7 steps incl. warm-up, MI355X; stream lanes overlap kernels, so per-kernel durations include contention)|" > $out/${tag}_kernel_stats_config5_w64_f16.txt
head -30 $out/${tag}_kernel_stats_config5_w64_f16.txt
