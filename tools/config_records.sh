#!/bin/bash
# Bench lines + rocprofv3 kernel-trace summaries for the BASELINE.json configs other than the headline:
#   config 2: W48 384x288, 3-frame (S=2), batch 8, fp32     config 4: W48 512x384, S=7 (DCN gather at 128x96)
#   config 5: W64 384x288, S=4, fp16 MFMA convs + fp32 losses
# >= 20 warm-up + >= 50 timed steps (bench.py defaults).   usage: tools/config_records.sh <outdir> [tag]
out=$1; tag=${2:-r02}; mkdir -p $out; out=$(cd $out && pwd); here=$(cd $(dirname $0)/.. && pwd)
cd /tmp; export TMPDIR=/tmp
run() {  # name, bench args...
  name=$1; shift
  python $here/bench.py --also none --no-frozen --no-cpu-baseline "$@" > $out/${tag}_bench_$name.json 2> $out/${tag}_bench_$name.err
  tail -c 900 $out/${tag}_bench_$name.json; echo
  rm -rf /tmp/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o bench -- python $here/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-frozen --also none "$@" > $out/${tag}_prof_$name.log 2>&1
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  python $here/tools/rocprof_summary.py $db 2>/dev/null | sed "1s|.*|# rocprofv3 --kernel-trace --stats summary of \`rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-frozen --also none $*\` (7 steps incl. warm-up, MI355X; stream lanes overlap kernels, so per-kernel durations include contention)|" > $out/${tag}_kernel_stats_$name.txt
  head -6 $out/${tag}_kernel_stats_$name.txt
}
run config2_s2_b8_f32 --sup 2 --batch 8 --dtype f32
run config4_512x384_s7_f32 --img-h 512 --img-w 384 --sup 7 --batch 4 --dtype f32
run config5_w64_f16 --width 64 --dtype f16
