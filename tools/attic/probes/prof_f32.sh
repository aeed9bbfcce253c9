out=gpurun_out/r03i; mkdir -p $out; here=$(pwd)
cd /tmp; export TMPDIR=/tmp
for dt in f32; do
  rm -rf /tmp/prof_$dt
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$dt -o bench -- python $here/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-frozen --dtype $dt --also none > $here/$out/prof_$dt.log 2>&1
  db=$(find /tmp/prof_$dt -name '*.db' | head -1)
  python $here/tools/rocprof_summary.py $db 2>/dev/null > $here/$out/kernel_stats_$dt.txt
  head -45 $here/$out/kernel_stats_$dt.txt | cut -c1-140
done
