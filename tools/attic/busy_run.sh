#!/bin/bash
here=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for dt in f32 bf16; do
  rm -rf /tmp/pg_$dt
  timeout 600 rocprofv3 --kernel-trace -d /tmp/pg_$dt -o bench -- python $here/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-frozen --dtype $dt --also none > /tmp/pg_$dt.log 2>&1
  db=$(find /tmp/pg_$dt -name '*.db' | head -1)
  echo "== $dt graph mode" >> $here/gpurun_out/r02_busy.txt
  tail -1 /tmp/pg_$dt.log | cut -c1-300 >> $here/gpurun_out/r02_busy.txt
  python $here/tools/busy.py $db 0.75 1.0 >> $here/gpurun_out/r02_busy.txt 2>&1
done
