#!/bin/bash
# rocprofv3 kernel trace of the bench step (graph replay) -> gpurun_out/<out>/trace_<dtype>.pkl.gz (start, end, name, stream, queue, grid)
out=$1; dt=${2:-bf16}; mkdir -p $out; out=$(cd $out && pwd); here=$(cd $(dirname $0)/.. && pwd)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_$dt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$dt -o bench -- python $here/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-frozen --dtype $dt --also none > $out/prof_$dt.log 2>&1
db=$(find /tmp/prof_$dt -name '*.db' | head -1)
python $here/tools/rocprof_summary.py $db 2>/dev/null > $out/kernel_stats_$dt.txt
python - <<PY
import sqlite3, gzip, pickle
c = sqlite3.connect("$db")
sel = ["start", "end", "name", "stream_id", "queue_id", "grid_x", "grid_y", "grid_z"]
rows = c.execute("select %s from kernels order by start" % ",".join(sel)).fetchall()
pickle.dump((sel, rows), gzip.open("$out/trace_$dt.pkl.gz", "wb"))
PY
