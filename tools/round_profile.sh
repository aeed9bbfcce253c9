#!/bin/bash
# Round-end measurement on the GPU box: GPU test suite, smoke, the default bench line, rocprofv3 kernel-trace summaries of
# the f32 and bf16 step (the bench command itself, graph replay), and the per-dtype kernel traces tools/in_step_summary.py
# turns into profiles/in_step.json.   usage: tools/round_profile.sh <outdir under gpurun_out/>
out=$1; mkdir -p $out; out=$(cd $out && pwd); here=$(cd $(dirname $0)/.. && pwd)
cd $here
python -m pytest tests -m gpu -x -q --durations=15 > $out/pytest_gpu.txt 2>&1; tail -22 $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.json
cd /tmp; export TMPDIR=/tmp
for dt in f32 bf16; do
  rm -rf /tmp/prof_$dt
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$dt -o bench -- python $here/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-frozen --dtype $dt --also none > $out/prof_$dt.log 2>&1
  db=$(find /tmp/prof_$dt -name '*.db' | head -1)
  python $here/tools/rocprof_summary.py $db 2>/dev/null | sed "1s|.*|# rocprofv3 --kernel-trace --stats summary of \`rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-frozen --dtype $dt --also none\` (hipGraph replay, 7 steps incl. warm-up + the capture's eager steps, MI355X; stream lanes overlap kernels, so per-kernel durations include contention)|" > $out/kernel_stats_$dt.txt
  head -8 $out/kernel_stats_$dt.txt
  python - <<PY
import sqlite3, gzip, pickle
c = sqlite3.connect("$db")
sel = ["start", "end", "name", "stream_id", "queue_id", "grid_x", "grid_y", "grid_z"]
rows = c.execute("select %s from kernels order by start" % ",".join(sel)).fetchall()
pickle.dump((sel, rows), gzip.open("$out/trace_$dt.pkl.gz", "wb"))
PY
done
