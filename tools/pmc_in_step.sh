#!/bin/bash
# L2 hit rate of the BatchNorm passes INSIDE the bf16 step (review r5 item 5): one rocprofv3 counter pass (TCC_HIT_sum, TCC_MISS_sum) over
# a short graph-mode bench run; per kernel name: calls, hits, misses, hit rate.  Counter collection serialises the kernels, so the cache
# state a launch finds is what its predecessors on the replayed order left.   usage: tools/pmc_in_step.sh <outdir> [dtype]
out=$1; dt=${2:-bf16}; mkdir -p $out; out=$(cd $out && pwd); here=$(cd $(dirname $0)/.. && pwd)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc_step_$dt
timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_step_$dt -o pmc -- python $here/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-frozen --dtype $dt --also none > $out/pmc_in_step_$dt.log 2>&1
f=$(find /tmp/pmc_step_$dt -name '*counter_collection.csv' | head -1)
python - "$f" <<'PY' > $out/pmc_in_step_$dt.txt
import csv, sys, collections, re
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
rows = list(csv.DictReader(open(sys.argv[1])))
disp = collections.defaultdict(dict)
for r in rows:
    disp[(r['Dispatch_Id'], r['Kernel_Name'])][r['Counter_Name']] = float(r['Counter_Value'])
for (d, name), c in disp.items():
    n = re.sub(r'^void ', '', name); n = re.sub(r'^_Z\d+', '', n)[:60]
    a = agg[n]; a[0] += 1; a[1] += c.get('TCC_HIT_sum', 0.0); a[2] += c.get('TCC_MISS_sum', 0.0)
print('# L2 (TCC) hits / misses per kernel over a 4-step bf16 bench run under counter collection; hit rate = hits / (hits + misses)')
print('%-62s %6s %14s %14s %8s' % ('kernel', 'calls', 'TCC_HIT', 'TCC_MISS', 'hit rate'))
for n, (c, h, m) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:40]:
    print('%-62s %6d %14.0f %14.0f %8.3f' % (n, c, h, m, h / max(h + m, 1.0)))
PY
head -30 $out/pmc_in_step_$dt.txt
