# GPU_MAX_HW_QUEUES sweep of the bench step (separate processes, one box, alternating).  Round 5: 4 (the runtime's default) is best
# by far -- bf16 20.1 ms, 8 -> 24.3, 6 -> 40.8, 16 -> 38.9; f32 45.8 / 50.0 / 58.5 / 56.9.
mkdir -p gpurun_out/r05_hwq; o=gpurun_out/r05_hwq
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-frozen --also none > /dev/null 2>&1
for rep in 1 2; do
for q in ${HWQ:-4 3 2}; do
  for dt in bf16; do
    GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-frozen --also none --dtype $dt 2>$o/err_$q.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('q=$q $dt rep$rep', j['value'], j['ms_per_step'])
" >> $o/hwq2.txt
  done
done
done
cat $o/hwq2.txt; tail -3 $o/err_2.txt
