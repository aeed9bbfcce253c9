# HIP runtime switches around graph execution, one bench process each (same box, alternating): NAME=VALUE pairs in $ENVS (space separated; "base" = none)
mkdir -p gpurun_out/r05_hipenv; o=gpurun_out/r05_hipenv; dt=${DT:-bf16}
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-frozen --also none > /dev/null 2>&1
for rep in 1 2; do
for e in ${ENVS:-base}; do
  if [ "$e" = base ]; then ev=""; else ev="$e"; fi
  env $ev timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-frozen --also none --dtype $dt 2>$o/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$e $dt rep$rep', j['value'], j['ms_per_step'])
" >> $o/hipenv_$dt.txt
done
done
cat $o/hipenv_$dt.txt
