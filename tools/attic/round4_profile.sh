#!/bin/bash
# Round-4 measurement on the GPU box: GPU test suite, smoke, default bench line, kernel traces (graph replay) of the f32 and bf16
# step, PMC passes of the dominant launches, per-launch benches, phase times.   usage: tools/round4_profile.sh <outdir under gpurun_out/> [quick]
out=$1; mkdir -p $out; out=$(cd $out && pwd); here=$(cd $(dirname $0)/.. && pwd)
cd $here
if [ "$2" != "quick" ]; then
  python -m pytest tests -m gpu -q --durations=10 > $out/pytest_gpu.txt 2>&1; tail -5 $out/pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
fi
python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 300 $out/bench_default.json; echo
for dt in f32 bf16; do tools/trace_dump.sh $out $dt; done
for w in conv_f32 conv_bf16 dcn_f32 dcn_bf16 dcnbwd_f32 dcnbwd_bf16; do
  tools/pmc_sets.sh $w $out/pmc_$w full > $out/pmc_$w.txt 2>&1
  rm -rf $out/pmc_$w        # keep the summaries, not the raw csv trees
done
python tools/bench_t5.py > $out/bench_t5.txt 2>&1
python tools/bench_t6.py > $out/bench_t6.txt 2>&1
python tools/bench_wg6.py > $out/bench_wg6.txt 2>&1
python tools/bench_layer1.py > $out/bench_layer1.txt 2>&1
python tools/bench_t4.py > $out/bench_t4_bf16.txt 2>&1
python tools/phase_times.py bf16 > $out/phase_times.txt 2>&1; python tools/phase_times.py f32 >> $out/phase_times.txt 2>&1
grep -v amdgpu $out/phase_times.txt | tail -14
