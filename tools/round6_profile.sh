#!/bin/bash
# Round-6 measurement on the GPU box: GPU test suite, smoke, default and driver-style bench lines, kernel traces (graph replay) of the
# f32 and bf16 step, PMC passes of the dominant launches (incl. the DCN kernels), per-launch benches, phase times (whole step and the
# stem + layer1 stretch).  Every step runs under `timeout`: a hung profiler must not eat the box.
# usage: tools/round6_profile.sh <outdir under gpurun_out/> [quick]
out=$1; mkdir -p $out; out=$(cd $out && pwd); here=$(cd $(dirname $0)/.. && pwd)
cd $here
if [ "$2" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $out/pytest_gpu.txt 2>&1; tail -5 $out/pytest_gpu.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
fi
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 300 $out/bench_default.json; echo
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_style.json 2> $out/bench_driver_style.err; tail -c 200 $out/bench_driver_style.json; echo
for dt in f32 bf16; do timeout 700 tools/trace_dump.sh $out $dt; done
for w in conv_f32 conv_bf16 dcn_f32 dcn_bf16 dcnbwd_f32 dcnbwd_bf16; do
  timeout 400 tools/pmc_sets.sh $w $out/pmc_$w full > $out/pmc_$w.txt 2>&1
  rm -rf $out/pmc_$w        # keep the summaries, not the raw csv trees
done
timeout 200 python tools/bench_t5.py > $out/bench_t5.txt 2>&1
timeout 200 python tools/bench_t6.py > $out/bench_t6.txt 2>&1
timeout 200 python tools/bench_wg6.py > $out/bench_wg6.txt 2>&1
timeout 200 python tools/bench_dcn_bwd.py > $out/bench_dcn_bwd.txt 2>&1
timeout 300 python tools/phase_times.py bf16 > $out/phase_times.txt 2>&1; timeout 300 python tools/phase_times.py f32 >> $out/phase_times.txt 2>&1
timeout 300 python tools/phase_stem.py bf16 > $out/phase_stem.txt 2>&1; timeout 300 python tools/phase_stem.py f32 >> $out/phase_stem.txt 2>&1
grep -v amdgpu $out/phase_times.txt | tail -14; grep -v amdgpu $out/phase_stem.txt | grep stretch
