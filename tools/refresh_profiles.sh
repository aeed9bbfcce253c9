#!/bin/bash
# Copy a tools/round4_profile.sh result (gpurun_out/<src>) into profiles/<dst> and rebuild the two derived files from it:
# profiles/in_step.json (tools/step_stats.py, one window for the kernel-stats tables and the in-step averages) and the entries of
# profiles/pmc_traffic.json that point into profiles/<dst> (tools/pmc_json.py: same key / kernel / note / algorithmic bytes, new counters).
# usage: tools/refresh_profiles.sh gpurun_out/r04_y r04_z
src=$1; dst=$2; here=$(cd $(dirname $0)/.. && pwd); cd $here
mkdir -p profiles/$dst
for f in bench_default.json bench_driver_style.json bench_dcn_bwd.txt phase_stem.txt bench_layer1.txt bench_t4_bf16.txt bench_t5.txt bench_t6.txt bench_wg6.txt phase_times.txt pytest_gpu.txt smoke.txt \
         pmc_conv_bf16.txt pmc_conv_f32.txt pmc_dcn_bf16.txt pmc_dcn_f32.txt pmc_dcnbwd_bf16.txt pmc_dcnbwd_f32.txt; do
  [ -s $src/$f ] && grep -v "amdgpu.ids" $src/$f > profiles/$dst/$f
done
python tools/step_stats.py profiles/$dst profiles/$dst/ f32=$src/trace_f32.pkl.gz bf16=$src/trace_bf16.pkl.gz > profiles/in_step.json
mv profiles/$dst/_kernel_stats_f32.txt profiles/$dst/kernel_stats_f32.txt; mv profiles/$dst/_kernel_stats_bf16.txt profiles/$dst/kernel_stats_bf16.txt
python - "$dst" <<'PY'
import json, subprocess, sys
dst = sys.argv[1]
d = json.load(open('profiles/pmc_traffic.json'))
for key, rec in list(d.items()):
    if not isinstance(rec, dict) or not str(rec.get('source', '')).startswith('profiles/%s/' % dst):
        continue
    kern = rec['workload'].split(',')[0].split(' ')[0]
    cmd = ['python', 'tools/pmc_json.py', key, rec['source'], kern, rec['workload'], str(rec['algorithmic_bytes'])]
    if 'expected_mfma_insts' in rec:
        cmd.append(str(rec['expected_mfma_insts']))
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    print('refreshed', key)
PY
