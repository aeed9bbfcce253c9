cp fami-pose_amd/libfami_hip.so /tmp/new.so
for v in new u1 new u1; do
  if [ $v = new ]; then cp /tmp/new.so fami-pose_amd/libfami_hip.so; else cp fami-pose_amd/_ab_u1.so fami-pose_amd/libfami_hip.so; fi
  echo "== $v"; python tools/bench_bn.py 2>&1 | grep -v amdgpu.ids | awk 'NR<=4 || (NR>=6 && NR<=9)'
  python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-frozen 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step f32 %.2f ms  bf16 %.2f ms' % (d['ms_per_step'], d['also_bf16']['ms_per_step']))"
done
cp /tmp/new.so fami-pose_amd/libfami_hip.so
