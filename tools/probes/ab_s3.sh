cp fami-pose_amd/libfami_hip.so /tmp/orig.so
for v in e1a0 e0a1 e0a0; do
  cp fami-pose_amd/_ab_$v.so fami-pose_amd/libfami_hip.so
  echo "== $v"
  python tools/probes/s3_check.py 2>&1 | grep "fp64"
  DT=f32 python tools/bench_t4.py 2>&1 | grep -v amdgpu.ids | cut -c1-20,80-260 | head -4
  python tools/ab_step.py f32 lds=31 lds=131 2>&1 | tail -2 | cut -c1-120
done
cp /tmp/orig.so fami-pose_amd/libfami_hip.so
