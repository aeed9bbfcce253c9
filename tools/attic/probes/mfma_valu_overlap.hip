// Probe: does VALU work hide under f32-input MFMAs (v_mfma_f32_16x16x4_f32 / 32x32x2) the way it does under bf16 MFMAs?
// Per loop iteration: NM independent MFMAs and NV independent v_fma_f32 (one wave per SIMD, or two).  If the pipes are
// separate, time(NM, NV) ~ max(time(NM, 0), time(0, NV)); if the f32 matrix op runs on the vector ALUs, it is the sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND, int NM, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  bf16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(a + i); hb[i] = (__bf16)(b + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        if (KIND == 0) acc[(r * NM + i) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[(r * NM + i) & 7], 0, 0, 0);
        else acc[(r * NM + i) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[(r * NM + i) & 7], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(r * NV + i) & 15]) : "v"(b), "v"(a));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> void run(const char* name, F launch, int nm, int nv, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 1; w <= 2; w *= 2) {
    const int grid = 256 * w;
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); launch(grid, iters); hipEventRecord(e1); hipEventSynchronize(e1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_iter_ns = ms * 1e6 / iters / 8 / w;   // per (NM MFMA + NV VALU) group per wave on the SIMD
    printf("%-28s waves/SIMD=%d  %.3f ms  %.1f ns per group (%d MFMA + %d VALU)\n", name, w, ms, per_iter_ns, nm, nv);
  }
}
#define RUN(KIND, NM, NV) run(#KIND " " #NM " mfma " #NV " valu", [&](int g, int it) { k<KIND, NM, NV><<<g, 256>>>(d, it); }, NM, NV, 4000)
int main() {
  float* d; hipMalloc(&d, 4 * 256 * 8192);
  RUN(0, 1, 0); RUN(0, 0, 4); RUN(0, 0, 8); RUN(0, 1, 4); RUN(0, 1, 8); RUN(0, 1, 16);
  RUN(1, 1, 0); RUN(1, 1, 2); RUN(1, 1, 4); RUN(1, 1, 8);
  return 0;
}
