// Probe: register-only MFMA rate on this box (f32 16x16x4 and bf16 16x16x32), wpc waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  bf16x8 ah, bh;
  for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(a + i); bh[i] = (__bf16)(b - i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 4 * 256 * 8192);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int wgs_per_cu = 1; wgs_per_cu <= 8; wgs_per_cu *= 2) {
      const int iters = mode == 0 ? 4000 : 16000, grid = 256 * wgs_per_cu;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) k<0><<<grid, 256>>>(d, iters); else k<1><<<grid, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)grid * 4 * iters * 8 * (mode == 0 ? 2048.0 : 16384.0);
      printf("%s waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", mode == 0 ? "f32 16x16x4 " : "bf16 16x16x32", wgs_per_cu, ms, flop / ms / 1e9);
    }
  return 0;
}
