// Probe: register-only f32 MFMA rate, 16x16x4 vs 32x32x2, by waves per SIMD and accumulator count.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> void run(const char* name, F launch, double flop_per_iter_per_wave, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 1; w <= 8; w *= 2) {
    const int grid = 256 * w;
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); launch(grid, iters); hipEventRecord(e1); hipEventSynchronize(e1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s waves/SIMD=%d %.3f ms %.1f TFLOP/s\n", name, w, ms, (double)grid * 4 * iters * flop_per_iter_per_wave / ms / 1e9);
  }
}
int main() {
  float* d; hipMalloc(&d, 4 * 256 * 8192);
  run("16x16x4 acc=4", [&](int g, int it) { k16<4><<<g, 256>>>(d, it); }, 4 * 2048.0, 8000);
  run("16x16x4 acc=8", [&](int g, int it) { k16<8><<<g, 256>>>(d, it); }, 8 * 2048.0, 4000);
  run("32x32x2 acc=2", [&](int g, int it) { k32<2><<<g, 256>>>(d, it); }, 2 * 4096.0, 8000);
  run("32x32x2 acc=4", [&](int g, int it) { k32<4><<<g, 256>>>(d, it); }, 4 * 4096.0, 4000);
  return 0;
}
