// Shared helpers for the fami HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FAMI_OK 0
#define FAMI_EARG (-1)    // bad argument
#define FAMI_ESHAPE (-2)  // unsupported shape / dtype
#define FAMI_EHIP (-3)    // HIP runtime error, see fami_last_error()

extern "C" void fami_set_error(const char* where, const char* what);

#define FAMI_CHECK_LAUNCH(name)                                   \
  do {                                                            \
    hipError_t e_ = hipGetLastError();                            \
    if (e_ != hipSuccess) {                                       \
      fami_set_error(name, hipGetErrorString(e_));                \
      return FAMI_EHIP;                                           \
    }                                                             \
  } while (0)

#define FAMI_REQUIRE(cond, name, msg)                             \
  do {                                                            \
    if (!(cond)) {                                                \
      fami_set_error(name, msg);                                  \
      return FAMI_EARG;                                           \
    }                                                             \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int fami_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// grid for a grid-stride elementwise kernel: enough blocks to fill 256 CUs x 8.
static inline int fami_ew_grid(long n, int block = 256) {
  long g = (n + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
