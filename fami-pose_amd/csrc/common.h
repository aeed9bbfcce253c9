// Shared helpers for the fami HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "route.h"

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
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- storage types: activations are fp32, bf16 or fp16 (fp32 arithmetic either way) ----------------------
typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 4-element vector load/store with conversion to/from f32 (bf16: one 8-byte access, v_cvt_pk_bf16_f32 RNE)
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const bf16_t* p) {
  return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(p), f32x4);
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, f32x4 v) {
  *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(v, bf16x4);
}
__device__ __forceinline__ f32x4 ld4(const f16_t* p) {
  return __builtin_convertvector(*reinterpret_cast<const f16x4*>(p), f32x4);
}
__device__ __forceinline__ void st4(f16_t* p, f32x4 v) {
  *reinterpret_cast<f16x4*>(p) = __builtin_convertvector(v, f16x4);
}
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 ld2(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
__device__ __forceinline__ f32x2 ld2(const bf16_t* p) {
  return __builtin_convertvector(*reinterpret_cast<const bf16x2*>(p), f32x2);
}
__device__ __forceinline__ void st2(float* p, f32x2 v) { *reinterpret_cast<f32x2*>(p) = v; }
__device__ __forceinline__ void st2(bf16_t* p, f32x2 v) {
  *reinterpret_cast<bf16x2*>(p) = __builtin_convertvector(v, bf16x2);
}
__device__ __forceinline__ f32x2 ld2(const f16_t* p) {
  return __builtin_convertvector(*reinterpret_cast<const f16x2*>(p), f32x2);
}
__device__ __forceinline__ void st2(f16_t* p, f32x2 v) {
  *reinterpret_cast<f16x2*>(p) = __builtin_convertvector(v, f16x2);
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = (bf16_t)v; }
__device__ __forceinline__ float ld1(const f16_t* p) { return (float)*p; }
__device__ __forceinline__ void st1(f16_t* p, float v) { *p = (f16_t)v; }

// the two 16-bit storage types on the 16x16x32 matrix-core instruction (8 K-elements per lane, fp32 accumulation)
template <typename H> struct H16;
template <> struct H16<bf16_t> {
  typedef bf16x8 x8;
  __device__ static __forceinline__ f32x4 mfma(x8 a, x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  // one 16-bit pattern -> f32
  __device__ static __forceinline__ float from_bits(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
};
template <> struct H16<f16_t> {
  typedef f16x8 x8;
  __device__ static __forceinline__ f32x4 mfma(x8 a, x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ float from_bits(unsigned short u) { return (float)__builtin_bit_cast(f16_t, u); }
};

// stamps the two C-ABI instances of an entry point whose body is a template over the activation type
#define FAMI_DTYPE_NAME_f32 float
#define FAMI_DTYPE_NAME_bf16 bf16_t
#define FAMI_DTYPE_NAME_f16 f16_t

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
// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4); every lane of the row ends with the total.
// quad_perm xor 1, quad_perm xor 2, row_half_mirror, row_mirror: four v_add_f32 with DPP modifiers, no LDS traffic.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
  return v;
}

// ---- BatchNorm slot rows (norm.hip two-launch forms; filled by the convolution epilogues of conv.hip when fused) ----
// slots = [BN_NS_MAX][2][C] fp64 sums (zero on entry) followed by [C] fp32 pivots (the value the forward sums are shifted
// by; written by the producer, so every consumer workgroup reads the same one).
#define BN_NS_MAX 8
static inline int bn_slots(int C) { return C <= 96 ? 8 : (C <= 192 ? 4 : 2); }
static inline long bn_slots_bytes(int C) { return (long)BN_NS_MAX * 2 * C * 8 + (long)((C + 3) & ~3) * 4; }
__host__ __device__ static inline float* bn_slots_pivot(void* slots, int C) {
  return reinterpret_cast<float*>(reinterpret_cast<double*>(slots) + (long)BN_NS_MAX * 2 * C);
}

// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own 4 MB L2:
// with the natural order, neighbouring pixel tiles -- which share their 3x3 halo rows -- sit in 8 different L2s and
// every input row is fetched from the fabric by several of them.  This remap hands XCD x the x-th contiguous eighth
// of the tile sequence (all output-channel blocks of a pixel tile adjacent), so halo re-fetch happens only at the
// 7 chunk borders.  bx/by: the logical (pixel tile, channel block) of this workgroup.
__device__ __forceinline__ void xcd_tile(int on, int& bx, int& by) {
  if (!on) {
    bx = blockIdx.x;
    by = blockIdx.y;
    return;
  }
  const int gx = gridDim.x, gy = gridDim.y;
  const int n = gx * gy;
  const int lin = blockIdx.y * gx + blockIdx.x;
  const int q = n >> 3, r = n & 7;
  const int x = lin & 7, l = lin >> 3;
  const int t = x * q + (x < r ? x : r) + l;  // position in the XCD-contiguous order
  bx = t / gy;
  by = t - bx * gy;
}

