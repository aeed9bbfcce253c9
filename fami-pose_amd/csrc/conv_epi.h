// Shared by the convolution translation units (conv.hip, conv_t4.hip): the optional BatchNorm work of an epilogue.
#pragma once
#include "common.h"

// Optional BatchNorm work in a convolution's epilogue (reference semantics
// posetimation/layers/basic_model.py:34-63: every conv is followed by a train-mode BatchNorm).  The output tile is in
// registers anyway, so the statistics pass over the tensor -- one launch and one HBM read per BatchNorm, forward and
// backward -- is folded in: per-channel partial sums of the workgroup's pixels go into the fp64 slot rows of the
// two-launch BatchNorm (common.h; norm.hip's apply passes fold the rows in their prologue), by global_atomic_add_f64.
//   mode 1 (forward conv -> BN):  sum (y - K), sum (y - K)^2 of the values as stored; K[c] = pivot_src[c] (the running
//                                 mean: any value near the mean conditions the variance) or 0; the first pixel tile
//                                 stores K behind the slot rows for the consumer.
//   mode 2 (input gradient -> the BN that produced this conv's input):  the epilogue holds dL/d(BN output) complete
//                                 (this launch is its last contribution), so dz = relu-mask(dy) is stored instead of dy
//                                 and sum dz, sum dz*xhat are taken; mask from the BN output (relu 1) or recomputed from
//                                 its input exactly as the forward apply pass computes it (relu 2).
struct EpiBN {
  double* slots;           // null: plain epilogue
  int ns, mode, relu, C;   // C = channels of the output tensor (row length of the slot rows)
  const float* pivot_src;  // mode 1
  const void* z;           // mode 2: BN input  [P][C] (activation storage type)
  const void* yr;          // mode 2, relu 1: BN output
  const float *mean, *invstd, *gamma, *beta;
};
static inline EpiBN epi_none() {
  EpiBN e;
  e.slots = nullptr; e.ns = 1; e.mode = 0; e.relu = 0; e.C = 0; e.pivot_src = nullptr; e.z = nullptr; e.yr = nullptr;
  e.mean = e.invstd = e.gamma = e.beta = nullptr;
  return e;
}
// The EpiBN block of a kernel's argument struct, read in the EPILOGUE through the kernarg segment pointer instead of
// through the by-value parameter: the compiler preloads every referenced kernel argument into SGPRs at the top of the
// kernel and keeps it there (+28 SGPRs through the main loop, one resident workgroup per CU less on the dgrad form).
// The empty asm makes the pointer opaque, so the loads cannot be hoisted above it.  `off` = offsetof(Args, e) (the
// argument struct is the kernel's only parameter: it starts at byte 0 of the segment).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) EpiBN* EpiPtr;   // constant address space: the field reads stay scalar loads
__device__ __forceinline__ EpiPtr epi_late(unsigned off) {
  const __attribute__((address_space(4))) char* k =
      (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + off;
  asm volatile("" : "+s"(k) : : "memory");
  return (EpiPtr)k;
}
#else
typedef const EpiBN* EpiPtr;
__device__ inline EpiPtr epi_late(unsigned) { return nullptr; }
#endif

// the forward apply pass's scale / shift (norm.hip bn_scale_shift): the recomputed ReLU mask must match it bit for bit
__device__ __forceinline__ void epi_scale_shift(float mean, float invstd, float gamma, float beta, float& sc, float& sf) {
  sc = invstd * gamma;
  sf = __builtin_fmaf(-mean, sc, beta);
}

// v rounded to storage type H and widened again (what a later pass over the stored tensor would read)
template <typename H>
__device__ __forceinline__ f32x4 ld4_round(f32x4 v) {
  if constexpr (sizeof(H) == 4) return v;
  else return __builtin_convertvector(__builtin_convertvector(v, H __attribute__((ext_vector_type(4)))), f32x4);
}


// Optional train-mode BatchNorm (+ReLU) applied to a convolution's INPUT while it is staged (conv_t4.hip forward,
// conv_wg16.hip weight gradient): y = relu(fma(z, sc, sf)) with sc = invstd*gamma, sf = fma(-mean, sc, beta) -- the
// arithmetic of norm.hip's apply pass, so the normalised tensor the reference materialises between conv1 and conv2 of a
// BasicBlock (basic_model.py:34-63) never exists in HBM.  Forward (slots != null): mean / invstd are folded from the slot
// rows the producing convolution's epilogue filled; workgroup (0, 0) stores them and advances the running statistics.
// Weight gradient (slots == null): mean / invstd are read.
struct XBN {
  const double* slots;   // forward: statistics of z (EpiBN mode 1 rows + pivots); null: use mean / invstd as they are
  const float *gamma, *beta;
  float *mean, *invstd, *running_mean, *running_var;
  long P;                // pixels the statistics were taken over
  float momentum, eps;
  int C, ns, on;         // channels of z, slot rows, 0 = no transform
  void* out;             // round 6 (conv_t6.hip XB instances): the normalised tensor is ALSO written here (null: never materialised)
};
static inline XBN xbn_none() {
  XBN x;
  x.slots = nullptr; x.gamma = x.beta = nullptr; x.mean = x.invstd = x.running_mean = x.running_var = nullptr;
  x.P = 0; x.momentum = 0.f; x.eps = 0.f; x.C = 0; x.ns = 0; x.on = 0; x.out = nullptr;
  return x;
}
// scale / shift of channel c into (sc, sf); forward form also publishes mean / invstd / running statistics when `publish`
__device__ __forceinline__ void xbn_channel(const XBN& x, int c, bool publish, float& sc, float& sf) {
  float muf, isf;
  if (x.slots) {
    double s = 0.0, q = 0.0;
    for (int k = 0; k < x.ns; ++k) {
      s += x.slots[(long)k * 2 * x.C + c];
      q += x.slots[(long)k * 2 * x.C + x.C + c];
    }
    const double invP = 1.0 / (double)x.P;
    const double dm = s * invP;
    double var = q * invP - dm * dm;
    if (var < 0.0) var = 0.0;
    const double mu = (double)bn_slots_pivot(const_cast<double*>(x.slots), x.C)[c] + dm;
    muf = (float)mu;
    isf = (float)(1.0 / sqrt(var + (double)x.eps));
    if (publish) {
      x.mean[c] = muf;
      x.invstd[c] = isf;
      if (x.running_mean) {
        const double unb = x.P > 1 ? var * (double)x.P / (double)(x.P - 1) : var;
        x.running_mean[c] = (float)((1.0 - x.momentum) * x.running_mean[c] + x.momentum * mu);
        x.running_var[c] = (float)((1.0 - x.momentum) * x.running_var[c] + x.momentum * unb);
      }
    }
  } else {
    muf = x.mean[c];
    isf = x.invstd[c];
  }
  epi_scale_shift(muf, isf, x.gamma[c], x.beta[c], sc, sf);
}
// 8 staged 16-bit values (one 16-byte piece, channels c0 .. c0+7) -> relu(fma(v, sc, sf)), tables in LDS
template <typename H>
__device__ __forceinline__ u32x4 xbn_piece(u32x4 raw, const float* sc, const float* sf) {
  typedef H hx8 __attribute__((ext_vector_type(8)));
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  f32x8 v = __builtin_convertvector(__builtin_bit_cast(hx8, raw), f32x8);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = fmaxf(__builtin_fmaf(v[j], sc[j], sf[j]), 0.f);
  return __builtin_bit_cast(u32x4, __builtin_convertvector(v, hx8));
}

// conv_t4.hip: register-blocked LDS 3x3 kernel for the 16-bit storage types.  half_kind: 0 bf16, 1 fp16, 2 f32.
// Returns 1 if launched, 0 if the shape is not eligible, < 0 on error.
int fami_try_conv3x3_t4(int half_kind, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci,
                        int Co, int KC, int NTt, int sgn, int relu, int accumulate, int out_f32, hipStream_t s,
                        const char* name, const EpiBN& epi, const XBN& xbn = xbn_none());
int fami_try_conv3x3_t4_dil(int half_kind, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci,
                            int Co, int KC, int NTt, int sgn, int relu, int accumulate, int out_f32, int dil, hipStream_t s,
                            const char* name);
void fami_conv_t4_tune(int on);
void fami_conv_t4_default_split(int on);
int fami_conv_t4_eligible16(int N, int H, int W, int Ci, int Co);
int fami_conv_t4_eligible_s3(int N, int H, int W, int Ci, int Co);

// conv_wg16.hip: 16-bit weight gradient of the centred k x k convolutions (k = 1 | 3, stride 1 | 2, any dilation).  fami_try_wgrad16 -> number of partial
// slabs [G][9][Ci][Co] written to `part` (reduce them with the caller's slab reduce), 0 = not eligible, < 0 = error.
long fami_wgrad16_slabs(int N, int H, int W, int Ci, int Co, int k, int st, int pad, int dil);
int fami_try_wgrad16(int half_kind, const void* x, const void* dy, float* part, long ws_bytes, int N, int H, int W, int Ci,
                     int Co, int k, int st, int pad, int dil, hipStream_t s, const char* name, const XBN& xbn = xbn_none());
void fami_wgrad16_tune(int on);

// conv_wgs3.hip: f32 weight gradient of the 3x3 stride-1 pad-1 convolutions on the bf16 matrix pipe (operands split into
// three bf16 terms, six products, fp32 accumulation).  Same contract as fami_try_wgrad16.
long fami_wgrad_s3_slabs(int N, int H, int W, int Ci, int Co);
int fami_try_wgrad_s3(const float* x, const float* dy, float* part, long ws_bytes, int N, int H, int W, int Ci, int Co,
                      hipStream_t s, const char* name, const XBN& xbn = xbn_none());
void fami_wgrad_s3_tune(int on);
void fami_wgrad_s3_default(int on);

// conv_t5.hip: persistent, unit-pipelined form of the 3x3 stride-1 kernels (round 4) and the pre-split f32 weight image it
// reads.  The split image follows the f32 fragment images of a 3x3 convolution with K % 16 == 0:
//   [tap][K/16][N/16][plane 3][n 16][k 16] bf16 -- w = plane0 + plane1 + plane2 exactly (the split of conv_t4.hip's t4_split)
// fami_split_image_elems -> its size in floats (0: no split image for this geometry).
long fami_split_image_elems(int kd, int nd, int taps);
void fami_pack_split_single(const float* w_oihw, float* split, int Co, int Ci, int taps, int mode, hipStream_t s);
void fami_pack_split_batch(const float* params, float* packed, const void* desc, int n, hipStream_t s);
int fami_try_conv3x3_t5(int half_kind, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci,
                        int Co, int KC, int NTt, int sgn, int relu, int accumulate, int out_f32, hipStream_t s,
                        const char* name, const EpiBN& epi, const XBN& xbn);
void fami_conv_t5_tune(int on);
int fami_conv_t5_eligible_s3(int N, int H, int W, int Ci, int Co);

// conv_t6.hip: weight-resident, DMA-staged 3x3 stride-1 kernel for the 16-bit types (round 4; 48 input channels, rows of 64 / 72 pixels).
// Same contract as fami_try_conv3x3_t4; no input BatchNorm (XBN) and no backward-statistics epilogue (EpiBN mode 2): returns 0 for those.
int fami_try_conv3x3_t6(int half_kind, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci,
                        int Co, int KC, int NTt, int sgn, int relu, int accumulate, int out_f32, hipStream_t s,
                        const char* name, const EpiBN& epi, const XBN& xbn);
extern "C" int fami_conv_t6_eligible(int N, int H, int W, int Ci, int Co);
void fami_conv_t6_tune(int on);
