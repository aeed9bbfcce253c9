// The stem's first convolution in the 16-bit modes: 3 -> 64 channels, 3x3, stride 2, pad 1 (posetimation/backbones/hrnet.py: conv1 of
// the stem, 384x288 frames -> 192x144 maps), forward (+ the BatchNorm statistics epilogue of conv_epi.h, mode 1).
// The implicit-GEMM kernel treats a tap as a K step of 32 channels, 3 of them real: nine MFMAs and nine fragment fetches (eight
// 2-byte loads each) per 16 pixels and channel tile for 27 multiply-adds per output -- 75 us per launch for an 85 MB output whose HBM
// floor is 17 us.  Here K is dense over (tap, channel) = 27 -> one K step of 32: a lane gathers its 8 K-values of its pixel with eight
// 2-byte loads ONCE per pixel tile and feeds them to four MFMAs (the 64 output channels); the weights come out of the packed
// fragment image into registers once per wave (lane (n, kq) of tap t holds W[n][ci = 0..2] at element n * 8 + ci of the tap's
// block); a wave walks pixel tiles (16 consecutive output pixels) with two tiles in flight.  The 27 products of an output meet in
// one MFMA instead of nine: fp32 sums in a different order, same operands (tests hold both against fp64).
#include "conv_epi.h"

struct StemArgs {
  EpiBN e;
  int emode;
  const void* x;      // [N,H,W,3]
  const void* wp;     // packed fragment image, mode 0: [tap][1][NTt][64][8]
  void* y;            // [N,Ho,Wo,64]
  const float* bias;
  int N, H, W, Ho, Wo, NTt;
  int P, ntiles;      // output pixels, 16-pixel tiles
  unsigned x_bytes;
};

typedef short stem_s16x8 __attribute__((ext_vector_type(8)));

template <typename H, int EM>
__global__ __launch_bounds__(256) void conv_stem1_fwd_kernel(StemArgs p) {
  typedef typename H16<H>::x8 hx8;
  constexpr int NT = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int W = p.W, HoWo = p.Ho * p.Wo;
  // this lane's 8 K-values: kk = kq * 8 + q -> (tap, ci); kk >= 27 is padding
  int dyq[8], dxq[8], offq[8];
  hx8 wf[NT];
  {
    const unsigned short* wq = reinterpret_cast<const unsigned short*>(p.wp);
    stem_s16x8 t[NT];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int kk = kq * 8 + q;
      const bool ok = kk < 27;
      const int tap = ok ? kk / 3 : 0, ci = ok ? kk - tap * 3 : 0;
      const int ky = tap / 3, kx = tap - ky * 3;
      dyq[q] = ok ? ky - 1 : 0x40000000;               // (never inside the image)
      dxq[q] = kx - 1;
      offq[q] = ((ky - 1) * W + (kx - 1)) * 3 + ci;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) t[nt][q] = ok ? (short)wq[(long)(tap * p.NTt + nt) * 512 + col * 8 + ci] : (short)0;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wf[nt] = __builtin_bit_cast(hx8, t[nt]);
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 bias4[NT], ek[NT], es[NT], eq[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co0 = nt * 16 + kq * 4;
    bias4[nt] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + co0) : z4;
    ek[nt] = (EM == 1 && p.e.pivot_src) ? *reinterpret_cast<const f32x4*>(p.e.pivot_src + co0) : z4;
    es[nt] = eq[nt] = z4;
  }
  auto gather = [&](int t, hx8& a, int& m, bool& valid) {
    m = t * 16 + col;
    valid = t < p.ntiles && m < p.P;
    const int mm = valid ? m : 0;
    const int n = mm / HoWo, r = mm - n * HoWo;
    const int oy = r / p.Wo, ox = r - oy * p.Wo;
    const int base = ((n * p.H + 2 * oy) * W + 2 * ox) * 3;
    stem_s16x8 v;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int iy = 2 * oy + dyq[q], ix = 2 * ox + dxq[q];
      const bool ok = valid && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)W;
      v[q] = (short)__builtin_amdgcn_raw_buffer_load_b16(rx, ok ? (unsigned)(base + offq[q]) * 2u : 0x80000000u, 0, 0);
    }
    a = __builtin_bit_cast(hx8, v);
  };
  auto emit = [&](const hx8& a, int m, bool valid) {
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = H16<H>::mfma(wf[nt], a, z4);
    if (!valid) return;
    H* yp = reinterpret_cast<H*>(p.y) + (long)m * 64 + kq * 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 v = acc[nt] + bias4[nt];
      st4(yp + nt * 16, v);
      if (EM == 1) {
        const f32x4 d = ld4_round<H>(v) - ek[nt];
        es[nt] += d;
        eq[nt] += d * d;
      }
    }
  };
  const int stride = gridDim.x * 4;
  for (int t = blockIdx.x * 4 + wave; t < p.ntiles; t += 2 * stride) {
    hx8 a0, a1;
    int m0, m1;
    bool v0, v1;
    gather(t, a0, m0, v0);
    gather(t + stride, a1, m1, v1);
    emit(a0, m0, v0);
    emit(a1, m1, v1);
  }
  if (EM == 1) {
    __shared__ float red[4][NT * 32];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = row16_sum(es[nt][r]), q = row16_sum(eq[nt][r]);
        if (col == 0) {
          red[wave][nt * 32 + kq * 4 + r] = s;
          red[wave][nt * 32 + 16 + kq * 4 + r] = q;
        }
      }
    __syncthreads();
    if (threadIdx.x < NT * 32) {
      const int tid = threadIdx.x, nt = tid >> 5, st = (tid >> 4) & 1, c16 = tid & 15;
      const int co = nt * 16 + c16;
      const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      const int eC = p.e.C;
      double* srow = p.e.slots + (long)(blockIdx.x % p.e.ns) * 2 * eC;
      unsafeAtomicAdd(srow + st * eC + co, (double)v);
      if (st == 0 && blockIdx.x == 0) bn_slots_pivot(p.e.slots, eC)[co] = p.e.pivot_src ? p.e.pivot_src[co] : 0.f;
    }
  }
}

// f32 storage: the same walk on the exact-f32 matrix instruction (v_mfma_f32_16x16x4_f32: lane (n | pixel, kq) holds ONE K-value per
// step) -- 27 -> seven K steps of 4, kk = 4 step + kq; weights from the f32 fragment image ([tap][1][NTt][64][4]: W[n][ci] of tap t at
// element n * 4 + ci).  The implicit GEMM spends nine steps of its 16-wide K group per pixel tile and 268 us on the launch (170 MB of
// output: floor 34 us).
template <int EM>
__global__ __launch_bounds__(256) void conv_stem1_fwd_f32_kernel(StemArgs p) {
  constexpr int NT = 4, NS = 7;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int W = p.W, HoWo = p.Ho * p.Wo;
  int dyq[NS], dxq[NS], offq[NS];
  float wf[NT][NS];
  {
    const float* wq = reinterpret_cast<const float*>(p.wp);
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const int kk = q * 4 + kq;
      const bool ok = kk < 27;
      const int tap = ok ? kk / 3 : 0, ci = ok ? kk - tap * 3 : 0;
      const int ky = tap / 3, kx = tap - ky * 3;
      dyq[q] = ok ? ky - 1 : 0x40000000;
      dxq[q] = kx - 1;
      offq[q] = ((ky - 1) * W + (kx - 1)) * 3 + ci;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wf[nt][q] = ok ? wq[(long)(tap * p.NTt + nt) * 256 + col * 4 + ci] : 0.f;
    }
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 bias4[NT], ek[NT], es[NT], eq[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co0 = nt * 16 + kq * 4;
    bias4[nt] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + co0) : z4;
    ek[nt] = (EM == 1 && p.e.pivot_src) ? *reinterpret_cast<const f32x4*>(p.e.pivot_src + co0) : z4;
    es[nt] = eq[nt] = z4;
  }
  auto gather = [&](int t, float (&a)[NS], int& m, bool& valid) {
    m = t * 16 + col;
    valid = t < p.ntiles && m < p.P;
    const int mm = valid ? m : 0;
    const int n = mm / HoWo, r = mm - n * HoWo;
    const int oy = r / p.Wo, ox = r - oy * p.Wo;
    const int base = ((n * p.H + 2 * oy) * W + 2 * ox) * 3;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const int iy = 2 * oy + dyq[q], ix = 2 * ox + dxq[q];
      const bool ok = valid && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)W;
      a[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? (unsigned)(base + offq[q]) * 4u : 0x80000000u, 0, 0));
    }
  };
  auto emit = [&](const float (&a)[NS], int m, bool valid) {
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc[nt] = z4;
#pragma unroll
      for (int q = 0; q < NS; ++q) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][q], a[q], acc[nt], 0, 0, 0);
    }
    if (!valid) return;
    float* yp = reinterpret_cast<float*>(p.y) + (long)m * 64 + kq * 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 v = acc[nt] + bias4[nt];
      st4(yp + nt * 16, v);
      if (EM == 1) {
        const f32x4 d = v - ek[nt];
        es[nt] += d;
        eq[nt] += d * d;
      }
    }
  };
  const int stride = gridDim.x * 4;
  for (int t = blockIdx.x * 4 + wave; t < p.ntiles; t += 2 * stride) {
    float a0[NS], a1[NS];
    int m0, m1;
    bool v0, v1;
    gather(t, a0, m0, v0);
    gather(t + stride, a1, m1, v1);
    emit(a0, m0, v0);
    emit(a1, m1, v1);
  }
  if (EM == 1) {
    __shared__ float red[4][NT * 32];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = row16_sum(es[nt][r]), q = row16_sum(eq[nt][r]);
        if (col == 0) {
          red[wave][nt * 32 + kq * 4 + r] = s;
          red[wave][nt * 32 + 16 + kq * 4 + r] = q;
        }
      }
    __syncthreads();
    if (threadIdx.x < NT * 32) {
      const int tid = threadIdx.x, nt = tid >> 5, st = (tid >> 4) & 1, c16 = tid & 15;
      const int co = nt * 16 + c16;
      const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      const int eC = p.e.C;
      double* srow = p.e.slots + (long)(blockIdx.x % p.e.ns) * 2 * eC;
      unsafeAtomicAdd(srow + st * eC + co, (double)v);
      if (st == 0 && blockIdx.x == 0) bn_slots_pivot(p.e.slots, eC)[co] = p.e.pivot_src ? p.e.pivot_src[co] : 0.f;
    }
  }
}

// [fami_route_t] g_stem1 (default 1)  // fami_conv_tune_lds(9000 / 9001): off / on
extern "C" void fami_conv_stem_tune(int on) { g_stem1 = on < 0 ? 1 : (on ? 1 : 0); }      // (declared inside conv.hip's extern "C" block)

// Returns 1 if launched, 0 if the shape is not this kernel's, < 0 on error.  half_kind: 0 bf16, 1 fp16, 2 f32 (out_f32 is then the storage type).
int fami_try_conv_stem1(int half_kind, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci, int Co,
                        int kh, int kw, int stride, int pad, int dil, int NTt, int relu, int accumulate, int out_f32, hipStream_t s,
                        const char* name, const EpiBN& epi) {
  if (!g_stem1 || Ci != 3 || Co != 64 || kh != 3 || kw != 3 || stride != 2 || pad != 1 || dil != 1 || relu || accumulate || (out_f32 && half_kind != 2)) return 0;
  const int emode = epi.slots ? epi.mode : 0;
  if (emode != 0 && emode != 1) return 0;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long P = (long)N * Ho * Wo, xb = (long)N * H * W * 3 * (half_kind == 2 ? 4 : 2);
  if (P >= (1L << 27) || xb >= (1L << 31)) return 0;
  StemArgs a;
  a.e = epi; a.emode = emode; a.x = x; a.wp = wp; a.y = y; a.bias = bias;
  a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.NTt = NTt; a.P = (int)P; a.ntiles = (int)((P + 15) / 16); a.x_bytes = (unsigned)xb;
  int g = (a.ntiles + 7) / 8;          // two tiles per wave and trip at least
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
#define FAMI_STEM_CASE(HT, em) hipLaunchKernelGGL((conv_stem1_fwd_kernel<HT, em>), dim3(g), dim3(256), 0, s, a)
  if (half_kind == 2) {
    if (emode) hipLaunchKernelGGL((conv_stem1_fwd_f32_kernel<1>), dim3(g), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_stem1_fwd_f32_kernel<0>), dim3(g), dim3(256), 0, s, a);
  } else if (half_kind == 1) { if (emode) FAMI_STEM_CASE(f16_t, 1); else FAMI_STEM_CASE(f16_t, 0); }
  else { if (emode) FAMI_STEM_CASE(bf16_t, 1); else FAMI_STEM_CASE(bf16_t, 0); }
#undef FAMI_STEM_CASE
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    fami_set_error(name, hipGetErrorString(err));
    return FAMI_EHIP;
  }
  return 1;
}
