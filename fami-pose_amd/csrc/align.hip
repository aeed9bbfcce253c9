// Temporal alignment kernels: the global bilinear shift and the modulated
// deformable convolution (DCNv2), NHWC fp32.  Both are gather kernels bound by
// HBM traffic; activations are NHWC so the cg channels of one offset group at
// one bilinear corner are contiguous (16 bytes for cg = 4).
//
// Replaces
//   kornia.geometry.warp_affine with M = [[1,0,tx],[0,1,ty]]   (Alignment_V15.py:133-135)
//   torchvision.ops.DeformConv2d(48,48,3,padding=3,dilation=3) (Alignment_V15.py:83,89,95,101;
//   calls :146,150,154,158), offset groups = offset.shape[1]/18, raw (un-sigmoided) masks.
//
// DCN forward: a workgroup owns 16 output pixels; offsets and masks (77 % of the algorithmic bytes) are streamed
// with fully coalesced loads, every bilinear corner is one 16-byte load, and the modulated samples are contracted
// with the weights on v_mfma_f32_16x16x4_f32 -- the sampled "column" never exists in HBM, so the traffic is
// input + offsets + masks + output (the algorithmic bytes of SURVEY.md 8d).  dcn_fwd_direct_kernel (default)
// feeds the samples to the MFMA from the registers of the lane that gathered them; dcn_fwd_kernel (fallback,
// A/B partner) stages them in an LDS column tile first.
#include "common.h"
#include <type_traits>

// ------------------------------------------------------------------ bilinear shift
template <typename T>
__global__ __launch_bounds__(256) void shift_fwd_kernel(const T* __restrict__ src, const float* __restrict__ t,
                                                        T* __restrict__ out, int B, int H, int W, int C) {
  const int CV = C >> 2;
  const long total = (long)B * H * W * CV;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    const float py = (float)y - t[b * 2 + 1], px = (float)x - t[b * 2 + 0];
    const float fy = floorf(py), fx = floorf(px);
    const float ly = py - fy, lx = px - fx, hy = 1.f - ly, hx = 1.f - lx;
    const int y0 = (int)fy, x0 = (int)fx;
    const T* base = src + (long)b * H * W * C + cv * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
      const float w = ((k >> 1) ? ly : hy) * ((k & 1) ? lx : hx);
      if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) acc += ld4(base + ((long)yy * W + xx) * C) * w;
    }
    st4(out + i * 4, acc);
  }
}

// grad wrt src, gather form (deterministic): every output pixel that touches (ys,xs) re-evaluates the
// forward weights exactly as shift_fwd_kernel does.
template <typename T>
__global__ __launch_bounds__(256) void shift_bwd_src_kernel(const T* __restrict__ gout,
                                                            const float* __restrict__ t, T* __restrict__ gsrc,
                                                            int B, int H, int W, int C, int accumulate) {
  const int CV = C >> 2;
  const long total = (long)B * H * W * CV;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int xs = (int)(p % W);
    p /= W;
    const int ys = (int)(p % H);
    const int b = (int)(p / H);
    const float ty = t[b * 2 + 1], tx = t[b * 2 + 0];
    const int yb = (int)floorf((float)ys + ty) - 1, xb = (int)floorf((float)xs + tx) - 1;
    const T* base = gout + (long)b * H * W * C + cv * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < 4; ++a) {
      const int y = yb + a;
      if ((unsigned)y >= (unsigned)H) continue;
      const float py = (float)y - ty;
      const float fy = floorf(py);
      const float ly = py - fy;
      const int y0 = (int)fy;
      float wy;
      if (y0 == ys) wy = 1.f - ly;
      else if (y0 + 1 == ys) wy = ly;
      else continue;
      for (int c = 0; c < 4; ++c) {
        const int x = xb + c;
        if ((unsigned)x >= (unsigned)W) continue;
        const float px = (float)x - tx;
        const float fx = floorf(px);
        const float lx = px - fx;
        const int x0 = (int)fx;
        float wx;
        if (x0 == xs) wx = 1.f - lx;
        else if (x0 + 1 == xs) wx = lx;
        else continue;
        acc += ld4(base + ((long)y * W + x) * C) * (wy * wx);
      }
    }
    if (accumulate) acc += ld4(gsrc + i * 4);
    st4(gsrc + i * 4, acc);
  }
}

// grad wrt (tx,ty): partial[b][blk][2]
template <typename T>
__global__ __launch_bounds__(256) void shift_bwd_t_kernel(const T* __restrict__ gout,
                                                          const T* __restrict__ src,
                                                          const float* __restrict__ t, float* __restrict__ partial,
                                                          int H, int W, int C) {
  __shared__ float red[2][4];
  const int b = blockIdx.y;
  const int CV = C >> 2;
  const long total = (long)H * W * CV;
  const float ty = t[b * 2 + 1], tx = t[b * 2 + 0];
  const T* sb = src + (long)b * H * W * C;
  const T* gb = gout + (long)b * H * W * C;
  float gx = 0.f, gy = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int x = (int)(p % W);
    const int y = (int)(p / W);
    const float py = (float)y - ty, px = (float)x - tx;
    const float fy = floorf(py), fx = floorf(px);
    const float ly = py - fy, lx = px - fx, hy = 1.f - ly, hx = 1.f - lx;
    const int y0 = (int)fy, x0 = (int)fx;
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
      v[k] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                 ? ld4(sb + ((long)yy * W + xx) * C + cv * 4)
                 : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const f32x4 g = ld4(gb + i * 4);
    const f32x4 dpy = hx * (v[2] - v[0]) + lx * (v[3] - v[1]);  // d out / d py
    const f32x4 dpx = hy * (v[1] - v[0]) + ly * (v[3] - v[2]);  // d out / d px
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gy -= g[k] * dpy[k];  // py = y - ty
      gx -= g[k] * dpx[k];
    }
  }
  gx = wave_sum(gx);
  gy = wave_sum(gy);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = gx;
    red[1][wave] = gy;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = partial + ((long)b * gridDim.x + blockIdx.x) * 2;
    o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}
__global__ void shift_bwd_t_finalize_kernel(const float* __restrict__ partial, int G, int B, float* gt,
                                            int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 2) return;
  const int b = i >> 1, k = i & 1;
  double s = 0.0;
  for (int g = 0; g < G; ++g) s += (double)partial[((long)b * G + g) * 2 + k];
  gt[i] = accumulate ? gt[i] + (float)s : (float)s;
}

// ------------------------------------------------------------------ DCN
template <typename T>
struct DcnArgs {
  const T* x;         // [B,H,W,C]
  const T* off;       // [B,Ho,Wo,2*G*K]
  const T* msk;       // [B,Ho,Wo,G*K]   (may be null => mask 1)
  const float* wp;    // packed [KS][NTt][64][4]
  const void* wp16;   // 16-bit image [ceil(KS/2)][NTt][64][8] (16-bit modes: contraction on the 16x16x32 matrix core)
  const float* bias;  // [Co] or null
  T* y;               // [B,Ho,Wo,Co]
  int B, H, W, C, Ho, Wo, Co, G, kh, kw, stride, pad, dil;
  int cg, KS, NTt, P;  // KS = ceil(C*K / 16): 16-wide k groups of the contraction
  int ostr, mstr;      // elements per pixel of the offset / mask tensors: 2GK / GK (two dense tensors) or 3GK / 3GK (ONE tensor
                       // [P][2GK offsets | GK masks], the output of the merged predictor: msk == off + 2GK)
};

// Column index of the contraction, TAP-major: kidx = ((tap*G + g)*q4 + q)*4 + c4  (channel c = g*cg + q*4 + c4,
// q4 = cg/4), KS16 = ceil(C*K / 16).  Tap-major so that the 16 sample blocks a workgroup gathers at a time are the
// G groups of one or two taps: together they consume whole NHWC pixels (all C channels of each touched pixel, i.e.
// whole cache lines) at positions that neighbouring output pixels share, instead of a 16-byte slice of every pixel
// of the window per step -- with the group-major order the window (~30 KB per workgroup in f32) was re-fetched from
// L2 on every step.
// packed[((ks16*NTt + nt)*64 + lane)*4 + t] = W[co = nt*16 + (lane&15)][kidx = (ks16*4 + (lane>>4))*4 + t]:
// lane (row, kq) of the A operand holds ONE float4 = kidx (ks16*4 + kq)*4 .. +3 of its pixel and feeds it to 4
// consecutive MFMAs; the weight image carries the same K permutation.
__global__ void dcn_pack_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int C, int K, int cg,
                                  int KS16, int NTt) {
  const long total = (long)KS16 * NTt * 256;
  const int CK = C * K, G = C / cg, q4 = cg >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const long r = i >> 8;
    const int nt = (int)(r % NTt), ks16 = (int)(r / NTt);
    const int kidx = (ks16 * 4 + (lane >> 4)) * 4 + t;
    const int co = nt * 16 + (lane & 15);
    float v = 0.f;
    if (kidx < CK && co < Co) {
      const int item = kidx >> 2, q = item % q4, tg = item / q4;
      const int g = tg % G, tap = tg / G;
      v = w[((long)co * C + g * cg + q * 4 + t) * K + tap];
    }
    wp[i] = v;
  }
}

// 16-bit image of the same matrix for the 16x16x32 matrix-core instruction (bf16 / fp16 modes): a lane's 8 K-values are
// its sample block of 16-step 2s (j < 4) and of 16-step 2s+1 (j >= 4) -- the two blocks the lane gathers per 32-step.
template <typename H>
__global__ void dcn_pack_w16_kernel(const float* __restrict__ w, H* __restrict__ wp, int Co, int C, int K, int cg, int KS32,
                                    int NTt) {
  const long total = (long)KS32 * NTt * 512;
  const int CK = C * K, G = C / cg, q4 = cg >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long r = i >> 9;
    const int nt = (int)(r % NTt), s32 = (int)(r / NTt);
    const int kidx = ((2 * s32 + (j >> 2)) * 4 + (lane >> 4)) * 4 + (j & 3);
    const int co = nt * 16 + (lane & 15);
    float v = 0.f;
    if (kidx < CK && co < Co) {
      const int item = kidx >> 2, q = item % q4, tg = item / q4;
      const int g = tg % G, tap = tg / G;
      v = w[((long)co * C + g * cg + q * 4 + (j & 3)) * K + tap];
    }
    wp[i] = (H)v;
  }
}

#define DCN_PIX 16  // output pixels per workgroup (= one MFMA row tile)

// a wave-uniform float kept in an SGPR (the scalar unit has no int->float conversion)
__device__ __forceinline__ float sgpr_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// Tail shared by the two forward kernels: the 4 waves hold partial [16, Co] tiles over their K slices; they meet in
// LDS, bias is added and the block is stored as one contiguous run.
template <typename T, int NT>
__device__ __forceinline__ void dcn_reduce_store(const DcnArgs<T>& p, const f32x4 (&acc)[NT], float* red, int m0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // D row = kq*4 + r (pixel), col = lane & 15 (channel of tile nt)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * (NT * 256) + (nt * 4 + r) * 64 + lane] = acc[nt][r];
  __syncthreads();
  for (int e = tid; e < DCN_PIX * NT * 16; e += 256) {
    const int co = e % (NT * 16), pix = e / (NT * 16);
    const int nt = co >> 4, c16 = co & 15;
    const int idx = (nt * 4 + (pix & 3)) * 64 + (pix >> 2) * 16 + c16;
    if (co < p.Co && m0 + pix < p.P) {
      float v = (red[idx] + red[NT * 256 + idx]) + (red[2 * NT * 256 + idx] + red[3 * NT * 256 + idx]);
      if (p.bias) v += p.bias[co];
      st1(p.y + (long)(m0 + pix) * p.Co + co, v);
    }
  }
}

// Contraction phase of dcn_fwd_kernel: y[16, Co] = col[16, C*K] x W^T + bias; the 4 waves split K.
template <typename T, int NT>
__device__ __forceinline__ void dcn_contract(const DcnArgs<T>& p, const float* col, float* red, int m0, int stride,
                                             int KS16) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = lane & 15, kq = lane >> 4;
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int ks = wave; ks < KS16; ks += 4) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(col + row * stride + (ks * 4 + kq) * 4);
    f32x4 bw[NT];
    const float* wb = p.wp + ((long)ks * p.NTt) * 256 + lane * 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bw[nt] = *reinterpret_cast<const f32x4*>(wb + (long)nt * 256);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bw[nt][t], acc[nt], 0, 0, 0);
  }
  dcn_reduce_store<T, NT>(p, acc, red, m0);
}

// Forward, LDS-column-tile form (the default until the register-fed kernel below; now the fallback and A/B
// partner).  Two phases per workgroup of DCN_PIX consecutive output pixels:
//  1. gather: one work item per (pixel, tap*group, 4-channel block) in the tap-major column order of the weight
//     image; each bilinear corner is one 16-byte load of the cg-contiguous NHWC channels.  The modulated samples
//     land in an LDS column tile col[pixel][kidx].
//  2. contraction: y[16, Co] = col[16, C*K] x W^T on v_mfma_f32_16x16x4_f32 (exact f32); the 4 waves split
//     K, partial tiles meet in LDS, bias is added and the [16, Co] block is stored as one contiguous run.
// The column tile never exists in HBM: traffic = x + offsets + masks + y (SURVEY.md 8d algorithmic bytes).
template <typename T, int NT>
__global__ __launch_bounds__(256) void dcn_fwd_kernel(DcnArgs<T> p) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x;
  const int K = p.kh * p.kw, GK = p.G * K, q4 = p.cg >> 2;
  const int KS16 = p.KS, stride = KS16 * 16 + 4;
  float* col = smem;                       // [DCN_PIX][stride]
  float* red = smem + DCN_PIX * stride;    // [4][NT*256]
  __shared__ int pcoord[DCN_PIX][4];       // b, oy*stride-pad, ox*stride-pad, valid
  int bxl, byl;
  xcd_tile(1, bxl, byl);  // neighbouring pixel tiles gather from the same input rows: keep them on one XCD's L2
  const int m0 = bxl * DCN_PIX;
  if (tid < DCN_PIX) {
    const int m = m0 + tid;
    const bool v = m < p.P;
    const int mm = v ? m : 0;
    const int HoWo = p.Ho * p.Wo;
    const int b = mm / HoWo, r = mm - b * HoWo;
    const int oy = r / p.Wo, ox = r - oy * p.Wo;
    pcoord[tid][0] = b;
    pcoord[tid][1] = oy * p.stride - p.pad;
    pcoord[tid][2] = ox * p.stride - p.pad;
    pcoord[tid][3] = v ? 1 : 0;
  }
  if (p.C * K < KS16 * 16) {               // zero the K tail (C*K not a multiple of 16)
    const int tail = KS16 * 16 - p.C * K;
    for (int i = tid; i < DCN_PIX * tail; i += 256) col[(i / tail) * stride + p.C * K + (i % tail)] = 0.f;
  }
  __syncthreads();

  const int per_pix = GK * q4;
  const int items = DCN_PIX * per_pix;
  // DCN_U items per thread per pass, software-pipelined by hand: all offset/mask loads of the pass are issued
  // first, then all 4*DCN_U corner loads, so a pass costs two memory latencies instead of 2*DCN_U.
  constexpr int DCN_U = 4;
  for (int base = tid; base < items; base += 256 * DCN_U) {
    int pixv[DCN_U], rv[DCN_U];
    f32x2 ov[DCN_U];
    float mv[DCN_U];
    bool okv[DCN_U];
#pragma unroll
    for (int u = 0; u < DCN_U; ++u) {
      const int i = base + u * 256;
      const bool in = i < items;
      const int ii = in ? i : 0;
      pixv[u] = ii / per_pix;
      rv[u] = ii - pixv[u] * per_pix;
      okv[u] = in && pcoord[pixv[u]][3];
      ov[u] = f32x2{0.f, 0.f};
      mv[u] = 1.f;
      if (okv[u]) {
        const long m = m0 + pixv[u];
        const int tg = rv[u] / q4, tap = tg / p.G;
        const int gt = (tg - tap * p.G) * K + tap;  // tap-major item -> (group, tap) index of the offset tensor
        ov[u] = ld2(p.off + m * p.ostr + gt * 2);
        if (p.msk) mv[u] = ld1(p.msk + m * p.mstr + gt);
      }
    }
    f32x4 a[DCN_U][4];
    float wgt[DCN_U][4];
#pragma unroll
    for (int u = 0; u < DCN_U; ++u) {
      const int tg = rv[u] / q4, q = rv[u] - tg * q4;
      const int tap = tg / p.G, g = tg - tap * p.G;
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      const float py = (float)(pcoord[pixv[u]][1] + ky * p.dil) + ov[u].x;
      const float px = (float)(pcoord[pixv[u]][2] + kx * p.dil) + ov[u].y;
      const float fy = floorf(py), fx = floorf(px);
      const float ly = py - fy, lx = px - fx, hy = 1.f - ly, hx = 1.f - lx;
      const int y0 = (int)fminf(fmaxf(fy, -4.f), (float)p.H + 2.f), x0 = (int)fminf(fmaxf(fx, -4.f), (float)p.W + 2.f);
      const T* cb = p.x + (long)pcoord[pixv[u]][0] * p.H * p.W * p.C + g * p.cg + q * 4;
      const bool yv0 = okv[u] && (unsigned)y0 < (unsigned)p.H, yv1 = okv[u] && (unsigned)(y0 + 1) < (unsigned)p.H;
      const bool xv0 = (unsigned)x0 < (unsigned)p.W, xv1 = (unsigned)(x0 + 1) < (unsigned)p.W;
      const long o00 = ((long)y0 * p.W + x0) * p.C;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      a[u][0] = (yv0 && xv0) ? ld4(cb + o00) : z;
      a[u][1] = (yv0 && xv1) ? ld4(cb + o00 + p.C) : z;
      a[u][2] = (yv1 && xv0) ? ld4(cb + o00 + (long)p.W * p.C) : z;
      a[u][3] = (yv1 && xv1) ? ld4(cb + o00 + (long)p.W * p.C + p.C) : z;
      wgt[u][0] = hy * hx; wgt[u][1] = hy * lx; wgt[u][2] = ly * hx; wgt[u][3] = ly * lx;
    }
#pragma unroll
    for (int u = 0; u < DCN_U; ++u) {
      if (base + u * 256 < items) {
        // same association as the oracle's sum over corners
        f32x4 v = ((a[u][0] * wgt[u][0] + a[u][1] * wgt[u][1]) + a[u][2] * wgt[u][2]) + a[u][3] * wgt[u][3];
        v *= mv[u];
        *reinterpret_cast<f32x4*>(col + pixv[u] * stride + rv[u] * 4) = v;
      }
    }
  }
  __syncthreads();

  dcn_contract<T, NT>(p, col, red, m0, stride, KS16);
}

// Forward, register-fed form (default; dcn_fwd_kernel above stays as the fallback for tensors of 4 GiB and more and
// as the A/B partner, fami_dcn_tune).  Ablations of dcn_fwd_kernel on the MI355X (tools/bench_dcn.py history: offset
// stream + arithmetic 12.6 us, + corner loads 15.5 us, + contraction 13.8 us = the 38-40 us of the whole kernel) show
// that its phases do not overlap: every workgroup of a CU gathers, meets at the barrier, then contracts, so the
// L1-bound gather and the MFMA-bound contraction are paid one after the other; neither cutting the gather's VALU work
// 3x nor prefetching the weight fragments moved the total.  Here the lane that gathers a sample is the lane that
// feeds it to the matrix core:
//  * v_mfma_f32_16x16x4_f32 takes A[row][k] from lane k*16 + row.  Lane (row, kq) of wave w therefore gathers, for
//    k group ks = w, w+4, .., the 4-channel sample block kidx = (ks*4 + kq)*4 .. +3 of pixel `row` -- exactly the
//    f32x4 the old kernel wrote to and re-read from the LDS column tile -- and issues the 4 MFMAs on it directly.
//    No column tile, no barrier between gather and contraction: the 16 waves of a CU run free, one wave's MFMAs
//    cover another wave's load latency.
//  * offsets and masks (77 % of the bytes) are still streamed fully coalesced: the tile's rows are one contiguous
//    run in HBM and are copied to LDS with 16-byte accesses before the waves read them in (row, tap) order.
//  * the (group, tap) decode is a per-workgroup LDS table; out-of-map corners zero the 1-D bilinear weight and clamp
//    the address (unconditional 16-byte loads, SGPR base + 32-bit byte offset); PF k groups per wave are in flight.
template <typename T, int NT, int PF, int MINW>
__global__ __launch_bounds__(256, MINW) void dcn_fwd_direct_kernel(DcnArgs<T> p) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = p.kh * p.kw, GK = p.G * K, q4 = p.cg >> 2, CK = p.C * K;
  const int KS16 = p.KS;
  const int nitem = KS16 * 4;  // 4-channel sample blocks per pixel, padded to whole k groups
  // two dense tensors: soff [DCN_PIX][2GK] then smsk [DCN_PIX][GK]; one merged tensor (p.ostr == 3GK): rows of 3GK, the
  // masks of a pixel behind its offsets -- the tile's rows are then ONE contiguous run of HBM
  const bool merged = p.ostr == 3 * GK;
  const int lo = merged ? 3 * GK : 2 * GK, lm = merged ? 3 * GK : GK;      // LDS row strides (elements)
  T* soff = reinterpret_cast<T*>(smem);
  T* smsk = merged ? soff + 2 * GK : soff + DCN_PIX * GK * 2;
  const int staged = (DCN_PIX * GK * 3 * (int)sizeof(T) + 15) & ~15;     // bytes
  int4* tapt = reinterpret_cast<int4*>(reinterpret_cast<char*>(smem) + staged);  // [nitem]
  float* red = smem;  // [4][NT*256], reuses the staged rows once every wave is done with them
  int bxl, byl;
  xcd_tile(1, bxl, byl);
  const int m0 = bxl * DCN_PIX;
  const int rows = min(DCN_PIX, p.P - m0);

  {  // offsets (/ + masks) of the tile: contiguous runs, 16-byte copies
    const int nb = rows * lo * (int)sizeof(T);
    const char* src = reinterpret_cast<const char*>(p.off + (long)m0 * p.ostr);
    char* dst = reinterpret_cast<char*>(soff);
    for (int i = tid * 16; i + 16 <= nb; i += 256 * 16) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
    for (int i = (nb & ~15) + tid * (int)sizeof(T); i < nb; i += 256 * (int)sizeof(T))
      *reinterpret_cast<T*>(dst + i) = *reinterpret_cast<const T*>(src + i);
  }
  if (p.msk && !merged) {
    const int nb = rows * GK * (int)sizeof(T);
    const char* src = reinterpret_cast<const char*>(p.msk + (long)m0 * GK);
    char* dst = reinterpret_cast<char*>(smsk);
    for (int i = tid * 16; i + 16 <= nb; i += 256 * 16) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
    for (int i = (nb & ~15) + tid * (int)sizeof(T); i < nb; i += 256 * (int)sizeof(T))
      *reinterpret_cast<T*>(dst + i) = *reinterpret_cast<const T*>(src + i);
  }
  for (int it = tid; it < nitem; it += 256) {
    int4 e = {0, 0, 0, 0};
    if (it * 4 < CK) {
      const int tg = it / q4, q = it - tg * q4;
      const int tap = tg / p.G, g = tg - tap * p.G;  // tap-major column order (dcn_pack_w_kernel)
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      e = int4{(g * p.cg + q * 4) * (int)sizeof(T), ky * p.dil, kx * p.dil, g * K + tap};
    }
    tapt[it] = e;
  }
  // this lane's pixel
  const int row = lane & 15, kq = lane >> 4;
  const bool valid = row < rows;
  int oy0, ox0;
  unsigned xoff;  // byte offset of the pixel's sample in x
  {
    const int mm = valid ? m0 + row : 0;
    const int HoWo = p.Ho * p.Wo;
    const int b = mm / HoWo, r = mm - b * HoWo;
    const int oy = r / p.Wo, ox = r - oy * p.Wo;
    oy0 = oy * p.stride - p.pad;
    ox0 = ox * p.stride - p.pad;
    xoff = (unsigned)b * (unsigned)(p.H * p.W * p.C) * (unsigned)sizeof(T);
  }
  const char* xbase = reinterpret_cast<const char*>(p.x);
  const unsigned WCb = p.W * p.C * (unsigned)sizeof(T), Cb = p.C * (unsigned)sizeof(T);
  const int Hm1 = p.H - 1, Wm1 = p.W - 1;
  const float Hf1 = sgpr_f((float)(p.H + 1)), Wf1 = sgpr_f((float)(p.W + 1));
  const T* myoff = soff + row * lo;
  const T* mymsk = smsk + row * lm;
  __syncthreads();

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // one sample block (4 channels of one tap / group) of this lane's pixel: the four corner loads and the folded weights
  auto gather = [&](int ks, f32x4 (&a)[4], float (&w1)[4]) {
    const int4 te = tapt[min(ks, KS16 - 1) * 4 + kq];
    const bool ok = valid && ks < KS16 && (ks * 4 + kq) * 4 < CK;
    const f32x2 ov = ld2(myoff + te.w * 2);
    const float mv = p.msk ? ld1(mymsk + te.w) : 1.f;
    const float py = (float)(oy0 + te.y) + ov.x;
    const float px = (float)(ox0 + te.z) + ov.y;
    const float fy = floorf(py), fx = floorf(px);
    const float ly = py - fy, lx = px - fx;
    const int y0 = (int)__builtin_amdgcn_fmed3f(fy, -2.f, Hf1), x0 = (int)__builtin_amdgcn_fmed3f(fx, -2.f, Wf1);
    const int y1 = y0 + 1, x1 = x0 + 1;
    w1[0] = (ok && (unsigned)y0 <= (unsigned)Hm1) ? (1.f - ly) * mv : 0.f;  // modulation folded into the row weights
    w1[1] = (ok && (unsigned)y1 <= (unsigned)Hm1) ? ly * mv : 0.f;
    w1[2] = (unsigned)x0 <= (unsigned)Wm1 ? 1.f - lx : 0.f;
    w1[3] = (unsigned)x1 <= (unsigned)Wm1 ? lx : 0.f;
    const unsigned r0 = xoff + __umul24(min(max(y0, 0), Hm1), WCb), r1 = xoff + __umul24(min(max(y1, 0), Hm1), WCb);
    const unsigned c0 = __umul24(min(max(x0, 0), Wm1), Cb) + (unsigned)te.x, c1 = __umul24(min(max(x1, 0), Wm1), Cb) + (unsigned)te.x;
    a[0] = ld4(reinterpret_cast<const T*>(xbase + (r0 + c0)));
    a[1] = ld4(reinterpret_cast<const T*>(xbase + (r0 + c1)));
    a[2] = ld4(reinterpret_cast<const T*>(xbase + (r1 + c0)));
    a[3] = ld4(reinterpret_cast<const T*>(xbase + (r1 + c1)));
  };
  // corner order of the oracle's sum
  auto blend = [&](const f32x4 (&a)[4], const float (&w1)[4]) {
    return ((a[0] * (w1[0] * w1[2]) + a[1] * (w1[0] * w1[3])) + a[2] * (w1[1] * w1[2])) + a[3] * (w1[1] * w1[3]);
  };
  if constexpr (sizeof(T) == 2) {
    // 16-bit modes: the contraction runs on v_mfma_f32_16x16x32_{bf16,f16} -- the MATRIX pipe -- so the gather's address
    // and bilinear arithmetic (vector pipe) overlaps it; with the exact-f32 MFMA (which IS the vector ALUs on gfx950) the
    // two added up (profiles/r02_probe_mfma_valu_overlap.txt).  A lane's two sample blocks of a 32-step (16-steps 2s and
    // 2s+1) are rounded to the storage type and form its 8 K-values; dcn_pack_w16_kernel orders the weights to match.
    typedef typename H16<T>::x8 hx8;
    typedef T hx4 __attribute__((ext_vector_type(4)));
    const int KS32 = (KS16 + 1) >> 1;
    const T* w16 = reinterpret_cast<const T*>(p.wp16);
    for (int s0 = wave; s0 < KS32; s0 += 4 * PF) {
      f32x4 a0[PF][4], a1[PF][4];
      float wa[PF][4], wb[PF][4];
      hx8 bw[PF][NT];
#pragma unroll
      for (int i = 0; i < PF; ++i) {       // PF 32-steps in flight: 8 * PF corner loads per lane before the first blend
        const int s32 = s0 + 4 * i;
        if (s32 < KS32) {
          gather(2 * s32, a0[i], wa[i]);
          gather(2 * s32 + 1, a1[i], wb[i]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            bw[i][nt] = *reinterpret_cast<const hx8*>(w16 + ((long)s32 * p.NTt + nt) * 512 + lane * 8);
        }
      }
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        if (s0 + 4 * i < KS32) {
          const hx4 v0 = __builtin_convertvector(blend(a0[i], wa[i]), hx4), v1 = __builtin_convertvector(blend(a1[i], wb[i]), hx4);
          const hx8 av = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = H16<T>::mfma(av, bw[i][nt], acc[nt]);
        }
      }
    }
  } else {
    for (int ks0 = wave; ks0 < KS16; ks0 += 4 * PF) {
      f32x4 bw[PF][NT];
      f32x4 a[PF][4];
      float w1[PF][4];  // mask*wy0, mask*wy1, wx0, wx1 (validity folded in)
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int ks = ks0 + 4 * i;
        if (ks < KS16) {
          gather(ks, a[i], w1[i]);
          const float* wb = p.wp + ((long)ks * p.NTt) * 256 + lane * 4;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bw[i][nt] = *reinterpret_cast<const f32x4*>(wb + (long)nt * 256);
        }
      }
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        if (ks0 + 4 * i < KS16) {
          const f32x4 v = blend(a[i], w1[i]);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[t], bw[i][nt][t], acc[nt], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();  // red aliases the staged offsets
  dcn_reduce_store<T, NT>(p, acc, red, m0);
}


// ------------------------------------------------------------------ DCN forward, LDS-window form (opt-in: fami_dcn_tune(2))
// Measured 29.8 us against the register-fed kernel's 32.0 us (f32, B = 4, 96x72x48; tools/bench_dcn_win.py, profiles/
// r02_dcn_window.txt) -- not the 2x its LDS gather was built for, because the gather was never the whole story:
// v_mfma_f32_16x16x4_f32 executes on the vector ALUs (tools/probes/mfma_valu_overlap.hip: MFMA + v_fma times ADD, unlike
// the bf16 matrix instructions), so the 1.15 GFLOP contraction (~10 us at the ~1.9 GHz the part sustains here) and the
// ~4.4 M wave-instructions of address / bilinear arithmetic (~5.5 us) share one pipe and the ~5 us prologue / epilogue
// of a 133 KB-window workgroup comes on top.  It stays opt-in: 7 % is not worth a second default path whose speed
// depends on the offsets staying within R pixels.
// The register-fed kernel above is bound by the L1's one-cache-line-per-clock rate: every one of its 11.9 M bilinear
// corner loads (B = 4) is a scattered 16-byte global access, 64 distinct lines per wave instruction (~19 us).  Here a
// workgroup of 16 waves owns a TH x TW tile of output pixels (<= 128) and
//   * copies the input WINDOW the tile can reach -- tile + dilation halo + R pixels of offset reach on every side --
//     once into LDS, coalesced (all of a thread's 16-byte pieces are requested before the first is stored), zero-filled
//     outside the image: the four corners of an in-window sample need no validity logic, zero padding IS the "corner
//     contributes only inside the map" rule.  Pixels are padded to C + 16 bytes so that neighbours start in different
//     bank groups;
//   * gathers every bilinear corner with one ds_read (16 bytes f32 / 8 bytes 16-bit) from that window; a sample whose
//     offset leaves the window (|offset| >= R: never for realistic offsets, 6e-5 of N(0,1) draws at R = 4) takes the
//     per-corner global path of the older kernels, so any offset magnitude stays exact;
//   * a PAIR of waves owns a 16-pixel sub-tile, each wave one half of K (4 + 3 quad groups at G*K = 108): four waves per
//     SIMD, so one wave's address / bilinear arithmetic (VALU) runs under another's MFMAs; the halves meet in LDS once
//     at the end.  The MFMA operand comes from the registers of the lane that gathered it, as in the register-fed kernel;
//   * offsets and masks (77 % of the bytes) are read straight from HBM one quad group ahead of their use.  Columns are
//     GROUP-major here (item = g*K + tap, the order of the offset tensor itself) and lane (pixel, kq) of a quad group
//     takes the 4 consecutive items 16Q + 4kq .. +3: its 4 offset pairs are one 32-byte run and its 4 masks one 16-byte
//     run, and the 4 kq lanes of a pixel cover 128 contiguous bytes -- whole lines per pixel;
//   * the (group, tap) decode of an item is one packed word of a per-workgroup LDS table (no integer division in the loop).
// Weight image: wq[((Q*4 + j)*NTt + nt)*256 + lane*4 + t] = W[nt*16 + (lane&15)][g*cg + t][tap], item 16Q + 4(lane>>4) + j.
template <typename T>
struct DcnWinArgs {
  const T* x;
  const T* off;
  const T* msk;       // may be null => 1
  const float* wq;
  const float* bias;
  T* y;
  int B, H, W, C, Ho, Wo, Co, G, K, kw, pad, dil;
  int TH, TW, tilesX, tilesY, R, WR, WC, PS;  // window rows / cols, LDS elements per window pixel (C + 16 bytes)
  int NQ, NTt, nsub;
  int ostr, mstr;                             // elements per pixel of off / msk (see DcnArgs)
  unsigned mul_row, mul_px;                   // exact-division multipliers (>> 24) by the pieces per window row / per pixel
  int abl;                                    // ablation bits (benchmarks): 1 no MFMA, 2 no LDS gather, 4 no offset stream, 8 no window fill
};

__global__ void dcn_pack_wq_kernel(const float* __restrict__ w, float* __restrict__ wq, int Co, int C, int K, int G,
                                   int NQ, int NTt) {
  const long total = (long)NQ * 4 * NTt * 256;
  const int cg = C / G;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i & 3), lane = (int)((i >> 2) & 63);
    long r = i >> 8;
    const int nt = (int)(r % NTt);
    r /= NTt;
    const int j = (int)(r & 3), Q = (int)(r >> 2);
    const int item = 16 * Q + 4 * (lane >> 4) + j, co = nt * 16 + (lane & 15);
    float v = 0.f;
    if (item < G * K && co < Co) {
      const int g = item / K, tap = item - g * K;
      v = w[((long)co * C + g * cg + t) * K + tap];
    }
    wq[i] = v;
  }
}

// KSPLIT = waves per 16-pixel sub-tile: 1 = 8 waves, each walks all of K; 2 = 16 waves (4 per SIMD), K halves met in LDS
template <typename T, int NT, bool ABL, int KSPLIT>
__global__ __launch_bounds__(512 * KSPLIT) void dcn_fwd_win_kernel(DcnWinArgs<T> p) {
  constexpr int DCN_WIN_THREADS = 512 * KSPLIT;
  const int abl = ABL ? p.abl : 0;   // benchmark instrumentation: compiled out of the production instance
  extern __shared__ __attribute__((aligned(16))) char wsm[];
  T* win = reinterpret_cast<T*>(wsm);                                        // [WR][WC][PS]
  const int winb = p.WR * p.WC * p.PS * (int)sizeof(T);                       // a multiple of 16
  unsigned* tab = reinterpret_cast<unsigned*>(wsm + winb);                   // [NQ*16] packed (ky*dil, kx*dil, g*4)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = lane & 15, kq = lane >> 4;
  int t, unused;
  xcd_tile(1, t, unused);                  // neighbouring tiles (overlapping windows) on one XCD's L2
  const int tx = t % p.tilesX;
  t /= p.tilesX;
  const int ty = t % p.tilesY, b = t / p.tilesY;
  const int oy0 = ty * p.TH, ox0 = tx * p.TW;
  const int wy0 = oy0 - p.pad - p.R, wx0 = ox0 - p.pad - p.R;   // window origin in input coordinates (stride 1)
  const int GK = p.G * p.K, GK2 = GK * 2;
  const T* xb = p.x + (long)b * p.H * p.W * p.C;
  const long mb = (long)b * p.Ho * p.Wo;

  // ---- this wave: sub-tile `sub` (16 output pixels), quad groups [Q0, NQ)
  const int sub = KSPLIT == 2 ? wave >> 1 : wave, half = KSPLIT == 2 ? wave & 1 : 0;
  const bool active = sub < p.nsub;          // wave-uniform
  const int Qh = (p.NQ + 1) >> 1;
  const int Q0 = half ? Qh : 0, NQ = (KSPLIT == 2 && !half) ? Qh : p.NQ;
  int oy, ox;
  bool valid;
  {
    const int i = sub * 16 + row;
    const int py = i / p.TW, px = i - py * p.TW;
    oy = oy0 + py;
    ox = ox0 + px;
    valid = active && i < p.TH * p.TW && oy < p.Ho && ox < p.Wo;
  }
  const long m = valid ? mb + (long)oy * p.Wo + ox : mb;
  struct OM { f32x4 o0, o1, mk; };           // one quad group's 4 offset pairs and 4 masks of this lane
  auto load_om = [&](int Q, OM& r) {
    if (abl & 4) { r.o0 = r.o1 = r.mk = f32x4{0.25f, 0.5f, 0.75f, 0.125f}; return; }
    int it0 = 16 * Q + 4 * kq;
    if (it0 + 4 > GK) it0 = GK - 4;                // padding items: any valid address (their weights are zero)
    const T* po = p.off + m * p.ostr + it0 * 2;
    r.o0 = ld4(po);
    r.o1 = ld4(po + 4);
    r.mk = p.msk ? ld4(p.msk + m * p.mstr + it0) : f32x4{1.f, 1.f, 1.f, 1.f};
  };
  OM om0, om1;
  om0.o0 = om0.o1 = om0.mk = om1.o0 = om1.o1 = om1.mk = f32x4{0.f, 0.f, 0.f, 0.f};
  if (active) {                                    // the offset stream starts before the window fill
    load_om(Q0 < NQ ? Q0 : NQ - 1, om0);
    load_om(Q0 + 1 < NQ ? Q0 + 1 : NQ - 1, om1);
  }

  // ---- item decode table
  for (int it = tid; it < p.NQ * 16; it += DCN_WIN_THREADS) {
    unsigned e = 0u;
    if (it < GK) {
      const int g = it / p.K, tap = it - g * p.K;
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      e = (unsigned)(ky * p.dil) | ((unsigned)(kx * p.dil) << 8) | ((unsigned)(g * 4) << 16);
    }
    tab[it] = e;
  }
  // ---- window fill: 16-byte pieces, zero outside the image; every request of a thread is in flight before its first store
  {
    constexpr int EPP = 16 / (int)sizeof(T);        // elements per piece
    const int ppp = p.C / EPP;                      // pieces per pixel
    const int rowp = p.WC * ppp;                    // pieces per window row
    const int total = p.WR * rowp;
    constexpr int FU = 8;
    for (int i0 = tid; i0 < total && !(abl & 8); i0 += DCN_WIN_THREADS * FU) {
      uint4 v[FU];
      int dst[FU];
#pragma unroll
      for (int u = 0; u < FU; ++u) {
        const int i = i0 + u * DCN_WIN_THREADS;
        v[u] = uint4{0u, 0u, 0u, 0u};
        dst[u] = -1;
        if (i < total) {
          const int wy = (int)(((unsigned)i * p.mul_row) >> 24);
          const int ir = i - wy * rowp;
          const int wx = (int)(((unsigned)ir * p.mul_px) >> 24), pc = ir - wx * ppp;
          const int yy = wy0 + wy, xx = wx0 + wx;
          dst[u] = (wy * p.WC + wx) * p.PS + pc * EPP;
          if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)
            v[u] = *reinterpret_cast<const uint4*>(xb + ((long)yy * p.W + xx) * p.C + pc * EPP);
        }
      }
#pragma unroll
      for (int u = 0; u < FU; ++u)
        if (dst[u] >= 0) *reinterpret_cast<uint4*>(win + dst[u]) = v[u];
    }
  }
  __syncthreads();
  if (KSPLIT == 1 && !active) return;

  const int rowbB = p.WC * p.PS * (int)sizeof(T), PSB = p.PS * (int)sizeof(T);   // window bytes per row / per pixel
  const char* winc = reinterpret_cast<const char*>(win);
  const int by = oy - p.pad, bx = ox - p.pad;
  const float Rlo = (float)wy0, Clo = (float)wx0;
  const float rmax = (float)(p.WR - 2), cmax = (float)(p.WC - 2);

  // modulated samples of quad group Q for this lane's 4 items, from the LDS window.  Branch-free: the window reads are
  // unconditional at clamped coordinates; bit j of the result flags item j as outside the window (fixed up by the caller).
  auto gather = [&](int Q, const OM& om, f32x4 (&val)[4]) -> unsigned {
    const int it0 = 16 * Q + 4 * kq;
    const bool live = valid && it0 < GK;             // G*K is a multiple of 4: a lane's 4 items are real or padding together
    const u32x4 te = *reinterpret_cast<const u32x4*>(tab + it0);
    unsigned oob = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned e = te[j];
      const float oyv = j < 2 ? om.o0[2 * j] : om.o1[2 * j - 4], oxv = j < 2 ? om.o0[2 * j + 1] : om.o1[2 * j - 3];
      const float mv = live ? om.mk[j] : 0.f;
      const float py = (float)(by + (int)(e & 255u)) + oyv, px = (float)(bx + (int)((e >> 8) & 255u)) + oxv;
      const float fy = floorf(py), fx = floorf(px);
      const float ly = py - fy, lx = px - fx;
      // window coordinates of the (y0, x0) corner (integer-valued floats: exact), clamped into the window
      const float ryf = fy - Rlo, rxf = fx - Clo;
      const float ryc = __builtin_amdgcn_fmed3f(ryf, 0.f, rmax), rxc = __builtin_amdgcn_fmed3f(rxf, 0.f, cmax);
      const bool inwin = ryf == ryc && rxf == rxc;
      const char* c00 = winc + (__mul24((int)ryc, rowbB) + __mul24((int)rxc, PSB) + (int)(e >> 16) * (int)sizeof(T));
      f32x4 a0, a1, a2, a3;
      if (abl & 2) { a0 = a1 = a2 = a3 = f32x4{ly, lx, mv, oyv}; }
      else {
        a0 = ld4(reinterpret_cast<const T*>(c00));
        a1 = ld4(reinterpret_cast<const T*>(c00 + PSB));
        a2 = ld4(reinterpret_cast<const T*>(c00 + rowbB));
        a3 = ld4(reinterpret_cast<const T*>(c00 + rowbB + PSB));
      }
      const float wy0m = (1.f - ly) * mv, wy1m = ly * mv, wx0 = 1.f - lx, wx1 = lx;
      // corner order and association of the oracle's sum (mask folded into the row weights, as dcn_fwd_direct_kernel)
      val[j] = ((a0 * (wy0m * wx0) + a1 * (wy0m * wx1)) + a2 * (wy1m * wx0)) + a3 * (wy1m * wx1);
      oob |= (!inwin && mv != 0.f) ? (1u << j) : 0u;
    }
    return oob;
  };
  // the rare sample that leaves the window: per-corner global loads with the image-bounds rule of the other kernels
  auto fixup = [&](int Q, const OM& om, f32x4 (&val)[4], unsigned oob) {
    const int it0 = 16 * Q + 4 * kq;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!((oob >> j) & 1u)) continue;
      const unsigned e = tab[it0 + j];
      const float oyv = j < 2 ? om.o0[2 * j] : om.o1[2 * j - 4], oxv = j < 2 ? om.o0[2 * j + 1] : om.o1[2 * j - 3];
      const float mv = om.mk[j];
      const float py = (float)(by + (int)(e & 255u)) + oyv, px = (float)(bx + (int)((e >> 8) & 255u)) + oxv;
      const float fy = floorf(py), fx = floorf(px);
      const float ly = py - fy, lx = px - fx;
      const int y0 = (int)fminf(fmaxf(fy, -4.f), (float)p.H + 2.f), x0 = (int)fminf(fmaxf(fx, -4.f), (float)p.W + 2.f);
      const bool yv0 = (unsigned)y0 < (unsigned)p.H, yv1 = (unsigned)(y0 + 1) < (unsigned)p.H;
      const bool xv0 = (unsigned)x0 < (unsigned)p.W, xv1 = (unsigned)(x0 + 1) < (unsigned)p.W;
      const T* cb = xb + (e >> 16);
      const long o00 = ((long)y0 * p.W + x0) * p.C;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const f32x4 a0 = (yv0 && xv0) ? ld4(cb + o00) : z;
      const f32x4 a1 = (yv0 && xv1) ? ld4(cb + o00 + p.C) : z;
      const f32x4 a2 = (yv1 && xv0) ? ld4(cb + o00 + (long)p.W * p.C) : z;
      const f32x4 a3 = (yv1 && xv1) ? ld4(cb + o00 + (long)p.W * p.C + p.C) : z;
      const float wy0m = (1.f - ly) * mv, wy1m = ly * mv, wx0 = 1.f - lx, wx1 = lx;
      val[j] = ((a0 * (wy0m * wx0) + a1 * (wy0m * wx1)) + a2 * (wy1m * wx0)) + a3 * (wy1m * wx1);
    }
  };

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // v_mfma_f32_16x16x4_f32 runs on the vector ALUs (tools/probes/mfma_valu_overlap.hip: N MFMAs + M v_fma take the SUM of
  // their times, unlike the bf16 matrix instructions), so gather arithmetic cannot hide under the contraction: the loop
  // only has to keep the memory requests ahead.  Request order matters (vmcnt counts requests in issue order): this quad
  // group's weight fragments FIRST, then the offsets / masks of quad group Q+2 -- the MFMAs then wait for all but the 3
  // youngest requests and the offset stream keeps flying under them.  Issued the other way round, every wait for a
  // weight fragment also waited for the HBM round trip of the prefetch.
  auto step = [&](int Q, OM& cur, OM& nxt) {   // consumes cur (quad group Q), refills it with quad group Q+2
    f32x4 val[4];
    const unsigned oob = gather(Q, cur, val);
    if (__builtin_amdgcn_ballot_w64(oob != 0u) != 0ull) fixup(Q, cur, val, oob);
    f32x4 bw[4][NT];
    const float* wb = p.wq + ((long)Q * 4 * p.NTt) * 256 + lane * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bw[j][nt] = (abl & 16) ? f32x4{0.5f, 0.25f, 0.125f, 1.f} : *reinterpret_cast<const f32x4*>(wb + (j * p.NTt + nt) * 256);
    __builtin_amdgcn_sched_barrier(0);
    load_om(Q + 2 < NQ ? Q + 2 : NQ - 1, cur);     // unconditional (the tail re-reads the last group): no branch to merge wait counts over
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (abl & 1) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] += val[j] * bw[j][nt];
      } else {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(val[j][tt], bw[j][nt][tt], acc[nt], 0, 0, 0);
      }
    }
  };
  for (int Q = Q0; active && Q < NQ && !(abl & 32); Q += 2) {   // two quad groups per trip: the offset registers alternate, no copies
    step(Q, om0, om1);
    if (Q + 1 < NQ) step(Q + 1, om1, om0);
  }
  float* red = reinterpret_cast<float*>(wsm);        // [nsub][NT*4][64]: aliases the window once every wave is done with it
  if (KSPLIT == 2) {
    __syncthreads();
    if (active && half) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(sub * NT * 4 + nt * 4 + r) * 64 + lane] = acc[nt][r];
    }
    __syncthreads();
    if (!active || half) return;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[nt][r] += red[(sub * NT * 4 + nt * 4 + r) * 64 + lane];
  }
  // D row = kq*4 + r (pixel of the sub-tile), col = lane & 15 (channel of tile nt): 64-byte runs per pixel
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = sub * 16 + kq * 4 + r;
    const int py = i / p.TW, px = i - py * p.TW;
    const int yy = oy0 + py, xx = ox0 + px;
    if (i >= p.TH * p.TW || yy >= p.Ho || xx >= p.Wo) continue;
    T* yp = p.y + (mb + (long)yy * p.Wo + xx) * p.Co;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = nt * 16 + row;
      if (co < p.Co) st1(yp + co, acc[nt][r] + (p.bias ? p.bias[co] : 0.f));
    }
  }
}

// ------------------------------------------------------------------ DCN backward (fused)
// One workgroup owns a TILE x TILE block of output pixels of one sample and one CHUNK of offset groups
// (GC groups = Cc channels = CKc = Cc*K columns, a multiple of 16).  Per 16-pixel sub-tile:
//   A. gcol[16, CKc] = dy[16, Co] x W[Co, CKc]  on v_mfma_f32_16x16x4_f32, kept in LDS (the column gradient
//      never exists in HBM);
//   B. one work item per (pixel, group*tap): coalesced offset/mask stream, four 16-byte corner loads, then
//        gmask = <gcol, sample>, goffset = <gcol*mask, d sample / d(y,x)>  (stored, 1 contiguous run / pixel)
//        col   = sample*mask  (overwrites gcol in LDS; flushed for the weight gradient)
//        gx   += gcol*mask*bilinear weights, scattered with LDS compare-and-swap adds (lds_add_f32) into a privatised input-gradient
//                region (tile + dilation halo + DCN_RO pixels of offset reach); samples that leave the region
//                fall back to global atomics, so any offset magnitude stays correct;
//   C. the col tile is flushed in contiguous runs.
// At the end the region is added to gx with coalesced global atomics: ~6x fewer atomics than scattering from
// the items, 16 consecutive channels per request instead of one address per lane, and no same-address pile-up.
#define DCN_TILE 8
#define DCN_RO 3
#define DCN_MASK_BOUND 8.f   // fixed-point LDS scatter: |mask| above this takes the global-atomic path (see dcn_bwd_kernel)

// f32 add into LDS as a compare-and-swap loop.  On gfx950 the native ds_add_f32 retires 0.33 lane-operations per clock
// per CU at this kernel's access pattern (tools/probes/lds_atomic.hip: ds_add_u32 4.7, ds_add_u64 2.5, racy read +
// write 7.3) -- the 47.8 M adds of a B=4 launch alone were 220 of its 292 us.  A ds_read + ds_cmpst_rtn loop (almost
// always one trip) measures 1.4 per clock, 4.2x the native instruction, with identical f32 arithmetic.
__device__ __forceinline__ void lds_add_f32(float* a, float v) {
  unsigned* ua = reinterpret_cast<unsigned*>(a);
  unsigned old = *reinterpret_cast<volatile unsigned*>(ua), assumed;
  do {
    assumed = old;
    old = atomicCAS(ua, assumed, __float_as_uint(__uint_as_float(assumed) + v));
  } while (old != assumed);
}
template <typename T>
struct DcnBwdArgs {
  const T* x;        // [B,H,W,C]
  const T* off;      // [B,Ho,Wo,2GK]
  const T* msk;      // [B,Ho,Wo,GK] or null
  const T* dy;       // [B,Ho,Wo,Co]
  const float* wpb;  // packed [nchunk][NTc][KSo][64][4]
  T* col;            // [P, C*K] or null
  float* gx;         // [B,H,W,C] fp32, accumulated (atomics) or null
  T* goff;           // [B,Ho,Wo,2GK] or null
  T* gmsk;           // [B,Ho,Wo,GK] or null
  int B, H, W, C, Ho, Wo, Co, G, kh, kw, stride, pad, dil, cg;
  int GC, NTc, KSo, tilesX, tilesY, RH, RW, acc_off;
  // deterministic mode (DET): the input gradient is accumulated in 64-bit fixed point -- integer adds are associative,
  // so the result does not depend on the order in which lanes / workgroups arrive -- into gfix [B,H,W,C]; the scale is
  // derived in-kernel from *amax_bits = bit pattern of max |dy| (fami_dcn_bwd_det_*)
  long long* gfix;
  const unsigned* amax_bits;
  const float* wnorm;   // 1 float: max over columns k of sum_co |W[co][k]| (tail of the backward weight image)
  int ostr, mstr;       // elements per pixel of off / goff and msk / gmsk (see DcnArgs)
  int fixl;             // 1 / 2: the LDS region accumulates in 64- / 32-bit fixed point (per-workgroup scale), flushed to gx as f32
  int abl;              // benchmarks (fami_dcn_tune(1024 + bits)): 1 = no region flush, 2 = no LDS adds, 4 = constant scale (no maxima pass)
};

// Fixed-point scale of the deterministic input-gradient accumulation: 2^(DCN_FIX_BITS - ceil(log2(max|dy|))).  A
// contribution is gcol*mask*bilinear weight with gcol = sum_co dy*W; the 63-bit accumulator leaves 2^(62-DCN_FIX_BITS)
// = 2^20 of headroom over max|dy| for |W|, |mask| and the number of samples landing on one input element, and resolves
// 2^-42 of max|dy| -- 18 bits below fp32's own resolution of that value.
#define DCN_FIX_BITS 42
__device__ __forceinline__ float dcn_fix_scale(const unsigned* amax_bits) {
  const float amax = __uint_as_float(*amax_bits);
  int e = 0;
  if (amax > 0.f) (void)frexpf(amax, &e);   // amax = m * 2^e, m in [0.5, 1)
  // clamp: for max|dy| below ~2^-85 (heavily down-scaled losses) 2^(42-e) overflows fp32 and every product v*scale is
  // inf / NaN; gradients that small keep 2^(42+80) of scale instead (still far inside their own fp32 resolution)
  e = e < -80 ? -80 : (e > 100 ? 100 : e);
  return ldexpf(1.f, DCN_FIX_BITS - e);
}
__device__ __forceinline__ void fix_add(long long* a, float v, float scale) {
  atomicAdd(reinterpret_cast<unsigned long long*>(a), (unsigned long long)(long long)__float2ll_rn(v * scale));
}

// one contribution (already scaled, |v| < 2^30) into a 64-bit LDS accumulator: 32-bit conversion, sign extension, ds_add_u64
__device__ __forceinline__ void lds_fix_add(long long* a, float v) {
  const int i = __float2int_rn(v);
  atomicAdd(reinterpret_cast<unsigned long long*>(a), (unsigned long long)(long long)i);
}

// max |x| over a tensor as a float bit pattern (max is order independent: atomicMax on the bits of non-negative floats)
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T* __restrict__ x, long n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = fabsf(ld1(x + i));
    m = v > m ? v : m;          // NaN never wins
  }
  // one atomic per workgroup (the host launches <= 256 of them): one per wave of a 2048-workgroup grid serialised 8192
  // same-address atomics, 95 us for a 1.3 M-element tensor
  __shared__ float red[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m > 0.f) atomicMax(out, __float_as_uint(fminf(m, 3.0e38f)));
  }
}
__global__ void zero_u64_kernel(unsigned long long* p, long n, unsigned* amax) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0ull;
  if (blockIdx.x == 0 && threadIdx.x == 0) *amax = 0u;
}
// gx (=|+=) fixed-point accumulator / scale
template <typename T>
__global__ __launch_bounds__(256) void fix_to_act_kernel(const long long* __restrict__ fix, T* __restrict__ gx, long n,
                                                         const unsigned* __restrict__ amax_bits, int accumulate) {
  const double inv = 1.0 / (double)dcn_fix_scale(amax_bits);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = (float)((double)fix[i] * inv);
    if (accumulate) v += ld1(gx + i);
    st1(gx + i, v);
  }
}

// wpb[((chunk*NTc + nt)*KSo + ks)*256 + lane*4 + t] = W[co = (ks*4 + (lane>>4))*4 + t][kidx = (chunk*NTc + nt)*16 + (lane&15)],
// kidx = c*K + tap (the OIHW order of weight.view(Co, C*K))
__global__ void dcn_pack_wb_kernel(const float* __restrict__ w, float* __restrict__ wpb, int Co, int C, int K, int cg,
                                   int NT, int KSo) {
  const long total = (long)NT * KSo * 256;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const long r = i >> 8;
    const int ks = (int)(r % KSo), nt = (int)(r / KSo);
    const int co = (ks * 4 + (lane >> 4)) * 4 + t, kidx = nt * 16 + (lane & 15);
    float v = 0.f;
    if (co < Co && kidx < C * K) v = w[(long)co * C * K + kidx];  // column order (channel, tap) == weight.view(Co, C*K)
    wpb[i] = v;
  }
}

// MODE 0: f32 LDS region through compare-and-swap adds (lds_add_f32), f32 global atomics.
// MODE 1 (deterministic): 64-bit fixed-point LDS region AND 64-bit fixed-point global accumulator (fami_dcn_bwd_det_*).
// MODE 2 (default): 64-bit fixed-point LDS region -- native ds_add_u64 retires 2.5 lane-operations per clock per CU against
//   the compare-and-swap loop's 1.4 at this access pattern (tools/probes/lds_atomic.hip) and needs a conversion and a shift
//   per contribution instead of the loop -- flushed to gx with f32 global atomics like MODE 0.  The scale is per
//   workgroup: 2^(30 - e) with 2^e > max|dy| (tile) x max_k sum_co |W[co][k]| x DCN_MASK_BOUND >= any |gcol x mask| it accepts,
//   so one contribution fits 31 bits, a cell (at most 64 pixels x 9 taps contributions) 41, and the bound's slack
//   (~sqrt(Co) for random signs) still leaves ~2^-24 of the largest contribution as resolution -- fp32's own.
// out[0] = max over columns k of sum_co |w[co][k]|  (w = weight.view(Co, CK)); out[1..3] = 0
__global__ __launch_bounds__(256) void dcn_wnorm_kernel(const float* __restrict__ w, float* __restrict__ out, int Co, int CK) {
  __shared__ float red[4];
  float m = 0.f;
  for (int k = threadIdx.x; k < CK; k += 256) {
    float sum = 0.f;
    for (int co = 0; co < Co; ++co) sum += fabsf(w[(long)co * CK + k]);
    m = fmaxf(m, sum);
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x < 4) out[threadIdx.x] = threadIdx.x == 0 ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : 0.f;
}

template <typename T, int KSO, int MODE>
__global__ __launch_bounds__(256) void dcn_bwd_kernel(DcnBwdArgs<T> p) {
  constexpr bool DET = MODE == 1, FIX32 = MODE == 3, FIXL = MODE == 2 || FIX32;
  constexpr int FIXB = FIX32 ? 20 : 30;   // bits of one contribution at the bound
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = lane & 15, kq = lane >> 4;
  const int K = p.kh * p.kw, GK = p.G * K, q4 = p.cg >> 2;
  const int CKc = p.NTc * 16, gstride = CKc + 4, Cc = p.GC * p.cg, CK = p.C * K;
  float* gcol = smem;                   // [16][gstride]
  float* region = smem + 16 * gstride;  // [RH][RW][Cc] (fp32), or the same shape in 64-bit fixed point (DET)
  long long* regfix = reinterpret_cast<long long*>(smem + 16 * gstride);   // gstride is a multiple of 4: 8-byte aligned
  int* reg32 = reinterpret_cast<int*>(smem + 16 * gstride);                 // MODE 3: 32-bit fixed point
  float fscale = DET ? dcn_fix_scale(p.amax_bits) : 0.f;
  float finv = 0.f;
  int t, chunk;
  xcd_tile(1, t, chunk);  // neighbouring tiles (overlapping halo / gather regions, all group chunks of a tile) on one XCD
  const int tx = t % p.tilesX;
  t /= p.tilesX;
  const int ty = t % p.tilesY, b = t / p.tilesY;
  const int oy0 = ty * DCN_TILE, ox0 = tx * DCN_TILE;
  const int ry0 = oy0 * p.stride - p.pad - DCN_RO, rx0 = ox0 * p.stride - p.pad - DCN_RO;
  const int rsize = p.RH * p.RW * Cc;
  if (FIX32) { for (int i = tid; i < rsize; i += 256) reg32[i] = 0; }
  else if (DET || FIXL) { for (int i = tid; i < rsize; i += 256) regfix[i] = 0ll; }
  else { for (int i = tid; i < rsize; i += 256) region[i] = 0.f; }
  const T* xb = p.x + (long)b * p.H * p.W * p.C;
  float* gxb = (!DET && p.gx) ? p.gx + (long)b * p.H * p.W * p.C : nullptr;
  long long* gfb = (DET && p.gfix) ? p.gfix + (long)b * p.H * p.W * p.C : nullptr;
  const int gtl_n = p.GC * K;  // (group, tap) pairs of this chunk
  if (FIXL && gxb && (p.abl & 4)) {
    fscale = 1048576.f;
    finv = 1.f / 1048576.f;
  } else if (FIXL && gxb) {
    // per-workgroup fixed-point scale from the tile's max |dy|, read with phase A's own access pattern: every wave sees the
    // whole tile, so a wave reduction is all it takes (no LDS, no barrier, 16-byte loads that phase A then finds in L2).
    // |mask| is bounded by the constant DCN_MASK_BOUND instead of a second maxima pass over the tile's masks (scalar loads:
    // the two passes together were 21 of the kernel's 169 us); a sample whose |mask| exceeds it goes to gx through the
    // global-atomic path that samples leaving the region already take.
    float mdy = 0.f;
    for (int sub = 0; sub < (DCN_TILE * DCN_TILE) / 16; ++sub) {
      const int py = oy0 + sub * 2 + (row >> 3), px = ox0 + (row & 7);
      if (py >= p.Ho || px >= p.Wo) continue;
      const long m = ((long)b * p.Ho + py) * p.Wo + px;
#pragma unroll
      for (int ks = 0; ks < KSO; ++ks) {
        const int c0 = (ks * 4 + kq) * 4;
        if (c0 + 3 < p.Co) {
          const f32x4 v = ld4(p.dy + m * p.Co + c0);
          mdy = fmaxf(fmaxf(mdy, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        } else {
          for (int q = 0; q < 4; ++q)
            if (c0 + q < p.Co) mdy = fmaxf(mdy, fabsf(ld1(p.dy + m * p.Co + c0 + q)));
        }
      }
    }
    mdy = wave_max(mdy);
    const float bound = mdy * (p.msk ? DCN_MASK_BOUND : 1.f) * p.wnorm[0];
    int e = -60;
    if (bound > 0.f && bound < 3.0e38f) (void)frexpf(bound, &e);   // bound = m * 2^e, m in [0.5, 1): every |gcol * mask| < 2^e
    e = e < -60 ? -60 : (e > 90 ? 90 : e);
    fscale = ldexpf(1.f, FIXB - e);
    finv = ldexpf(1.f, e - FIXB);
  }
  __syncthreads();

  for (int sub = 0; sub < (DCN_TILE * DCN_TILE) / 16; ++sub) {
    // ---- A: column gradient of this sub-tile (2 rows x 8 cols)
    {
      const int py = oy0 + sub * 2 + (row >> 3), px = ox0 + (row & 7);
      const bool pv = py < p.Ho && px < p.Wo;
      const long m = ((long)b * p.Ho + py) * p.Wo + px;
      f32x4 a[KSO];
#pragma unroll
      for (int ks = 0; ks < KSO; ++ks) {
        const int c0 = (ks * 4 + kq) * 4;
        a[ks] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (pv) {
          if (c0 + 3 < p.Co) a[ks] = ld4(p.dy + m * p.Co + c0);
          else
            for (int q = 0; q < 4; ++q)
              if (c0 + q < p.Co) a[ks][q] = ld1(p.dy + m * p.Co + c0 + q);
        }
      }
      for (int nt = wave; nt < p.NTc; nt += 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* wb = p.wpb + ((long)(chunk * p.NTc + nt) * KSO) * 256 + lane * 4;
#pragma unroll
        for (int ks = 0; ks < KSO; ++ks) {
          const f32x4 bw = *reinterpret_cast<const f32x4*>(wb + ks * 256);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][q], bw[q], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) gcol[(kq * 4 + r) * gstride + nt * 16 + row] = acc[r];
      }
    }
    __syncthreads();
    // ---- B: gather / scatter
    for (int i = tid; i < 16 * gtl_n; i += 256) {
      const int pix = i / gtl_n, gtl = i - pix * gtl_n;
      const int py = oy0 + sub * 2 + (pix >> 3), px = ox0 + (pix & 7);
      if (py >= p.Ho || px >= p.Wo) continue;
      const long m = ((long)b * p.Ho + py) * p.Wo + px;
      const int gt = chunk * gtl_n + gtl;
      const int g = gt / K, tap = gt - g * K;
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      const f32x2 o = ld2(p.off + m * p.ostr + gt * 2);
      const float mk = p.msk ? ld1(p.msk + m * p.mstr + gt) : 1.f;
      const float sy = (float)(py * p.stride - p.pad + ky * p.dil) + o.x;
      const float sx = (float)(px * p.stride - p.pad + kx * p.dil) + o.y;
      const float fy = floorf(sy), fx = floorf(sx);
      const float ly = sy - fy, lx = sx - fx, hy = 1.f - ly, hx = 1.f - lx;
      // clamp so the int conversion cannot overflow for wild offsets; clamped values are out of range anyway
      const int y0 = (int)fminf(fmaxf(fy, -4.f), (float)p.H + 2.f), x0 = (int)fminf(fmaxf(fx, -4.f), (float)p.W + 2.f);
      const bool yv0 = (unsigned)y0 < (unsigned)p.H, yv1 = (unsigned)(y0 + 1) < (unsigned)p.H;
      const bool xv0 = (unsigned)x0 < (unsigned)p.W, xv1 = (unsigned)(x0 + 1) < (unsigned)p.W;
      const bool v00 = yv0 && xv0, v01 = yv0 && xv1, v10 = yv1 && xv0, v11 = yv1 && xv1;
      const long o00 = ((long)y0 * p.W + x0) * p.C;
      const long o01 = o00 + p.C, o10 = o00 + (long)p.W * p.C, o11 = o10 + p.C;
      const float w00 = hy * hx, w01 = hy * lx, w10 = ly * hx, w11 = ly * lx;
      // region coordinates of the (y0, x0) corner; the four corners are inside iff 0 <= r < R-1
      const int ry = y0 - ry0, rx = x0 - rx0;
      const bool inreg = ry >= 0 && ry + 1 < p.RH && rx >= 0 && rx + 1 < p.RW && (!FIXL || fabsf(mk) <= DCN_MASK_BOUND);
      float gm = 0.f, gpy = 0.f, gpx = 0.f;
      for (int q = 0; q < q4; ++q) {
        const int cl = (gtl / K) * p.cg + q * 4;  // channel within the chunk
        const T* cb = xb + chunk * Cc + cl;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 a00 = v00 ? ld4(cb + o00) : z;
        const f32x4 a01 = v01 ? ld4(cb + o01) : z;
        const f32x4 a10 = v10 ? ld4(cb + o10) : z;
        const f32x4 a11 = v11 ? ld4(cb + o11) : z;
        float* gp = gcol + pix * gstride + cl * K + tap;  // column (channel, tap): 4 channels are K apart
        const f32x4 gc = {gp[0], gp[K], gp[2 * K], gp[3 * K]};
        const f32x4 val = ((a00 * w00 + a01 * w01) + a10 * w10) + a11 * w11;
#pragma unroll
        for (int c = 0; c < 4; ++c) gp[c * K] = val[c] * mk;
        const f32x4 gv = gc * mk;
        const f32x4 dpy = hx * (a10 - a00) + lx * (a11 - a01);
        const f32x4 dpx = hy * (a01 - a00) + ly * (a11 - a10);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          gm += gc[c] * val[c];
          gpy += gv[c] * dpy[c];
          gpx += gv[c] * dpx[c];
        }
        if (DET) {
          if (gfb) {
            if (inreg) {
              long long* r00 = regfix + ((long)ry * p.RW + rx) * Cc + cl;
#pragma unroll
              for (int c0 = 0; c0 < 4; ++c0) {
                const int c = (c0 + pix) & 3;
                const float gvc = c == 0 ? gv[0] : c == 1 ? gv[1] : c == 2 ? gv[2] : gv[3];
                if (v00) fix_add(r00 + c, gvc * w00, fscale);
                if (v01) fix_add(r00 + Cc + c, gvc * w01, fscale);
                if (v10) fix_add(r00 + p.RW * Cc + c, gvc * w10, fscale);
                if (v11) fix_add(r00 + p.RW * Cc + Cc + c, gvc * w11, fscale);
              }
            } else {
              long long* g00 = gfb + chunk * Cc + cl;
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                if (v00) fix_add(g00 + o00 + c, gv[c] * w00, fscale);
                if (v01) fix_add(g00 + o01 + c, gv[c] * w01, fscale);
                if (v10) fix_add(g00 + o10 + c, gv[c] * w10, fscale);
                if (v11) fix_add(g00 + o11 + c, gv[c] * w11, fscale);
              }
            }
          }
        } else if (FIX32 && gxb) {
          if (inreg) {
            int* r00 = reg32 + (ry * p.RW + rx) * Cc + cl;
            const f32x4 gs = gv * fscale;
#pragma unroll
            for (int c0 = 0; c0 < 4; ++c0) {
              const int c = (c0 + pix) & 3;
              const float gvc = c == 0 ? gs[0] : c == 1 ? gs[1] : c == 2 ? gs[2] : gs[3];
              if (v00) atomicAdd(r00 + c, __float2int_rn(gvc * w00));
              if (v01) atomicAdd(r00 + Cc + c, __float2int_rn(gvc * w01));
              if (v10) atomicAdd(r00 + p.RW * Cc + c, __float2int_rn(gvc * w10));
              if (v11) atomicAdd(r00 + p.RW * Cc + Cc + c, __float2int_rn(gvc * w11));
            }
          } else {
            float* g00 = gxb + chunk * Cc + cl;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (v00) unsafeAtomicAdd(g00 + o00 + c, gv[c] * w00);
              if (v01) unsafeAtomicAdd(g00 + o01 + c, gv[c] * w01);
              if (v10) unsafeAtomicAdd(g00 + o10 + c, gv[c] * w10);
              if (v11) unsafeAtomicAdd(g00 + o11 + c, gv[c] * w11);
            }
          }
        } else if (FIXL && gxb) {
          if (p.abl & 2) {
          } else if (inreg) {
            long long* r00 = regfix + ((long)ry * p.RW + rx) * Cc + cl;
            const f32x4 gs = gv * fscale;
#pragma unroll
            for (int c0 = 0; c0 < 4; ++c0) {
              const int c = (c0 + pix) & 3;                       // neighbours start on different words (see below)
              const float gvc = c == 0 ? gs[0] : c == 1 ? gs[1] : c == 2 ? gs[2] : gs[3];
              if (v00) lds_fix_add(r00 + c, gvc * w00);
              if (v01) lds_fix_add(r00 + Cc + c, gvc * w01);
              if (v10) lds_fix_add(r00 + p.RW * Cc + c, gvc * w10);
              if (v11) lds_fix_add(r00 + p.RW * Cc + Cc + c, gvc * w11);
            }
          } else {
            float* g00 = gxb + chunk * Cc + cl;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (v00) unsafeAtomicAdd(g00 + o00 + c, gv[c] * w00);
              if (v01) unsafeAtomicAdd(g00 + o01 + c, gv[c] * w01);
              if (v10) unsafeAtomicAdd(g00 + o10 + c, gv[c] * w10);
              if (v11) unsafeAtomicAdd(g00 + o11 + c, gv[c] * w11);
            }
          }
        } else if (gxb) {
          if (inreg) {
            float* r00 = region + ((long)ry * p.RW + rx) * Cc + cl;
            // Neighbouring pixels of a wave often land on the same input position at the same tap; with every lane on
            // channel c in step c those lanes collide in the compare-and-swap and retry.  Pixel `pix` walks the
            // channels starting at pix & 3, so neighbours are on different words in any one step.
#pragma unroll
            for (int c0 = 0; c0 < 4; ++c0) {
              const int c = (c0 + pix) & 3;
              const float gvc = c == 0 ? gv[0] : c == 1 ? gv[1] : c == 2 ? gv[2] : gv[3];
              if (v00) lds_add_f32(r00 + c, gvc * w00);
              if (v01) lds_add_f32(r00 + Cc + c, gvc * w01);
              if (v10) lds_add_f32(r00 + p.RW * Cc + c, gvc * w10);
              if (v11) lds_add_f32(r00 + p.RW * Cc + Cc + c, gvc * w11);
            }
          } else {
            float* g00 = gxb + chunk * Cc + cl;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (v00) unsafeAtomicAdd(g00 + o00 + c, gv[c] * w00);
              if (v01) unsafeAtomicAdd(g00 + o01 + c, gv[c] * w01);
              if (v10) unsafeAtomicAdd(g00 + o10 + c, gv[c] * w10);
              if (v11) unsafeAtomicAdd(g00 + o11 + c, gv[c] * w11);
            }
          }
        }
      }
      if (p.goff) {
        T* qo = p.goff + m * p.ostr + gt * 2;
        f32x2 go = {gpy, gpx};
        if (p.acc_off) go += ld2(qo);
        st2(qo, go);
      }
      if (p.gmsk) {
        T* qm = p.gmsk + m * p.mstr + gt;
        st1(qm, p.acc_off ? ld1(qm) + gm : gm);
      }
    }
    __syncthreads();
    // ---- C: flush the modulated samples (weight-gradient operand)
    if (p.col) {
      const int v4 = CKc >> 2;
      for (int e = tid; e < 16 * v4; e += 256) {
        const int pix = e / v4, j = e - pix * v4;
        const int py = oy0 + sub * 2 + (pix >> 3), px = ox0 + (pix & 7);
        if (py >= p.Ho || px >= p.Wo) continue;
        const long m = ((long)b * p.Ho + py) * p.Wo + px;
        const long kbase = (long)chunk * CKc + j * 4;
        if (kbase + 3 < CK) st4(p.col + m * CK + kbase, *reinterpret_cast<const f32x4*>(gcol + pix * gstride + j * 4));
      }
    }
    __syncthreads();
  }
  // ---- region -> gx
  if (DET) {
    if (gfb) {
      for (int e = tid; e < p.RH * p.RW * Cc; e += 256) {
        const int c = e % Cc, pos = e / Cc;
        const int ry = pos / p.RW, rx = pos - ry * p.RW;
        const int gy = ry0 + ry, gxx = rx0 + rx;
        if ((unsigned)gy >= (unsigned)p.H || (unsigned)gxx >= (unsigned)p.W) continue;
        const long long v = regfix[e];
        if (v != 0ll)
          atomicAdd(reinterpret_cast<unsigned long long*>(gfb + ((long)gy * p.W + gxx) * p.C + chunk * Cc + c), (unsigned long long)v);
      }
    }
  } else if (FIXL && gxb) {
    if (p.abl & 1) return;
    for (int e = tid; e < p.RH * p.RW * Cc; e += 256) {
      const long long v = FIX32 ? (long long)reg32[e] : regfix[e];
      if (v == 0ll) continue;
      const int c = e % Cc, pos = e / Cc;
      const int ry = pos / p.RW, rx = pos - ry * p.RW;
      const int gy = ry0 + ry, gxx = rx0 + rx;
      if ((unsigned)gy >= (unsigned)p.H || (unsigned)gxx >= (unsigned)p.W) continue;
      unsafeAtomicAdd(gxb + ((long)gy * p.W + gxx) * p.C + chunk * Cc + c, (float)v * finv);
    }
  } else if (gxb) {
    const int c4n = Cc >> 2;
    for (int e = tid; e < p.RH * p.RW * c4n; e += 256) {
      const int c4 = e % c4n, pos = e / c4n;
      const int ry = pos / p.RW, rx = pos - ry * p.RW;
      const int gy = ry0 + ry, gxx = rx0 + rx;
      if ((unsigned)gy >= (unsigned)p.H || (unsigned)gxx >= (unsigned)p.W) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(region + (long)pos * Cc + c4 * 4);
      float* dst = gxb + ((long)gy * p.W + gxx) * p.C + chunk * Cc + c4 * 4;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (v[c] != 0.f) unsafeAtomicAdd(dst + c, v[c]);
    }
  }
}

// ------------------------------------------------------------------ DCN backward, register-fed form (round 5; default)
// dcn_bwd_kernel above spends its time in vector instructions (PMC: 36 M wave instructions per B = 4 launch, 770 per
// 64 (pixel, group, tap) items: three runtime integer divisions and 64-bit address arithmetic per item, per-corner
// predication around every load and every LDS add, the column gradient through an LDS tile with three barriers per 16
// pixels, every channel chunk of a tile in a workgroup of its own that repeats phase A's loads).  This kernel is the
// forward's register-fed scheme (dcn_fwd_direct_kernel) run backwards:
//   * a wave owns a 2 x 8 sub-tile of an 8 x 8 pixel tile and walks the chunk's (tap, group) items four at a time -- lane
//     (pixel, kq) owns item 4q + kq of quad q.  The column gradient of the quad, gcol^T[16 columns][16 pixels] = W^T dy^T, comes
//     out of the matrix core in exactly that shape: accumulator register r of lane (pixel, kq) is channel r of item 4q + kq at
//     its pixel.  No LDS tile, no barrier between the GEMM and the gather; the dy fragments stay in registers for the whole
//     kernel (f32: v_mfma_f32_16x16x4_f32 with the reduction index permuted so that a lane's Co / 4 values are contiguous in
//     memory; 16-bit storage: v_mfma_f32_16x16x32 on the matrix pipe, the weights rounded to the storage type as in the forward);
//   * the lane gathers its item's four corners exactly as the forward does (clamped addresses, zeroed weights for corners
//     outside the map, 32-bit byte offsets, a decode table instead of divisions), forms the modulated sample (weight-gradient
//     operand `col`, written as one 16-byte store per lane in the kernel's own column order: fami_dcn_col_dw_unpermute_f32 puts
//     the weight gradient back into OIHW order), the mask gradient, the two offset gradients, and
//   * scatters gcol x mask x bilinear weight into the workgroup's fixed-point LDS region with UNCONDITIONAL adds: region cells
//     outside the map are simply never flushed, so no corner needs a predicate; a sample whose corners leave the region
//     (offset beyond DCN2_RO pixels) or whose |mask| exceeds the fixed-point bound takes per-corner global atomics.
// Channel chunks (GC groups per workgroup) only bound the region's LDS footprint, so that 4-6 workgroups share a CU.
#define DCN2_RO 3      // offset reach of the LDS region beyond the taps (as DCN_RO of the general kernel)
template <typename T>
struct DcnBwd2Args {
  const T* x;
  const T* off;
  const T* msk;        // may be null => 1
  const T* dy;
  const float* wimg;   // f32 storage: [nchunk][NQ][NCO][64][4]; 16-bit: [nchunk][NQ][NS][64][8] (fp32 values, converted in the kernel)
  T* col;              // [P][colw] or null
  float* gx;           // fp32, accumulated with atomics, or null
  T* goff;
  T* gmsk;
  const float* wnorm;
  int B, H, W, C, Ho, Wo, Co, G, K, kw, pad, dil, ostr, mstr, acc_off;
  int GC, nchunk, NQ, colw;
  int tilesX, tilesY, RH, RW, CS;     // region rows / columns, cells per region position (GC * 4 + 1: neighbours in different banks)
};

// item li of a chunk (tap-major: li = tap * GC + gl) -> column block of the original weight.view(Co, C*K) order
__device__ __host__ __forceinline__ int dcn2_kidx(int li, int r, int chunk, int GC, int K) {
  const int tap = li / GC, gl = li - tap * GC;
  return ((chunk * GC + gl) * 4 + r) * K + tap;
}
// f32 image: wimg[(((chunk*NQ + q)*NCO + j4)*64 + lane)*4 + t] = W[co = (lane>>4)*NCO*4 + j4*4 + t][column of row lane&15 of quad q]
// 16-bit image: wimg[(((chunk*NQ + q)*NS + m)*64 + lane)*8 + i] = W[co = m*32 + (lane>>4)*8 + i][same column]   (0 past Co / past the items)
__global__ void dcn_pack_wb2_kernel(const float* __restrict__ w, float* __restrict__ img, int Co, int C, int K, int GC,
                                    int nchunk, int NQ, int NJ, int half) {
  const int per = half ? 512 : 256;
  const long total = (long)nchunk * NQ * NJ * per;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i % per);
    long r = i / per;
    const int j = (int)(r % NJ);
    r /= NJ;
    const int q = (int)(r % NQ), chunk = (int)(r / NQ);
    const int lane = half ? e >> 3 : e >> 2, t = half ? e & 7 : e & 3;
    const int row = lane & 15, kq = lane >> 4;
    const int co = half ? j * 32 + kq * 8 + t : kq * NJ * 4 + j * 4 + t;
    const int li = 4 * q + (row >> 2);
    float v = 0.f;
    if (co < Co && li < GC * K) v = w[(long)co * C * K + dcn2_kidx(li, row & 3, chunk, GC, K)];
    img[i] = v;
  }
}
// dw[co][c*K + tap] (=|+=) dwp[co][column of the kernel's order]   (dwp: [Co][colw] from the 1x1 weight gradient over `col`)
__global__ void dcn_dw_unpermute_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Co, int K, int GC,
                                        int nchunk, int NQ, int colw, int CK, int accumulate) {
  const long total = (long)Co * nchunk * NQ * 16;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i & 15);
    long r = i >> 4;
    const int q = (int)(r % NQ);
    r /= NQ;
    const int chunk = (int)(r % nchunk), co = (int)(r / nchunk);
    const int li = 4 * q + (row >> 2);
    if (li >= GC * K) continue;
    const int k = dcn2_kidx(li, row & 3, chunk, GC, K);
    const float v = dwp[(long)co * colw + (chunk * NQ + q) * 16 + row];
    float* d = dw + (long)co * CK + k;
    *d = accumulate ? *d + v : v;
  }
}

// floor(v + 0.5) as an integer in one instruction (v_cvt_rpi_i32_f32; __float2int_rn is v_rndne + v_cvt)
__device__ __forceinline__ int dcn2_rint(float v) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ void dcn2_add64(long long* a, float v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(a), (unsigned long long)(long long)dcn2_rint(v));
}

// STG: the offset / mask gradients of a wave's 16 pixels are collected in a wave-private LDS buffer and written with consecutive
// lanes on consecutive elements at the end (PMC: one 8- / 4-byte store per lane and item was 218 MB of write requests per f32
// launch against 90 MB of algorithmic writes -- every partial store its own 32-byte sector); costs 13.8 KB of LDS per workgroup
template <typename T, int NCO, bool FIX32, bool STG>
__global__ __launch_bounds__(256, 3) void dcn_bwd2_kernel(DcnBwd2Args<T> p) {
  constexpr bool HALF = sizeof(T) == 2;
  constexpr int FIXB = FIX32 ? 20 : 30;            // bits of one contribution at the bound (see dcn_bwd_kernel)
  constexpr int NS = (NCO + 1) / 2;                // 32-wide reduction steps of the 16-bit matrix instruction
  extern __shared__ __attribute__((aligned(16))) char smem2[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, kq = lane >> 4;
  const int nli = p.GC * p.K, Cc = p.GC * 4, CS = p.CS;
  const int rcells = p.RH * p.RW * CS;
  long long* reg64 = reinterpret_cast<long long*>(smem2);
  int* reg32 = reinterpret_cast<int*>(smem2);
  char* after = smem2 + (((size_t)rcells * (FIX32 ? 4 : 8) + 15) & ~(size_t)15);
  int4* tapt = reinterpret_cast<int4*>(after);                                  // [NQ * 4]
  float* wmax = reinterpret_cast<float*>(after + (size_t)p.NQ * 4 * sizeof(int4));   // [4]
  const int grun = 3 * nli;                                                          // staged gradient elements per pixel: 2 nli offsets | nli masks
  T* gst = reinterpret_cast<T*>(after + (size_t)p.NQ * 4 * sizeof(int4) + 16) + (STG ? wave * 16 * grun : 0);   // [16 px][grun]
  int t, chunk;
  xcd_tile(1, t, chunk);      // neighbouring tiles and all chunks of a tile on one XCD
  const int tx = t % p.tilesX;
  t /= p.tilesX;
  const int ty = t % p.tilesY, b = t / p.tilesY;
  const int oy = ty * 8 + 2 * wave + (px >> 3), ox = tx * 8 + (px & 7);
  const bool pv = oy < p.Ho && ox < p.Wo;
  const long m = pv ? ((long)b * p.Ho + oy) * p.Wo + ox : (long)b * p.Ho * p.Wo;
  const int ry0 = ty * 8 - p.pad - DCN2_RO, rx0 = tx * 8 - p.pad - DCN2_RO;

  if (FIX32) { for (int i = tid; i < rcells; i += 256) reg32[i] = 0; }
  else { for (int i = tid; i < rcells; i += 256) reg64[i] = 0ll; }
  for (int it = tid; it < p.NQ * 4; it += 256) {
    int4 e = {0, 0, 0, 0};
    if (it < nli) {
      const int tap = it / p.GC, gl = it - tap * p.GC;
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      e = int4{(chunk * Cc + gl * 4) * (int)sizeof(T), ky * p.dil, kx * p.dil, (chunk * p.GC + gl) * p.K + tap};
    }
    tapt[it] = e;
  }
  // ---- dy fragments of this lane's pixel (kept for the whole kernel) and the tile's max |dy| (fixed-point scale)
  typedef typename std::conditional<HALF, T, __bf16>::type HT;
  typedef HT hx8 __attribute__((ext_vector_type(8)));
  f32x4 dyf[HALF ? 1 : NCO];
  hx8 dyh[HALF ? NS : 1];
  float mdy = 0.f;
  if constexpr (HALF) {
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
      const int co0 = s2 * 32 + kq * 8;
      hx8 v;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (HT)0.f;
      if (pv && co0 < p.Co) v = *reinterpret_cast<const hx8*>(p.dy + m * p.Co + co0);
      dyh[s2] = v;
#pragma unroll
      for (int i = 0; i < 8; ++i) mdy = fmaxf(mdy, fabsf((float)v[i]));
    }
  } else {
#pragma unroll
    for (int j = 0; j < NCO; ++j) {
      dyf[j] = pv ? ld4(p.dy + m * p.Co + kq * NCO * 4 + j * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      mdy = fmaxf(fmaxf(mdy, fmaxf(fabsf(dyf[j][0]), fabsf(dyf[j][1]))), fmaxf(fabsf(dyf[j][2]), fabsf(dyf[j][3])));
    }
  }
  mdy = wave_max(mdy);
  if (lane == 0) wmax[wave] = mdy;
  __syncthreads();
  float fscale, finv;
  {
    mdy = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const float bound = mdy * (p.msk ? DCN_MASK_BOUND : 1.f) * p.wnorm[0];
    int e = -60;
    if (bound > 0.f && bound < 3.0e38f) (void)frexpf(bound, &e);   // every |gcol * mask| < 2^e
    e = e < -60 ? -60 : (e > 90 ? 90 : e);
    fscale = ldexpf(1.f, FIXB - e);
    finv = ldexpf(1.f, e - FIXB);
  }

  const char* xbase = reinterpret_cast<const char*>(p.x) + (long)b * p.H * p.W * p.C * sizeof(T);
  float* gxb = p.gx ? p.gx + (long)b * p.H * p.W * p.C : nullptr;
  const unsigned WCb = p.W * p.C * (unsigned)sizeof(T), Cb = p.C * (unsigned)sizeof(T);
  const int Hm1 = p.H - 1, Wm1 = p.W - 1;
  const float Hf1 = sgpr_f((float)(p.H + 1)), Wf1 = sgpr_f((float)(p.W + 1));
  const int sy0 = oy - p.pad, sx0 = ox - p.pad;
  const T* offp = p.off + m * p.ostr;
  const T* mskp = p.msk ? p.msk + m * p.mstr : nullptr;

  // per-lane bases (32-bit element offsets inside the loop)
  T* const goffp = p.goff ? p.goff + m * p.ostr : nullptr;
  T* const gmskp = p.gmsk ? p.gmsk + m * p.mstr : nullptr;
  T* const colp = (p.col && pv) ? p.col + m * p.colw + chunk * p.NQ * 16 + kq * 4 : nullptr;
  const int rwcs = p.RW * CS;
  // offsets / mask of the quad in flight are requested one quad ahead
  int4 te = tapt[kq];
  f32x2 ov = pv ? ld2(offp + te.w * 2) : f32x2{0.f, 0.f};
  float mv = (pv && mskp) ? ld1(mskp + te.w) : (pv ? 1.f : 0.f);
  for (int q = 0; q < p.NQ; ++q) {
    const int li = 4 * q + kq;
    const bool iv = pv && li < nli;
    const int4 tc = te;
    const f32x2 o = ov;
    const float mk = iv ? mv : 0.f;
    // ---- the item's sample position and its four corner loads FIRST: they do not depend on the column gradient, so the matrix
    // instructions below run while the loads are in flight
    const float sy = (float)(sy0 + tc.y) + o.x, sx = (float)(sx0 + tc.z) + o.y;
    const float fy = floorf(sy), fx = floorf(sx);
    const float ly = sy - fy, lx = sx - fx, hy = 1.f - ly, hx = 1.f - lx;
    const int y0 = (int)__builtin_amdgcn_fmed3f(fy, -2.f, Hf1), x0 = (int)__builtin_amdgcn_fmed3f(fx, -2.f, Wf1);
    const bool yv0 = (unsigned)y0 <= (unsigned)Hm1, yv1 = (unsigned)(y0 + 1) <= (unsigned)Hm1;
    const bool xv0 = (unsigned)x0 <= (unsigned)Wm1, xv1 = (unsigned)(x0 + 1) <= (unsigned)Wm1;
    const unsigned r0 = __umul24(min(max(y0, 0), Hm1), WCb), r1 = __umul24(min(max(y0 + 1, 0), Hm1), WCb);
    const unsigned c0 = __umul24(min(max(x0, 0), Wm1), Cb) + (unsigned)tc.x, c1 = __umul24(min(max(x0 + 1, 0), Wm1), Cb) + (unsigned)tc.x;
    const f32x4 a00 = ld4(reinterpret_cast<const T*>(xbase + (r0 + c0))), a01 = ld4(reinterpret_cast<const T*>(xbase + (r0 + c1)));
    const f32x4 a10 = ld4(reinterpret_cast<const T*>(xbase + (r1 + c0))), a11 = ld4(reinterpret_cast<const T*>(xbase + (r1 + c1)));
    if (q + 1 < p.NQ) {
      te = tapt[4 * (q + 1) + kq];
      ov = pv ? ld2(offp + te.w * 2) : f32x2{0.f, 0.f};
      mv = (pv && mskp) ? ld1(mskp + te.w) : (pv ? 1.f : 0.f);
    }
    // ---- column gradient of the quad: gc[r] = sum_co dy[pixel][co] W[co][channel r of item li]   (independent accumulators)
    f32x4 gc;
    if constexpr (HALF) {
      const float* wb = p.wimg + ((long)(chunk * p.NQ + q) * NS) * 512 + lane * 8;
      f32x4 g2[NS];
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wb + s2 * 512), w1 = *reinterpret_cast<const f32x4*>(wb + s2 * 512 + 4);
        const hx8 wv = {(HT)w0[0], (HT)w0[1], (HT)w0[2], (HT)w0[3], (HT)w1[0], (HT)w1[1], (HT)w1[2], (HT)w1[3]};
        g2[s2] = H16<HT>::mfma(wv, dyh[s2], f32x4{0.f, 0.f, 0.f, 0.f});
      }
      gc = g2[0];
#pragma unroll
      for (int s2 = 1; s2 < NS; ++s2) gc += g2[s2];
    } else {
      const float* wb = p.wimg + ((long)(chunk * p.NQ + q) * NCO) * 256 + lane * 4;
      f32x4 wv[NCO], g2[NCO];
#pragma unroll
      for (int j = 0; j < NCO; ++j) wv[j] = *reinterpret_cast<const f32x4*>(wb + j * 256);
#pragma unroll
      for (int j = 0; j < NCO; ++j) g2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int j = 0; j < NCO; ++j) g2[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j][tt], dyf[j][tt], g2[j], 0, 0, 0);
      gc = g2[0];
#pragma unroll
      for (int j = 1; j < NCO; ++j) gc += g2[j];
    }
    const float wy0 = yv0 ? hy : 0.f, wy1 = yv1 ? ly : 0.f, wx0 = xv0 ? hx : 0.f, wx1 = xv1 ? lx : 0.f;
    const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
    // corner order of the oracle's sum (and of the forward)
    const f32x4 val = ((a00 * w00 + a01 * w01) + a10 * w10) + a11 * w11;
    if (colp) st4(colp + q * 16, val * mk);
    // d sample / d position: zero-padded corners, the derivative of each 1-D weight keeps its sign
    const f32x4 z00 = (yv0 && xv0) ? a00 : f32x4{0.f, 0.f, 0.f, 0.f}, z01 = (yv0 && xv1) ? a01 : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 z10 = (yv1 && xv0) ? a10 : f32x4{0.f, 0.f, 0.f, 0.f}, z11 = (yv1 && xv1) ? a11 : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 gv = gc * mk;
    const f32x4 dpy = hx * (z10 - z00) + lx * (z11 - z01);
    const f32x4 dpx = hy * (z01 - z00) + ly * (z11 - z10);
    float gm = 0.f, gpy = 0.f, gpx = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      gm += gc[c] * val[c];
      gpy += gv[c] * dpy[c];
      gpx += gv[c] * dpx[c];
    }
    if (STG) {
      if (li < nli) {                                  // (pixels outside the map stage zeros that are never written out)
        const int gtl = tc.w - chunk * nli;            // (group, tap) index within the chunk: the memory order of the run
        st2(gst + px * grun + gtl * 2, f32x2{gpy, gpx});
        st1(gst + px * grun + 2 * nli + gtl, gm);
      }
    } else if (iv) {
      if (goffp) {
        T* qo = goffp + tc.w * 2;
        f32x2 go = {gpy, gpx};
        if (p.acc_off) go += ld2(qo);
        st2(qo, go);
      }
      if (gmskp) {
        T* qm = gmskp + tc.w;
        st1(qm, p.acc_off ? ld1(qm) + gm : gm);
      }
    }
    // ---- input gradient
    if (gxb && iv) {
      const int ry = (int)fy - ry0, rx = (int)fx - rx0;       // (fy, fx inside the clamp whenever the region test passes)
      const bool inreg = fy == (float)y0 && fx == (float)x0 && ry >= 0 && ry + 1 < p.RH && rx >= 0 && rx + 1 < p.RW &&
                         fabsf(mk) <= DCN_MASK_BOUND;
      const int cl = (tc.x / (int)sizeof(T)) - chunk * Cc;      // channel within the chunk
      if (inreg) {
        const f32x4 gs = gv * fscale;
        const int cell = (ry * p.RW + rx) * CS + cl;
        const float u00 = hy * hx, u01 = hy * lx, u10 = ly * hx, u11 = ly * lx;   // cells outside the map are never flushed
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if constexpr (FIX32) {
            atomicAdd(reg32 + cell + c, dcn2_rint(gs[c] * u00));
            atomicAdd(reg32 + cell + CS + c, dcn2_rint(gs[c] * u01));
            atomicAdd(reg32 + cell + rwcs + c, dcn2_rint(gs[c] * u10));
            atomicAdd(reg32 + cell + rwcs + CS + c, dcn2_rint(gs[c] * u11));
          } else {
            dcn2_add64(reg64 + cell + c, gs[c] * u00);
            dcn2_add64(reg64 + cell + CS + c, gs[c] * u01);
            dcn2_add64(reg64 + cell + rwcs + c, gs[c] * u10);
            dcn2_add64(reg64 + cell + rwcs + CS + c, gs[c] * u11);
          }
        }
      } else {
        float* g00 = gxb + tc.x / (int)sizeof(T);
        const long o00 = ((long)y0 * p.W + x0) * p.C;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (yv0 && xv0) unsafeAtomicAdd(g00 + o00 + c, gv[c] * w00);
          if (yv0 && xv1) unsafeAtomicAdd(g00 + o00 + p.C + c, gv[c] * w01);
          if (yv1 && xv0) unsafeAtomicAdd(g00 + o00 + (long)p.W * p.C + c, gv[c] * w10);
          if (yv1 && xv1) unsafeAtomicAdd(g00 + o00 + (long)p.W * p.C + p.C + c, gv[c] * w11);
        }
      }
    }
  }
  if (STG && (p.goff || p.gmsk)) {
    // ---- the wave's staged offset / mask gradients -> goff / gmsk: lane l takes element l, l + 64, ... of the 16 runs
    // (the wave wrote them itself: no barrier, only the LDS counter)
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
    const int tyw = ty * 8 + 2 * wave;
    const float rrun2 = 1.f / (float)(2 * nli), rrun1 = 1.f / (float)nli;
    if (p.goff) {
      for (int e = lane; e < 16 * 2 * nli; e += 64) {
        const int q16 = (int)(((float)e + 0.5f) * rrun2), i = e - q16 * 2 * nli;
        const int oyq = tyw + (q16 >> 3), oxq = tx * 8 + (q16 & 7);
        if (oyq >= p.Ho || oxq >= p.Wo) continue;
        T* dst = p.goff + (((long)b * p.Ho + oyq) * p.Wo + oxq) * p.ostr + chunk * 2 * nli + i;
        const float v = ld1(gst + q16 * grun + i);
        st1(dst, p.acc_off ? ld1(dst) + v : v);
      }
    }
    if (p.gmsk) {
      for (int e = lane; e < 16 * nli; e += 64) {
        const int q16 = (int)(((float)e + 0.5f) * rrun1), i = e - q16 * nli;
        const int oyq = tyw + (q16 >> 3), oxq = tx * 8 + (q16 & 7);
        if (oyq >= p.Ho || oxq >= p.Wo) continue;
        T* dst = p.gmsk + (((long)b * p.Ho + oyq) * p.Wo + oxq) * p.mstr + chunk * nli + i;
        const float v = ld1(gst + q16 * grun + 2 * nli + i);
        st1(dst, p.acc_off ? ld1(dst) + v : v);
      }
    }
  }
  if (!gxb) return;
  __syncthreads();
  // ---- region -> gx (f32 atomics; positions outside the map hold what zero padding discards)
  // consecutive lanes = consecutive channels of one position: the atomics of a wave instruction fall into a few cache lines (one
  // position per lane -- 64 lines per instruction -- ran the whole kernel at 300 us: the atomic units work per line request)
  const float rcc = 1.f / (float)Cc, rrw = 1.f / (float)p.RW;      // exact quotients of (n + 0.5) * (1 / d) for n < 2^20
  for (int e = tid; e < p.RH * p.RW * Cc; e += 256) {
    const int pos = (int)(((float)e + 0.5f) * rcc), c = e - pos * Cc;
    const long long v = FIX32 ? (long long)reg32[pos * CS + c] : reg64[pos * CS + c];
    if (v == 0ll) continue;
    const int ry = (int)(((float)pos + 0.5f) * rrw), rx = pos - ry * p.RW;
    const int gy = ry0 + ry, gxx = rx0 + rx;
    if ((unsigned)gy >= (unsigned)p.H || (unsigned)gxx >= (unsigned)p.W) continue;
    unsafeAtomicAdd(gxb + ((long)gy * p.W + gxx) * p.C + chunk * Cc + c, (float)v * finv);
  }
}

// ------------------------------------------------------------------ host side (templates over the storage type)
template <typename T>
static int shift_fwd_impl(const T* src, const float* t, T* out, int B, int H, int W, int C, hipStream_t s,
                          const char* nm) {
  FAMI_REQUIRE(src && t && out && B > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0, nm, "bad argument");
  hipLaunchKernelGGL(shift_fwd_kernel<T>, dim3(fami_ew_grid((long)B * H * W * (C / 4))), dim3(256), 0, s, src, t, out, B, H, W, C);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}

template <typename T>
static int shift_bwd_impl(const T* gout, const T* src, const float* t, T* gsrc, float* gt, int B, int H, int W, int C,
                          int acc_src, int acc_t, float* ws, hipStream_t s, const char* nm) {
  FAMI_REQUIRE(gout && src && t && B > 0 && (C % 4) == 0, nm, "bad argument");
  if (gsrc) {
    hipLaunchKernelGGL(shift_bwd_src_kernel<T>, dim3(fami_ew_grid((long)B * H * W * (C / 4))), dim3(256), 0, s, gout, t, gsrc, B, H, W, C, acc_src);
    FAMI_CHECK_LAUNCH(nm);
  }
  if (gt) {
    FAMI_REQUIRE(ws, nm, "workspace required for gt");
    long g = ((long)H * W * (C / 4) + 255) / 256;
    if (g > 256) g = 256;
    hipLaunchKernelGGL(shift_bwd_t_kernel<T>, dim3((int)g, B), dim3(256), 0, s, gout, src, t, ws, H, W, C);
    FAMI_CHECK_LAUNCH(nm);
    hipLaunchKernelGGL(shift_bwd_t_finalize_kernel, dim3(fami_cdiv(B * 2, 64)), dim3(64), 0, s, ws, (int)g, B, gt, acc_t);
    FAMI_CHECK_LAUNCH(nm);
  }
  return FAMI_OK;
}

// [fami_route_t] g_dcn_gather (default -1)  // fami_dcn_tune: 0 = dcn_fwd_kernel (LDS column tile), 1 = dcn_fwd_direct_kernel, 2 = dcn_fwd_win_kernel where eligible, -1 default (= 1)
// [fami_route_t] g_dcn_abl (default 0)
// [fami_route_t] g_dcn_win_r (default 0)  // fami_dcn_tune(32 + r): force the window's offset reach (benchmarks); 0 = largest that fits, up to 4

// LDS-window forward: tile / window plan.  Eligible: stride 1, 4 channels per offset group (every HRNet width),
// G*K a multiple of 4, C a multiple of 16 bytes' worth of elements, window (+ decode table) within 156 KB.
struct DcnWinPlan { int ok, TH, TW, R, WR, WC, PS, tilesX, tilesY, NQ, nsub; unsigned mul_row, mul_px; size_t lds; };
// m with (i * m) >> 24 == i / d for every 0 <= i < n (0 if there is none below 2^32 / n)
static unsigned dcn_div_mul(int d, int n) {
  const unsigned long m = ((1ul << 24) + d - 1) / d;
  if (m * (unsigned long)n >= (1ul << 32)) return 0;
  if ((m * d - (1ul << 24)) * (unsigned long)n >= (1ul << 24)) return 0;   // error term stays below one quotient step
  return (unsigned)m;
}
static DcnWinPlan dcn_win_plan(int B, int Ho, int Wo, int C, int G, int kh, int kw, int stride, int dil, int esz) {
  DcnWinPlan q;
  q.ok = 0;
  const int GK = G * kh * kw;
  if (stride != 1 || C != 4 * G || (C * esz) % 16 != 0 || GK < 4 || (GK & 3) || (kh - 1) * dil > 255 || (kw - 1) * dil > 255) return q;
  static const int cand[][2] = {{8, 16}, {6, 18}, {8, 8}, {4, 16}, {6, 12}, {4, 8}};   // <= 128 pixels = 8 sub-tiles = 16 waves
  double best = 1e30;
  const int PS = C + 16 / esz;   // + 16 bytes: neighbouring pixels start in different bank groups
  q.NQ = fami_cdiv(GK, 16);
  const int ppp = C * esz / 16;
  for (int Rr = (g_dcn_win_r ? g_dcn_win_r : 4); Rr >= (g_dcn_win_r ? g_dcn_win_r : 2); --Rr) {
    for (auto& c : cand) {
      const int TH = c[0] < Ho ? c[0] : Ho, TW = c[1] < Wo ? c[1] : Wo;
      const int WR = TH + (kh - 1) * dil + 2 * Rr, WC = TW + (kw - 1) * dil + 2 * Rr;
      const size_t lds = (size_t)WR * WC * PS * esz + (size_t)q.NQ * 16 * 4;
      if (lds > 156 * 1024) continue;
      const int nsub = fami_cdiv(TH * TW, 16);
      if ((size_t)nsub * fami_cdiv(96, 16) * 4 * 64 * 4 > (size_t)WR * WC * PS * esz) continue;   // the K-half reduction reuses the window
      const unsigned mr = dcn_div_mul(WC * ppp, WR * WC * ppp), mp = dcn_div_mul(ppp, WC * ppp);
      if (!mr || !mp) continue;
      const int tX = fami_cdiv(Wo, TW), tY = fami_cdiv(Ho, TH);
      const long wgs = (long)tX * tY * B;
      // cost model: rounds over 256 CUs x (K work of the sub-tiles actually populated + window fill)
      const double cost = (double)fami_cdiv(wgs, 256) * (9.0 * nsub / 8.0 + lds / 65536.0) * ((double)nsub * 16 / (TH * TW));
      if (cost < best) {
        best = cost;
        q.ok = 1; q.TH = TH; q.TW = TW; q.R = Rr; q.WR = WR; q.WC = WC; q.PS = PS; q.tilesX = tX; q.tilesY = tY;
        q.nsub = nsub; q.lds = lds; q.mul_row = mr; q.mul_px = mp;
      }
    }
    if (q.ok) break;      // the largest reach that fits
  }
  return q;
}

template <typename T, int NT, bool ABL, int KSPLIT>
static void dcn_fwd_win_launch1(const DcnWinArgs<T>& a, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)dcn_fwd_win_kernel<T, NT, ABL, KSPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((dcn_fwd_win_kernel<T, NT, ABL, KSPLIT>), grid, dim3(512 * KSPLIT), lds, s, a);
}
// [fami_route_t] g_dcn_ksplit (default 1)  // fami_dcn_tune(256 + k): 2 = the 16-wave K-split build (measured slower: 35.5 vs 29.8 us)
template <typename T, int NT>
static void dcn_fwd_win_launch(const DcnWinArgs<T>& a, dim3 grid, size_t lds, hipStream_t s) {
  if (NT == 3 && sizeof(T) == 4 && a.abl) {   // tools/bench_dcn_win.py
    if (g_dcn_ksplit == 2) dcn_fwd_win_launch1<T, NT == 3 ? NT : 3, sizeof(T) == 4, 2>(a, grid, lds, s);
    else dcn_fwd_win_launch1<T, NT == 3 ? NT : 3, sizeof(T) == 4, 1>(a, grid, lds, s);
  } else if (g_dcn_ksplit == 2) dcn_fwd_win_launch1<T, NT, false, 2>(a, grid, lds, s);
  else dcn_fwd_win_launch1<T, NT, false, 1>(a, grid, lds, s);
}
// [fami_route_t] g_dcn_pf (default 0)  // fami_dcn_tune(16 + 2): the 2-k-groups-in-flight x 4-waves-per-SIMD build of the direct kernel (benchmarks)

template <typename T, int NT, int PF, int MINW>
static void dcn_fwd_direct_launch(const DcnArgs<T>& a, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)dcn_fwd_direct_kernel<T, NT, PF, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((dcn_fwd_direct_kernel<T, NT, PF, MINW>), grid, dim3(256), lds, s, a);
}

template <typename T, int NT>
static void dcn_fwd_launch(const DcnArgs<T>& a, dim3 grid, size_t lds, size_t lds_direct, bool direct, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)dcn_fwd_kernel<T, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  if (!direct) {
    hipLaunchKernelGGL((dcn_fwd_kernel<T, NT>), grid, dim3(256), lds, s, a);
    return;
  }
  // Measured (tools/bench_dcn.py, B=4 96x72 48 ch): 1 or 2 k groups in flight x 4..7 waves per SIMD all land within
  // +-1.5 us in f32 (31-34 us: the corner loads run at the L1's one-line-per-cycle rate), bf16 25.0 (1 x 7) vs 27.6 us
  // (2 x 4).  Default: one k group per wave in flight, 7 workgroups per CU (62 VGPRs, 22.5 KB LDS) -- the 6.75 tiles a
  // CU owns at B=4 are then all resident at once.  Wide outputs (NT > 3) carry 4*NT accumulator + 4*NT weight registers.
  if (NT > 3)
    dcn_fwd_direct_launch<T, NT, 1, 4>(a, grid, lds_direct, s);
  else if (g_dcn_pf == 2)
    dcn_fwd_direct_launch<T, NT, 2, 4>(a, grid, lds_direct, s);
  else
    dcn_fwd_direct_launch<T, NT, 1, 7>(a, grid, lds_direct, s);
}

static inline long dcn_f32_image_elems(int Co, int C, int kh, int kw, int G) {
  const long nt = fami_cdiv(Co, 16);
  return (long)fami_cdiv((long)C * kh * kw, 16) * nt * 256 + (long)fami_cdiv((long)G * kh * kw, 16) * 4 * nt * 256;
}
template <typename T>
static int dcn_fwd_impl(const T* x, const T* off, const T* msk, const float* wp, const float* bias, T* y, int B, int H,
                        int W, int C, int Co, int G, int kh, int kw, int stride, int pad, int dil, hipStream_t s,
                        const char* nm, bool merged = false) {
  FAMI_REQUIRE(x && off && wp && y && B > 0 && G > 0 && C % G == 0, nm, "bad argument");
  DcnArgs<T> a;
  // merged: `off` is ONE tensor [P][2GK offsets | GK masks] (the merged predictor's output); msk is derived here
  a.ostr = (merged ? 3 : 2) * G * kh * kw;
  a.mstr = (merged ? 3 : 1) * G * kh * kw;
  if (merged) msk = off + 2 * G * kh * kw;
  a.x = x; a.off = off; a.msk = msk; a.wp = wp; a.bias = bias; a.y = y;
  a.B = B; a.H = H; a.W = W; a.C = C; a.Co = Co; a.G = G; a.kh = kh; a.kw = kw;
  a.stride = stride; a.pad = pad; a.dil = dil;
  // the 16-bit image follows the two fp32 images (fami_dcn_packed_weight_elems); written by fami_dcn_pack_weight_bf16/_f16
  a.wp16 = wp + dcn_f32_image_elems(Co, C, kh, kw, G);
  a.Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1;
  a.Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
  a.cg = C / G;
  if ((a.cg % 4) != 0 || Co > 96) {
    fami_set_error(nm, "channels per offset group must be a multiple of 4 and Co <= 96");
    return FAMI_ESHAPE;
  }
  a.KS = fami_cdiv((long)C * kh * kw, 16);
  a.NTt = fami_cdiv(Co, 16);
  const long P = (long)B * a.Ho * a.Wo;
  FAMI_REQUIRE(P < (1L << 31), nm, "size out of range");
  a.P = (int)P;
  if (g_dcn_gather == 2) {
    const DcnWinPlan q = dcn_win_plan(B, a.Ho, a.Wo, C, G, kh, kw, stride, dil, (int)sizeof(T));
    if (q.ok && a.NTt <= 4 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(off) | reinterpret_cast<uintptr_t>(msk)) & 15) == 0) {
      DcnWinArgs<T> w;
      w.x = x; w.off = off; w.msk = msk; w.bias = bias; w.y = y;
      w.wq = wp + (long)a.KS * a.NTt * 256;      // the quad-order image follows the tap-major one
      w.B = B; w.H = H; w.W = W; w.C = C; w.Ho = a.Ho; w.Wo = a.Wo; w.Co = Co; w.G = G; w.K = kh * kw; w.kw = kw;
      w.pad = pad; w.dil = dil;
      w.TH = q.TH; w.TW = q.TW; w.tilesX = q.tilesX; w.tilesY = q.tilesY; w.R = q.R; w.WR = q.WR; w.WC = q.WC; w.PS = q.PS;
      w.NQ = q.NQ; w.NTt = a.NTt; w.nsub = q.nsub; w.mul_row = q.mul_row; w.mul_px = q.mul_px; w.abl = g_dcn_abl;
      w.ostr = a.ostr; w.mstr = a.mstr;
      const dim3 grid(q.tilesX * q.tilesY * B);
      switch (a.NTt) {
        case 1: dcn_fwd_win_launch<T, 1>(w, grid, q.lds, s); break;
        case 2: dcn_fwd_win_launch<T, 2>(w, grid, q.lds, s); break;
        case 3: dcn_fwd_win_launch<T, 3>(w, grid, q.lds, s); break;
        default: dcn_fwd_win_launch<T, 4>(w, grid, q.lds, s); break;
      }
      FAMI_CHECK_LAUNCH(nm);
      return FAMI_OK;
    }
  }
  const dim3 grid(fami_cdiv(P, DCN_PIX));
  // the direct kernel addresses x with 32-bit byte offsets (24-bit row / column products)
  const size_t lds = ((size_t)DCN_PIX * (a.KS * 16 + 4) + 4 * (size_t)a.NTt * 256) * sizeof(float);
  size_t lds_direct = (((size_t)DCN_PIX * G * kh * kw * 3 * sizeof(T) + 15) & ~(size_t)15) + (size_t)a.KS * 4 * sizeof(int4);
  if (lds_direct < 4 * (size_t)a.NTt * 256 * sizeof(float)) lds_direct = 4 * (size_t)a.NTt * 256 * sizeof(float);
  const bool direct = g_dcn_gather != 0 && (long)B * H * W * C * sizeof(T) < (1L << 32) &&
                      (long)H * W * C * sizeof(T) < (1L << 24) && lds_direct <= 150 * 1024;
  if (!direct && lds > 150 * 1024) {
    fami_set_error(nm, "C*kh*kw too large for the LDS column tile");
    return FAMI_ESHAPE;
  }
  switch (a.NTt) {
    case 1: dcn_fwd_launch<T, 1>(a, grid, lds, lds_direct, direct, s); break;
    case 2: dcn_fwd_launch<T, 2>(a, grid, lds, lds_direct, direct, s); break;
    case 3: dcn_fwd_launch<T, 3>(a, grid, lds, lds_direct, direct, s); break;
    case 4: dcn_fwd_launch<T, 4>(a, grid, lds, lds_direct, direct, s); break;
    case 5: dcn_fwd_launch<T, 5>(a, grid, lds, lds_direct, direct, s); break;
    default: dcn_fwd_launch<T, 6>(a, grid, lds, lds_direct, direct, s); break;
  }
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}

// ---- register-fed backward (dcn_bwd2_kernel): plan shared by the weight pack, the launch and the callers' `col` width
// [fami_route_t] g_dcn_bwd2 (default 1)  // fami_dcn_tune(2048 / 2049): off / on
// [fami_route_t] g_dcn_bwd2_cap (default 36)  // fami_dcn_tune(4096 + KB): LDS budget of the fixed-point region (decides the groups per workgroup; benchmarks -- set BEFORE the weight pack)
// [fami_route_t] g_dcn_bwd_abl (default 0)  // fami_dcn_tune(1024 + bits): ablations of the general backward kernel (benchmarks)
// [fami_route_t] g_dcn_bwd_scatter (default -1)  // fami_dcn_tune(512 + m): 0 = f32 compare-and-swap LDS adds, 1 / default = fixed-point LDS adds (64-bit for f32, 32-bit for 16-bit storage), 2 = 64-bit for every type
// the register-fed kernel takes the default scatter (64-bit cells for f32 storage, 32-bit for the 16-bit types); every other
// setting of the knobs above selects a form only the general kernel has
static inline bool dcn_bwd2_on(int esz) {
  return g_dcn_bwd2 && g_dcn_bwd_scatter != 0 && !(esz == 2 && g_dcn_bwd_scatter == 2) && !g_dcn_bwd_abl;
}
struct DcnBwd2Plan { int ok, GC, nchunk, NQ, colw, NJ, RH, RW, CS, stg; size_t lds; };
static DcnBwd2Plan dcn_bwd2_plan(int C, int Co, int G, int kh, int kw, int stride, int dil, int esz) {
  DcnBwd2Plan q;
  q.ok = 0; q.colw = C * kh * kw;
  const int K = kh * kw;
  if (!g_dcn_bwd2 || stride != 1 || G <= 0 || C != 4 * G || (Co % 16) != 0 || Co > 96 || K > 64) return q;
  q.RH = 7 + (kh - 1) * dil + 2 + 2 * DCN2_RO;
  q.RW = 7 + (kw - 1) * dil + 2 + 2 * DCN2_RO;
  // groups per workgroup: the largest divisor of G whose fixed-point region (64-bit cells for f32 storage, 32-bit for the
  // 16-bit types) stays within g_dcn_bwd2_cap KB at the head's dilation of 3.  The kernel is latency-bound, so the budget buys
  // resident waves: measured on the B = 4, 96 x 72, 48-channel launch (tools/bench_dcn_bwd.py, region budget 64 / 48 / 36 / 24 KB):
  // f32 102.6 / 89.9 / 82.3 / 98.4 us (groups per workgroup 4 / 3 / 2 / 1: below 2 the item quads are a quarter padding),
  // bf16 107.6 / 108.0 / 72.5 / 74.8 us (6 / 6 / 4 / 3) -- against 166 / 127 us for dcn_bwd_kernel.  Default 36 KB: five
  // workgroups (20 waves) per CU.  Chosen WITHOUT looking at the actual dilation: the weight image (packed before any launch)
  // depends on it.
  const int RHc = 7 + (kh - 1) * 3 + 2 + 2 * DCN2_RO, RWc = 7 + (kw - 1) * 3 + 2 + 2 * DCN2_RO;
  q.GC = 0;
  for (int gc = G; gc >= 1; --gc) {
    if (G % gc) continue;
    if ((size_t)RHc * RWc * (gc * 4 + 1) * (esz == 4 ? 8 : 4) <= (size_t)g_dcn_bwd2_cap * 1024) { q.GC = gc; break; }
  }
  if (!q.GC) return q;
  q.nchunk = G / q.GC;
  q.NQ = (q.GC * K + 3) / 4;
  q.colw = q.nchunk * q.NQ * 16;
  q.NJ = esz == 4 ? Co / 16 : (Co + 31) / 32;
  q.CS = q.GC * 4 + 1;
  q.stg = g_dcn_bwd2_stage;
  q.lds = (((size_t)q.RH * q.RW * q.CS * (esz == 4 ? 8 : 4) + 15) & ~(size_t)15) + (size_t)q.NQ * 4 * 16 + 16 +
          (q.stg ? (size_t)4 * 16 * 3 * q.GC * K * esz : 0);
  q.ok = q.lds <= 64 * 1024;     // (a larger dilation than the head's: the general kernel)
  if (!q.ok) q.colw = C * kh * kw;
  return q;
}
static inline long dcn_bwd_old_image_elems(int Co, int C, int kh, int kw) {
  return (long)fami_cdiv((long)C * kh * kw, 16) * fami_cdiv(Co, 16) * 256 + 4;
}
// image of the register-fed kernel for storage size esz (4 | 2): floats
static inline long dcn_bwd2_image_elems(const DcnBwd2Plan& q, int esz) {
  return q.ok ? (long)q.nchunk * q.NQ * q.NJ * (esz == 4 ? 256 : 512) : 0;
}
template <typename T, int NCO, bool STG>
static void dcn_bwd2_launch1(const DcnBwd2Args<T>& a, dim3 grid, size_t lds, hipStream_t s) {
  constexpr bool F32 = sizeof(T) == 2;       // 16-bit storage: 32-bit fixed-point region
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)dcn_bwd2_kernel<T, NCO, F32, STG>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((dcn_bwd2_kernel<T, NCO, F32, STG>), grid, dim3(256), lds, s, a);
}
template <typename T, int NCO>
static void dcn_bwd2_launch(const DcnBwd2Args<T>& a, dim3 grid, size_t lds, hipStream_t s, int stg) {
  if (stg) dcn_bwd2_launch1<T, NCO, true>(a, grid, lds, s);
  else dcn_bwd2_launch1<T, NCO, false>(a, grid, lds, s);
}

static int dcn_bwd_chunk_groups(int G, int cg, int K) {
  // smallest group count whose column span cg*K*GC is a multiple of 16 and divides G
  for (int gc = 1; gc <= G; ++gc)
    if (G % gc == 0 && (gc * cg * K) % 16 == 0) return gc;
  return 0;
}

template <typename T, int KSO, int MODE>
static void dcn_bwd_launch1(const DcnBwdArgs<T>& a, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)dcn_bwd_kernel<T, KSO, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((dcn_bwd_kernel<T, KSO, MODE>), grid, dim3(256), lds, s, a);
}
template <typename T, int KSO>
static void dcn_bwd_launch(const DcnBwdArgs<T>& a, dim3 grid, size_t lds, hipStream_t s) {
  if (a.gfix) dcn_bwd_launch1<T, KSO, 1>(a, grid, lds, s);
  else if (a.fixl == 2) {
    if constexpr (sizeof(T) == 2) dcn_bwd_launch1<T, KSO, 3>(a, grid, lds, s);
  } else if (a.fixl) dcn_bwd_launch1<T, KSO, 2>(a, grid, lds, s);
  else dcn_bwd_launch1<T, KSO, 0>(a, grid, lds, s);
}

template <typename T>
static int dcn_bwd_impl(const T* x, const T* off, const T* msk, const T* dy, const float* wpb, T* col, float* gx,
                        T* goff, T* gmsk, int B, int H, int W, int C, int Co, int G, int kh, int kw, int stride,
                        int pad, int dil, int acc_off, hipStream_t s, const char* nm, long long* gfix = nullptr,
                        const unsigned* amax_bits = nullptr, bool merged = false, bool general = false) {
  // general: the caller sized `col` for dcn_bwd_kernel ([P][C K], tap-major: what fami_dcn_bwd_col_width / _col_permuted report
  // for the deterministic mode) -- the register-fed kernel, which writes its own column order q.colw wide, must not take it
  FAMI_REQUIRE(x && off && dy && wpb && B > 0 && G > 0 && C % G == 0, nm, "bad argument");
  DcnBwdArgs<T> a;
  // merged: off / goff are ONE tensor each, [P][2GK offsets | GK masks] (merged predictor output and its gradient)
  a.ostr = (merged ? 3 : 2) * G * kh * kw;
  a.mstr = (merged ? 3 : 1) * G * kh * kw;
  if (merged) {
    msk = off + 2 * G * kh * kw;
    gmsk = goff ? goff + 2 * G * kh * kw : nullptr;
  }
  a.gfix = gfix; a.amax_bits = amax_bits;
  a.x = x; a.off = off; a.msk = msk; a.dy = dy; a.wpb = wpb; a.col = col; a.gx = gx; a.goff = goff; a.gmsk = gmsk;
  a.B = B; a.H = H; a.W = W; a.C = C; a.Co = Co; a.G = G; a.kh = kh; a.kw = kw;
  a.stride = stride; a.pad = pad; a.dil = dil; a.acc_off = acc_off;
  a.Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1;
  a.Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
  a.cg = C / G;
  const int K = kh * kw;
  {
    // register-fed kernel (default): the non-deterministic fixed-point scatter on the shapes it is built for
    const DcnBwd2Plan q = dcn_bwd2_plan(C, Co, G, kh, kw, stride, dil, (int)sizeof(T));
    if (q.ok && !gfix && !general && dcn_bwd2_on((int)sizeof(T))) {
      // (32-bit byte offsets inside a frame, 24-bit row / column products; no fallback here: the callers sized `col` for this kernel)
      FAMI_REQUIRE((long)H * W * C * sizeof(T) < (1L << 32) && (long)W * C * sizeof(T) < (1L << 24), nm, "frame too large");
      DcnBwd2Args<T> n;
      n.x = x; n.off = off; n.msk = msk; n.dy = dy; n.col = col; n.gx = gx; n.goff = goff; n.gmsk = gmsk;
      const long oldn = dcn_bwd_old_image_elems(Co, C, kh, kw);
      n.wnorm = wpb + oldn - 4;
      const DcnBwd2Plan q4 = dcn_bwd2_plan(C, Co, G, kh, kw, stride, dil, 4);
      n.wimg = wpb + oldn + (sizeof(T) == 4 ? 0 : dcn_bwd2_image_elems(q4, 4));
      n.B = B; n.H = H; n.W = W; n.C = C; n.Ho = a.Ho; n.Wo = a.Wo; n.Co = Co; n.G = G; n.K = K; n.kw = kw;
      n.pad = pad; n.dil = dil; n.ostr = a.ostr; n.mstr = a.mstr; n.acc_off = acc_off;
      n.GC = q.GC; n.nchunk = q.nchunk; n.NQ = q.NQ; n.colw = q.colw;
      n.tilesX = fami_cdiv(a.Wo, 8); n.tilesY = fami_cdiv(a.Ho, 8); n.RH = q.RH; n.RW = q.RW; n.CS = q.CS;
      const dim3 grid(n.tilesX * n.tilesY * B, q.nchunk);
      switch (Co / 16) {
        case 1: dcn_bwd2_launch<T, 1>(n, grid, q.lds, s, q.stg); break;
        case 2: dcn_bwd2_launch<T, 2>(n, grid, q.lds, s, q.stg); break;
        case 3: dcn_bwd2_launch<T, 3>(n, grid, q.lds, s, q.stg); break;
        case 4: dcn_bwd2_launch<T, 4>(n, grid, q.lds, s, q.stg); break;
        case 5: dcn_bwd2_launch<T, 5>(n, grid, q.lds, s, q.stg); break;
        default: dcn_bwd2_launch<T, 6>(n, grid, q.lds, s, q.stg); break;
      }
      FAMI_CHECK_LAUNCH(nm);
      return FAMI_OK;
    }
  }
  a.GC = (a.cg % 4 == 0) ? dcn_bwd_chunk_groups(G, a.cg, K) : 0;
  a.KSo = fami_cdiv(Co, 16);
  if (a.GC == 0 || a.KSo > 6) {
    fami_set_error(nm, "unsupported channel grouping (need cg % 4 == 0, a 16-aligned group chunk, Co <= 96)");
    return FAMI_ESHAPE;
  }
  a.NTc = a.GC * a.cg * K / 16;
  a.tilesX = fami_cdiv(a.Wo, DCN_TILE); a.tilesY = fami_cdiv(a.Ho, DCN_TILE);
  a.RH = (DCN_TILE - 1) * stride + (kh - 1) * dil + 2 + 2 * DCN_RO;
  a.RW = (DCN_TILE - 1) * stride + (kw - 1) * dil + 2 + 2 * DCN_RO;
  a.wnorm = wpb + (long)fami_cdiv((long)C * K, 16) * a.KSo * 256;       // written by fami_dcn_pack_weight_bwd_f32 behind the image
  a.fixl = (!gfix && gx && g_dcn_bwd_scatter != 0) ? 1 : 0;
  // 16-bit storage: a 32-bit fixed-point region (20 bits per contribution at the bound: ~2^-17 of the largest one, far
  // below bf16 / fp16 resolution) -- half the LDS (4 workgroups per CU instead of 2) and ds_add_u32 (4.7 against 2.5
  // lane-operations per clock per CU); fami_dcn_tune(514) keeps the 64-bit region
  if (a.fixl && sizeof(T) == 2 && g_dcn_bwd_scatter != 2) a.fixl = 2;
  a.abl = g_dcn_bwd_abl;
  size_t lds = ((size_t)16 * (a.NTc * 16 + 4) + (size_t)a.RH * a.RW * a.GC * a.cg * ((gfix || a.fixl == 1) ? 2 : 1)) * sizeof(float);
  if (a.fixl == 1 && lds > 150 * 1024) {                                  // the f32 region is half the size
    a.fixl = 0;
    lds = ((size_t)16 * (a.NTc * 16 + 4) + (size_t)a.RH * a.RW * a.GC * a.cg) * sizeof(float);
  }
  if (lds > 150 * 1024) {
    fami_set_error(nm, "tile does not fit LDS");
    return FAMI_ESHAPE;
  }
  const dim3 grid(a.tilesX * a.tilesY * B, G / a.GC);
  switch (a.KSo) {
    case 1: dcn_bwd_launch<T, 1>(a, grid, lds, s); break;
    case 2: dcn_bwd_launch<T, 2>(a, grid, lds, s); break;
    case 3: dcn_bwd_launch<T, 3>(a, grid, lds, s); break;
    case 4: dcn_bwd_launch<T, 4>(a, grid, lds, s); break;
    case 5: dcn_bwd_launch<T, 5>(a, grid, lds, s); break;
    default: dcn_bwd_launch<T, 6>(a, grid, lds, s); break;
  }
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}

// Deterministic variant: |dy| maximum -> fixed-point scale, 64-bit integer accumulation of the input gradient, one
// conversion pass into gx (activation type, =|+=).  ws: 8*B*H*W*C + 16 bytes.
template <typename T>
static int dcn_bwd_det_impl(const T* x, const T* off, const T* msk, const T* dy, const float* wpb, T* col, T* gx,
                            T* goff, T* gmsk, int B, int H, int W, int C, int Co, int G, int kh, int kw, int stride,
                            int pad, int dil, int acc_off, int acc_x, void* ws, hipStream_t s, const char* nm,
                            bool merged = false) {
  FAMI_REQUIRE(ws && dy, nm, "bad argument");
  const long n = (long)B * H * W * C;
  long long* gfix = gx ? reinterpret_cast<long long*>(ws) : nullptr;
  unsigned* amax = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + n * 8);
  if (!gx) return dcn_bwd_impl<T>(x, off, msk, dy, wpb, col, nullptr, goff, gmsk, B, H, W, C, Co, G, kh, kw, stride, pad, dil, acc_off, s, nm, nullptr, nullptr, merged, true);      // (no input gradient: nothing order-dependent is left, but `col` keeps the deterministic mode's layout)
  const int Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
  hipLaunchKernelGGL(zero_u64_kernel, dim3(fami_ew_grid(n)), dim3(256), 0, s, reinterpret_cast<unsigned long long*>(gfix), n, amax);
  FAMI_CHECK_LAUNCH(nm);
  const long ndy = (long)B * Ho * Wo * Co;
  hipLaunchKernelGGL(absmax_kernel<T>, dim3(fami_ew_grid(ndy) < 256 ? fami_ew_grid(ndy) : 256), dim3(256), 0, s, dy, ndy, amax);
  FAMI_CHECK_LAUNCH(nm);
  const int rc = dcn_bwd_impl<T>(x, off, msk, dy, wpb, col, nullptr, goff, gmsk, B, H, W, C, Co, G, kh, kw, stride, pad, dil, acc_off, s, nm, gfix, amax, merged);
  if (rc != FAMI_OK) return rc;
  hipLaunchKernelGGL(fix_to_act_kernel<T>, dim3(fami_ew_grid(n)), dim3(256), 0, s, gfix, gx, n, amax, acc_x);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}

extern "C" int fami_dcn_pack_weight_f32(const float* w_oihw, float* wp, int Co, int C, int kh, int kw, int G, hipStream_t s);
template <typename H>
static int dcn_pack_w16_impl(const char* nm, const float* w_oihw, float* wp, int Co, int C, int kh, int kw, int G, hipStream_t s) {
  const int rc = fami_dcn_pack_weight_f32(w_oihw, wp, Co, C, kh, kw, G, s);
  if (rc != FAMI_OK) return rc;
  const int K = kh * kw, KS32 = fami_cdiv(fami_cdiv((long)C * K, 16), 2), NTt = fami_cdiv(Co, 16);
  H* w16 = reinterpret_cast<H*>(wp + dcn_f32_image_elems(Co, C, kh, kw, G));
  hipLaunchKernelGGL(dcn_pack_w16_kernel<H>, dim3(fami_ew_grid((long)KS32 * NTt * 512)), dim3(256), 0, s, w_oihw, w16, Co, C, K, C / G, KS32, NTt);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}
extern "C" {

long fami_dcn_bwd_det_workspace(int B, int H, int W, int C) { return (long)B * H * W * C * 8 + 16; }

long fami_shift_workspace(int B) { return (long)B * 256 * 2 * (long)sizeof(float); }

// two fp32 images: tap-major (dcn_fwd_kernel / dcn_fwd_direct_kernel) followed by quad-group order (dcn_fwd_win_kernel);
// behind them the 16-bit tap-major image of the 16-bit modes (256 floats' worth per 32-step and channel tile)
long fami_dcn_packed_weight_elems(int Co, int C, int kh, int kw, int G) {
  const long nt = fami_cdiv(Co, 16);
  return dcn_f32_image_elems(Co, C, kh, kw, G) + (long)fami_cdiv(fami_cdiv((long)C * kh * kw, 16), 2) * nt * 256;
}

// forward weight image (fp32 for both activation types: the contraction runs on the exact f32 MFMA)
int fami_dcn_pack_weight_f32(const float* w_oihw, float* wp, int Co, int C, int kh, int kw, int G, hipStream_t s) {
  FAMI_REQUIRE(w_oihw && wp && G > 0 && C % G == 0 && ((C / G) % 4) == 0, "fami_dcn_pack_weight_f32", "channels per offset group must be a multiple of 4");
  const int K = kh * kw, cg = C / G, KS16 = fami_cdiv((long)C * K, 16), NTt = fami_cdiv(Co, 16);
  const long total = (long)KS16 * NTt * 256;
  hipLaunchKernelGGL(dcn_pack_w_kernel, dim3(fami_ew_grid(total)), dim3(256), 0, s, w_oihw, wp, Co, C, K, cg, KS16, NTt);
  FAMI_CHECK_LAUNCH("fami_dcn_pack_weight_f32");
  const int NQ = fami_cdiv((long)G * K, 16);
  hipLaunchKernelGGL(dcn_pack_wq_kernel, dim3(fami_ew_grid((long)NQ * 4 * NTt * 256)), dim3(256), 0, s, w_oihw, wp + total, Co, C, K, G, NQ, NTt);
  FAMI_CHECK_LAUNCH("fami_dcn_pack_weight_f32/quad");
  return FAMI_OK;
}
// the fp32 images plus the 16-bit image the bf16 / fp16 forward contracts with (wp: fami_dcn_packed_weight_elems floats)
int fami_dcn_pack_weight_bf16(const float* w_oihw, float* wp, int Co, int C, int kh, int kw, int G, hipStream_t s) {
  return dcn_pack_w16_impl<bf16_t>("fami_dcn_pack_weight_bf16", w_oihw, wp, Co, C, kh, kw, G, s);
}
int fami_dcn_pack_weight_f16(const float* w_oihw, float* wp, int Co, int C, int kh, int kw, int G, hipStream_t s) {
  return dcn_pack_w16_impl<f16_t>("fami_dcn_pack_weight_f16", w_oihw, wp, Co, C, kh, kw, G, s);
}

// benchmarks / tests: 0 = dcn_fwd_kernel (LDS column tile), 1 = dcn_fwd_direct_kernel (register-fed MFMA), -1 = default;
// 16 + 2 / 16 + 0 = the (2 k groups in flight, 4 waves per SIMD) build of the direct kernel / the default build
int fami_dcn_tune(int gather) {
  if (gather < 0) {   // every DCN knob back to its default
    g_dcn_bwd_abl = 0; g_dcn_bwd_scatter = -1; g_dcn_ksplit = 1; g_dcn_abl = 0; g_dcn_win_r = 0; g_dcn_pf = 0; g_dcn_gather = -1; g_dcn_bwd2 = 1; g_dcn_bwd2_cap = 36; g_dcn_bwd2_stage = 1;
    return FAMI_OK;
  }
  if (gather >= 8192) g_dcn_bwd2_stage = gather - 8192;
  else if (gather >= 4096) g_dcn_bwd2_cap = gather - 4096;
  else if (gather >= 2048) g_dcn_bwd2 = gather - 2048;
  else if (gather >= 1024) g_dcn_bwd_abl = gather - 1024;
  else if (gather >= 512) g_dcn_bwd_scatter = gather - 512;
  else if (gather >= 256) g_dcn_ksplit = gather - 256;
  else if (gather >= 64) g_dcn_abl = gather - 64;
  else if (gather >= 32) g_dcn_win_r = gather - 32;       // benchmarks: offset reach of the window kernel (0 = automatic)
  else if (gather >= 16) g_dcn_pf = gather - 16;
  else g_dcn_gather = gather;
  return FAMI_OK;
}

// the image of dcn_pack_wb_kernel + 4 floats: [0] = max_k sum_co |W[co][k]| (fixed-point scale bound of dcn_bwd_kernel)
long fami_dcn_packed_weight_bwd_elems(int Co, int C, int kh, int kw, int G) {
  // + the two images of the register-fed kernel (f32 storage / 16-bit storage: other chunking, other MFMA shape); sized for the
  // head's geometry (stride 1, dilation = padding): the plan does not depend on anything else
  long n = dcn_bwd_old_image_elems(Co, C, kh, kw);
  for (int esz = 4; esz >= 2; esz -= 2) {
    const int keep = g_dcn_bwd2;
    g_dcn_bwd2 = 1;
    n += dcn_bwd2_image_elems(dcn_bwd2_plan(C, Co, G, kh, kw, 1, 3, esz), esz);
    g_dcn_bwd2 = keep;
  }
  return n;
}
/* columns of the `col` matrix fami_dcn_bwd_* writes for this geometry and storage size: C*kh*kw (weight.view(Co, C*K) order) on
 * the general kernel, more (the register-fed kernel's own column order, padded) otherwise -- then the 1x1 weight gradient over
 * col is [Co][cols] and fami_dcn_col_dw_unpermute_f32 brings it into OIHW order */
long fami_dcn_bwd_col_width(int C, int Co, int G, int kh, int kw, int stride, int dil, int elem_bytes, int deterministic) {
  if (deterministic || !dcn_bwd2_on(elem_bytes)) return (long)C * kh * kw;
  return dcn_bwd2_plan(C, Co, G, kh, kw, stride, dil, elem_bytes).colw;
}
/* 1: col is in the register-fed kernel's column order (fami_dcn_col_dw_unpermute_f32 needed), 0: weight.view(Co, C*K) order.  (The
 * widths alone do not tell: with four groups per workgroup the permuted matrix is exactly C*K wide.) */
int fami_dcn_bwd_col_permuted(int C, int Co, int G, int kh, int kw, int stride, int dil, int elem_bytes, int deterministic) {
  if (deterministic || !dcn_bwd2_on(elem_bytes)) return 0;
  return dcn_bwd2_plan(C, Co, G, kh, kw, stride, dil, elem_bytes).ok ? 1 : 0;
}
int fami_dcn_col_dw_unpermute_f32(const float* dwp, float* dw, int Co, int C, int G, int kh, int kw, int stride, int dil,
                                  int elem_bytes, int accumulate, hipStream_t s) {
  FAMI_REQUIRE(dwp && dw, "fami_dcn_col_dw_unpermute_f32", "bad argument");
  const DcnBwd2Plan q = dcn_bwd2_plan(C, Co, G, kh, kw, stride, dil, elem_bytes);
  FAMI_REQUIRE(q.ok, "fami_dcn_col_dw_unpermute_f32", "geometry not on the register-fed backward kernel");
  hipLaunchKernelGGL(dcn_dw_unpermute_kernel, dim3(fami_ew_grid((long)Co * q.colw)), dim3(256), 0, s, dwp, dw, Co, kh * kw, q.GC,
                     q.nchunk, q.NQ, q.colw, C * kh * kw, accumulate);
  FAMI_CHECK_LAUNCH("fami_dcn_col_dw_unpermute_f32");
  return FAMI_OK;
}

// weight image of the backward column-gradient GEMM (see dcn_pack_wb_kernel)
int fami_dcn_pack_weight_bwd_f32(const float* w_oihw, float* wpb, int Co, int C, int kh, int kw, int G,
                                 hipStream_t s) {
  FAMI_REQUIRE(w_oihw && wpb && G > 0 && C % G == 0 && ((C / G) % 4) == 0, "fami_dcn_pack_weight_bwd_f32", "channels per offset group must be a multiple of 4");
  const int K = kh * kw, cg = C / G, NT = fami_cdiv((long)C * K, 16), KSo = fami_cdiv(Co, 16);
  const long total = (long)NT * KSo * 256;
  hipLaunchKernelGGL(dcn_pack_wb_kernel, dim3(fami_ew_grid(total)), dim3(256), 0, s, w_oihw, wpb, Co, C, K, cg, NT, KSo);
  FAMI_CHECK_LAUNCH("fami_dcn_pack_weight_bwd_f32");
  hipLaunchKernelGGL(dcn_wnorm_kernel, dim3(1), dim3(256), 0, s, w_oihw, wpb + total, Co, C * K);
  FAMI_CHECK_LAUNCH("fami_dcn_pack_weight_bwd_f32/norm");
  float* img = wpb + total + 4;
  for (int esz = 4; esz >= 2; esz -= 2) {
    const int keep = g_dcn_bwd2;
    g_dcn_bwd2 = 1;
    const DcnBwd2Plan q = dcn_bwd2_plan(C, Co, G, kh, kw, 1, 3, esz);
    g_dcn_bwd2 = keep;
    const long n = dcn_bwd2_image_elems(q, esz);
    if (n > 0) {
      hipLaunchKernelGGL(dcn_pack_wb2_kernel, dim3(fami_ew_grid(n)), dim3(256), 0, s, w_oihw, img, Co, C, K, q.GC, q.nchunk, q.NQ, q.NJ,
                         esz == 2 ? 1 : 0);
      FAMI_CHECK_LAUNCH("fami_dcn_pack_weight_bwd_f32/register-fed image");
      img += n;
    }
  }
  return FAMI_OK;
}

#define FAMI_ALIGN_ABI(sfx, T)                                                                                         \
  /* out[b,y,x,:] = bilinear(src[b], y - t[b,1], x - t[b,0]) ; t device fp32 [B,2] = (tx,ty) */                        \
  int fami_shift_bilinear_fwd_##sfx(const T* src, const float* t, T* out, int B, int H, int W, int C, hipStream_t s) { \
    return shift_fwd_impl<T>(src, t, out, B, H, W, C, s, "fami_shift_bilinear_fwd_" #sfx);                             \
  }                                                                                                                    \
  /* gsrc (=|+=) d/dsrc ; gt[B,2] fp32 (=|+=) d/d(tx,ty).  Either output may be null. */                               \
  int fami_shift_bilinear_bwd_##sfx(const T* gout, const T* src, const float* t, T* gsrc, float* gt, int B, int H,     \
                                    int W, int C, int acc_src, int acc_t, float* ws, hipStream_t s) {                  \
    return shift_bwd_impl<T>(gout, src, t, gsrc, gt, B, H, W, C, acc_src, acc_t, ws, s, "fami_shift_bilinear_bwd_" #sfx); \
  }                                                                                                                    \
  /* y[B,Ho,Wo,Co] = deform_conv2d(x[B,H,W,C], off[B,Ho,Wo,2GK], msk[B,Ho,Wo,GK], W, bias) */                          \
  int fami_dcn_fwd_##sfx(const T* x, const T* off, const T* msk, const float* wp, const float* bias, T* y, int B,      \
                         int H, int W, int C, int Co, int G, int kh, int kw, int stride, int pad, int dil,             \
                         hipStream_t s) {                                                                              \
    return dcn_fwd_impl<T>(x, off, msk, wp, bias, y, B, H, W, C, Co, G, kh, kw, stride, pad, dil, s, "fami_dcn_fwd_" #sfx); \
  }                                                                                                                    \
  /* Backward of fami_dcn_fwd wrt x, offsets and masks, plus the modulated-sample matrix `col` [P, C*K] (column     */ \
  /* order (channel, tap) == weight.view(Co, C*K)) for the weight gradient dW[co,k] = sum_p dy[p,co]*col[p,k].      */ \
  /* gx is an FP32 buffer ACCUMULATED with atomics (zero it first); goff/gmsk (=|+=) per acc_off; each may be null. */ \
  int fami_dcn_bwd_##sfx(const T* x, const T* off, const T* msk, const T* dy, const float* wpb, T* col, float* gx,     \
                         T* goff, T* gmsk, int B, int H, int W, int C, int Co, int G, int kh, int kw, int stride,      \
                         int pad, int dil, int acc_off, hipStream_t s) {                                               \
    return dcn_bwd_impl<T>(x, off, msk, dy, wpb, col, gx, goff, gmsk, B, H, W, C, Co, G, kh, kw, stride, pad, dil,     \
                           acc_off, s, "fami_dcn_bwd_" #sfx);                                                          \
  }                                                                                                                    \
  /* Run-to-run deterministic form of fami_dcn_bwd: the input gradient is accumulated in 64-bit fixed point (integer  */ \
  /* atomics: order independent) and written to gx (activation type, =|+= per acc_x) by a conversion pass.            */ \
  int fami_dcn_bwd_det_##sfx(const T* x, const T* off, const T* msk, const T* dy, const float* wpb, T* col, T* gx,     \
                             T* goff, T* gmsk, int B, int H, int W, int C, int Co, int G, int kh, int kw, int stride,  \
                             int pad, int dil, int acc_off, int acc_x, void* ws, hipStream_t s) {                      \
    return dcn_bwd_det_impl<T>(x, off, msk, dy, wpb, col, gx, goff, gmsk, B, H, W, C, Co, G, kh, kw, stride, pad, dil, \
                               acc_off, acc_x, ws, s, "fami_dcn_bwd_det_" #sfx);                                       \
  }                                                                                                                    \
  /* The same three entry points for offsets and masks held in ONE tensor om [B,Ho,Wo,3GK] = per pixel (2GK offsets |   */ \
  /* GK masks): the output of the two predictor convolutions of Alignment_V15.py:144-158 run as one 48 -> 324 convolution */ \
  /* (gom: its gradient, same layout).                                                                                   */ \
  int fami_dcn_fwd_om_##sfx(const T* x, const T* om, const float* wp, const float* bias, T* y, int B, int H, int W,    \
                            int C, int Co, int G, int kh, int kw, int stride, int pad, int dil, hipStream_t s) {       \
    return dcn_fwd_impl<T>(x, om, nullptr, wp, bias, y, B, H, W, C, Co, G, kh, kw, stride, pad, dil, s,                \
                           "fami_dcn_fwd_om_" #sfx, true);                                                             \
  }                                                                                                                    \
  int fami_dcn_bwd_om_##sfx(const T* x, const T* om, const T* dy, const float* wpb, T* col, float* gx, T* gom, int B,  \
                            int H, int W, int C, int Co, int G, int kh, int kw, int stride, int pad, int dil,          \
                            int acc_om, hipStream_t s) {                                                               \
    return dcn_bwd_impl<T>(x, om, nullptr, dy, wpb, col, gx, gom, nullptr, B, H, W, C, Co, G, kh, kw, stride, pad, dil,\
                           acc_om, s, "fami_dcn_bwd_om_" #sfx, nullptr, nullptr, true);                                \
  }                                                                                                                    \
  int fami_dcn_bwd_det_om_##sfx(const T* x, const T* om, const T* dy, const float* wpb, T* col, T* gx, T* gom, int B,  \
                                int H, int W, int C, int Co, int G, int kh, int kw, int stride, int pad, int dil,      \
                                int acc_om, int acc_x, void* ws, hipStream_t s) {                                      \
    return dcn_bwd_det_impl<T>(x, om, nullptr, dy, wpb, col, gx, gom, nullptr, B, H, W, C, Co, G, kh, kw, stride, pad, \
                               dil, acc_om, acc_x, ws, s, "fami_dcn_bwd_det_om_" #sfx, true);                          \
  }
FAMI_ALIGN_ABI(f32, float)
FAMI_ALIGN_ABI(bf16, bf16_t)
FAMI_ALIGN_ABI(f16, f16_t)
#undef FAMI_ALIGN_ABI

}  // extern "C"
