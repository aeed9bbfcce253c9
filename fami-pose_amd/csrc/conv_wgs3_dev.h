// Device code of conv_wgs3.hip (the split-product f32 weight-gradient kernel), shared with conv_pair.hip.
#pragma once
#include "conv_epi.h"
#include <type_traits>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4w __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 ws3_frag(s16x4 lo, s16x4 hi) {
  const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
  const u32x4 t = {a.x, a.y, b.x, b.y};
  return __builtin_bit_cast(bf16x8, t);
}
// 4 f32 -> three bf16x4 terms at dst, dst + plane, dst + 2 * plane
__device__ __forceinline__ void ws3_split_store(char* dst, int plane, u32x4 raw) {
  const f32x4 v = __builtin_bit_cast(f32x4, raw);
  const bf16x4w h0 = __builtin_convertvector(v, bf16x4w);
  const f32x4 r1 = v - __builtin_convertvector(h0, f32x4);          // exact
  const bf16x4w h1 = __builtin_convertvector(r1, bf16x4w);
  const f32x4 r2 = r1 - __builtin_convertvector(h1, f32x4);         // exact, and representable in bf16
  const bf16x4w h2 = __builtin_convertvector(r2, bf16x4w);
  *reinterpret_cast<bf16x4w*>(dst) = h0;
  *reinterpret_cast<bf16x4w*>(dst + plane) = h1;
  *reinterpret_cast<bf16x4w*>(dst + 2 * plane) = h2;
}

#define WS3_THREADS 512
#define WS3_WAVES 8
#define WS3_NXS 8     // patch positions per thread and run

struct Wgs3Args {
  XBN xb;           // BatchNorm + ReLU applied to X while it is staged (the convolution's input was never materialised; xb.on)
  const float* x;   // [N,H,W,Ci]
  const float* dy;  // [N,H,W,Co]
  float* part;      // [G][9][Ci][Co]
  int N, H, W, Ci, Co;
  int BT;           // 16-pixel tiles per run (<= 16)
  int bpf;          // runs per frame
  int nsub;         // runs per workgroup (accumulators persist)
  int NB;           // runs in total (N * bpf)
  int ciBlocks, coBlocks;
  int PW;           // W + 2: patch row length in positions
  int xpl, ypl;     // LDS bytes of one X plane / one dY plane
  int yrows;        // dY rows of a plane: whole K steps (the tail rows are zeros)
};

// (the body is a device function of the block coordinates so that conv_pair.hip can run it beside an input-gradient body in one launch)
template <int CIT, int COT, int NYS>
__device__ __forceinline__ void conv_wgrad_s3_body(const Wgs3Args& p, const int bx, const int by, const int gx, const int gy) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPW = (CIT * 9 + WS3_WAVES - 1) / WS3_WAVES;   // (ci tile, tap) pairs per wave
  constexpr int XPC = CIT * 4, YPC = COT * 4;                   // 16-byte f32 pieces per position / pixel
  constexpr int XPS = CIT == 2 ? 96 : CIT * 32, YPS = COT == 2 ? 96 : COT * 32;   // LDS bytes per position / pixel in one plane (an odd number of 32-byte blocks)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kq = lane >> 4;
  const int rsel = l16 >> 2, piece = l16 & 3;   // this lane hands pixel row `rsel` (of 4), channels piece*4..+3 to the reads
  int g, byl;
  {   // common.h xcd_tile(1, ...) on the body's own grid: XCD-contiguous order of the (x, y) blocks
    const int n = gx * gy, lin = by * gx + bx;
    const int q = n >> 3, r = n & 7, xc = lin & 7, l = lin >> 3;
    const int t = xc * q + (xc < r ? xc : r) + l;
    g = t / gy;
    byl = t - g * gy;
  }
  const int cob = byl % p.coBlocks, cib = byl / p.coBlocks;
  const int HW = p.H * p.W;
  char* xbuf = smem;
  char* ybuf = smem + 3 * p.xpl;

  int poff[NPW], ptap[NPW], pci[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int q = wave + WS3_WAVES * i;
    const bool ok = q < CIT * 9;
    pci[i] = ok ? q / 9 : 0;
    ptap[i] = ok ? q - pci[i] * 9 : -1;
    const int t = ok ? ptap[i] : 0;
    // the patch starts at image row y0 - 1, column -1: tap (ky, kx) of pixel (y, x) sits at patch row y - y0 + ky, column x + kx
    poff[i] = ((t / 3) * p.PW + (t % 3)) * XPS + pci[i] * 32 + piece * 8;
  }
  f32x4 acc[NPW][COT];
#pragma unroll
  for (int i = 0; i < NPW; ++i)
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const char* xg = reinterpret_cast<const char*>(p.x) + (long)cib * (CIT * 64);
  const char* yg = reinterpret_cast<const char*>(p.dy) + (long)cob * (COT * 64);
  // staging plan, computed once per thread (conv_wg16.hip): piece xpc of patch positions xp0, xp0 + XS, ...; piece ypc of
  // dY pixels yp0, yp0 + YS, ...
  constexpr int XS = WS3_THREADS / XPC, YS = WS3_THREADS / YPC;
  const int xpc = tid % XPC, xp0 = tid / XPC, ypc = tid % YPC, yp0 = tid / YPC;
  const bool xthr = xp0 < XS, ythr = yp0 < YS;
  int xrow[WS3_NXS], xgo[WS3_NXS];
  {
    int r = xp0 / p.PW, c = xp0 - r * p.PW;
    const int dr = XS / p.PW, dc = XS - dr * p.PW;
#pragma unroll
    for (int u = 0; u < WS3_NXS; ++u) {
      const bool colok = c >= 1 && c <= p.W;                          // columns 0 and W + 1 are the zero border
      xrow[u] = (xthr && colok) ? r : 0x40000000;
      xgo[u] = ((r * p.W + c - 1) * p.Ci) * 4 + xpc * 16;
      c += dc;
      r += dr;
      if (c >= p.PW) {
        c -= p.PW;
        r += 1;
      }
    }
  }
  const int ygo = (yp0 * p.Co) * 4 + ypc * 16;
  u32x4 prx[WS3_NXS], pry[NYS];
  unsigned xvalid = 0;   // bit u: sweep u's piece was loaded (a zero-border / outside piece stays zero under XBN)
  auto run_geo = [&](int b, int& img, int& q0, int& q1, int& y0, int& nrow) {
    img = b / p.bpf;
    q0 = (b - img * p.bpf) * p.BT * 16;
    q1 = min(q0 + p.BT * 16, HW);
    y0 = q0 / p.W;
    nrow = (q1 - 1) / p.W - y0 + 3;
  };
  auto fetch = [&](int b) {
    int img, q0, q1, y0, nrow;
    run_geo(b, img, q0, q1, y0, nrow);
    const int r0 = y0 - 1;
    const char* xr = xg + ((long)(img * p.H + r0) * p.W) * p.Ci * 4;
#pragma unroll
    for (int u = 0; u < WS3_NXS; ++u) {
      prx[u] = u32x4{0u, 0u, 0u, 0u};
      const bool ld = xrow[u] < nrow && (unsigned)(r0 + xrow[u]) < (unsigned)p.H;
      if (ld) prx[u] = *reinterpret_cast<const u32x4*>(xr + xgo[u]);
      xvalid = (xvalid & ~(1u << u)) | ((ld ? 1u : 0u) << u);
    }
    const char* yr = yg + ((long)img * HW + q0) * p.Co * 4;
    const int M = q1 - q0;
#pragma unroll
    for (int u = 0; u < NYS; ++u) {
      pry[u] = u32x4{0u, 0u, 0u, 0u};
      if (ythr && yp0 + u * YS < M) pry[u] = *reinterpret_cast<const u32x4*>(yr + ygo + (long)u * YS * p.Co * 4);
    }
  };
  // XBN: scale / shift of this workgroup's CIT*16 input channels in LDS behind the planes
  float* xsc = reinterpret_cast<float*>(smem + 3 * (p.xpl + p.ypl));
  float* xsf = xsc + CIT * 16;
  if (p.xb.on) {
    if (tid < CIT * 16) {
      float a, b;
      xbn_channel(p.xb, cib * (CIT * 16) + tid, false, a, b);
      xsc[tid] = a;
      xsf[tid] = b;
    }
    __syncthreads();
  }
  auto stash = [&](int b) {                            // registers of run b -> the three planes
    int img, q0, q1, y0, nrow;
    run_geo(b, img, q0, q1, y0, nrow);
    const int npos = nrow * p.PW;
#pragma unroll
    for (int u = 0; u < WS3_NXS; ++u)
      if (xthr && xp0 + u * XS < npos) {
        u32x4 v = prx[u];
        if (p.xb.on && ((xvalid >> u) & 1u)) {
          f32x4 t = __builtin_bit_cast(f32x4, v);
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] = fmaxf(__builtin_fmaf(t[j], xsc[xpc * 4 + j], xsf[xpc * 4 + j]), 0.f);
          v = __builtin_bit_cast(u32x4, t);
        }
        ws3_split_store(xbuf + (xp0 + u * XS) * XPS + xpc * 8, p.xpl, v);
      }
#pragma unroll
    for (int u = 0; u < NYS; ++u)
      if (ythr && yp0 + u * YS < p.yrows) ws3_split_store(ybuf + (yp0 + u * YS) * YPS + ypc * 8, p.ypl, pry[u]);
  };
  const int b0 = g * p.nsub;
  if (b0 < p.NB) fetch(b0);
  const int dyq = 32 / p.W, dxr = 32 - dyq * p.W;
  for (int sub = 0; sub < p.nsub; ++sub) {
    const int b = b0 + sub;
    if (b >= p.NB) break;
    if (sub > 0) __syncthreads();      // the previous run has been multiplied by every wave
    stash(b);
    __syncthreads();
    int img, q0, q1, y0, nrow;
    run_geo(b, img, q0, q1, y0, nrow);
    const int M = q1 - q0;
    if (sub + 1 < p.nsub && b + 1 < p.NB) fetch(b + 1);   // in flight while this run is multiplied

    // this lane's two pixels of the current K step (local index pl = ks*32 + h*16 + kq*4 + rsel), kept incrementally.
    // The 32 lanes a transposing read serves together are two kq groups: with kq*8 + h*4 the second group's four pixel rows
    // sat 8 rows = a multiple of 256 bytes behind the first's and hit the same banks (SQ_LDS_BANK_CONFLICT 38 % of the LDS
    // cycles); rows kq*4 .. kq*4+3 of both groups are eight consecutive rows = eight distinct 32-byte bank blocks when a
    // row is an odd number of them (16 or 48 channels; 32-channel rows are padded to 48)
    int py[2], pxx[2], pl[2], ya[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      pl[h] = kq * 4 + h * 16 + rsel;   // (any pixel <-> K-slot map serves as long as both operands use it; this one is bank-conflict free)
      const int q = q0 + pl[h];
      py[h] = q / p.W;
      pxx[h] = q - py[h] * p.W;
      ya[h] = pl[h] * YPS + piece * 8;
    }
    const int ksteps = (M + 31) >> 5;
    auto kstep = [&](auto npc) {
      constexpr int NP = decltype(npc)::value;
      int xb[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool pin = pl[h] < M;      // pixels past the run meet a zero dY row; their X address only has to stay inside the patch
        xb[h] = pin ? ((py[h] - y0) * p.PW + pxx[h]) * XPS : 0;
        pl[h] += 32;
        pxx[h] += dxr;
        py[h] += dyq;
        if (pxx[h] >= p.W) {
          pxx[h] -= p.W;
          py[h] += 1;
        }
      }
      bf16x8 bfr[COT][3];
#pragma unroll
      for (int c = 0; c < COT; ++c)
#pragma unroll
        for (int pn = 0; pn < 3; ++pn) {
          const char* yb = ybuf + pn * p.ypl + c * 32;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(yb + ya[0]));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(yb + ya[1]));
          bfr[c][pn] = ws3_frag(lo, hi);
        }
      ya[0] += 32 * YPS;
      ya[1] += 32 * YPS;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        bf16x8 afr[3];
#pragma unroll
        for (int pn = 0; pn < 3; ++pn) {
          const char* xq = xbuf + pn * p.xpl + poff[i];
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xq + xb[0]));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xq + xb[1]));
          afr[pn] = ws3_frag(lo, hi);
        }
#pragma unroll
        for (int c = 0; c < COT; ++c) {
          f32x4 a = acc[i][c];
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[2], bfr[c][0], a, 0, 0, 0);   // low-order products first
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[1], bfr[c][1], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[0], bfr[c][2], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[1], bfr[c][0], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[0], bfr[c][1], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[0], bfr[c][0], a, 0, 0, 0);
          acc[i][c] = a;
        }
      }
    };
    const bool full = ptap[NPW - 1] >= 0;   // wave-uniform: does this wave use its last pair slot?
    if (full) { for (int ks = 0; ks < ksteps; ++ks) kstep(std::integral_constant<int, NPW>()); }
    else { for (int ks = 0; ks < ksteps; ++ks) kstep(std::integral_constant<int, (NPW > 1 ? NPW - 1 : 1)>()); }
  }

  // D row = kq*4 + r (ci), col = l16 (co)  ->  slab [g][tap][ci][co]
  float* slab = p.part + (long)g * 9 * p.Ci * p.Co;
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    if (ptap[i] < 0) continue;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
      const int co = cob * (COT * 16) + c * 16 + l16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = cib * (CIT * 16) + pci[i] * 16 + kq * 4 + r;
        slab[((long)ptap[i] * p.Ci + ci) * p.Co + co] = acc[i][c][r];
      }
    }
  }
}

template <int CIT, int COT, int NYS>
__global__ __launch_bounds__(WS3_THREADS) void conv_wgrad_s3_kernel(Wgs3Args p) {
  conv_wgrad_s3_body<CIT, COT, NYS>(p, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}
