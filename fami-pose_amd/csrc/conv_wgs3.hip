// Weight gradient of the 3x3 stride-1 pad-1 convolutions in f32 storage on the bf16 matrix pipe ("split-product", the
// weight-gradient side of conv_t4.hip's S3 instance).
//   dW[tap][ci][co] = sum_p X[p + tap][ci] * dY[p][co]     (nn.Conv2d autograd of posetimation/layers/basic_model.py:25-63,
//                                                            the HRNet branch convolutions of backbones/hrnet.py:17-172)
// Both operands are split EXACTLY into three bf16 terms (x = x0 + x1 + x2) on the way into LDS, one plane array per term;
// the six products x0d0, x1d0, x0d1, x2d0, x1d1, x0d2 are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (the three dropped
// ones are below 2^-24 of the product; accuracy against fp64 as the exact-f32 kernels', tests/test_kernels_gpu.py).  The
// exact-f32 kernels (conv_wgrad_taps_lin_f32 and friends) run v_mfma_f32_16x16x4_f32 at a sixteenth of the matrix pipe's
// rate and were 17 % of the f32 step's kernel time.
// Structure as conv_wg16.hip: the reduction runs over PIXELS, both MFMA operands are built with transposing LDS reads
// (ds_read_b64_tr_b16), X is staged as a patch with a zero border column on either side so that a tap is a wave-uniform
// LDS offset, (input-channel tile, tap) pairs are dealt to 8 waves, one partial slab [9][Ci][Co] per workgroup.  Unlike
// it the LDS holds ONE run (three planes of an f32 patch are 1.5x its bytes): the next run is fetched into registers while
// this one is multiplied, split and stored between two barriers.
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

template <int CIT, int COT, int NYS>
__global__ __launch_bounds__(WS3_THREADS) void conv_wgrad_s3_kernel(Wgs3Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPW = (CIT * 9 + WS3_WAVES - 1) / WS3_WAVES;   // (ci tile, tap) pairs per wave
  constexpr int XPC = CIT * 4, YPC = COT * 4;                   // 16-byte f32 pieces per position / pixel
  constexpr int XPS = CIT == 2 ? 96 : CIT * 32, YPS = COT == 2 ? 96 : COT * 32;   // LDS bytes per position / pixel in one plane (an odd number of 32-byte blocks)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kq = lane >> 4;
  const int rsel = l16 >> 2, piece = l16 & 3;   // this lane hands pixel row `rsel` (of 4), channels piece*4..+3 to the reads
  int g, byl;
  xcd_tile(1, g, byl);
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

// ------------------------------------------------------------------ host side
struct Wgs3Plan { int ok, CIT, COT, NYS, BT, bpf, nsub, G, ciBlocks, coBlocks, xpl, ypl, yrows; size_t lds; };
// [fami_route_t] g_wgs3 (default 1), g_wgs3_bt (default 0), g_wgs3_target (default 0)
static Wgs3Plan wgs3_plan(int N, int H, int W, int Ci, int Co) {
  Wgs3Plan q;
  q.ok = 0;
  if (!g_wgs3 || (Ci % 16) || (Co % 16)) return q;
  q.CIT = Ci % 48 == 0 ? 3 : (Ci % 32 == 0 ? 2 : 1);
  q.COT = Co % 48 == 0 ? 3 : (Co % 32 == 0 ? 2 : 1);
  q.ciBlocks = Ci / (16 * q.CIT);
  q.coBlocks = Co / (16 * q.COT);
  const int HW = H * W, FT = (HW + 15) / 16, PW = W + 2;
  const long blocks = (long)q.ciBlocks * q.coBlocks;
  const int XS = WS3_THREADS / (4 * q.CIT), YS = WS3_THREADS / (4 * q.COT);
  // (9 tiles = 144 pixels = whole rows of every map of the path (widths 72, 36, 18, 9): aligned runs carry one halo row less)
  const int cand[10] = {16, 14, 12, 10, 9, 8, 6, 4, 3, 2};
  q.BT = 0;
  for (int i = 0; i < 10 && !q.BT; ++i) {
    int bt = g_wgs3_bt > 0 ? g_wgs3_bt : cand[i];
    if (bt > FT) bt = FT;                 // a frame smaller than the candidate is one run: size the LDS and the sweeps for it
    // output rows a run can touch: a whole frame, or runs of whole rows (aligned); any other length may straddle one row more
    const long orows = bt == FT ? H : ((bt * 16) % W == 0 ? (bt * 16) / W : (bt * 16 + W - 2) / W + 1);
    const long npos = (orows + 2) * (long)PW;
    const int yrows = (bt * 16 + 31) / 32 * 32;
    const int xps = q.CIT == 2 ? 96 : 32 * q.CIT, yps = q.COT == 2 ? 96 : 32 * q.COT;
    const size_t lds = 3 * ((size_t)npos * xps + (size_t)yrows * yps) + 2 * 48 * sizeof(float);   // + the XBN table
    const int nys = (yrows + YS - 1) / YS;      // the zero tail rows are stored too
    if (npos <= (long)WS3_NXS * XS && nys <= 7 && bt <= 16 && lds <= 158 * 1024) {
      q.BT = bt;
      q.xpl = (int)(npos * xps);
      q.yrows = yrows;
      q.ypl = yrows * yps;
      q.lds = lds;
      q.NYS = nys <= 4 ? 4 : 7;
    }
    if (g_wgs3_bt > 0) break;
  }
  if (!q.BT) return q;
  q.bpf = (FT + q.BT - 1) / q.BT;
  const long NB = (long)N * q.bpf;
  // one workgroup per CU at a time (see conv_wg16.hip), so never more than 256; 192 measured better inside the step (fewer
  // partial slabs to write and reduce, the idle CUs go to other lanes: f32 step 51.0 -> 50.65 ms in both A/B orders) although
  // 256 wins per launch
  const long target = g_wgs3_target > 0 ? g_wgs3_target : 192;
  long G = target / blocks;
  if (G > NB) G = NB;
  if (G < 1) G = 1;
  q.nsub = (int)((NB + G - 1) / G);
  q.G = (int)((NB + q.nsub - 1) / q.nsub);
  q.ok = (long)N * HW < (1L << 31) && (long)N * H * W * Ci * 4 < (1L << 31) && (long)N * H * W * Co * 4 < (1L << 31) &&
         q.G < 65536 && blocks < 65536;
  return q;
}

long fami_wgrad_s3_slabs(int N, int H, int W, int Ci, int Co) {
  const Wgs3Plan q = wgs3_plan(N, H, W, Ci, Co);
  return q.ok ? q.G : 0;
}

// -> number of partial slabs written to `part` ([G][9][Ci][Co] fp32), 0 if the shape is not eligible, < 0 on error
int fami_try_wgrad_s3(const float* x, const float* dy, float* part, long ws_bytes, int N, int H, int W, int Ci, int Co,
                      hipStream_t s, const char* name, const XBN& xbn) {
  const Wgs3Plan q = wgs3_plan(N, H, W, Ci, Co);
  if (!q.ok || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) != 0) return 0;
  if (ws_bytes < (long)q.G * Co * Ci * 9 * (long)sizeof(float)) {
    fami_set_error(name, "workspace too small");
    return FAMI_EARG;
  }
  Wgs3Args a;
  a.xb = xbn;
  a.x = x; a.dy = dy; a.part = part;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.BT = q.BT; a.bpf = q.bpf; a.nsub = q.nsub; a.NB = N * q.bpf;
  a.ciBlocks = q.ciBlocks; a.coBlocks = q.coBlocks;
  a.PW = W + 2; a.xpl = q.xpl; a.ypl = q.ypl; a.yrows = q.yrows;
  const dim3 grid(q.G, a.ciBlocks * a.coBlocks);
  bool ok = false;
#define FAMI_WS3_CASE(cit, cot, nys)                                                                                    \
  if (q.CIT == cit && q.COT == cot && q.NYS == nys) {                                                                   \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv_wgrad_s3_kernel<cit, cot, nys>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((conv_wgrad_s3_kernel<cit, cot, nys>), grid, dim3(WS3_THREADS), q.lds, s, a);                    \
    ok = true;                                                                                                            \
  }
#define FAMI_WS3_ROW(nys)                                                                                               \
  FAMI_WS3_CASE(1, 1, nys) FAMI_WS3_CASE(1, 2, nys) FAMI_WS3_CASE(1, 3, nys) FAMI_WS3_CASE(2, 1, nys)               \
  FAMI_WS3_CASE(2, 2, nys) FAMI_WS3_CASE(2, 3, nys) FAMI_WS3_CASE(3, 1, nys) FAMI_WS3_CASE(3, 2, nys) FAMI_WS3_CASE(3, 3, nys)
  FAMI_WS3_ROW(4) FAMI_WS3_ROW(7)
#undef FAMI_WS3_ROW
#undef FAMI_WS3_CASE
  if (!ok) return 0;
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    fami_set_error(name, hipGetErrorString(err));
    return FAMI_EHIP;
  }
  return q.G;
}
// benchmarks / tests: 0 / 1 off / on, 100 + bt forces the tiles per run, 1000 + n the workgroup target, < 0 defaults
// [fami_route_t] g_wgs3_default (default 1)  // fami_tune_defaults (FAMI_F32_SPLIT)
void fami_wgrad_s3_default(int on) { g_wgs3_default = on ? 1 : 0; }
void fami_wgrad_s3_tune(int on) {
  if (on < 0) { g_wgs3 = g_wgs3_default; g_wgs3_bt = 0; g_wgs3_target = 0; }
  else if (on <= 1) g_wgs3 = on;
  else if (on >= 1000) g_wgs3_target = on - 1000;
  else if (on >= 100) g_wgs3_bt = on - 100;
}
