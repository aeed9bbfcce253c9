// Weight gradient of the 3x3 stride-1 pad-1 convolutions in the 16-bit storage modes, round-3 form.
//   dW[tap][ci][co] = sum_p X[p + tap][ci] * dY[p][co]     (nn.Conv2d autograd of posetimation/layers/basic_model.py:25-63,
//                                                            the HRNet branch convolutions of backbones/hrnet.py:17-172)
// Same arithmetic as conv_wgrad_h_kernel (conv.hip): the reduction runs over PIXELS, both MFMA operands are built with
// gfx950's transposing LDS reads (ds_read_b64_tr_b16), (input-channel tile, tap) pairs are dealt to the waves, fp32
// accumulators, one partial slab [9][Ci][Co] per workgroup reduced afterwards.  What changed, from the graph-mode trace of
// the bf16 step (36-40 us per launch at every branch shape, 16 % of the step's kernel time):
//   * the staging of a pixel run was synchronous and four 16-byte loads deep per thread -- five to ten dependent HBM round
//     trips per run.  Now the NEXT run's X patch and dY rows are fetched into registers before the current run is
//     multiplied (one barrier pair per run, as conv_t4.hip);
//   * X is staged as a PATCH (image rows y0-1 .. y1+1, a zero border column on either side), so a tap is a wave-uniform
//     LDS offset: the per-pair bounds tests (8 VALU per pair and pixel) and the per-step division of the pixel index are
//     gone; a pixel run is any <= 256 consecutive pixels of ONE frame;
//   * 8 waves per workgroup: 27 pairs at 48 input channels are 4+4+4+3+3+3+3+3 instead of 7+7+7+6 (48 instead of 84
//     accumulator registers), twice the loads in flight per workgroup.
#include "conv_epi.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#define WG16_THREADS 512
#define WG16_WAVES 8
#define WG16_PM 10   // 16-byte staging pieces per thread and run

struct Wg16Args {
  const void* x;    // [N,H,W,Ci]
  const void* dy;   // [N,H,W,Co]
  float* part;      // [G][9][Ci][Co]
  int N, H, W, Ci, Co;
  int BT;           // 16-pixel tiles per run (<= 16)
  int bpf;          // runs per frame
  int nsub;         // runs per workgroup (accumulators persist)
  int NB;           // runs in total (N * bpf)
  int ciBlocks, coBlocks;
  int PW;           // W + 2
  int xps, yps;     // LDS bytes per patch position / per dY pixel
  int xbytes;       // LDS bytes reserved for the patch
};

template <typename H, int CIT, int COT>
__global__ __launch_bounds__(WG16_THREADS) void conv_wgrad16_kernel(Wg16Args p) {
  typedef typename H16<H>::x8 hx8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPW = (CIT * 9 + WG16_WAVES - 1) / WG16_WAVES;   // (ci tile, tap) pairs per wave
  constexpr int XPC = CIT * 2, YPC = COT * 2;                     // 16-byte pieces per position / pixel
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kq = lane >> 4;
  const int rsel = l16 >> 2, piece = l16 & 3;   // this lane hands pixel row `rsel` (of 4), channels piece*4..+3 to the reads
  int g, byl;
  xcd_tile(1, g, byl);
  const int cob = byl % p.coBlocks, cib = byl / p.coBlocks;
  char* xt = smem;                      // patch [(rows + 2) * PW][xps]
  char* yt = smem + p.xbytes;           // [BT*16][yps]
  char* zrow = yt + p.BT * 16 * p.yps;  // 32 zero bytes
  const int HW = p.H * p.W;

  // pairs of this wave: q = wave + 8*i -> (ci tile, tap); LDS offset of the pair relative to a pixel's own position
  int poff[NPW], ptap[NPW], pci[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int q = wave + WG16_WAVES * i;
    const bool ok = q < CIT * 9;
    pci[i] = ok ? q / 9 : 0;
    ptap[i] = ok ? q - pci[i] * 9 : -1;
    const int t = ok ? ptap[i] : 4;
    poff[i] = ((t / 3 - 1) * p.PW + (t % 3 - 1)) * p.xps + pci[i] * 32 + piece * 8;
  }
  f32x4 acc[NPW][COT];
#pragma unroll
  for (int i = 0; i < NPW; ++i)
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const char* xg = reinterpret_cast<const char*>(p.x) + (long)cib * (CIT * 32);
  const char* yg = reinterpret_cast<const char*>(p.dy) + (long)cob * (COT * 32);
  u32x4 pr[WG16_PM];
  // geometry of run b (wave-uniform): frame, first / last pixel in the frame, first patch row
  auto run_geo = [&](int b, int& img, int& q0, int& q1, int& y0, int& nx) {
    img = b / p.bpf;
    q0 = (b - img * p.bpf) * p.BT * 16;
    q1 = min(q0 + p.BT * 16, HW);
    y0 = q0 / p.W;
    const int y1 = (q1 - 1) / p.W;
    nx = (y1 - y0 + 3) * p.PW * XPC;   // patch pieces
  };
  auto fetch = [&](int b) {
    int img, q0, q1, y0, nx;
    run_geo(b, img, q0, q1, y0, nx);
    const int ny = (q1 - q0) * YPC;
#pragma unroll
    for (int u = 0; u < WG16_PM; ++u) {
      const int i = tid + u * WG16_THREADS;
      pr[u] = u32x4{0u, 0u, 0u, 0u};
      if (i < nx) {
        const int pos = i / XPC, pc = i - pos * XPC;
        const int r = pos / p.PW, c = pos - r * p.PW;
        const int gy = y0 - 1 + r, gx = c - 1;
        if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
          pr[u] = *reinterpret_cast<const u32x4*>(xg + ((long)(img * p.H + gy) * p.W + gx) * p.Ci * 2 + pc * 16);
      } else if (i < nx + ny) {
        const int k = i - nx;
        const int px = k / YPC, pc = k - px * YPC;
        pr[u] = *reinterpret_cast<const u32x4*>(yg + ((long)img * HW + q0 + px) * p.Co * 2 + pc * 16);
      }
    }
  };
  const int b0 = g * p.nsub;
  if (b0 < p.NB) fetch(b0);
  for (int sub = 0; sub < p.nsub; ++sub) {
    const int b = b0 + sub;
    if (b >= p.NB) break;
    int img, q0, q1, y0, nx;
    run_geo(b, img, q0, q1, y0, nx);
    const int M = q1 - q0, ny = M * YPC;
    if (sub > 0) __syncthreads();   // the previous run has been consumed
#pragma unroll
    for (int u = 0; u < WG16_PM; ++u) {
      const int i = tid + u * WG16_THREADS;
      if (i < nx) {
        const int pos = i / XPC, pc = i - pos * XPC;
        *reinterpret_cast<u32x4*>(xt + pos * p.xps + pc * 16) = pr[u];
      } else if (i < nx + ny) {
        const int k = i - nx;
        const int px = k / YPC, pc = k - px * YPC;
        *reinterpret_cast<u32x4*>(yt + px * p.yps + pc * 16) = pr[u];
      }
    }
    if (tid < 2) *reinterpret_cast<u32x4*>(zrow + tid * 16) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (sub + 1 < p.nsub && b + 1 < p.NB) fetch(b + 1);   // in flight while this run is multiplied

    // this lane's two pixels of the current K step (local index pl = ks*32 + kq*8 + h*4 + rsel): image coordinates kept
    // incrementally (a K step advances a pixel by 32: dyq rows + dxr columns, one conditional carry)
    int py[2], pxx[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = q0 + kq * 8 + h * 4 + rsel;
      py[h] = q / p.W;
      pxx[h] = q - py[h] * p.W;
    }
    const int dyq = 32 / p.W, dxr = 32 - dyq * p.W;
    const int ksteps = (M + 31) >> 5;
    for (int ks = 0; ks < ksteps; ++ks) {
      int pl[2], xb[2];
      bool pin[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        pl[h] = ks * 32 + kq * 8 + h * 4 + rsel;
        pin[h] = pl[h] < M;
        // patch position of the pixel; pixels past the run contribute through a zero dY row, so X may be anything in range
        const int yy = pin[h] ? py[h] - y0 + 1 : 1, xx = pin[h] ? pxx[h] + 1 : 1;
        xb[h] = (yy * p.PW + xx) * p.xps;
        pxx[h] += dxr;
        py[h] += dyq;
        if (pxx[h] >= p.W) {
          pxx[h] -= p.W;
          py[h] += 1;
        }
      }
      hx8 bfr[COT];
#pragma unroll
      for (int c = 0; c < COT; ++c) {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_s16x4*)(pin[0] ? yt + pl[0] * p.yps + c * 32 + piece * 8 : zrow + piece * 8));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_s16x4*)(pin[1] ? yt + pl[1] * p.yps + c * 32 + piece * 8 : zrow + piece * 8));
        s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bfr[c] = __builtin_bit_cast(hx8, t);
      }
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        if (ptap[i] < 0) continue;   // wave-uniform
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xt + xb[0] + poff[i]));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xt + xb[1] + poff[i]));
        s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const hx8 afr = __builtin_bit_cast(hx8, t);
#pragma unroll
        for (int c = 0; c < COT; ++c) acc[i][c] = H16<H>::mfma(afr, bfr[c], acc[i][c]);
      }
    }
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
struct Wg16Plan { int ok, CIT, COT, BT, bpf, nsub, G; size_t lds; int xps, yps, xbytes; };
static int g_wg16 = 1, g_wg16_bt = 0, g_wg16_target = 0;
static Wg16Plan wg16_plan(int N, int H, int W, int Ci, int Co) {
  Wg16Plan q;
  q.ok = 0;
  if (!g_wg16 || (Ci % 16) || (Co % 16)) return q;
  q.CIT = Ci % 48 == 0 ? 3 : (Ci % 32 == 0 ? 2 : 1);
  q.COT = Co % 48 == 0 ? 3 : (Co % 32 == 0 ? 2 : 1);
  const int HW = H * W, FT = (HW + 15) / 16;
  const long blocks = (long)(Ci / (16 * q.CIT)) * (Co / (16 * q.COT));
  // LDS rows are unpadded (see conv_wgrad_h_kernel): the 4 pixel rows of a transposing read are 32 * CIT bytes apart
  q.xps = 32 * q.CIT;
  q.yps = 32 * q.COT;
  const int cand[6] = {16, 14, 12, 10, 8, 4};
  q.BT = 0;
  for (int i = 0; i < 6 && !q.BT; ++i) {
    const int bt = g_wg16_bt > 0 ? g_wg16_bt : cand[i];
    const long npos = (long)((bt * 16 + W - 2) / W + 3) * (W + 2);
    const long pieces = npos * 2 * q.CIT + (long)bt * 16 * 2 * q.COT;
    const size_t lds = (size_t)npos * q.xps + (size_t)bt * 16 * q.yps + 32;
    if (pieces <= (long)WG16_PM * WG16_THREADS && lds <= 76 * 1024) {
      q.BT = bt > FT ? FT : bt;
      q.xbytes = (int)(npos * q.xps);
      q.lds = lds;
    }
    if (g_wg16_bt > 0) break;
  }
  if (!q.BT) return q;
  q.bpf = (FT + q.BT - 1) / q.BT;
  const long NB = (long)N * q.bpf;
  // runs per workgroup: the kernel holds 140-190 VGPRs x 8 waves, i.e. ONE workgroup per CU at a time -- one workgroup
  // more than 256 costs a whole second round (A: 270 workgroups 41.7 us, 180 workgroups 30.5 us; tools/bench_wg16.py)
  const long target = g_wg16_target > 0 ? g_wg16_target : 256;
  long G = target / blocks;
  if (G > NB) G = NB;
  if (G < 1) G = 1;
  q.nsub = (int)((NB + G - 1) / G);
  q.G = (int)((NB + q.nsub - 1) / q.nsub);
  q.ok = (long)N * HW < (1L << 31) && q.G < 65536;
  return q;
}

long fami_wgrad16_slabs(int N, int H, int W, int Ci, int Co) {
  const Wg16Plan q = wg16_plan(N, H, W, Ci, Co);
  return q.ok ? q.G : 0;
}

template <typename HT>
static int wg16_launch(const Wg16Plan& q, const void* x, const void* dy, float* part, int N, int H, int W, int Ci, int Co,
                       hipStream_t s) {
  Wg16Args a;
  a.x = x; a.dy = dy; a.part = part;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.BT = q.BT; a.bpf = q.bpf; a.nsub = q.nsub; a.NB = N * q.bpf;
  a.ciBlocks = Ci / (16 * q.CIT); a.coBlocks = Co / (16 * q.COT);
  a.PW = W + 2; a.xps = q.xps; a.yps = q.yps; a.xbytes = q.xbytes;
  const dim3 grid(q.G, a.ciBlocks * a.coBlocks);
  bool ok = false;
#define FAMI_WG16_CASE(cit, cot)                                                                                          \
  if (q.CIT == cit && q.COT == cot) {                                                                                     \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv_wgrad16_kernel<HT, cit, cot>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((conv_wgrad16_kernel<HT, cit, cot>), grid, dim3(WG16_THREADS), q.lds, s, a);                       \
    ok = true;                                                                                                            \
  }
  FAMI_WG16_CASE(1, 1) FAMI_WG16_CASE(1, 2) FAMI_WG16_CASE(1, 3) FAMI_WG16_CASE(2, 1) FAMI_WG16_CASE(2, 2) FAMI_WG16_CASE(2, 3)
  FAMI_WG16_CASE(3, 1) FAMI_WG16_CASE(3, 2) FAMI_WG16_CASE(3, 3)
#undef FAMI_WG16_CASE
  return ok ? 1 : 0;
}

// -> number of partial slabs written to `part` ([G][9][Ci][Co] fp32), 0 if the shape is not eligible, < 0 on error
int fami_try_wgrad16(int half_kind, const void* x, const void* dy, float* part, long ws_bytes, int N, int H, int W, int Ci,
                     int Co, hipStream_t s, const char* name) {
  const Wg16Plan q = wg16_plan(N, H, W, Ci, Co);
  if (!q.ok || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) != 0) return 0;
  if (ws_bytes < (long)q.G * Co * Ci * 9 * (long)sizeof(float)) {
    fami_set_error(name, "workspace too small");
    return FAMI_EARG;
  }
  const int rc = half_kind == 1 ? wg16_launch<f16_t>(q, x, dy, part, N, H, W, Ci, Co, s)
                                : wg16_launch<bf16_t>(q, x, dy, part, N, H, W, Ci, Co, s);
  if (!rc) return 0;
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    fami_set_error(name, hipGetErrorString(err));
    return FAMI_EHIP;
  }
  return q.G;
}
// benchmarks / tests: 0 / 1 off / on, 100 + bt forces the tiles per run, 1000 + n the workgroup target, < 0 defaults
void fami_wgrad16_tune(int on) {
  if (on < 0) { g_wg16 = 1; g_wg16_bt = 0; g_wg16_target = 0; }
  else if (on <= 1) g_wg16 = on;
  else if (on >= 1000) g_wg16_target = on - 1000;
  else if (on >= 100) g_wg16_bt = on - 100;
}
