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
#include "conv_wg6_dev.h"
#include "conv_pair.h"

struct Wg16Args {
  XBN xb;           // BatchNorm + ReLU applied to X while it is staged (the convolution's input was never materialised; xb.on)
  const void* x;    // [N,H,W,Ci]
  const void* dy;   // [N,Ho,Wo,Co]
  float* part;      // [G][taps][Ci][Co]
  int N, H, W, Ci, Co;
  int Ho, Wo;       // output map (= H, W for the stride-1 same-size convolutions)
  int st, dil, pad; // stride (1 | 2), dilation, padding (= dil for 3x3, 0 for 1x1: the kernel is centred)
  int BT;           // 16-pixel tiles per run (<= 16)
  int bpf;          // runs per frame
  int nsub;         // runs per workgroup (accumulators persist)
  int NB;           // runs in total (N * bpf)
  int ciBlocks, coBlocks;
  int PW;           // W + 2 * pad: patch row length in positions
  int xps, yps;     // LDS bytes per patch position / per dY pixel
  int xbytes;       // LDS bytes reserved for the patch
  int abl;          // benchmarks (fami_conv_tune_wgrad_lds(22000 + bits), WRONG results): 1 no K loop, 2 no slab store, 4 no staging after the first run
};

template <typename H, int CIT, int COT, int TAPS>
__global__ __launch_bounds__(WG16_THREADS) void conv_wgrad16_kernel(Wg16Args p) {
  typedef typename H16<H>::x8 hx8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KW = TAPS == 9 ? 3 : 1;
  constexpr int NPW = (CIT * TAPS + WG16_WAVES - 1) / WG16_WAVES;   // (ci tile, tap) pairs per wave
  constexpr int XPC = CIT * 2, YPC = COT * 2;                     // 16-byte pieces per position / pixel
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kq = lane >> 4;
  const int rsel = l16 >> 2, piece = l16 & 3;   // this lane hands pixel row `rsel` (of 4), channels piece*4..+3 to the reads
  int g, byl;
  xcd_tile(1, g, byl);
  const int cob = byl % p.coBlocks, cib = byl / p.coBlocks;
  const int HW = p.Ho * p.Wo;   // output pixels of a frame: the runs and the K dimension walk these

  // pairs of this wave: q = wave + 8*i -> (ci tile, tap); LDS offset of the pair relative to a pixel's own position
  int poff[NPW], ptap[NPW], pci[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int q = wave + WG16_WAVES * i;
    const bool ok = q < CIT * TAPS;
    pci[i] = ok ? q / TAPS : 0;
    ptap[i] = ok ? q - pci[i] * TAPS : -1;
    const int t = ok ? ptap[i] : 0;
    // the patch starts at input row st*y0 - pad, column -pad: tap (ky, kx) of output pixel (y, x) sits at patch row
    // st*(y - y0) + ky*dil, column st*x + kx*dil
    poff[i] = ((t / KW) * p.dil * p.PW + (t % KW) * p.dil) * p.xps + pci[i] * 32 + piece * 8;
  }
  f32x4 acc[NPW][COT];
#pragma unroll
  for (int i = 0; i < NPW; ++i)
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const char* xg = reinterpret_cast<const char*>(p.x) + (long)cib * (CIT * 32);
  const char* yg = reinterpret_cast<const char*>(p.dy) + (long)cob * (COT * 32);
  // ---- staging plan, computed ONCE per thread: the kernel was instruction-issue bound (SQ_ACTIVE_INST_ANY 45 % of the
  // wave cycles), and the per-piece index arithmetic of every run's fetch (two divisions per 16-byte piece) was half of it.
  // A thread owns piece xpc of patch positions xp0, xp0 + XS, ... and piece ypc of dY pixels yp0, yp0 + YS, ...: its patch
  // (row, column) per sweep does not depend on the run, so a run's fetch is one add and two compares per piece.
  constexpr int XS = WG16_THREADS / XPC, YS = WG16_THREADS / YPC;     // positions / pixels per sweep
  constexpr int NXS = WG16_XSWEEPS, NYS = (288 + YS - 1) / YS;   // runs of up to 18 tiles
  const int xpc = tid % XPC, xp0 = tid / XPC, ypc = tid % YPC, yp0 = tid / YPC;
  const bool xthr = xp0 < XS, ythr = yp0 < YS;                        // the last threads of the block own no piece
  int xrow[NXS], xgo[NXS];   // patch row of the sweep's position; global byte offset relative to the run's first patch row
  {
    int r = xp0 / p.PW, c = xp0 - r * p.PW;
    const int dr = XS / p.PW, dc = XS - dr * p.PW;
#pragma unroll
    for (int u = 0; u < NXS; ++u) {
      const bool colok = c >= p.pad && c < p.W + p.pad;               // the pad columns on either side are the zero border
      xrow[u] = (xthr && colok) ? r : 0x40000000;                     // never a valid row
      xgo[u] = ((r * p.W + c - p.pad) * p.Ci) * 2 + xpc * 16;
      c += dc;
      r += dr;
      if (c >= p.PW) {
        c -= p.PW;
        r += 1;
      }
    }
  }
  const int ygo = (yp0 * p.Co) * 2 + ypc * 16;
  // output-channel tail of the last block (Co % 16 != 0: 216 / 108-channel offset and mask predictors): a piece holds 8
  // channels; a piece with only 4 left is an 8-byte load, pieces past Co stay zero
  const int yc0 = cob * (COT * 16) + ypc * 8;
  const int ykind = yc0 + 8 <= p.Co ? 2 : (yc0 + 4 <= p.Co ? 1 : 0);
  u32x4 prx[NXS], pry[NYS];
  unsigned xvalid = 0;   // bit u: sweep u's piece was loaded (a zero-border / outside piece stays zero under XBN)
  // geometry of run b (wave-uniform): frame, first / last pixel in the frame, first patch row, patch rows
  auto run_geo = [&](int b, int& img, int& q0, int& q1, int& y0, int& nrow) {
    img = b / p.bpf;
    q0 = (b - img * p.bpf) * p.BT * 16;
    q1 = min(q0 + p.BT * 16, HW);
    y0 = q0 / p.Wo;
    nrow = p.st * ((q1 - 1) / p.Wo - y0) + 2 * p.pad + 1;
  };
  auto fetch = [&](int b) {
    int img, q0, q1, y0, nrow;
    run_geo(b, img, q0, q1, y0, nrow);
    const int r0 = p.st * y0 - p.pad;                                       // first patch row (may lie above the image)
    const char* xr = xg + ((long)(img * p.H + r0) * p.W) * p.Ci * 2;
#pragma unroll
    for (int u = 0; u < NXS; ++u) {
      prx[u] = u32x4{0u, 0u, 0u, 0u};
      const bool ld = xrow[u] < nrow && (unsigned)(r0 + xrow[u]) < (unsigned)p.H;
      if (ld) prx[u] = *reinterpret_cast<const u32x4*>(xr + xgo[u]);
      xvalid = (xvalid & ~(1u << u)) | ((ld ? 1u : 0u) << u);
    }
    const char* yr = yg + ((long)img * HW + q0) * p.Co * 2;
    const int M = q1 - q0;
    // ONE load kind per loop: with the 16-byte / 8-byte choice inside the loop hipcc branched around each load and put an
    // s_waitcnt vmcnt(0) behind it -- the four dY loads of a run were four DEPENDENT HBM round trips in every fetch (found in
    // the ISA in round 4; the 8 patch loads above always were in flight together).  The 8-byte form only exists for output-
    // channel tails (Co % 8 != 0: the DCN predictors), a kernel-uniform condition.
    if ((p.Co & 7) == 0) {
#pragma unroll
      for (int u = 0; u < NYS; ++u) {
        pry[u] = u32x4{0u, 0u, 0u, 0u};
        if (ythr && ykind == 2 && yp0 + u * YS < M) pry[u] = *reinterpret_cast<const u32x4*>(yr + ygo + (long)u * YS * p.Co * 2);
      }
    } else {
#pragma unroll
      for (int u = 0; u < NYS; ++u) {
        pry[u] = u32x4{0u, 0u, 0u, 0u};
        if (ythr && yp0 + u * YS < M) {
          const char* src = yr + ygo + (long)u * YS * p.Co * 2;
          if (ykind == 2) pry[u] = *reinterpret_cast<const u32x4*>(src);
          else if (ykind == 1) {
            const u32x2 h2 = *reinterpret_cast<const u32x2*>(src);
            pry[u] = u32x4{h2.x, h2.y, 0u, 0u};
          }
        }
      }
    }
  };
  // Two LDS buffers: while run b is multiplied out of one, run b+1's registers (fetched at the top of the run) are
  // written into the other between the two halves of the K loop -- the global-load latency and the LDS stores sit
  // behind MFMAs, and the workgroup meets once per run.  dY rows past the run's last pixel are stored as zeros, so the
  // partial last K step needs no select on the dY side.
  const int yrows = ((p.BT * 16 + 31) >> 5) << 5;     // dY rows of a buffer: whole K steps (the tail rows are zeros)
  const int bufsz = p.xbytes + yrows * p.yps;         // one buffer: patch + dY rows (16-byte multiple)
  // XBN: scale / shift of this workgroup's CIT*16 input channels in LDS behind the two buffers
  float* xsc = reinterpret_cast<float*>(smem + 2 * bufsz);
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
  auto stash = [&](int b, char* buf) {                 // registers of run b -> LDS
    int img, q0, q1, y0, nrow;
    run_geo(b, img, q0, q1, y0, nrow);
    const int npos = nrow * p.PW;
#pragma unroll
    for (int u = 0; u < NXS; ++u)
      if (xthr && xp0 + u * XS < npos) {
        u32x4 v = prx[u];
        if (p.xb.on && ((xvalid >> u) & 1u)) v = xbn_piece<H>(v, xsc + xpc * 8, xsf + xpc * 8);
        *reinterpret_cast<u32x4*>(buf + (xp0 + u * XS) * p.xps + xpc * 16) = v;
      }
#pragma unroll
    for (int u = 0; u < NYS; ++u)
      if (ythr && yp0 + u * YS < yrows) *reinterpret_cast<u32x4*>(buf + p.xbytes + (yp0 + u * YS) * p.yps + ypc * 16) = pry[u];
  };
  const int b0 = g * p.nsub;
  if (b0 < p.NB) {
    fetch(b0);
    stash(b0, smem);
  }
  __syncthreads();
  const int dyq = 32 / p.Wo, dxr = 32 - dyq * p.Wo;
  for (int sub = 0; sub < p.nsub; ++sub) {
    const int b = b0 + sub;
    if (b >= p.NB) break;
    int img, q0, q1, y0, nrow;
    run_geo(b, img, q0, q1, y0, nrow);
    const int M = q1 - q0;
    char* xt = smem + (sub & 1) * bufsz;
    const bool more = sub + 1 < p.nsub && b + 1 < p.NB && !(p.abl & 4);
    if (more) fetch(b + 1);   // in flight while the first half of this run is multiplied

    // this lane's two pixels of the current K step (local index pl = ks*32 + kq*8 + h*4 + rsel): image coordinates kept
    // incrementally (a K step advances a pixel by 32: dyq rows + dxr columns, one conditional carry); the dY address
    // advances by 32 rows per step
    int py[2], pxx[2], pl[2], ya[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      pl[h] = kq * 4 + h * 16 + rsel;   // (any pixel <-> K-slot map serves as long as both operands use it; this one is bank-conflict free)
      const int q = q0 + pl[h];
      py[h] = q / p.Wo;
      pxx[h] = q - py[h] * p.Wo;
      ya[h] = p.xbytes + pl[h] * p.yps + piece * 8;
    }
    const int ksteps = (p.abl & 1) ? 0 : (M + 31) >> 5;
    // one K step with NP live pairs (compile-time: the per-pair "is this slot used" branch kept every pair's LDS reads
    // behind the previous pair's MFMAs -- ds_read x2, s_waitcnt, 3 MFMAs, four times over).  All fragments of the step
    // are requested first, then multiplied.
    auto kstep = [&](auto npc) {
      constexpr int NP = decltype(npc)::value;
      int xb[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // pixels past the run meet a zero dY row; their X address only has to stay inside the patch
        const bool pin = pl[h] < M;
        xb[h] = pin ? (p.st * (py[h] - y0) * p.PW + p.st * pxx[h]) * p.xps : 0;
        pl[h] += 32;
        pxx[h] += dxr;
        py[h] += dyq;
        if (pxx[h] >= p.Wo) {
          pxx[h] -= p.Wo;
          py[h] += 1;
        }
      }
      hx8 bfr[COT], afr[NP];
#pragma unroll
      for (int c = 0; c < COT; ++c) {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xt + ya[0] + c * 32));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xt + ya[1] + c * 32));
        bfr[c] = frag_of<hx8>(lo, hi);
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xt + xb[0] + poff[i]));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xt + xb[1] + poff[i]));
        afr[i] = frag_of<hx8>(lo, hi);
      }
      ya[0] += 32 * p.yps;
      ya[1] += 32 * p.yps;
#pragma unroll
      for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int c = 0; c < COT; ++c) acc[i][c] = H16<H>::mfma(afr[i], bfr[c], acc[i][c]);
    };
    const bool full = ptap[NPW - 1] >= 0;   // wave-uniform: does this wave use its last pair slot?
    const int kh = (ksteps + 1) >> 1;
    if (full) { for (int ks = 0; ks < kh; ++ks) kstep(std::integral_constant<int, NPW>()); }
    else { for (int ks = 0; ks < kh; ++ks) kstep(std::integral_constant<int, (NPW > 1 ? NPW - 1 : 1)>()); }
    if (more) stash(b + 1, smem + ((sub + 1) & 1) * bufsz);
    if (full) { for (int ks = kh; ks < ksteps; ++ks) kstep(std::integral_constant<int, NPW>()); }
    else { for (int ks = kh; ks < ksteps; ++ks) kstep(std::integral_constant<int, (NPW > 1 ? NPW - 1 : 1)>()); }
    __syncthreads();   // this buffer is free, the other one is complete
  }

  // D row = kq*4 + r (ci), col = l16 (co)  ->  slab [g][tap][ci][co]
  float* slab = p.part + (long)g * TAPS * p.Ci * p.Co;
  if (p.abl & 2) return;
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    if (ptap[i] < 0) continue;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
      const int co = cob * (COT * 16) + c * 16 + l16;
      if (co >= p.Co) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = cib * (CIT * 16) + pci[i] * 16 + kq * 4 + r;
        slab[((long)ptap[i] * p.Ci + ci) * p.Co + co] = acc[i][c][r];
      }
    }
  }
}

static long long* g_wg6_dbg = nullptr;
extern "C" void fami_wgrad6_debug(void* buf) { g_wg6_dbg = reinterpret_cast<long long*>(buf); }
struct Wg6Plan { int ok, UR, upf, M, KS, nunits, G, XI, YI, blocks, CIT, COT, Ho, Wo, PR, XJ, coBlocks; size_t lds; };
// [fami_route_t] g_wg6_s2 (default 1)  // fami_conv_tune_wgrad_lds(23004 / 23005): stride-2 launches off / on
// [fami_route_t] g_wg6_dil (default 1)  // fami_conv_tune_wgrad_lds(23008 / 23009): the dilated (48 -> 216 / 108, dilation 3) launches off / on
// [fami_route_t] g_wg6_c42 (default 1)
// [fami_route_t] g_wg6_c4 (default 1)  // fami_conv_tune_wgrad_lds(23002 / 23003): the 64-channel blocks off / on
// [fami_route_t] g_wg6_target_c4 (default 160)  // fami_conv_tune_wgrad_lds(26000 + n): workgroup target of the launches with 64-channel input blocks (0: wg6_target); bf16 W48 step 19.47 / 19.39 / 19.43 / 19.34 ms at 80 / 120 / 160 / 240, W64 fp16 29.82 / 29.50 / 29.38 at 80 / 140 / 180
// [fami_route_t] g_wg6 (default 1), g_wg6_nu (default 0), g_wg6_target (default 80)  // fami_conv_tune_wgrad_lds(23000 / 23001): off / on; 23100 + n: units per workgroup; 23400 + n: workgroup target
static Wg6Plan wg6_plan(int N, int H, int W, int Ci, int Co, int k, int st, int pad, int dil, bool pair = false) {      // pair: the weight-gradient half of a combined launch (conv_pair.h)
  Wg6Plan q;
  q.ok = 0;
  // dilated form (the DCN offset / mask predictors of the head, Alignment_V15.py:83-101: 48 -> 216 / 108, dilation = padding = 3,
  // stride 1): 48-channel blocks with a channel tail, units of two output rows (2 + 2 dil patch rows)
  const bool dilated = dil > 1;
  if (!g_wg6 || k != 3 || !(st == 1 || (st == 2 && g_wg6_s2)) || pad != dil || (dilated && (!g_wg6_dil || st != 1 || dil > 3 || Ci % 48 != 0 || Co % 4 != 0))) return q;
  q.Ho = (H + 2 * pad - 2 * dil - 1) / st + 1;
  q.Wo = (W + 2 * pad - 2 * dil - 1) / st + 1;
  q.CIT = Ci % 48 == 0 ? 3 : (Ci % 64 == 0 ? 4 : 0);
  q.COT = (Co % 48 == 0 || dilated) ? 3 : (Co % 64 == 0 ? 4 : 0);
  q.XJ = dilated ? 8 : WG6_XJ;
  if (!q.CIT || !q.COT || (q.CIT == 3 && q.COT == 4) || (q.CIT == 4 && !g_wg6_c4)) return q;
  // 64 x 64 blocks only fit the LDS twice with two-row units (five K steps, a barrier per 144 pixels, 17 spilled registers);
  // 64 x 32 blocks take four-row units (nine K steps): fami_conv_tune_wgrad_lds(23006 / 23007) off / on
  if (q.CIT == 4 && q.COT == 4 && g_wg6_c42) q.COT = 2;       // (3 x 4 is not instantiated: no layer of the path has it)
  // rows per unit: about 288 pixels (nine K steps) of whole rows, H a multiple, two buffers in the LDS
  const int RG = (W + 2 * dil) * 2 * q.CIT;
  q.UR = 0;
  for (int ur = q.Ho; ur >= 1; --ur) {
    if (q.Ho % ur != 0 || ur * q.Wo > 288) continue;
    const int M = ur * q.Wo, KS = (M + 31) / 32, PR = st * (ur - 1) + 2 * dil + 1;
    if (!(KS == 9 || KS == 8 || KS == 7 || KS == 5 || KS == 4 || (KS == 3 && q.CIT == 4 && q.COT == 2)) || KS * 32 - M > 24) continue;      // (a quarter of the last step may be padding, not more)
    if (q.CIT == 4 && !(KS == 5 || ((KS == 9 || KS == 4 || KS == 3) && q.COT == 2))) continue;                                                              // (instantiated: 64-channel blocks with five K steps)
    const int XI = (PR * RG + 63) / 64, YI = KS * q.COT;
    if (XI > 8 * q.XJ || YI > 8 * WG6_YJ || 2 * (size_t)(XI + YI) * 1024 > 160 * 1024) continue;
    if (dilated && KS != 5) continue;                    // (instantiated: five K steps)
    q.UR = ur; q.M = M; q.KS = KS; q.XI = XI; q.YI = YI; q.PR = PR;
    break;
  }
  if (!q.UR) return q;
  q.lds = 2 * (size_t)(q.XI + q.YI) * 1024;
  q.upf = q.Ho / q.UR;
  q.coBlocks = (Co + 16 * q.COT - 1) / (16 * q.COT);
  q.blocks = (Ci / (16 * q.CIT)) * q.coBlocks;
  const long NU = (long)N * q.upf;
  // units per workgroup: about g_wg6_target workgroups in the launch (the other stream lanes use the CUs a launch leaves, and a
  // workgroup's 83 KB partial slab -- written, then read by the reduce -- is the kernel's largest HBM item).  Alone 240 is the
  // fastest (19.9 us with the reduce at 48 channels against 21.1 / 25.0 / 31.1 at 160 / 96 / 64); inside the bf16 step
  // (tools/ab_env.py) 80: 22.44 ms against 22.58 (120) and 23.08 (240); 48 and 64 equal to 80
  // ... of 64-channel blocks (round 5): HRNet-W64's launches are 1.8 x the work of W48's -- inside the W64 fp16 step 80 / 120 / 160 / 240 / 320
  // workgroups 29.77 / 29.45 / 29.42 / 29.79 / 30.55 ms
  int tgt6 = q.CIT == 4 && g_wg6_target_c4 > 0 ? g_wg6_target_c4 : g_wg6_target;
  if (pair && g_pair_wg6_target > 0) tgt6 = g_pair_wg6_target;
  long nu = g_wg6_nu > 0 ? g_wg6_nu : (NU * q.blocks + tgt6 - 1) / tgt6;
  if (nu < 1) nu = 1;
  if (nu > NU) nu = NU;
  q.nunits = (int)nu;
  q.G = (int)((NU + nu - 1) / nu);
  q.ok = (long)H * W * Ci * 2 < (1L << 31) && (long)H * W * Co * 2 < (1L << 31) && q.G < 65536 && q.blocks < 65536;
  return q;
}

// ------------------------------------------------------------------ 1x1 weight gradient of the wide layers ("wg1", round 4)
// dW[ci][co] = sum_p X[p][ci] dY[p][co] for the 1x1 convolutions of stage 1 (64 <-> 256 channels @96x72, 20 frames: operands of
// 17.7 and 70.8 MB in bf16).  conv_wgrad16_kernel excludes them by a measured rule and the scalar-operand kernel they fell back to
// takes 65-82 us per launch alone (87-123 us in the step's tail, where it is the busiest thing on the weight-gradient stream)
// against an HBM floor of 7-18 us.  A 1x1 convolution has no spatial structure: the pixels are one flat axis, a unit is 288 of
// them (nine K steps), a workgroup owns a 64 x 64 channel block and a run of consecutive units, and a unit's X slice and dY
// slice (128 bytes per pixel each) are copied by LDS DMA into one of two buffers while the previous unit is multiplied --
// conv_wgrad6_kernel's pipeline without the patch.  Wave w owns input tile w / 2 and output tiles 2 (w % 2), + 1.
struct Wg1Args {
  const void* x;    // [P][Ci]
  const void* dy;   // [P][Co]
  float* part;      // [G][Ci][Co]
  long P;
  int Ci, Co, coBlocks;
  int nunits, NU;   // units per workgroup, units in total (ceil(P / 288))
};
#define WG1_M 288
#define WG1_I 36      // DMA instructions (1 KiB) per operand and unit: 288 pixels x 128 bytes

template <typename H>
__global__ __launch_bounds__(WG16_THREADS, 1) void conv_wgrad1_kernel(Wg1Args p) {
  typedef typename H16<H>::x8 hx8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PS = 128, KS = WG1_M / 32, XB = WG1_I * 1024, BUFSZ = 2 * WG1_I * 1024, PJ = (WG1_I + WG16_WAVES - 1) / WG16_WAVES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kq = lane >> 4;
  const int rsel = l16 >> 2, piece = l16 & 3;
  int job;
  {
    const int n = gridDim.x, lin = blockIdx.x;
    const int q = n >> 3, r = n & 7, xc = lin & 7, l = lin >> 3;
    job = xc * q + (xc < r ? xc : r) + l;
  }
  const int cob = blockIdx.y % p.coBlocks, cib = blockIdx.y / p.coBlocks;
  const int u0 = job * p.nunits;
  const int nunits = min(p.nunits, p.NU - u0);
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
  // operands past 2 GiB: the buffer descriptor is rebased per unit (a unit is 288 pixels: its offsets stay small)
  // DMA plan: granule q of a unit -> (pixel, 16-byte piece of the 128-byte slice)
  int xo[PJ], yo[PJ], pixj[PJ];
#pragma unroll
  for (int j = 0; j < PJ; ++j) {
    const int q = (wave + WG16_WAVES * j) * 64 + lane;
    const int pix = q >> 3, c = q & 7;
    pixj[j] = pix;
    xo[j] = (pix * p.Ci + cib * 64 + c * 8) * 2;
    yo[j] = (pix * p.Co + cob * 64 + c * 8) * 2;
  }
  auto dma_piece = [&](int ug, unsigned buf, int k) {      // k < PJ: X, else dY
    const long p0 = (long)ug * WG1_M;
    const long left = p.P - p0;                              // pixels from the unit's first to the tensor's end
    const int j = k < PJ ? k : k - PJ;
    const int i = wave + WG16_WAVES * j;
    if (i < WG1_I) {
      if (k < PJ) {
        const wg6_i32x4 rx = wg6_rsrc(reinterpret_cast<const char*>(p.x) + p0 * p.Ci * 2, WG1_M * p.Ci * 2);
        wg6_dma16(rx, pixj[j] < left ? (unsigned)xo[j] : 0x80000000u, buf + i * 1024);
      } else {
        const wg6_i32x4 ry = wg6_rsrc(reinterpret_cast<const char*>(p.dy) + p0 * p.Co * 2, WG1_M * p.Co * 2);
        wg6_dma16(ry, pixj[j] < left ? (unsigned)yo[j] : 0x80000000u, buf + XB + i * 1024);
      }
    }
  };
  if (nunits > 0) {
#pragma unroll
    for (int k = 0; k < 2 * PJ; ++k) dma_piece(u0, lds0, k);
  }
  const int cit = wave >> 1, cot0 = (wave & 1) * 2;
  f32x4 acc[2];
  acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  // this lane's two pixels of K step 0 (local index pl = ks*32 + kq*4 + h*16 + rsel, conv_wgrad16_kernel's map)
  const int pl0 = kq * 4 + rsel;
  const int xa0 = pl0 * PS + cit * 32 + piece * 8, ya0 = XB + pl0 * PS + cot0 * 32 + piece * 8;

  for (int u = 0; u < nunits; ++u) {
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's share of unit u has landed
    __builtin_amdgcn_s_barrier();         // ... everybody's, and every wave has left unit u - 1
    asm volatile("" ::: "memory");
    const bool more = u + 1 < nunits;
    const unsigned nbuf = lds0 + ((u + 1) & 1) * BUFSZ;
    const char* xt = smem + (u & 1) * BUFSZ;
    hx8 afr[2], bfr[2][2];
    auto load = [&](int ks, int set) {
      const char* xp = xt + xa0 + ks * 32 * PS;
      const char* yp = xt + ya0 + ks * 32 * PS;
      {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xp));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xp + 16 * PS));
        afr[set] = frag_of<hx8>(lo, hi);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(yp + c * 32));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(yp + 16 * PS + c * 32));
        bfr[set][c] = frag_of<hx8>(lo, hi);
      }
    };
    load(0, 0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) load(ks + 1, (ks + 1) & 1);
      if (more) {
        dma_piece(u0 + u + 1, nbuf, ks);                   // (KS = 9 <= 2 PJ = 10)
        if (ks == KS - 1) {
#pragma unroll
          for (int k = KS; k < 2 * PJ; ++k) dma_piece(u0 + u + 1, nbuf, k);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[c] = H16<H>::mfma(afr[ks & 1], bfr[ks & 1][c], acc[c]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // D row = kq*4 + r (ci), col = l16 (co)  ->  slab [job][ci][co]
  float* slab = p.part + (long)job * p.Ci * p.Co;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int co = cob * 64 + (cot0 + c) * 16 + l16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci = cib * 64 + cit * 16 + kq * 4 + r;
      slab[(long)ci * p.Co + co] = acc[c][r];
    }
  }
}

struct Wg1Plan { int ok, nunits, G, blocks; long NU; };
// [fami_route_t] g_wg1 (default 1), g_wg1_target (default 192)  // inside the bf16 step (tools/ab_env.py): 48 / 96 / 192 workgroups 22.63 / 22.54 / 22.45 ms against 22.96 without the kernel      // fami_conv_tune_wgrad_lds(24000 / 24001): off / on; 24100 + n: workgroup target
static Wg1Plan wg1_plan(int N, int H, int W, int Ci, int Co, int k, int st, int pad, int dil) {
  Wg1Plan q;
  q.ok = 0;
  if (!g_wg1 || k != 1 || st != 1 || pad != 0 || (Ci % 64) != 0 || (Co % 64) != 0) return q;
  const long P = (long)N * H * W;
  q.blocks = (Ci / 64) * (Co / 64);
  q.NU = (P + WG1_M - 1) / WG1_M;
  // the layers this kernel is for: large pixel counts (the fuse-layer 1x1 convolutions of the low-resolution maps stay on
  // conv_wgrad16_kernel, 14 us per launch)
  if (P < 65536) return q;
  long nu = (q.NU * q.blocks + g_wg1_target - 1) / g_wg1_target;
  if (nu < 1) nu = 1;
  if (nu > q.NU) nu = q.NU;
  q.nunits = (int)nu;
  q.G = (int)((q.NU + nu - 1) / nu);
  q.ok = q.G < 65536 && q.blocks < 65536 && (long)WG1_M * Ci * 2 < (1L << 31) && (long)WG1_M * Co * 2 < (1L << 31);
  return q;
}

// ------------------------------------------------------------------ weight gradient of the stem's first convolution ("wgs", round 4)
// conv1: 3 -> 64 channels, 3x3, stride 2 (hrnet.py:573-578).  K = 27 fits no 16-channel tile, so the launch fell back to the scalar-
// operand kernel: 183 us -- the LAST weight gradient of the backward pass (it needs the last input gradient), i.e. on the step's
// critical path in front of the final reduce and Adam.  Here: dW[k = tap * 3 + c][co] = sum_p Xcol[p][k] dY[p][co] over the flat
// output-pixel axis, units of 288 pixels (whole output rows); the unit's dY rows (128 bytes per pixel, contiguous) come by LDS DMA,
// its im2col tile Xcol [288][32] (27 used) is gathered by the threads two bytes at a time (x is 13 MB: the gather hits L2) into
// registers while the previous unit is multiplied and stored to the other buffer behind it; wave w owns k tile w / 4, output tile w % 4.
struct WgsArgs {
  const void* x;    // [N,H,W,3]
  const void* dy;   // [N,Ho,Wo,64]
  float* part;      // [G][9][3][64]
  int N, H, W, Ho, Wo;
  int UR, upf;      // output rows per unit (UR * Wo = 288), units per frame
  int nunits, NU;
};
#define WGS_IT 2      // (pixel, tap row) items per thread and unit: 288 * 3 = 864 <= 2 * 512

template <typename H>
__global__ __launch_bounds__(WG16_THREADS, 1) void conv_wgrad_stem_kernel(WgsArgs p) {
  typedef typename H16<H>::x8 hx8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int M = 288, KS = M / 32, PSX = 64, PSY = 128, XB = M * PSX, YI = M * PSY / 1024, BUFSZ = XB + M * PSY;
  constexpr int YJ = (YI + WG16_WAVES - 1) / WG16_WAVES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kq = lane >> 4;
  const int rsel = l16 >> 2, piece = l16 & 3;
  int job;
  {
    const int n = gridDim.x, lin = blockIdx.x;
    const int q = n >> 3, r = n & 7, xc = lin & 7, l = lin >> 3;
    job = xc * q + (xc < r ? xc : r) + l;
  }
  const int u0 = job * p.nunits;
  const int nunits = min(p.nunits, p.NU - u0);
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
  // zero both Xcol tiles once (columns 27 .. 31 stay zero; the gather rewrites the others)
  for (int i = tid; i < 2 * (XB / 16); i += WG16_THREADS) {
    const int b = i / (XB / 16), o = i - b * (XB / 16);
    *reinterpret_cast<u32x4*>(smem + b * BUFSZ + o * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  const long xfb = (long)p.H * p.W * 3 * 2, yfb = (long)p.Ho * p.Wo * 64 * 2;
  // gather plan: item = (pixel of the unit, tap row ky); nine consecutive 16-bit values of input row 2 oy + ky - 1 from column
  // 2 ox - 1
  int ipix[WGS_IT], iky[WGS_IT];
#pragma unroll
  for (int t = 0; t < WGS_IT; ++t) {
    const int it = tid + t * WG16_THREADS;
    ipix[t] = it < M * 3 ? it / 3 : -1;
    iky[t] = it - (it / 3) * 3;
  }
  unsigned short gv[WGS_IT][9];
  auto gather = [&](int ug) {
    const int img = ug / p.upf, oy0 = (ug - img * p.upf) * p.UR;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x)) + (long)img * xfb, 0, (int)xfb, 0x00020000);
#pragma unroll
    for (int t = 0; t < WGS_IT; ++t) {
      const int pix = ipix[t] < 0 ? 0 : ipix[t];
      const int oy = oy0 + pix / p.Wo, ox = pix - (pix / p.Wo) * p.Wo;
      const int iy = 2 * oy + iky[t] - 1;
      const bool rowok = ipix[t] >= 0 && (unsigned)iy < (unsigned)p.H;
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        const int ix = 2 * ox - 1 + e / 3;
        const bool ok = rowok && (unsigned)ix < (unsigned)p.W;
        gv[t][e] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rx, ok ? (unsigned)(((iy * p.W + ix) * 3 + e % 3) * 2) : 0x80000000u, 0, 0);
      }
    }
  };
  auto scatter = [&](char* buf) {
#pragma unroll
    for (int t = 0; t < WGS_IT; ++t)
      if (ipix[t] >= 0) {
        unsigned short* d = reinterpret_cast<unsigned short*>(buf + ipix[t] * PSX + iky[t] * 18);
#pragma unroll
        for (int e = 0; e < 9; ++e) d[e] = gv[t][e];
      }
  };
  auto dma_y = [&](int ug, unsigned buf) {
    const int img = ug / p.upf, oy0 = (ug - img * p.upf) * p.UR;
    const wg6_i32x4 ry = wg6_rsrc(reinterpret_cast<const char*>(p.dy) + (long)img * yfb, (int)yfb);
#pragma unroll
    for (int j = 0; j < YJ; ++j) {
      const int i = wave + WG16_WAVES * j;
      if (i < YI) wg6_dma16(ry, (unsigned)(oy0 * p.Wo * PSY + i * 1024 + lane * 16), buf + XB + i * 1024);
    }
  };
  if (nunits > 0) {
    gather(u0);
    dma_y(u0, lds0);
    __syncthreads();                 // (the zero fill is complete before the first scatter)
    scatter(smem);
  }
  const int kt = wave >> 2, ct = wave & 3;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int pl0 = kq * 4 + rsel;
  const int xa0 = pl0 * PSX + kt * 32 + piece * 8, ya0 = XB + pl0 * PSY + ct * 32 + piece * 8;
  for (int u = 0; u < nunits; ++u) {
    __syncthreads();                 // vmcnt(0) lgkmcnt(0) + barrier: unit u's dY rows have landed, its Xcol tile is stored, unit u - 1 is done
    const bool more = u + 1 < nunits;
    const char* xt = smem + (u & 1) * BUFSZ;
    if (more) {
      gather(u0 + u + 1);
      dma_y(u0 + u + 1, lds0 + ((u + 1) & 1) * BUFSZ);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const char* xp = xt + xa0 + ks * 32 * PSX;
      const char* yp = xt + ya0 + ks * 32 * PSY;
      const hx8 a = frag_of<hx8>(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xp)),
                                 __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xp + 16 * PSX)));
      const hx8 b = frag_of<hx8>(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(yp)),
                                 __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(yp + 16 * PSY)));
      acc = H16<H>::mfma(a, b, acc);
    }
    if (more) scatter(smem + ((u + 1) & 1) * BUFSZ);
  }
  // D row = kq*4 + r (k = tap * 3 + c), col = l16 (co)  ->  slab [job][tap][c][co] = [job][k][co]
  float* slab = p.part + (long)job * 27 * 64;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = kt * 16 + kq * 4 + r;
    if (k < 27) slab[k * 64 + ct * 16 + l16] = acc[r];
  }
}

struct WgsPlan { int ok, UR, upf, nunits, G; long NU; };
// [fami_route_t] g_wgs (default 1)  // fami_conv_tune_wgrad_lds(25000 / 25001): off / on
static WgsPlan wgs_plan(int N, int H, int W, int Ci, int Co, int k, int st, int pad, int dil) {
  WgsPlan q;
  q.ok = 0;
  if (!g_wgs || k != 3 || st != 2 || pad != 1 || dil != 1 || Ci != 3 || Co != 64) return q;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if (288 % Wo != 0 || Ho % (288 / Wo) != 0) return q;
  q.UR = 288 / Wo;
  q.upf = Ho / q.UR;
  q.NU = (long)N * q.upf;
  long nu = (q.NU + 255) / 256;
  q.nunits = (int)nu;
  q.G = (int)((q.NU + nu - 1) / nu);
  q.ok = (long)H * W * 6 < (1L << 31) && (long)Ho * Wo * 128 < (1L << 31) && q.G < 65536;
  return q;
}

// ------------------------------------------------------------------ host side
// Geometry covered: k x k with k in {1, 3}, stride 1 | 2, any dilation, padding = dilation * (k - 1) / 2 (centred kernels:
// every convolution of the path), Ci % 16 == 0, Co % 4 == 0 (output-channel tails are zero-filled on the way into LDS).
struct Wg16Plan { int ok, CIT, COT, taps, BT, bpf, nsub, G, Ho, Wo, ciBlocks, coBlocks; size_t lds; int xps, yps, xbytes; };
// [fami_route_t] g_wg16_abl (default 0)
// [fami_route_t] g_wg16 (default 1), g_wg16_bt (default 0), g_wg16_target (default 0), g_wg16_general (default 1), g_wg16_bt18 (default 0)  // 18-tile aligned runs: per launch 27.6 -> 25.0 us (48 ch @96x72), inside the bf16 step 26.10 -> 26.24 / 25.99 -> 26.09 ms: off
static Wg16Plan wg16_plan(int N, int H, int W, int Ci, int Co, int k, int st, int pad, int dil) {
  Wg16Plan q;
  q.ok = 0;
  if (!g_wg16 || (Ci % 16) || (Co % 4) || !(k == 1 || k == 3) || !(st == 1 || st == 2) || pad != dil * (k - 1) / 2) return q;
  if (!g_wg16_general && !(k == 3 && st == 1 && dil == 1 && Co % 16 == 0)) return q;
  q.taps = k * k;
  q.Ho = (H + 2 * pad - dil * (k - 1) - 1) / st + 1;
  q.Wo = (W + 2 * pad - dil * (k - 1) - 1) / st + 1;
  q.CIT = Ci % 48 == 0 ? 3 : (Ci % 32 == 0 ? 2 : 1);
  q.COT = Co > 32 ? 3 : (Co > 16 ? 2 : 1);
  if (Co % 48 != 0 && Co % 32 == 0 && Co <= 64) q.COT = 2;
  q.ciBlocks = Ci / (16 * q.CIT);
  q.coBlocks = (Co + 16 * q.COT - 1) / (16 * q.COT);
  const int HW = q.Ho * q.Wo, FT = (HW + 15) / 16;
  const long blocks = (long)q.ciBlocks * q.coBlocks;
  // every channel block re-stages the pixels: the wide 1x1 convolutions of stage 1 on the full-resolution map (64 -> 256,
  // 256 -> 64 @96x72: 12 blocks x 138 240 pixels) measured 105-110 us against 80 us for the scalar-operand kernel
  if (k == 1 && (long)N * HW * blocks > 500000) return q;
  // LDS rows are unpadded (see conv_wgrad_h_kernel): the 4 pixel rows of a transposing read are 32 * CIT bytes apart
  // a row is an odd number of 32-byte bank blocks (32-channel rows padded to 48): the eight pixel rows a transposing read's
  // 32 lanes touch then hit eight distinct blocks (see the K-step's pixel map)
  q.xps = q.CIT == 2 ? 96 : 32 * q.CIT;
  q.yps = q.COT == 2 ? 96 : 32 * q.COT;
  const int PW = W + 2 * pad;
  // 18 tiles = 288 pixels = whole rows of every map of the path (widths 72, 36, 18): an aligned run carries one halo row
  // less (6 patch rows for 4 output rows of the 96x72 map against 7 for 3.6)
  const int cand[9] = {18, 16, 14, 12, 10, 8, 6, 4, 2};
  q.BT = 0;
  for (int i = 0; i < 9 && !q.BT; ++i) {
    const int bt = g_wg16_bt > 0 ? g_wg16_bt : cand[i];
    if (bt > 16 && ((bt * 16) % q.Wo != 0 || !g_wg16_bt18)) {
      if (g_wg16_bt > 0) break;
      continue;
    }
    const long orows = (bt * 16) % q.Wo == 0 ? (bt * 16) / q.Wo : (bt * 16 + q.Wo - 2) / q.Wo + 1;   // output rows a run can touch (aligned runs: exactly)
    const long npos = (st * (orows - 1) + 2 * pad + 1) * (long)PW;       // patch positions
    const size_t lds = 2 * ((size_t)npos * q.xps + (size_t)((bt * 16 + 31) / 32 * 32) * q.yps) + 2 * 48 * sizeof(float);   // two buffers + the XBN table
    if (npos <= (long)WG16_XSWEEPS * (WG16_THREADS / (2 * q.CIT)) && bt * 16 <= 288 && lds <= 150 * 1024) {
      q.BT = bt > FT ? FT : bt;
      q.xbytes = (int)(npos * q.xps);
      q.lds = lds;
    }
    if (g_wg16_bt > 0) break;
  }
  if (!q.BT) return q;
  q.bpf = (FT + q.BT - 1) / q.BT;
  const long NB = (long)N * q.bpf;
  // runs per workgroup: the kernel holds 140-230 VGPRs x 8 waves, i.e. ONE workgroup per CU at a time -- one workgroup
  // more than 256 costs a whole second round (A: 270 workgroups 41.7 us, 180 workgroups 30.5 us; tools/bench_wg16.py)
  // Round 4 (tools/ab_env.py: one graph per setting, replayed alternately on one box, +-0.1 ms): a target of 128 workgroups beats
  // 256 inside the bf16 step (24.81 vs 24.96, 25.19 vs 25.30 ms; 64: 25.79) although a launch alone is slower -- half the partial
  // slabs (the slab store + reduce are 41 % of a launch: tools/abl_wg16.py) and half the per-workgroup prologues, and the other
  // stream lanes use the CUs it leaves
  const long target = g_wg16_target > 0 ? g_wg16_target : 128;
  long G = target / blocks;
  if (G > NB) G = NB;
  if (G < 1) G = 1;
  q.nsub = (int)((NB + G - 1) / G);
  q.G = (int)((NB + q.nsub - 1) / q.nsub);
  q.ok = (long)N * HW < (1L << 31) && (long)N * H * W * Ci * 2 < (1L << 31) && q.G < 65536 && blocks < 65536;
  return q;
}

long fami_wgrad16_slabs(int N, int H, int W, int Ci, int Co, int k, int st, int pad, int dil) {
  const Wg16Plan q = wg16_plan(N, H, W, Ci, Co, k, st, pad, dil);
  const Wg6Plan q6 = wg6_plan(N, H, W, Ci, Co, k, st, pad, dil);      // (workspace sizing: whichever kernel takes the launch)
  const Wg1Plan q1 = wg1_plan(N, H, W, Ci, Co, k, st, pad, dil);
  const WgsPlan qs = wgs_plan(N, H, W, Ci, Co, k, st, pad, dil);
  long g = q.ok ? q.G : 0;
  if (qs.ok && qs.G > g) g = qs.G;
  if (q6.ok && q6.G > g) g = q6.G;
  if (q1.ok && q1.G > g) g = q1.G;
  if (g_pair_wg6_target > 0) {                                          // ... also as the half of a combined launch
    const Wg6Plan q6p = wg6_plan(N, H, W, Ci, Co, k, st, pad, dil, true);
    if (q6p.ok && q6p.G > g) g = q6p.G;
  }
  return g;
}

template <typename HT>
static int wg16_launch(const Wg16Plan& q, const void* x, const void* dy, float* part, int N, int H, int W, int Ci, int Co,
                       int st, int pad, int dil, hipStream_t s, const XBN& xbn) {
  Wg16Args a;
  a.xb = xbn;
  a.x = x; a.dy = dy; a.part = part;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.Ho = q.Ho; a.Wo = q.Wo; a.st = st; a.dil = dil; a.pad = pad;
  a.BT = q.BT; a.bpf = q.bpf; a.nsub = q.nsub; a.NB = N * q.bpf;
  a.ciBlocks = q.ciBlocks; a.coBlocks = q.coBlocks;
  a.PW = W + 2 * pad; a.xps = q.xps; a.yps = q.yps; a.xbytes = q.xbytes;
  a.abl = g_wg16_abl;
  const dim3 grid(q.G, a.ciBlocks * a.coBlocks);
  bool ok = false;
#define FAMI_WG16_CASE(cit, cot, TP)                                                                                    \
  if (q.CIT == cit && q.COT == cot && q.taps == TP) {                                                                   \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv_wgrad16_kernel<HT, cit, cot, TP>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((conv_wgrad16_kernel<HT, cit, cot, TP>), grid, dim3(WG16_THREADS), q.lds, s, a);                 \
    ok = true;                                                                                                            \
  }
#define FAMI_WG16_ROW(TP)                                                                                               \
  FAMI_WG16_CASE(1, 1, TP) FAMI_WG16_CASE(1, 2, TP) FAMI_WG16_CASE(1, 3, TP) FAMI_WG16_CASE(2, 1, TP)             \
  FAMI_WG16_CASE(2, 2, TP) FAMI_WG16_CASE(2, 3, TP) FAMI_WG16_CASE(3, 1, TP) FAMI_WG16_CASE(3, 2, TP) FAMI_WG16_CASE(3, 3, TP)
  FAMI_WG16_ROW(9) FAMI_WG16_ROW(1)
#undef FAMI_WG16_ROW
#undef FAMI_WG16_CASE
  return ok ? 1 : 0;
}

// the instance for (KS, CIT, COT, XJ) -> 1 launched, 0 none
static int wg6_dispatch(const Wg6Args& a, dim3 grid, size_t lds, int half_kind, int KS, int CIT, int COT, int XJ, hipStream_t s) {
  bool ok6 = false;
#define FAMI_WG6_CASE(ks, cit, cot, xj)                                                                                   \
  if (!ok6 && KS == ks && CIT == cit && COT == cot && XJ == xj) {                                                         \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv_wgrad6_kernel<bf16_t, ks, cit, cot, xj>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      (void)hipFuncSetAttribute((const void*)conv_wgrad6_kernel<f16_t, ks, cit, cot, xj>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    if (half_kind == 1) hipLaunchKernelGGL((conv_wgrad6_kernel<f16_t, ks, cit, cot, xj>), grid, dim3(WG16_THREADS), lds, s, a);         \
    else hipLaunchKernelGGL((conv_wgrad6_kernel<bf16_t, ks, cit, cot, xj>), grid, dim3(WG16_THREADS), lds, s, a);                       \
    ok6 = true;                                                                                                           \
  }
  FAMI_WG6_CASE(9, 3, 3, WG6_XJ) FAMI_WG6_CASE(8, 3, 3, WG6_XJ) FAMI_WG6_CASE(7, 3, 3, WG6_XJ) FAMI_WG6_CASE(5, 3, 3, WG6_XJ) FAMI_WG6_CASE(4, 3, 3, WG6_XJ)
  FAMI_WG6_CASE(5, 4, 4, WG6_XJ) FAMI_WG6_CASE(5, 4, 3, WG6_XJ) FAMI_WG6_CASE(9, 4, 2, WG6_XJ) FAMI_WG6_CASE(5, 4, 2, WG6_XJ) FAMI_WG6_CASE(4, 4, 2, WG6_XJ) FAMI_WG6_CASE(3, 4, 2, WG6_XJ)      /* (4, 4, 2: 512 channels @12x9, HRNet-W64's fourth branch) */
  FAMI_WG6_CASE(5, 3, 3, 8)
#undef FAMI_WG6_CASE
  return ok6 ? 1 : 0;
}
// conv_pair.hip: a recorded weight-gradient half as the single launch it would have been
int fami_wg6_pair_replay(const PairHalf& h, hipStream_t s) {
  if (h.kind != 16) return 0;
  Wg6Args a;
  memcpy(&a, h.args, sizeof(a));
  return wg6_dispatch(a, dim3(h.gx, h.gy), h.lds, h.half_kind, h.v[0], h.v[1], h.v[2], h.v[3], s);
}
// conv_pair.hip: would conv_wgrad6_kernel take this launch (v = KS, CIT, COT, XJ)?  (the stem / wide-1x1 kernels come first, as below)
int fami_wg6_pair_probe(int N, int H, int W, int Ci, int Co, int k, int st, int pad, int dil, int* v) {
  if (wgs_plan(N, H, W, Ci, Co, k, st, pad, dil).ok || wg1_plan(N, H, W, Ci, Co, k, st, pad, dil).ok) return 0;
  const Wg6Plan q6 = wg6_plan(N, H, W, Ci, Co, k, st, pad, dil, true);
  if (!q6.ok) return 0;
  v[0] = q6.KS; v[1] = q6.CIT; v[2] = q6.COT; v[3] = q6.XJ;
  return 16;
}
// -> number of partial slabs written to `part` ([G][k*k][Ci][Co] fp32), 0 if the shape is not eligible, < 0 on error
int fami_try_wgrad16(int half_kind, const void* x, const void* dy, float* part, long ws_bytes, int N, int H, int W, int Ci,
                     int Co, int k, int st, int pad, int dil, hipStream_t s, const char* name, const XBN& xbn) {
  if (!xbn.on && (reinterpret_cast<uintptr_t>(dy) & 15) == 0) {
    const WgsPlan qs = wgs_plan(N, H, W, Ci, Co, k, st, pad, dil);
    if (qs.ok && ws_bytes >= (long)qs.G * 27 * 64 * (long)sizeof(float)) {
      WgsArgs a;
      a.x = x; a.dy = dy; a.part = part; a.N = N; a.H = H; a.W = W; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
      a.UR = qs.UR; a.upf = qs.upf; a.nunits = qs.nunits; a.NU = (int)qs.NU;
      static bool attrs = false;
      if (!attrs) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_stem_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_stem_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attrs = true;
      }
      const size_t ldss = (size_t)2 * (288 * 64 + 288 * 128);
      if (half_kind == 1) hipLaunchKernelGGL((conv_wgrad_stem_kernel<f16_t>), dim3(qs.G), dim3(WG16_THREADS), ldss, s, a);
      else hipLaunchKernelGGL((conv_wgrad_stem_kernel<bf16_t>), dim3(qs.G), dim3(WG16_THREADS), ldss, s, a);
      hipError_t errs = hipGetLastError();
      if (errs != hipSuccess) {
        fami_set_error(name, hipGetErrorString(errs));
        return FAMI_EHIP;
      }
      return qs.G;
    }
  }
  if (!xbn.on && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0) {
    const Wg1Plan q1 = wg1_plan(N, H, W, Ci, Co, k, st, pad, dil);
    if (q1.ok && ws_bytes >= (long)q1.G * Ci * Co * (long)sizeof(float)) {
      Wg1Args a;
      a.x = x; a.dy = dy; a.part = part; a.P = (long)N * H * W; a.Ci = Ci; a.Co = Co; a.coBlocks = Co / 64;
      a.nunits = q1.nunits; a.NU = (int)q1.NU;
      static bool attr1 = false;
      if (!attr1) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad1_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad1_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr1 = true;
      }
      const dim3 grid1(q1.G, q1.blocks);
      const size_t lds1 = (size_t)4 * WG1_I * 1024;
      if (half_kind == 1) hipLaunchKernelGGL((conv_wgrad1_kernel<f16_t>), grid1, dim3(WG16_THREADS), lds1, s, a);
      else hipLaunchKernelGGL((conv_wgrad1_kernel<bf16_t>), grid1, dim3(WG16_THREADS), lds1, s, a);
      hipError_t err1 = hipGetLastError();
      if (err1 != hipSuccess) {
        fami_set_error(name, hipGetErrorString(err1));
        return FAMI_EHIP;
      }
      return q1.G;
    }
    const Wg6Plan q6 = wg6_plan(N, H, W, Ci, Co, k, st, pad, dil, fami_pair_capture() != nullptr);
    if (q6.ok && ws_bytes >= (long)q6.G * 9 * Ci * Co * (long)sizeof(float)) {
      Wg6Args a;
      a.x = x; a.dy = dy; a.part = part; a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
      a.st = st; a.Ho = q6.Ho; a.Wo = q6.Wo; a.PR = q6.PR;
      a.UR = q6.UR; a.upf = q6.upf; a.M = q6.M; a.nunits = q6.nunits; a.NU = N * q6.upf; a.coBlocks = q6.coBlocks; a.dil = dil;
      a.PW = W + 2 * dil; a.RG = (W + 2 * dil) * 2 * q6.CIT; a.q512 = 512 / a.RG; a.r512 = 512 % a.RG; a.dyq = 32 / q6.Wo; a.dxr = 32 % q6.Wo;
      a.XI = q6.XI; a.YI = q6.YI; a.dbg = g_wg6_dbg;
      const dim3 grid(q6.G, q6.blocks);
      if (PairCapture* pc = fami_pair_capture()) {      // conv_pair.h: recorded, launched by fami_conv2d_bwd_pair_*
        pair_record(pc->b, 16, half_kind, a, grid, q6.lds, q6.KS, q6.CIT, q6.COT, q6.XJ);
        pc->b.slabs = q6.G;
        return q6.G;
      }
      const bool ok6 = wg6_dispatch(a, grid, q6.lds, half_kind, q6.KS, q6.CIT, q6.COT, q6.XJ, s) != 0;
      if (ok6) {
        hipError_t err6 = hipGetLastError();
        if (err6 != hipSuccess) {
          fami_set_error(name, hipGetErrorString(err6));
          return FAMI_EHIP;
        }
        return q6.G;
      }
    }
  }
  const Wg16Plan q = wg16_plan(N, H, W, Ci, Co, k, st, pad, dil);
  if (!q.ok || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) != 0) return 0;
  if (ws_bytes < (long)q.G * Co * Ci * k * k * (long)sizeof(float)) {
    fami_set_error(name, "workspace too small");
    return FAMI_EARG;
  }
  const int rc = half_kind == 1 ? wg16_launch<f16_t>(q, x, dy, part, N, H, W, Ci, Co, st, pad, dil, s, xbn)
                                : wg16_launch<bf16_t>(q, x, dy, part, N, H, W, Ci, Co, st, pad, dil, s, xbn);
  if (!rc) return 0;
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    fami_set_error(name, hipGetErrorString(err));
    return FAMI_EHIP;
  }
  return q.G;
}
// benchmarks / tests: 0 / 1 off / on, 2 / 3: only the 3x3 stride-1 shapes / every covered geometry, 100 + bt forces the
// tiles per run, 1000 + n the workgroup target, < 0 defaults
void fami_wgrad16_tune(int on) {
  if (on < 0) { g_wg16 = 1; g_wg16_abl = 0; g_wg16_bt = 0; g_wg16_target = 0; g_wg16_general = 1; g_wg16_bt18 = 0; g_wg6 = 1; g_wg6_c4 = 1; g_wg6_c42 = 1; g_wg6_s2 = 1; g_wg6_dil = 1; g_wgs = 1; g_wg6_nu = 0; g_wg6_target = 80; g_wg6_target_c4 = 160; g_wg1 = 1; g_wg1_target = 192; g_pair_wg6_target = 64; }
  else if (on >= 7000 && on < 8000) g_pair_wg6_target = on - 7000;  // (27000 + workgroup target of the weight-gradient half of a combined launch, conv_pair.hip; 0: the targets below)
  else if (on >= 6000 && on < 7000) g_wg6_target_c4 = on - 6000;  // (26000 + workgroup target of the launches with 64-channel blocks; 0: the common target)
  else if (on == 4000 || on == 4001) g_wg1 = on - 4000;           // (fami_conv_tune_wgrad_lds(24000 / 24001): the DMA-staged wide 1x1 kernel off / on)
  else if (on >= 4100 && on < 5000) g_wg1_target = on - 4100;     // (24100 + workgroup target)
  else if (on == 5000 || on == 5001) g_wgs = on - 5000;           // (fami_conv_tune_wgrad_lds(25000 / 25001): the stem conv1 kernel off / on)
  else if (on == 3008 || on == 3009) g_wg6_dil = on - 3008;
  else if (on == 3006 || on == 3007) g_wg6_c42 = on - 3006;
  else if (on == 3004 || on == 3005) g_wg6_s2 = on - 3004;
  else if (on == 3002 || on == 3003) g_wg6_c4 = on - 3002;
  else if (on == 3000 || on == 3001) g_wg6 = on - 3000;           // (fami_conv_tune_wgrad_lds(23000 / 23001): the DMA-staged 48-channel kernel off / on)
  else if (on >= 3100 && on < 3400) g_wg6_nu = on - 3100;        // (23100 + units per workgroup)
  else if (on >= 3400 && on < 4000) g_wg6_target = on - 3400;    // (23400 + workgroup target)
  else if (on <= 1) g_wg16 = on;
  else if (on <= 3) g_wg16_general = on - 2;
  else if (on == 4 || on == 5) g_wg16_bt18 = on - 4;      // 18-tile aligned runs off / on
  else if (on >= 2000) g_wg16_abl = on - 2000;
  else if (on >= 1000) g_wg16_target = on - 1000;
  else if (on >= 100) g_wg16_bt = on - 100;
}
