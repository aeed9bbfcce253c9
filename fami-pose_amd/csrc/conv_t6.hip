// Weight-resident, DMA-staged 3x3 convolution (stride 1, pad 1) for the 16-bit storage types, round 4: forward and input
// gradient of the HRNet high-resolution branch (48 channels @96x72: posetimation/backbones/hrnet.py:17-172 via
// layers/basic_model.py:25-63), the most numerous launch of the bf16 step (133 per step) and the slowest of its four
// branch shapes on the band kernel (conv_t4.hip: 17.3 us alone, 13.1 us chip-filled, against 13.0 us at 96 / 192 channels).
// What the band kernel pays for at 48 channels: a 32-channel K chunk pads K = 48 to 64 (a quarter of the MFMAs multiply
// zeros), twelve 16-pixel tiles on eight waves (four waves carry two, four carry one), the 27 KB weight slab of a chunk
// re-staged for every 192 pixels, and everything a workgroup needs fetched through registers before its first MFMA.
// Here
//   * K is DENSE: the reduction index runs over (tap, 8-channel granule) = 54 granules -> 13.5 chunks of 32 (14 MFMAs
//     per tile and channel tile instead of 18); a lane quarter (kq) of one MFMA may sit on a different tap than its
//     neighbour -- the per-lane LDS offset koff[k] carries tap shift and channel granule together;
//   * the whole weight image of the job's 48 output channels (42 KiB) is LDS-RESIDENT, gathered by DMA
//     (buffer_load_dwordx4 ... lds, per-lane source offsets into the packed fragment image) in the dense order;
//   * a workgroup owns a band of RB whole rows of one frame; its patch (RB + 2 rows, zero border columns, 96-byte
//     positions, unpadded: conflict-free for ds_read_b128, see below) is ONE contiguous LDS region filled by DMA too --
//     border and out-of-image granules carry an out-of-range buffer offset (the buffer unit writes zeros), so there is no
//     store phase, no VGPR staging and no zeroing;
//   * the band is multiplied in UNITS of four rows (288 pixels = 18 tiles at W = 72; two rows where the band is not a multiple of
//     four): the weight slab and the rows of unit 0 are requested at the top of the kernel, the rows of unit u + 1 behind the
//     barrier that opens unit u; a unit opens with s_waitcnt vmcnt(0) on the wave's own queue + one barrier;
//   * wave w owns tiles w and w + 8 (all three channel tiles); the remaining two tiles' six (tile, channel tile) pairs go to
//     waves 0-5: per SIMD (waves s, s + 4) 14 / 14 / 13 / 13 pairs;
//   * a unit's results are written behind the NEXT unit's barrier (no store sits in front of a vmcnt wait); the fragments of
//     K chunk k + 1 are requested before chunk k is multiplied (scheduler fenced with sched_barrier: left alone it requests
//     a fragment one to two MFMAs ahead of its use).
// Measured (tools/bench_t6.py, tools/trace_t6.py with -DFAMI_T6_TRACE): 48 -> 48 @96x72, 20 frames 18.0 -> 11.2 us; of a
// workgroup's 17.5 k cycles 6.4 k pass before the first MFMA (kernel arguments, 96 KB through the CU's 64 B / clk vector-memory
// path, the second wave of each SIMD 1.1 k cycles behind the first), the two units take 4.1 k each for 3.1 k of MFMA on the
// busiest SIMD.  Earlier forms: two-row units with every DMA up front and counted vmcnt waits 14.5 us (LDS-read bound: three
// weight fragments per pixel fragment), global_load_lds with a zero constant for the border instead of buffer loads 13.3 us
// (5.7 k cycles of address arithmetic and divergent branches in front of the first wait).
// LDS bank check, 96-byte positions (6 granules of 16 bytes), ds_read_b128 served in lane groups {0-3, 12-15, 20-27}, ...:
// the eight lanes of one kq in a group read granules (pos0 + col) * 6 + c -> bank quads {0, 6, 12, 2, 8, 14, 4, 10} + const,
// the eight lanes of the neighbouring kq read granule c + 1 (or granule 0 of the next tap: 6 is even) -> the odd quads:
// sixteen distinct bank quads per group.  (The one tile of a unit that straddles the two rows is 2-way conflicted on two quads.)
#include "conv_t6_dev.h"
#include "conv_pair.h"

// ---- plan + launch
// [fami_route_t] g_use_t6 (default 1)  // fami_conv_tune_lds(8000 / 8001): off / on
// [fami_route_t] g_t6_rows (default 0)  // fami_conv_tune_lds(8100 + RB): force the rows per band (benchmarks)
// [fami_route_t] g_t6_min_jobs (default 96)  // fami_conv_tune_lds(8400 + n): only launches of >= n jobs
// [fami_route_t] g_t6_mt (default 0)  // fami_conv_tune_lds(8201 / 8202): units of two / four rows (0: four where the band allows)
static long long* g_t6_dbg = nullptr;
extern "C" void fami_conv_t6_debug(void* buf) { g_t6_dbg = reinterpret_cast<long long*>(buf); }

struct T6Plan { int ok, RB, bands, pj, TU, MT; size_t lds; };
static T6Plan t6_plan(int N, int H, int W, int Ci, int Co) {
  T6Plan q;
  q.ok = 0;
  constexpr int G = 6, NT = 3, NK = (9 * G + 3) / 4, WJ = (NK * NT + 7) / 8;
  if (!g_use_t6 || Ci != 8 * G || Co % (16 * NT) != 0) return q;
  if ((2 * W) % 16 != 0 || (H & 1)) return q;
  const int TU2 = 2 * W / 16;                      // tiles of two rows
  if (TU2 < 8 || TU2 > 9) return q;
  const int RG = (W + 2) * G;
  const size_t wbytes = (size_t)WJ * 8 * 1024, lds_cap = 160 * 1024;
  double best = 1e30;
  q.RB = 0;
  for (int RB = 2; RB <= H && RB <= 16; RB += 2) {
    if (g_t6_rows > 0 && RB != g_t6_rows) continue;
    if (H % RB != 0) continue;
    const int instr = ((RB + 2) * RG + 63) / 64, pj = (instr + 7) / 8;
    if (wbytes + (size_t)pj * 8192 + 8 * (NT + 1) * 32 * 4 + 1024 > lds_cap) break;
    const long jobs = (long)N * (H / RB) * (Co / (16 * NT));
    const double cost = (double)((jobs + 255) / 256) * (RB / 2 + 3.0);     // rounds x (rows + prologue)
    if (cost < best - 1e-9) {
      best = cost;
      q.RB = RB;
      q.pj = pj;
    }
  }
  if (!q.RB) return q;
  q.bands = H / q.RB;
  if (g_t6_rows == 0 && (long)N * q.bands * (Co / (16 * NT)) < g_t6_min_jobs) return q;
  q.MT = (q.RB % 4 == 0 && g_t6_mt != 1) ? 2 : 1;
  q.TU = TU2 * q.MT;
  q.lds = wbytes + (size_t)q.pj * 8192;
  const size_t red = (size_t)8 * (NT + 1) * 32 * 4;
  if ((size_t)q.pj * 8192 < red) q.lds = wbytes + red;
  q.lds += 1024;                                   // EpiBN mode 2's channel table (behind pj * 8 KiB of patch)
  q.ok = 1;
  return q;
}

// the instance for (MT, EX, ACC, EM) -> 1 launched, 0 none
template <typename HT>
static int t6_dispatch(const ConvT6Args& a, dim3 grid, size_t lds, int MT, int ex_, bool acc_, hipStream_t s, bool xb = false) {
  bool ok = false;
  if (xb) {      // the input-BatchNorm instances (forward: never accumulating, no backward-statistics epilogue)
#define FAMI_T6X_CASE(mt, ex, em)                                                                                         \
  if (!ok && MT == mt && ex_ == ex && !acc_ && a.emode == em) {                                                           \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv3x3_t6_kernel<HT, 6, 3, mt, ex, false, em, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((conv3x3_t6_kernel<HT, 6, 3, mt, ex, false, em, true>), grid, dim3(T6_THREADS), lds, s, a);        \
    ok = true;                                                                                                            \
  }
    FAMI_T6X_CASE(1, 1, 0) FAMI_T6X_CASE(1, 1, 1) FAMI_T6X_CASE(1, 0, 0) FAMI_T6X_CASE(1, 0, 1)
    FAMI_T6X_CASE(2, 1, 0) FAMI_T6X_CASE(2, 1, 1) FAMI_T6X_CASE(2, 0, 0) FAMI_T6X_CASE(2, 0, 1)
#undef FAMI_T6X_CASE
    return ok ? 1 : 0;
  }
#define FAMI_T6_CASE(mt, ex, ac, em)                                                                                      \
  if (!ok && MT == mt && ex_ == ex && acc_ == ac && a.emode == em) {                                                   \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv3x3_t6_kernel<HT, 6, 3, mt, ex, ac, em>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((conv3x3_t6_kernel<HT, 6, 3, mt, ex, ac, em>), grid, dim3(T6_THREADS), lds, s, a);                 \
    ok = true;                                                                                                            \
  }
  FAMI_T6_CASE(1, 1, false, 0) FAMI_T6_CASE(1, 1, false, 1) FAMI_T6_CASE(1, 1, true, 0) FAMI_T6_CASE(1, 1, false, 2) FAMI_T6_CASE(1, 1, true, 2)
  FAMI_T6_CASE(1, 0, false, 0) FAMI_T6_CASE(1, 0, false, 1) FAMI_T6_CASE(1, 0, true, 0) FAMI_T6_CASE(1, 0, false, 2) FAMI_T6_CASE(1, 0, true, 2)
  FAMI_T6_CASE(2, 1, false, 0) FAMI_T6_CASE(2, 1, false, 1) FAMI_T6_CASE(2, 1, true, 0) FAMI_T6_CASE(2, 1, false, 2) FAMI_T6_CASE(2, 1, true, 2)
  FAMI_T6_CASE(2, 0, false, 0) FAMI_T6_CASE(2, 0, false, 1) FAMI_T6_CASE(2, 0, true, 0) FAMI_T6_CASE(2, 0, false, 2) FAMI_T6_CASE(2, 0, true, 2)
#undef FAMI_T6_CASE
  return ok ? 1 : 0;
}

template <typename HT>
static int t6_launch(const T6Plan& q, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci,
                     int Co, int KC, int NTt, int sgn, int relu, int accumulate, int out_f32, hipStream_t s, const EpiBN& epi,
                     const XBN& xbn = xbn_none()) {
  ConvT6Args a;
  a.xb = xbn; a.xout = xbn.out;
  a.e = epi; a.emode = epi.slots ? epi.mode : 0;
  a.x = x; a.wimg = wp; a.y = y; a.bias = bias;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.KC = KC; a.NTt = NTt;
  a.sgn = sgn; a.relu = relu; a.accumulate = accumulate; a.out_f32 = out_f32;
  a.RB = q.RB; a.bands = q.bands; a.PW = W + 2; a.RG = (W + 2) * 6; a.REMP = (q.TU - 8 * q.MT) * 3; a.pj = q.pj; a.dbg = g_t6_dbg;
  a.q512 = 512 / a.RG; a.r512 = 512 % a.RG;
  const dim3 grid(N * q.bands, Co / 48);
  const bool acc_ = accumulate != 0;
  const int ex_ = q.TU > 8 * q.MT ? 1 : 0;
  if (PairCapture* pc = fami_pair_capture()) {      // conv_pair.h: recorded, launched by fami_conv2d_bwd_pair_*
    pair_record(pc->a, 6, std::is_same<HT, f16_t>::value ? 1 : 0, a, grid, q.lds, 6, 3, q.MT, ex_, acc_ ? 1 : 0, a.emode);
    return 1;
  }
  return t6_dispatch<HT>(a, grid, q.lds + (xbn.on ? 512 : 0), q.MT, ex_, acc_, s, xbn.on != 0);      // (+ the input BatchNorm's scale / shift table)
}

struct T7Plan { int ok, RB, bands, TU, MT, EX, PI, jpw, G, SG, NT; size_t lds; };      // SG: granules of a phase's slice (6 | 4); NT: channel tiles per workgroup (3 | 4)
// [fami_route_t] g_t7_c64 (default 1)  // fami_conv_tune_lds(8502 / 8503): the 32-channel-phase instances (64-multiple layers: HRNet-W64, stage 1's 64 -> 64) off / on
// [fami_route_t] g_t7_target (default 240)  // fami_conv_tune_lds(8700 + n): workgroups of a launch (jobs are dealt consecutively).  Round 4: 120; round 5 (statistics in an LDS table, no spills): bf16 step 19.90 / 19.87 / 19.80 / 19.80 / 19.79 ms at 120 / 160 / 200 / 240 / 290, W64 fp16 29.52 -> 29.20 at 240
// [fami_route_t] g_use_t7 (default 1)  // fami_conv_tune_lds(8500 / 8501): off / on
// [fami_route_t] g_t7_rows (default 0)  // fami_conv_tune_lds(8600 + RB): force the rows per band (benchmarks)
static T7Plan t7_plan(int N, int H, int W, int Ci, int Co) {
  T7Plan q;
  q.ok = 0;
  if (!g_use_t7) return q;
  if (Ci >= 96 && Ci % 48 == 0 && Co % 48 == 0) { q.SG = 6; q.NT = 3; }
  else if (g_t7_c64 && Ci >= 64 && Ci % 64 == 0 && Co % 64 == 0) { q.SG = 4; q.NT = 4; }
  else return q;
  const int CB = 16 * q.NT;                           // output channels of a workgroup
  const int WIK = ((9 * q.SG + 3) / 4) * q.NT;        // KiB of a phase's weight slab
  const int RG = (W + 2) * q.SG;
  double best = 1e30;
  for (int RB = 1; RB <= H; ++RB) {
    if (H % RB != 0) continue;
    if (g_t7_rows > 0 && RB != g_t7_rows) continue;
    const int npix = RB * W, TU = (npix + 15) / 16;
    int MT, EX;
    if (TU <= 8) { MT = 1; EX = 0; }
    else if (TU <= 10) { MT = 1; EX = 1; }
    else if (TU >= 16 && TU <= 18) { MT = 2; EX = TU > 16 ? 1 : 0; }
    else continue;
    const int PI = ((RB + 2) * RG + 63) / 64;
    if (PI > 8 * T7_PJ) continue;
    const size_t lds = 2 * (size_t)(WIK + PI) * 1024 + 1024 + 512;     // (+ EpiBN mode 2's channel table, the workgroup's statistics table)
    if (lds > 160 * 1024) continue;
    if (EX && (TU - 8 * MT) * q.NT > 8) continue;     // (the tiles past the 8 MT-th are dealt one (tile, channel tile) pair per wave)
    const long jobs = (long)N * (H / RB) * (Co / CB);
    if (jobs > 512 && g_t7_rows == 0) continue;       // (more than one round of workgroups: the band kernel's two workgroups per CU win, e.g. 96 -> 48 @64x64 19.2 vs 14.3 us)
    // rounds x (MFMA tiles of the busiest SIMD + a fixed cost per phase-set)
    const int per_simd = TU <= 8 ? (TU > 4 ? 2 : 1) * q.NT : (MT * 2 * q.NT + (EX ? 1 : 0));
    const double c2 = (double)((jobs + 255) / 256) * (per_simd + 2.0);      // ~ time of the launch
    if (c2 < best - 1e-9) {
      best = c2;
      q.RB = RB; q.TU = TU; q.MT = MT; q.EX = EX; q.PI = PI; q.lds = lds;
      q.ok = 1;
    }
  }
  if (!q.ok) return q;
  q.bands = H / q.RB;
  {
    const long njobs = (long)N * q.bands;
    long tgt = g_t7_target / (Co / CB);
    if (tgt < 1) tgt = 1;
    q.jpw = (int)((njobs + tgt - 1) / tgt);
    q.G = (int)((njobs + q.jpw - 1) / q.jpw);
  }
  return q;
}
template <typename HT>
static int t7_dispatch(const ConvT7Args& a, dim3 grid, size_t lds, int SG, int MT, int EX, bool acc_, hipStream_t s) {
  bool ok = false;
#define FAMI_T7_CASE_G(sg, nt, mt, ex, ac, em)                                                                            \
  if (!ok && SG == sg && MT == mt && EX == ex && acc_ == ac && a.emode == em) {                                        \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv3x3_t7_kernel<HT, sg, nt, mt, ex, ac, em>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((conv3x3_t7_kernel<HT, sg, nt, mt, ex, ac, em>), grid, dim3(T6_THREADS), lds, s, a);               \
    ok = true;                                                                                                            \
  }
#define FAMI_T7_CASE(mt, ex, ac, em) FAMI_T7_CASE_G(6, 3, mt, ex, ac, em) FAMI_T7_CASE_G(4, 4, mt, ex, ac, em)
  FAMI_T7_CASE(1, 1, false, 0) FAMI_T7_CASE(1, 1, false, 1) FAMI_T7_CASE(1, 1, true, 0) FAMI_T7_CASE(1, 1, false, 2) FAMI_T7_CASE(1, 1, true, 2)
  FAMI_T7_CASE(1, 0, false, 0) FAMI_T7_CASE(1, 0, false, 1) FAMI_T7_CASE(1, 0, true, 0) FAMI_T7_CASE(1, 0, false, 2) FAMI_T7_CASE(1, 0, true, 2)
  FAMI_T7_CASE(2, 1, false, 0) FAMI_T7_CASE(2, 1, false, 1) FAMI_T7_CASE(2, 1, true, 0) FAMI_T7_CASE(2, 1, false, 2) FAMI_T7_CASE(2, 1, true, 2)
  FAMI_T7_CASE(2, 0, false, 0) FAMI_T7_CASE(2, 0, false, 1) FAMI_T7_CASE(2, 0, true, 0) FAMI_T7_CASE(2, 0, false, 2) FAMI_T7_CASE(2, 0, true, 2)
#undef FAMI_T7_CASE
#undef FAMI_T7_CASE_G
  return ok ? 1 : 0;
}
template <typename HT>
static int t7_launch(const T7Plan& q, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci,
                     int Co, int KC, int NTt, int sgn, int accumulate, hipStream_t s, const EpiBN& epi) {
  ConvT7Args a;
  a.e = epi; a.emode = epi.slots ? epi.mode : 0;
  a.x = x; a.wimg = wp; a.y = y; a.bias = bias;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.KC = KC; a.NTt = NTt; a.sgn = sgn; a.accumulate = accumulate;
  a.RB = q.RB; a.bands = q.bands; a.PW = W + 2; a.RG = (W + 2) * q.SG; a.q512 = 512 / a.RG; a.r512 = 512 % a.RG;
  a.nph = Ci / (8 * q.SG); a.TU = q.TU; a.npix = q.RB * W; a.REMP = q.TU > 8 * q.MT ? (q.TU - 8 * q.MT) * q.NT : 0; a.PI = q.PI; a.jpw = q.jpw;
  const dim3 grid(q.G, Co / (16 * q.NT));
  const bool acc_ = accumulate != 0;
  if (PairCapture* pc = fami_pair_capture()) {      // conv_pair.h
    pair_record(pc->a, 7, std::is_same<HT, f16_t>::value ? 1 : 0, a, grid, q.lds, q.SG, q.NT, q.MT, q.EX, acc_ ? 1 : 0, a.emode);
    return 1;
  }
  return t7_dispatch<HT>(a, grid, q.lds, q.SG, q.MT, q.EX, acc_, s);
}

// conv_pair.hip: a recorded input-gradient half as the single launch it would have been -> 1 launched, 0 no instance
int fami_t6_pair_replay(const PairHalf& h, hipStream_t s) {
  const dim3 grid(h.gx, h.gy);
  if (h.kind == 6) {
    ConvT6Args a;
    memcpy(&a, h.args, sizeof(a));
    return h.half_kind == 1 ? t6_dispatch<f16_t>(a, grid, h.lds, h.v[2], h.v[3], h.v[4] != 0, s) : t6_dispatch<bf16_t>(a, grid, h.lds, h.v[2], h.v[3], h.v[4] != 0, s);
  }
  if (h.kind == 7) {
    ConvT7Args a;
    memcpy(&a, h.args, sizeof(a));
    return h.half_kind == 1 ? t7_dispatch<f16_t>(a, grid, h.lds, h.v[0], h.v[2], h.v[3], h.v[4] != 0, s)
                            : t7_dispatch<bf16_t>(a, grid, h.lds, h.v[0], h.v[2], h.v[3], h.v[4] != 0, s);
  }
  return 0;
}

// Returns 1 if launched, 0 if the shape is not eligible, < 0 on error.  half_kind: 0 bf16, 1 fp16.
int fami_try_conv3x3_t6(int half_kind, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci,
                        int Co, int KC, int NTt, int sgn, int relu, int accumulate, int out_f32, hipStream_t s,
                        const char* name, const EpiBN& epi, const XBN& xbn) {
  if (half_kind > 1 || (epi.slots && epi.mode != 1 && epi.mode != 2)) return 0;
  // an input BatchNorm (XBN) only in the materialising form of the 48-channel kernel: forward, fresh output, no backward-statistics epilogue
  if (xbn.on && (!xbn.out || sgn < 0 || accumulate || (epi.slots && epi.mode == 2) || Ci != 48 || t7_plan(N, H, W, Ci, Co).ok ||
                 (reinterpret_cast<uintptr_t>(xbn.out) & 15) != 0)) return 0;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (reinterpret_cast<uintptr_t>(wp) & 15) != 0) return 0;
  if (out_f32 || relu || (accumulate && epi.slots && epi.mode == 1)) return 0;                      // (forward-only fused ReLU / fp32 heatmap outputs stay on conv_t4)
  if ((long)H * W * Ci * 2 >= (1L << 31) || (long)9 * KC * NTt * 1024 >= (1L << 31)) return 0;
  if (KC * 32 < Ci || NTt * 16 < Co) return 0;
  const T7Plan q7 = t7_plan(N, H, W, Ci, Co);
  if (q7.ok) {
    const int rc7 = half_kind == 1 ? t7_launch<f16_t>(q7, x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, accumulate, s, epi)
                                   : t7_launch<bf16_t>(q7, x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, accumulate, s, epi);
    if (!rc7) return 0;
    hipError_t err7 = hipGetLastError();
    if (err7 != hipSuccess) {
      fami_set_error(name, hipGetErrorString(err7));
      return FAMI_EHIP;
    }
    return 1;
  }
  const T6Plan q = t6_plan(N, H, W, Ci, Co);
  if (!q.ok) return 0;
  int rc;
  if (half_kind == 1) rc = t6_launch<f16_t>(q, x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, out_f32, s, epi, xbn);
  else rc = t6_launch<bf16_t>(q, x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, out_f32, s, epi, xbn);
  if (!rc) return 0;
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    fami_set_error(name, hipGetErrorString(err));
    return FAMI_EHIP;
  }
  return 1;
}
// conv_pair.hip: which instance would take this launch (v = SG, NT, MT, EX) -> 6 | 7 | 0
int fami_t6_pair_probe(int N, int H, int W, int Ci, int Co, int* v) {
  const T7Plan q7 = t7_plan(N, H, W, Ci, Co);
  if (q7.ok) {
    v[0] = q7.SG; v[1] = q7.NT; v[2] = q7.MT; v[3] = q7.EX;
    return 7;
  }
  const T6Plan q = t6_plan(N, H, W, Ci, Co);
  if (!q.ok) return 0;
  v[0] = 6; v[1] = 3; v[2] = q.MT; v[3] = q.TU > 8 * q.MT ? 1 : 0;
  return 6;
}
// would the 48-channel kernel take this FORWARD convolution with the BatchNorm + ReLU in front of it inside the launch (XB instances)?
extern "C" int fami_conv2d_fwd_bnin_ok(int N, int H, int W, int Ci, int Co) {
  return g_bn_in && Ci == 48 && !t7_plan(N, H, W, Ci, Co).ok && t6_plan(N, H, W, Ci, Co).ok ? 1 : 0;
}
// 1: the 48-channel kernel takes it, 2: the phased kernel (48-channel phases), 3: the phased kernel with 32-channel phases, 0: neither
extern "C" int fami_conv_t6_eligible(int N, int H, int W, int Ci, int Co) {
  const T7Plan q7 = t7_plan(N, H, W, Ci, Co);
  if (q7.ok) return q7.SG == 4 ? 3 : 2;
  return t6_plan(N, H, W, Ci, Co).ok ? 1 : 0;
}
void fami_conv_t6_tune(int on) {
  if (on < 0) { g_use_t6 = 1; g_t6_rows = 0; g_t6_min_jobs = 96; g_t6_mt = 0; g_use_t7 = 1; g_t7_rows = 0; g_t7_target = 240; g_t7_c64 = 1; g_bwd_pair = 1; g_bn_in = 1; }
  else if (on == 8996 || on == 8997) g_bn_in = on - 8996;                 // the input-BatchNorm instances off / on
  else if (on == 8998 || on == 8999) g_bwd_pair = on - 8998;              // conv_pair.hip
  else if (on == 8502 || on == 8503) g_t7_c64 = on - 8502;
  else if (on >= 8700 && on < 8996) g_t7_target = on - 8700;
  else if (on == 8500 || on == 8501) g_use_t7 = on - 8500;
  else if (on >= 8600 && on < 8700) g_t7_rows = on - 8600;
  else if (on >= 8200 && on <= 8202) g_t6_mt = on - 8200;
  else if (on == 8000 || on == 8001) g_use_t6 = on - 8000;
  else if (on >= 8400 && on < 8500) g_t6_min_jobs = on - 8400;
  else if (on >= 8100 && on < 8200) g_t6_rows = on - 8100;
}
