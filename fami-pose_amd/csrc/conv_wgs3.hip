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
#include "conv_wgs3_dev.h"
#include "conv_pair.h"

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
  // round 6 (most of these launches now share their launch with the input gradient, conv_pair.h): 128 -- f32 step 96 / 128 / 144 / 160 /
  // 192 / 256 workgroups 44.51 / 44.07 / - / 44.40 / 44.29 / 44.83 ms and, second box, - / 44.19 / 44.43 / - / 44.66 / - (tools/ab_env.py)
  const long target = g_wgs3_target > 0 ? g_wgs3_target : 128;
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

static int wgs3_dispatch(const Wgs3Args& a, dim3 grid, size_t lds, int CIT, int COT, int NYS, hipStream_t s) {
  bool ok = false;
#define FAMI_WS3_CASE(cit, cot, nys)                                                                                    \
  if (!ok && CIT == cit && COT == cot && NYS == nys) {                                                                  \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv_wgrad_s3_kernel<cit, cot, nys>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((conv_wgrad_s3_kernel<cit, cot, nys>), grid, dim3(WS3_THREADS), lds, s, a);                      \
    ok = true;                                                                                                            \
  }
#define FAMI_WS3_ROW(nys)                                                                                               \
  FAMI_WS3_CASE(1, 1, nys) FAMI_WS3_CASE(1, 2, nys) FAMI_WS3_CASE(1, 3, nys) FAMI_WS3_CASE(2, 1, nys)               \
  FAMI_WS3_CASE(2, 2, nys) FAMI_WS3_CASE(2, 3, nys) FAMI_WS3_CASE(3, 1, nys) FAMI_WS3_CASE(3, 2, nys) FAMI_WS3_CASE(3, 3, nys)
  FAMI_WS3_ROW(4) FAMI_WS3_ROW(7)
#undef FAMI_WS3_ROW
#undef FAMI_WS3_CASE
  return ok ? 1 : 0;
}
// conv_pair.hip: a recorded weight-gradient half as the single launch it would have been; which instance would take a launch
int fami_wgs3_pair_replay(const PairHalf& h, hipStream_t s) {
  if (h.kind != 13) return 0;
  Wgs3Args a;
  memcpy(&a, h.args, sizeof(a));
  return wgs3_dispatch(a, dim3(h.gx, h.gy), h.lds, h.v[0], h.v[1], h.v[2], s);
}
int fami_wgs3_pair_probe(int N, int H, int W, int Ci, int Co, int* v) {
  const Wgs3Plan q = wgs3_plan(N, H, W, Ci, Co);
  if (!q.ok) return 0;
  v[0] = q.CIT; v[1] = q.COT; v[2] = q.NYS;
  return 13;
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
  if (PairCapture* pc = fami_pair_capture()) {      // conv_pair.h: recorded, launched by fami_conv2d_bwd_pair_f32
    pair_record(pc->b, 13, 2, a, grid, q.lds, q.CIT, q.COT, q.NYS, 0);
    pc->b.slabs = q.G;
    return q.G;
  }
  const bool ok = wgs3_dispatch(a, grid, q.lds, q.CIT, q.COT, q.NYS, s) != 0;
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
