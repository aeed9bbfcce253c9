// Persistent, unit-pipelined 3x3 convolution (stride 1, pad 1), round 4: forward and input gradient of the HRNet branch
// convolutions (posetimation/backbones/hrnet.py:17-172 via layers/basic_model.py:25-63) -- the f32-storage split-product
// arithmetic of conv_t4.hip (three bf16 terms per operand, six products on the bf16 matrix pipe, fp32 accumulation) and
// the 16-bit storage types -- restructured around what the round-3 kernel's trace showed:
//   * conv_t4's workgroup = one band: prologue (first fetch), 2-24 chunks of [store -> barrier -> taps -> barrier], store
//     of the results; with 100 KB of LDS one workgroup per CU, so none of these phases overlapped anything, and a launch
//     was 1.4 "rounds" of workgroups on 256 CUs (360 workgroups: the second round 41 % full);
//   * every workgroup re-split the same f32 weights into bf16 terms (42 % of the staging arithmetic, a third of the LDS
//     stores).
// Here
//   * the grid is PERSISTENT: <= 256 workgroups walk a job list (frame, band of R whole rows, output-channel block) sized
//     so that every workgroup gets the same number of jobs (no tail round);
//   * the weights come PRE-SPLIT from the packed image (three bf16 planes per weight, written once per step by the
//     weight pack: fami_pack_split_*) and go global -> LDS by DMA (global_load_lds_dwordx4): no registers, no VALU, no
//     ds_write for 49 % of what a chunk stages;
//   * the K loop is cut into UNITS of one tap row of one 16-channel chunk (split-product form; a whole chunk for the
//     16-bit types): while unit u is multiplied out of weight slab u & 1, the DMA of slab u + 1 is in flight into the
//     other one, and the NEXT chunk's activation patch (fetched into registers one chunk ahead, split while it is
//     stored) is written into the second patch buffer in the middle of the current chunk -- one barrier per unit, the
//     pipeline runs across chunk AND job boundaries, so a job's prologue / epilogue hide behind its neighbours' MFMAs.
// Same summation order per output element as conv_t4's split-product instance: results are bitwise equal to it
// (tests/test_kernels_gpu.py::test_persistent_conv_is_bitwise_the_band_kernel).
#include "conv_epi.h"
#include "conv_t5_dev.h"
#include "conv_pair.h"
#include <type_traits>

// (t5_bf16x4: conv_t5_dev.h)

// ------------------------------------------------------------------ the pre-split weight image
// [tap][K/16][N/16][plane 3][n 16][k 16] bf16 (a (tap, K group, N tile) block is 1536 bytes = what the kernel copies);
// mode 0: K = Cin, N = Cout; mode 1 (input gradient): K = Cout, N = Cin -- the orientation of the f32 fragment image
// it follows.
long fami_split_image_elems(int kd, int nd, int taps) {
  return (taps == 9 && kd % 16 == 0) ? (long)taps * (kd / 16) * ((nd + 15) / 16) * 384 : 0;
}
__device__ __forceinline__ void t5_split3(float v, __bf16& h0, __bf16& h1, __bf16& h2) {
  h0 = (__bf16)v;
  const float r1 = v - (float)h0;     // exact
  h1 = (__bf16)r1;
  const float r2 = r1 - (float)h1;    // exact, representable
  h2 = (__bf16)r2;
}
struct T5PackDesc { long src, dst; int Co, Ci, taps, mode; };
// One (K group, N tile) block at a time through LDS: its source is 16 output-channel rows of 16 * 9 contiguous floats (OIHW
// keeps a (co, ci) pair's taps adjacent) -- read coalesced, written as 8-byte runs of a plane row (the element-by-element
// form with its stride-9 reads and 2-byte stores took 477 us per launch for the step's 3x3 images, on the critical path at
// the top of the step).
__device__ __forceinline__ void t5_pack_split_image(const float* __restrict__ w, __bf16* __restrict__ sp, int Co, int Ci,
                                                    int mode, int first_block, int block_stride, float* tile) {
  const int kd = mode == 0 ? Ci : Co, nd = mode == 0 ? Co : Ci;
  const int KC = kd / 16, NTt = (nd + 15) / 16;
  constexpr int PITCH = 16 * 9 + 1;
  for (int b = first_block; b < KC * NTt; b += block_stride) {
    const int kc = b / NTt, nt = b - kc * NTt;
    const int co0 = mode == 0 ? nt * 16 : kc * 16, ci0 = mode == 0 ? kc * 16 : nt * 16;
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * 144; i += 256) {
      const int r = i / 144, c = i - r * 144;
      const int co = co0 + r, ci = ci0 + c / 9;
      tile[r * PITCH + c] = (co < Co && ci < Ci) ? w[((long)co * Ci + ci0) * 9 + c] : 0.f;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 9 * 64; o += 256) {          // (tap, n, k quad) -> four values of a plane row
      const int tap = o >> 6, n = (o >> 2) & 15, k4 = (o & 3) * 4;
      t5_bf16x4 h0, h1, h2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k4 + j;
        const float v = mode == 0 ? tile[n * PITCH + k * 9 + tap] : tile[k * PITCH + n * 9 + tap];
        __bf16 a0, a1, a2;
        t5_split3(v, a0, a1, a2);
        h0[j] = a0; h1[j] = a1; h2[j] = a2;
      }
      __bf16* blk = sp + ((long)(tap * KC + kc) * NTt + nt) * 768 + n * 16 + k4;
      *reinterpret_cast<t5_bf16x4*>(blk) = h0;
      *reinterpret_cast<t5_bf16x4*>(blk + 256) = h1;
      *reinterpret_cast<t5_bf16x4*>(blk + 512) = h2;
    }
  }
}
__global__ __launch_bounds__(256) void t5_pack_split_kernel(const float* __restrict__ w, __bf16* __restrict__ sp, int Co, int Ci, int mode) {
  __shared__ float tile[16 * (16 * 9 + 1)];
  t5_pack_split_image(w, sp, Co, Ci, mode, blockIdx.x, gridDim.x, tile);
}
__global__ __launch_bounds__(256) void t5_pack_split_batch_kernel(const float* __restrict__ params, float* __restrict__ packed,
                                                                   const T5PackDesc* __restrict__ desc) {
  __shared__ float tile[16 * (16 * 9 + 1)];
  const T5PackDesc d = desc[blockIdx.y];
  const int kd = d.mode == 0 ? d.Ci : d.Co, nd = d.mode == 0 ? d.Co : d.Ci;
  if (d.taps != 9 || (kd % 16) != 0) return;
  const long n16 = (long)9 * (kd / 16) * ((nd + 15) / 16) * 256;      // the f32 fragment image in front (no 32x32 image for 3x3)
  t5_pack_split_image(params + d.src, reinterpret_cast<__bf16*>(packed + d.dst + n16), d.Co, d.Ci, d.mode, blockIdx.x, gridDim.x, tile);
}
void fami_pack_split_single(const float* w_oihw, float* split, int Co, int Ci, int taps, int mode, hipStream_t s) {
  const int kd = mode == 0 ? Ci : Co, nd = mode == 0 ? Co : Ci;
  if (fami_split_image_elems(kd, nd, taps) == 0) return;
  int blocks = (kd / 16) * ((nd + 15) / 16);
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(t5_pack_split_kernel, dim3(blocks), dim3(256), 0, s, w_oihw, reinterpret_cast<__bf16*>(split), Co, Ci, mode);
}
void fami_pack_split_batch(const float* params, float* packed, const void* desc, int n, hipStream_t s) {
  hipLaunchKernelGGL(t5_pack_split_batch_kernel, dim3(48, n), dim3(256), 0, s, params, packed, reinterpret_cast<const T5PackDesc*>(desc));
}


// ------------------------------------------------------------------ plan + launch
static long long* g_t5_dbg = nullptr;   // fami_conv_t5_debug (FAMI_T5_TRACE builds)
extern "C" void fami_conv_t5_debug(void* buf) { g_t5_dbg = reinterpret_cast<long long*>(buf); }
// [fami_route_t] g_use_t5 (default 1)  // fami_conv_tune_lds(7000 / 7001): off / on
// [fami_route_t] g_t5_rows (default 0)  // fami_conv_tune_lds(7100 + R): force the rows per band (benchmarks)
// [fami_route_t] g_t5_maxwg (default 256)  // fami_conv_tune_lds(7500 + n): at most 8 n workgroups in the persistent grid (benchmarks; 7599: one job per workgroup)
// [fami_route_t] g_t5_h16 (default 0)  // fami_conv_tune_lds(7010 / 7011): the 16-bit instances off / on
// [fami_route_t] g_t5_abl (default 0)  // fami_conv_tune_lds(7700 + n): ablation, see ConvT5Args.abl_chunks
// [fami_route_t] g_t5_min_jobs (default 200)  // fami_conv_tune_lds(7600 + n): only launches of >= n jobs
// [fami_route_t] g_t5_min_tiles (default 0)  // fami_conv_tune_lds(7400 + n): only frames of >= 8 n tiles (7401: >= 1)  // fami_conv_tune_lds(7400 + n): only frames of >= n tiles (benchmarks / routing experiments)

struct T5Plan { int ok, NT, R, bands, cblocks, njobs, G, npos; size_t lds; };
template <bool S3>
static T5Plan t5_plan(int N, int H, int W, int Ci, int Co, bool xbn) {
  T5Plan q;
  q.ok = 0;
  q.NT = Co % 48 == 0 ? 3 : 0;      // (64-wide channel blocks: three pixel tiles x four channel tiles spill; conv_t4 keeps those)
  if (!q.NT || !g_use_t5) return q;
  if (S3 ? (Ci % 16) != 0 : (Ci % 8) != 0) return q;
  q.cblocks = Co / (16 * q.NT);
  const int PW = W + 2;
  const int blk = S3 ? 1536 : 1024, ps = S3 ? 96 : 80;
  const size_t fixed = (size_t)(S3 ? 9 : 18) * q.NT * blk + (size_t)T5_WAVES * q.NT * 32 * 4 + (xbn ? (size_t)2 * Ci * 4 : 0);   // the two weight regions, the EpiBN exchange, the XBN tables
  // rows per band: the least (rounds of the persistent grid) x (MFMA work of the busiest SIMD per band); waves w and w + 4 share
  // a SIMD, wave w owns tiles w, w + 8, w + 16; + a fixed cost per band (pipeline bubbles at the job boundary, halo rows, epilogue)
  auto simd_cost = [](int tiles) {
    int simd = 0;
    for (int w = 0; w < 4; ++w) {
      int t = 0;
      for (int k = 0; k < T5_MTT; ++k) t += (w + 8 * k < tiles) + (w + 4 + 8 * k < tiles);
      simd = t > simd ? t : simd;
    }
    return simd;
  };
  double best = 1e30;
  q.R = 0;
  for (int R = 1; R <= H; ++R) {
    if (g_t5_rows > 0 && R != g_t5_rows) continue;
    const int tiles = (R * W + 15) / 16;
    const long npos = (long)(R + 2) * PW;
    if (tiles > T5_WAVES * T5_MTT - 6 || npos * 4 > (long)T5_PM * T5_THREADS) break;      // 18 tiles: waves 0, 1 carry three
    if (2 * (size_t)npos * ps + fixed > 160 * 1024) break;
    // (rounds of the persistent grid) x (a band's cost): a launch that is in flight alone -- 57 % of the f32 step's wall time has
    // exactly one kernel in flight -- is as long as its busiest workgroup.  (Ranking by CU time alone, i.e. without the rounds,
    // picked 7-row bands on the 48x36 maps: 280 jobs on 140 workgroups, 56 us per launch against 36 us; inside the step the two
    // rankings measured the same, 49.3 vs 49.6 ms.)
    const long njobs_r = (long)N * ((H + R - 1) / R) * q.cblocks;
    const double cost = (double)((njobs_r + g_t5_maxwg - 1) / g_t5_maxwg) * (simd_cost(tiles) + 0.35);
    if (cost < best - 1e-9 || (cost < best + 1e-9 && R > q.R)) {
      best = cost;
      q.R = R;
    }
  }
  if (!q.R) return q;
  q.bands = (H + q.R - 1) / q.R;
  const long njobs = (long)N * q.bands * q.cblocks;
  if (njobs >= (1L << 30)) return q;
  q.njobs = (int)njobs;
  // Launches that cannot give every CU a job (the head's 4-frame convolutions: 96 jobs) and the low-resolution maps stay on
  // the band kernel.  Per launch the 12x9 maps lose (one 7-tile band per frame: 69 vs 79 us) and the 24x18 maps win (52 -> 47 us),
  // but inside the step (tools/ab_env.py: graphs captured per routing and replayed alternately on one box, +-0.1 ms) the 24x18
  // maps lose: no persistent kernel 47.90 ms, frames >= 16 tiles 48.24, >= 32 tiles (96x72 and 48x36) **47.09**, 96x72 only 47.18.
  if (g_t5_rows == 0 && (njobs < g_t5_min_jobs || (long)((H * W + 15) / 16) < (g_t5_min_tiles ? g_t5_min_tiles : 32))) return q;
  const long rounds = (njobs + g_t5_maxwg - 1) / g_t5_maxwg;
  long G = (njobs + rounds - 1) / rounds;          // every workgroup the same number of jobs (+- 1)
  if (G >= 8) G = (G + 7) / 8 * 8;
  if (G > g_t5_maxwg) G = g_t5_maxwg >= 8 ? g_t5_maxwg / 8 * 8 : g_t5_maxwg;
  if (G > njobs) G = njobs;
  q.G = (int)G;
  q.npos = (q.R + 2) * PW;
  q.lds = 2 * (size_t)q.npos * ps + fixed;
  q.ok = 1;
  return q;
}

template <typename HT, bool S3>
static int t5_dispatch(const ConvT5Args& a, int G, size_t lds, int NT, hipStream_t s) {
  bool ok = false;
#define FAMI_T5_CASE(nt)                                                                                                  \
  if (NT == nt) {                                                                                                         \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv3x3_t5_kernel<HT, nt, S3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((conv3x3_t5_kernel<HT, nt, S3>), dim3(G), dim3(T5_THREADS), lds, s, a);                            \
    ok = true;                                                                                                            \
  }
  FAMI_T5_CASE(3)
#undef FAMI_T5_CASE
  return ok ? 1 : 0;
}
template <typename HT, bool S3>
static int t5_launch(const T5Plan& q, const void* x, const void* wimg, const float* bias, void* y, int N, int H, int W, int Ci,
                     int Co, int KC, int NTt, int sgn, int relu, int accumulate, int out_f32, hipStream_t s, const EpiBN& epi,
                     const XBN& xbn) {
  ConvT5Args a;
  a.e = epi; a.emode = epi.slots ? epi.mode : 0; a.xb = xbn;
  a.x = x; a.wimg = wimg; a.y = y; a.bias = bias;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.R = q.R; a.bands = q.bands; a.cblocks = q.cblocks; a.njobs = q.njobs;
  a.PW = W + 2; a.KC = KC; a.NTt = NTt; a.sgn = sgn; a.relu = relu; a.accumulate = accumulate; a.out_f32 = out_f32;
  a.nposmax = q.npos;
  a.dbg = g_t5_dbg;
  a.abl_chunks = g_t5_abl;
  a.ppl = q.npos * 32;
  a.patch_bytes = q.npos * (S3 ? 96 : 80);
  if constexpr (S3) {
    if (PairCapture* pc = fami_pair_capture()) {      // conv_pair.h: recorded, launched by fami_conv2d_bwd_pair_f32
      pair_record(pc->a, 5, 2, a, dim3(q.G), q.lds, q.NT, 0, 0, 0);
      return 1;
    }
  }
  return t5_dispatch<HT, S3>(a, q.G, q.lds, q.NT, s);
}
// conv_pair.hip: a recorded input-gradient half (f32 split-product instance) as the single launch it would have been
int fami_t5_pair_replay(const PairHalf& h, hipStream_t s) {
  if (h.kind != 5) return 0;
  ConvT5Args a;
  memcpy(&a, h.args, sizeof(a));
  return t5_dispatch<float, true>(a, (int)h.gx, h.lds, h.v[0], s);
}
// conv_pair.hip: would the persistent split-product kernel take this f32 launch (v[0] = NT)?
int fami_t5_pair_probe(int N, int H, int W, int Ci, int Co, int* v) {
  const T5Plan q = t5_plan<true>(N, H, W, Ci, Co, false);
  if (!q.ok) return 0;
  v[0] = q.NT;
  return 5;
}

// Returns 1 if launched, 0 if the shape is not eligible, < 0 on error.  half_kind: 0 bf16, 1 fp16, 2 f32 (split products).
int fami_try_conv3x3_t5(int half_kind, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci,
                        int Co, int KC, int NTt, int sgn, int relu, int accumulate, int out_f32, hipStream_t s,
                        const char* name, const EpiBN& epi, const XBN& xbn) {
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return 0;
  int rc = 0;
  if (half_kind == 2) {
    const T5Plan q = t5_plan<true>(N, H, W, Ci, Co, xbn.on != 0);
    if (!q.ok || KC * 16 != Ci) return 0;
    // the split image follows the f32 fragment image [9][KC][NTt][64][4] (fami_packed_weight_elems)
    const char* wimg = reinterpret_cast<const char*>(wp) + (size_t)9 * KC * NTt * 1024;
    rc = t5_launch<float, true>(q, x, wimg, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, 1, s, epi, xbn);
  } else {
    if (!g_t5_h16) return 0;
    const T5Plan q = t5_plan<false>(N, H, W, Ci, Co, xbn.on != 0);
    if (!q.ok || KC * 32 < Ci) return 0;
    rc = half_kind == 1 ? t5_launch<f16_t, false>(q, x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, out_f32, s, epi, xbn)
                        : t5_launch<bf16_t, false>(q, x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, out_f32, s, epi, xbn);
  }
  if (!rc) return 0;
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    fami_set_error(name, hipGetErrorString(err));
    return FAMI_EHIP;
  }
  return 1;
}
int fami_conv_t5_eligible_s3(int N, int H, int W, int Ci, int Co) { return t5_plan<true>(N, H, W, Ci, Co, true).ok; }
extern "C" int fami_conv_t5_eligible(int N, int H, int W, int Ci, int Co) { return fami_conv_t5_eligible_s3(N, H, W, Ci, Co); }
void fami_conv_t5_tune(int on) {
  if (on < 0) { g_use_t5 = 1; g_t5_rows = 0; g_t5_maxwg = 256; g_t5_min_tiles = 0; g_t5_min_jobs = 200; g_t5_abl = 0; g_t5_h16 = 0; }
  else if (on >= 7700) g_t5_abl = on - 7700;
  else if (on >= 7600) g_t5_min_jobs = on - 7600;
  else if (on == 7000 || on == 7001) g_use_t5 = on - 7000;
  else if (on == 7010 || on == 7011) g_t5_h16 = on - 7010;
  else if (on >= 7500 && on < 7600) g_t5_maxwg = (on - 7500) * 8;
  else if (on >= 7400) g_t5_min_tiles = on == 7401 ? 1 : (on - 7400) * 8;      // 7401: every frame; else 8 n tiles
  else if (on >= 7100) g_t5_rows = on - 7100;
}
