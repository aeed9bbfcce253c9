// One launch for the two backward kernels of a 3x3 stride-1 convolution in the 16-bit storage modes (round 6; see conv_pair.h):
// the backward of posetimation/layers/basic_model.py:44-63 (conv1 / conv2 of every BasicBlock of the HRNet branches,
// backbones/hrnet.py:17-172) needs dX = conv_transpose(dY, W) and dW = X (*) dY; both read dY, neither reads the other.
// The grid is [ weight-gradient workgroups (padded to a multiple of 8) | input-gradient workgroups ]: workgroups are dispatched
// in flat-id order, so the longer-running weight-gradient workgroups start first and the input-gradient workgroups fill the
// remaining CUs and the CUs that free up.  A workgroup runs exactly one of the two bodies (a scalar branch on the block id);
// registers and LDS of the launch are the larger of the two, and both single kernels already were one 512-thread workgroup
// per CU.  The padding keeps blockIdx % 8 (the XCD a workgroup lands on) equal to the value the input-gradient body derives its
// XCD-contiguous job order from.
#include "conv_t6_dev.h"
#include "conv_wg6_dev.h"
#include "conv_t5_dev.h"
#include "conv_wgs3_dev.h"
#include "conv_pair.h"
#include "route.h"

static thread_local PairCapture* g_pair_capture = nullptr;
PairCapture*& fami_pair_capture() { return g_pair_capture; }

int fami_t6_pair_probe(int N, int H, int W, int Ci, int Co, int* v);
int fami_wg6_pair_probe(int N, int H, int W, int Ci, int Co, int k, int st, int pad, int dil, int* v);
int fami_t6_pair_replay(const PairHalf& h, hipStream_t s);
int fami_wg6_pair_replay(const PairHalf& h, hipStream_t s);
int fami_t5_pair_probe(int N, int H, int W, int Ci, int Co, int* v);
int fami_wgs3_pair_probe(int N, int H, int W, int Ci, int Co, int* v);
int fami_t5_pair_replay(const PairHalf& h, hipStream_t s);
int fami_wgs3_pair_replay(const PairHalf& h, hipStream_t s);

struct PairGeo {
  int nb, nbp;     // weight-gradient workgroups, the same rounded up to a multiple of 8
  int bgx, agx;    // grid.x of the weight-gradient / input-gradient single kernel
  int bgy, agy;    // grid.y of the two single kernels
};

// A = conv3x3_t6_body (48 input channels), B = conv_wgrad6_body.  The input-gradient arguments come FIRST: the body reads its
// EpiBN block through the kernarg segment pointer at offsetof(ConvT6Args, e) (conv_epi.h epi_late).
template <typename H, int MT, int EX, bool ACC, int EM, int KS, int CIT, int COT>
__global__ __launch_bounds__(T6_THREADS, 1) void bwd_pair_t6_kernel(ConvT6Args a, Wg6Args b, PairGeo g) {
  const int lin = blockIdx.x;
  if (lin < g.nbp) {
    if (lin >= g.nb) return;
    conv_wgrad6_body<H, KS, CIT, COT, WG6_XJ>(b, lin % g.bgx, lin / g.bgx, g.bgx);
  } else {
    const int l = lin - g.nbp;
    conv3x3_t6_body<H, 6, 3, MT, EX, ACC, EM>(a, l % g.agx, l / g.agx, g.agx);
  }
}
// A = conv3x3_t7_body (phases of 8 SG input channels)
template <typename H, int SG, int NT, int MT, int EX, bool ACC, int EM, int KS, int CIT, int COT>
__global__ __launch_bounds__(T6_THREADS, 1) void bwd_pair_t7_kernel(ConvT7Args a, Wg6Args b, PairGeo g) {
  const int lin = blockIdx.x;
  if (lin < g.nbp) {
    if (lin >= g.nb) return;
    conv_wgrad6_body<H, KS, CIT, COT, WG6_XJ>(b, lin % g.bgx, lin / g.bgx, g.bgx);
  } else {
    const int l = lin - g.nbp;
    conv3x3_t7_body<H, SG, NT, MT, EX, ACC, EM>(a, l % g.agx, l / g.agx, g.agx);
  }
}

// f32 storage: A = conv3x3_t5_body<float, 3, true> (the persistent split-product kernel: its workgroups deal themselves the jobs by
// workgroup index, so the body gets the index within its own part of the grid), B = conv_wgrad_s3_body
template <int NT, int CIT, int COT, int NYS>
__global__ __launch_bounds__(T5_THREADS, 2) void bwd_pair_t5_kernel(ConvT5Args a, Wgs3Args b, PairGeo g) {
  const int lin = blockIdx.x;
  if (lin < g.nbp) {
    if (lin >= g.nb) return;
    conv_wgrad_s3_body<CIT, COT, NYS>(b, lin % g.bgx, lin / g.bgx, g.bgx, g.bgy);
  } else {
    conv3x3_t5_body<float, NT, true>(a, lin - g.nbp, g.agx);
  }
}
#define PAIR_SHAPES_F32(X) \
  X(3, 3, 3, 4)      /* 48 -> 48 @96x72 and 96 -> 96 @48x36 (48-channel blocks, runs of nine tiles) */ \
  X(3, 3, 3, 7)      /* the 64x48 / 32x24 maps of the 512x384 configuration */

// The combined instances: (input-gradient kernel, SG, NT, MT, EX) x (KS, CIT, COT) of the layer shapes of the path -- every
// (ACC, EM) pair the backward pass uses (EM 0: plain, 2: backward BatchNorm statistics in the epilogue) for bf16 and fp16.
//   W48 @384x288: branch 0 (48 ch, 96x72) t6 MT 2 EX 1 / KS 9; branches 1-3 and the 512x384 maps: see PAIR_SHAPES below
#define PAIR_SHAPES(X)            \
  X(6, 6, 3, 2, 1, 9, 3, 3)       /* 48 ch @96x72, >= 8 frames (branch 0 of every stage; B >= 8 heads) */            \
  X(6, 6, 3, 1, 1, 9, 3, 3)       /* 48 ch @96x72, 4 frames (the head's aggregation blocks at B = 4) */              \
  X(7, 6, 3, 2, 1, 9, 3, 3)       /* 96 ch @48x36 */                                                                 \
  X(7, 6, 3, 1, 1, 7, 3, 3)       /* 192 ch @24x18, 20 frames */                                                     \
  X(7, 6, 3, 1, 0, 7, 3, 3)       /* 192 ch @24x18, 24 frames (config 2) */                                          \
  X(7, 6, 3, 1, 0, 4, 3, 3)       /* 384 ch @12x9 */                                                                 \
  X(7, 4, 4, 2, 1, 9, 4, 2)       /* HRNet-W64: 64 ch @96x72, 128 ch @48x36; stage 1's 64 -> 64 */                   \
  X(7, 4, 4, 1, 1, 5, 4, 2)       /* 256 ch @24x18 */                                                                \
  X(7, 4, 4, 1, 0, 4, 4, 2)       /* 512 ch @12x9 */

static bool pair_shape_known(int kind, const int* va, const int* vb) {
#define X(k, sg, nt, mt, ex, ks, cit, cot) \
  if (kind == k && va[0] == sg && va[1] == nt && va[2] == mt && va[3] == ex && vb[0] == ks && vb[1] == cit && vb[2] == cot && vb[3] == WG6_XJ) return true;
  PAIR_SHAPES(X)
#undef X
  return false;
}

template <typename HT, int K, int SG, int NT, int MT, int EX, bool ACC, int EM, int KS, int CIT, int COT>
static void pair_launch_one(const PairCapture& c, const PairGeo& g, unsigned blocks, size_t lds, hipStream_t s) {
  Wg6Args b;
  memcpy(&b, c.b.args, sizeof(b));
  if constexpr (K == 6) {
    ConvT6Args a;
    memcpy(&a, c.a.args, sizeof(a));
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)bwd_pair_t6_kernel<HT, MT, EX, ACC, EM, KS, CIT, COT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
    hipLaunchKernelGGL((bwd_pair_t6_kernel<HT, MT, EX, ACC, EM, KS, CIT, COT>), dim3(blocks), dim3(T6_THREADS), lds, s, a, b, g);
  } else {
    ConvT7Args a;
    memcpy(&a, c.a.args, sizeof(a));
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)bwd_pair_t7_kernel<HT, SG, NT, MT, EX, ACC, EM, KS, CIT, COT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
    hipLaunchKernelGGL((bwd_pair_t7_kernel<HT, SG, NT, MT, EX, ACC, EM, KS, CIT, COT>), dim3(blocks), dim3(T6_THREADS), lds, s, a, b, g);
  }
}

template <typename HT>
static bool pair_launch_typed(const PairCapture& c, const PairGeo& g, unsigned blocks, size_t lds, hipStream_t s) {
  const int* va = c.a.v;
  const int* vb = c.b.v;
  const bool acc = va[4] != 0;
  const int em = va[5];
#define X(k, sg, nt, mt, ex, ks, cit, cot)                                                                                         \
  if (c.a.kind == k && va[0] == sg && va[1] == nt && va[2] == mt && va[3] == ex && vb[0] == ks && vb[1] == cit && vb[2] == cot) { \
    if (!acc && em == 0) { pair_launch_one<HT, k, sg, nt, mt, ex, false, 0, ks, cit, cot>(c, g, blocks, lds, s); return true; }    \
    if (acc && em == 0) { pair_launch_one<HT, k, sg, nt, mt, ex, true, 0, ks, cit, cot>(c, g, blocks, lds, s); return true; }      \
    if (!acc && em == 2) { pair_launch_one<HT, k, sg, nt, mt, ex, false, 2, ks, cit, cot>(c, g, blocks, lds, s); return true; }    \
    if (acc && em == 2) { pair_launch_one<HT, k, sg, nt, mt, ex, true, 2, ks, cit, cot>(c, g, blocks, lds, s); return true; }      \
    return false;                                                                                                                  \
  }
  PAIR_SHAPES(X)
#undef X
  return false;
}

// [fami_route_t] g_bwd_pair (default 1)  // fami_conv_tune_lds(8998 / 8999): the combined launch off (the two recorded halves run as single launches) / on
// What was recorded runs: as one launch where both halves are there and a combined instance exists, as single launches otherwise.
// -> 0 ok (combined), 1 ok (singles), < 0: a recorded half has no instance (cannot happen for plans of this library)
static bool pair_f32_known(const int* va, const int* vb) {
#define X(nt, cit, cot, nys) if (va[0] == nt && vb[0] == cit && vb[1] == cot && vb[2] == nys) return true;
  PAIR_SHAPES_F32(X)
#undef X
  return false;
}
static bool pair_launch_f32(const PairCapture& c, const PairGeo& g, unsigned blocks, size_t lds, hipStream_t s) {
  ConvT5Args a;
  Wgs3Args b;
  memcpy(&a, c.a.args, sizeof(a));
  memcpy(&b, c.b.args, sizeof(b));
#define X(nt, cit, cot, nys)                                                                                                         \
  if (c.a.v[0] == nt && c.b.v[0] == cit && c.b.v[1] == cot && c.b.v[2] == nys) {                                                    \
    static bool attr = false;                                                                                                        \
    if (!attr) {                                                                                                                     \
      (void)hipFuncSetAttribute((const void*)bwd_pair_t5_kernel<nt, cit, cot, nys>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr = true;                                                                                                                   \
    }                                                                                                                                \
    hipLaunchKernelGGL((bwd_pair_t5_kernel<nt, cit, cot, nys>), dim3(blocks), dim3(T5_THREADS), lds, s, a, b, g);                    \
    return true;                                                                                                                     \
  }
  PAIR_SHAPES_F32(X)
#undef X
  return false;
}

int fami_pair_launch(const PairCapture& c, hipStream_t s) {
  if (c.a.kind == 5 || c.b.kind == 13) {      // f32 storage
    if (c.a.kind == 5 && c.b.kind == 13 && g_bwd_pair) {
      PairGeo g;
      g.nb = (int)(c.b.gx * c.b.gy);
      g.nbp = (g.nb + 7) & ~7;
      g.bgx = (int)c.b.gx;
      g.bgy = (int)c.b.gy;
      g.agx = (int)c.a.gx;
      g.agy = (int)c.a.gy;
      const size_t lds = c.a.lds > c.b.lds ? c.a.lds : c.b.lds;
      const unsigned blocks = (unsigned)g.nbp + c.a.gx * c.a.gy;
      if (pair_f32_known(c.a.v, c.b.v) && pair_launch_f32(c, g, blocks, lds, s)) return 0;
    }
    if (c.a.kind == 5 && !fami_t5_pair_replay(c.a, s)) return -1;
    if (c.b.kind == 13 && !fami_wgs3_pair_replay(c.b, s)) return -1;
    return 1;
  }
  const bool both = c.a.kind != 0 && c.b.kind == 16 && c.a.half_kind == c.b.half_kind && c.b.v[3] == WG6_XJ;
  if (both && g_bwd_pair && pair_shape_known(c.a.kind, c.a.v, c.b.v)) {
    PairGeo g;
    g.nb = (int)(c.b.gx * c.b.gy);
    g.nbp = (g.nb + 7) & ~7;
    g.bgx = (int)c.b.gx;
    g.bgy = (int)c.b.gy;
    g.agx = (int)c.a.gx;
    g.agy = (int)c.a.gy;
    const unsigned blocks = (unsigned)g.nbp + c.a.gx * c.a.gy;
    const size_t lds = c.a.lds > c.b.lds ? c.a.lds : c.b.lds;
    const bool ok = c.a.half_kind == 1 ? pair_launch_typed<f16_t>(c, g, blocks, lds, s) : pair_launch_typed<bf16_t>(c, g, blocks, lds, s);
    if (ok) return 0;
  }
  if (c.a.kind != 0 && !fami_t6_pair_replay(c.a, s)) return -1;
  if (c.b.kind != 0 && !fami_wg6_pair_replay(c.b, s)) return -1;
  return 1;
}

// Would fami_conv2d_bwd_pair_* run this convolution's backward as ONE launch (16-bit storage)?  Geometry of the forward convolution.
extern "C" int fami_conv2d_bwd_pair_ok(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil) {
  if (!g_bwd_pair || kh != 3 || kw != 3 || stride != 1 || pad != 1 || dil != 1) return 0;
  int va[6] = {0, 0, 0, 0, 0, 0}, vb[6] = {0, 0, 0, 0, 0, 0};
  const int ka = fami_t6_pair_probe(N, H, W, Co, Ci, va);      // the input gradient is a convolution of dY (Co channels) into Ci
  if (!ka) return 0;
  if (!fami_wg6_pair_probe(N, H, W, Ci, Co, 3, 1, 1, 1, vb)) return 0;
  return pair_shape_known(ka, va, vb) ? 1 : 0;
}
extern "C" int fami_conv2d_bwd_pair_ok_f32(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil) {
  if (!g_bwd_pair || kh != 3 || kw != 3 || stride != 1 || pad != 1 || dil != 1) return 0;
  int va[6] = {0, 0, 0, 0, 0, 0}, vb[6] = {0, 0, 0, 0, 0, 0};
  if (!fami_wgs3_pair_probe(N, H, W, Ci, Co, vb)) return 0;
  if (fami_t5_pair_probe(N, H, W, Co, Ci, va)) return pair_f32_known(va, vb) ? 1 : 0;
  return 0;      // (the band kernel's launches -- the low-resolution maps -- stay two launches: combined they measured slower, see conv_pair.h)
}
// probe for tools / tests (f32 storage): out = kind (5 | 0), NT, 0, 0 | CIT, COT, NYS
extern "C" int fami_conv2d_bwd_pair_key_f32(int N, int H, int W, int Ci, int Co, int* out) {
  int va[6] = {0, 0, 0, 0, 0, 0}, vb[6] = {0, 0, 0, 0, 0, 0};
  const int ka = fami_t5_pair_probe(N, H, W, Co, Ci, va);
  const int kb = fami_wgs3_pair_probe(N, H, W, Ci, Co, vb);
  out[0] = ka;
  for (int i = 0; i < 3; ++i) { out[1 + i] = va[i]; out[4 + i] = vb[i]; }
  return ka && kb ? 1 : 0;
}
// probe for tools / tests: the (kind, SG, NT, MT, EX | KS, CIT, COT, XJ) key of the two halves -> out[9]; returns 1 if both halves exist
extern "C" int fami_conv2d_bwd_pair_key(int N, int H, int W, int Ci, int Co, int* out) {
  int va[6] = {0, 0, 0, 0, 0, 0}, vb[6] = {0, 0, 0, 0, 0, 0};
  const int ka = fami_t6_pair_probe(N, H, W, Co, Ci, va);
  const int kb = fami_wg6_pair_probe(N, H, W, Ci, Co, 3, 1, 1, 1, vb);
  out[0] = ka;
  for (int i = 0; i < 4; ++i) { out[1 + i] = va[i]; out[5 + i] = vb[i]; }
  return ka && kb ? 1 : 0;
}
