// Register-blocked LDS-staged 3x3 convolution (stride 1, pad 1) for bf16 / fp16 storage: forward and input gradient of the
// HRNet branch convolutions (posetimation/backbones/hrnet.py:17-172 via layers/basic_model.py:25-63) in the 16-bit modes.
#include "conv_epi.h"
#include <type_traits>

// ------------------------------------------------------------------ register-blocked LDS 3x3 for 16-bit storage ("t4")
// Round-3 replacement of conv3x3_lds_kernel for the 16-bit modes.  In the graph-mode trace of the bf16 step
// (profiles/r03_trace_*) the 96 / 192 / 384-channel branch convolutions ran 31-40 us per 5.7 GFLOP launch -- 10x off
// both rooflines -- and were 20 % of the step's kernel time.  That kernel read one 1 KiB LDS fragment per MFMA (the LDS
// peak of 256 B/clk/CU exactly), synchronised the workgroup once per tap row and staged every channel chunk
// synchronously.  This one
//   * register-blocks up to 4 pixel tiles x NT channel tiles per wave: (4 + NT) fragments feed 4*NT MFMAs
//     (0.58 KiB of LDS traffic per MFMA at NT = 3);
//   * stages a 32-channel chunk of the activation patch AND the nine taps' weight fragments of that chunk in one
//     go, so the workgroup meets twice per chunk (not per tap row) with 9 * 4 * NT MFMAs per wave in between;
//   * fetches the next chunk's patch and weights into registers before it starts multiplying the current one;
//   * keeps the patch with a zero border column on either side (row stride W + 2): a tap is a wave-uniform LDS
//     offset and needs no per-lane mask, and a pixel tile may straddle image rows, so a band is any run of up to 256
//     consecutive pixels of a frame (no row alignment, no ninth-tile special case);
//   * needs 27 KiB (weights) + <= 41 KiB (patch) of LDS: two or three workgroups per CU, and workgroups of DIFFERENT
//     stream lanes can share a CU (the old kernel took 150 KiB).
// MFMA A = weights (rows = 16 output channels), B = activations (cols = 16 pixels), as in the kernels above.
struct ConvT4Args {
  EpiBN e;
  int emode;
  XBN xb;             // BatchNorm + ReLU applied to the input while it is staged (16-bit types; xb.on)
  const void* x;      // [N,H,W,Ci]
  const void* wp;     // packed weights [tap][KC][NTt][64][8]
  void* y;            // [N,H,W,Co]
  const float* bias;  // [Co] or null
  int N, H, W, Ci, Co;
  int BT, bands;      // 16-pixel tiles per band (<= 16), bands per frame
  int PW, PS;         // patch row length in positions (W + 2), bytes per position (80)
  int KC, NTt;
  int sgn, relu, accumulate, out_f32;
  int dil;            // dilation (= padding: the centred 3x3 kernels of the path; 1 for the HRNet blocks, 3 for the DCN predictors)
  int patch_bytes;
  long long* dbg;     // FAMI_T4_TRACE builds: phase timestamps of one workgroup (null otherwise)
};

// storage-type traits: a fragment is 16 bytes per lane in every case -- 8 K-values of a 32-channel chunk for the 16-bit
// types (one v_mfma_f32_16x16x32), 4 K-values of a 16-channel chunk for f32 (four v_mfma_f32_16x16x4_f32, exact f32:
// the K order (lane >> 4) * 4 + t is the same permutation on both operands, as in conv_igemm_f32).  A chunk is 64 bytes
// of a pixel either way, so the staging, the patch layout and the weight slabs (1 KiB blocks) are shared.
template <typename T> struct T4Traits {
  typedef typename H16<T>::x8 frag;
  __device__ static __forceinline__ f32x4 mma(const frag& w, const frag& a, f32x4 acc) { return H16<T>::mfma(w, a, acc); }
};
template <> struct T4Traits<float> {
  typedef f32x4 frag;
  __device__ static __forceinline__ f32x4 mma(const frag& w, const frag& a, f32x4 acc) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t], a[t], acc, 0, 0, 0);
    return acc;
  }
};

#define T4_THREADS 512   // 8 waves: the staging registers per thread halve, 2 pixel tiles x NT channel tiles per wave
#define T4_WAVES (T4_THREADS / 64)
#define T4_MT (16 / T4_WAVES)   // pixel tiles per wave (a band has <= 16)
#define T4_PMAX 5   // most 16-byte patch pieces per thread and chunk (<= 640 positions x 4 pieces / 512 threads); template PM <= it

// S3 (H = float only): f32 storage, products on the bf16 matrix pipe.  Every f32 operand is split while it is staged into
// three bf16 terms, x = x0 + x1 + x2 EXACTLY (8 + 8 + 8 mantissa bits; bf16 has f32's exponent range), and the six products
// x0w0, x1w0, x0w1, x2w0, x1w1, x0w2 are accumulated in fp32 -- the three dropped ones are below 2^-24 of the product.
// Measured against fp64 (tools/probes/split_mfma.hip, MI355X): error <= the v_mfma_f32_16x16x4_f32 chain's at K = 448 and
// K = 3456 (6.3e-7 vs 4.2e-7, 1.4e-6 vs 2.0e-6 of the result's maximum), i.e. this IS an f32 convolution, at 3 matrix-pipe
// MFMAs (16 cycles each) per 16 channels where the exact-f32 MFMA spends 4 x 32 cycles on the VALU-rate pipe.
// LDS: one array per plane, a row (a patch position or a weight row) is 16 channels bf16 = 32 bytes, unpadded.  A
// ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, ...: eight rows at channel offset 0 and eight other rows at
// offset 16 bytes -- with 32-byte rows those are the 16 distinct bank quads (the first layout, planes interleaved in
// 112-byte rows, was 2-way conflicted on 7 lanes of every group: 54 us per launch at 48 channels, LDS-bound).
// One v_mfma_f32_16x16x32_bf16 takes K = 16 channels of plane a next to 16 channels of plane b, i.e. two of the six
// products: X(0|0) W(0|1) = x0w0 + x0w1, X(1|1) W(0|1) = x1w0 + x1w1, X(0|2) W(2|0) = x0w2 + x2w0 -- three activation
// fragments per pixel tile and two weight fragments per channel tile (12 LDS reads per 18 MFMAs at 2 x 3 tiles; the
// kernel is LDS-read bound before it is MFMA bound).
#define T4_S3_ROW 32
// The low-order MFMAs (x1 w0 + x1 w1, x0 w2 + x2 w0) have their own accumulators: the high-order sum then takes one fp32
// rounding per 16 channels instead of three.  Measured against fp64 (tools/probes/s3_check.py, 12 shapes): error 1.0-2.5x
// the exact-f32 MFMA path's (3.0e-7 vs 1.6e-7 ... 1.26e-6 vs 5.0e-7 of the result's maximum); with one accumulator 2-4x.
// (Splitting the next chunk in registers during the tap loop, so that only the LDS stores sit between the barriers, was
// tried: 236 VGPRs, same time per launch and per step -- the split is not what the staging phase waits for.)
#define T4_S3_ACC2 1
// Where the time of a launch goes (48 -> 48 channels @96x72, 20 frames, 44 us; ablation builds of this kernel, MI355X):
// two of the three MFMAs removed -12 us, weight stores removed -4.7, split arithmetic -3.3, global loads after chunk 0 -2,
// patch stores -1.7; requesting a tap's fragments a whole tap ahead (below) and two 4-wave workgroups per CU instead of
// one 8-wave workgroup changed nothing.  With 100 KiB of LDS per workgroup there is one workgroup and two waves per
// SIMD on a CU, and the phases of a chunk (loads -> split -> LDS stores -> barrier -> LDS reads -> MFMAs) add up instead of
// overlapping: MFMA floor 13.7 us + LDS reads 12 us + LDS stores 4 us + staging.
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
struct T4Split { bf16x4v h[3]; };
__device__ __forceinline__ T4Split t4_split(u32x4 raw) {
  T4Split o;
  const f32x4 v = __builtin_bit_cast(f32x4, raw);
  o.h[0] = __builtin_convertvector(v, bf16x4v);
  const f32x4 r1 = v - __builtin_convertvector(o.h[0], f32x4);      // exact
  o.h[1] = __builtin_convertvector(r1, bf16x4v);
  const f32x4 r2 = r1 - __builtin_convertvector(o.h[1], f32x4);     // exact, and representable in bf16
  o.h[2] = __builtin_convertvector(r2, bf16x4v);
  return o;
}
__device__ __forceinline__ void t4_split_store(char* dst, int plane_bytes, const T4Split& o) {
  *reinterpret_cast<bf16x4v*>(dst) = o.h[0];
  *reinterpret_cast<bf16x4v*>(dst + plane_bytes) = o.h[1];
  *reinterpret_cast<bf16x4v*>(dst + 2 * plane_bytes) = o.h[2];
}

// WV: waves per workgroup (8; a 4-wave S3 form with two workgroups per CU measured the same and is not instantiated).
// MTT: pixel tiles per wave (2; the split-product instance also 3 / 4: bands of 24 / 32 tiles, fewer LDS fragment reads per MFMA)
// PC (split-product instance only): producer / consumer waves.  Waves 0 .. WV/2-1 multiply (2 pixel tiles each: bands of
// <= 8 tiles), waves WV/2 .. WV-1 stage: they fetch, split and store chunk c + 1 into a SECOND LDS buffer while chunk c is
// multiplied out of the first; one barrier per chunk, no store phase on the multiplying waves' timeline.  Measured
// (tools/bench_t4.py "s3pc"): two buffers leave room for 8-tile bands only -- 1080 workgroups at 48 channels @96x72, each
// paying its prologue and the 41 KB weight staging for 128 pixels -- 65 us against 44-52; 59-62 against 68 us on the 12x9
// maps (one band, 24 chunks); f32 step equal with it on the one-band maps.  Off (fami_conv_tune_lds(61 / 62) to try).
template <typename H, int NT, int PM, bool S3 = false, int WV = 8, int MTT = 2, bool PC = false>
__global__ __launch_bounds__(WV * 64, S3 ? 2 : (NT == 3 ? 4 : 3)) void conv3x3_t4_kernel(ConvT4Args p) {
  constexpr int THREADS = WV * 64;
  static_assert(!PC || (S3 && MTT == 2), "producer / consumer waves: split-product instance, 2 tiles per wave");
  constexpr int CW = PC ? WV / 2 : WV;                 // waves that multiply
  constexpr int STH = PC ? (WV - CW) * 64 : THREADS;   // threads that stage
  typedef typename T4Traits<H>::frag frag;
  static_assert(!S3 || sizeof(H) == 4, "the split instance takes f32 storage");
  constexpr int SZ = (int)sizeof(H), CHN = 64 / SZ, PCN = 16 / SZ;   // bytes per element, channels per chunk / per piece
  constexpr int WBLK = S3 ? 16 * T4_S3_ROW : 1024;                   // LDS bytes of one (tap, channel tile) weight block (S3: per plane)
  constexpr int WPL = 9 * NT * WBLK;                                 // S3: bytes of one weight plane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WPC = 9 * NT * 64;               // 16-byte weight pieces per chunk
  constexpr int WR = (WPC + STH - 1) / STH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = PC && wave >= CW;              // wave-uniform
  const int stid = PC ? tid - CW * 64 : tid;           // staging thread index (negative on the multiplying waves of PC)
  const int col = lane & 15, kq = lane >> 4;
  int bxl, byl;
  xcd_tile(1, bxl, byl);   // neighbouring bands (shared halo rows) on one XCD's L2
  const int img = bxl / p.bands, bnd = bxl - img * p.bands;
  const int HW = p.H * p.W;
  const int p0 = bnd * p.BT * 16, p1 = min(p0 + p.BT * 16, HW);
  const int ntile = (p1 - p0 + 15) >> 4;
  const int y0 = p0 / p.W, y1 = (p1 - 1) / p.W;
  const int dil = p.dil;
  const int npos = (y1 - y0 + 1 + 2 * dil) * p.PW;   // patch rows y0-dil .. y1+dil, columns -dil .. W-1+dil (PW = W + 2 dil)
  const int ntg0 = byl * NT;
  char* patch = smem;
  char* wbuf = smem + p.patch_bytes;
  const int mtw = producer ? 0 : (ntile - wave + CW - 1) / CW;   // pixel tiles of this wave: wave, wave + CW, ... (wave-uniform)

  int base[MTT];
#pragma unroll
  for (int mt = 0; mt < MTT; ++mt) {
    const int pp = min(p0 + ((producer ? 0 : wave) + CW * mt) * 16 + col, p1 - 1);   // lanes past the band re-read its last pixel (never stored)
    const int ry = pp / p.W, rx = pp - ry * p.W;
    base[mt] = ((ry - y0 + dil) * p.PW + rx + dil) * p.PS + (S3 ? 0 : kq * 16);
  }
  // S3: per-lane byte offset of fragment m inside a row: K 0..15 (kq 0, 1) from one plane, K 16..31 (kq 2, 3) from another
  const int s3h = kq >> 1, s3l = (kq & 1) * 16;
  const int ppl = p.patch_bytes / 3;                                 // S3: bytes of one patch plane
  const int xo0 = s3l, xo1 = ppl + s3l, xo2 = (s3h ? 2 : 0) * ppl + s3l;                 // X(0|0), X(1|1), X(0|2)
  const int wo0 = (s3h ? 1 : 0) * WPL + s3l + col * T4_S3_ROW, wo1 = (s3h ? 0 : 2) * WPL + s3l + col * T4_S3_ROW;   // W(0|1), W(2|0)
  f32x4 acc[MTT][NT];
#pragma unroll
  for (int mt = 0; mt < MTT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc2[S3 && T4_S3_ACC2 ? MTT : 1][S3 && T4_S3_ACC2 ? NT : 1];
  if constexpr (S3 && T4_S3_ACC2) {
#pragma unroll
    for (int mt = 0; mt < MTT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // chunk-invariant part of the patch staging: global byte offset of this thread's pieces (-1: zero border / outside)
  const char* xg = reinterpret_cast<const char*>(p.x);
  const char* wg = reinterpret_cast<const char*>(p.wp);
  const int npiece = npos * 4;
  int goff[PM];
#pragma unroll
  for (int u = 0; u < PM; ++u) {
    const int i = stid + u * STH;
    goff[u] = -1;
    if (stid >= 0 && i < npiece) {
      const int pos = i >> 2, pc = i & 3;
      const int r = pos / p.PW, c = pos - r * p.PW;
      const int gy = y0 - dil + r, gx = c - dil;
      if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
        goff[u] = (((img * p.H + gy) * p.W + gx) * p.Ci + pc * PCN) * SZ;
    }
  }
  const int nchunk = (p.Ci + CHN - 1) / CHN;
  // (PC: two register sets, so that chunk c + 2 is requested BEFORE chunk c + 1 is split and stored)
  u32x4 pr[PC ? 2 : 1][PM], wr[PC ? 2 : 1][WR];
  typedef std::integral_constant<int, 0> Set0;
  typedef std::integral_constant<int, (PC ? 1 : 0)> Set1;
  auto fetch = [&](int c, auto setc) {
    constexpr int SET = decltype(setc)::value;
#pragma unroll
    for (int u = 0; u < PM; ++u) {
      pr[SET][u] = u32x4{0u, 0u, 0u, 0u};
      const int pc = (stid + u * STH) & 3;
      if (goff[u] >= 0 && c * CHN + pc * PCN < p.Ci) pr[SET][u] = *reinterpret_cast<const u32x4*>(xg + goff[u] + c * 64);
    }
#pragma unroll
    for (int u = 0; u < WR; ++u) {
      const int i = stid + u * STH;
      wr[SET][u] = u32x4{0u, 0u, 0u, 0u};
      if (stid >= 0 && i < WPC) {
        const int blk = i >> 6;                    // blk = tap*NT + nt
        // S3: thread j of a block takes the image's lane (k quarter j & 3, row j >> 2): consecutive threads then store
        // consecutive 8-byte runs of a plane row
        const int l = S3 ? (((i & 3) << 4) | ((i >> 2) & 15)) : (i & 63);
        const int tap = blk / NT, nt = blk - tap * NT;
        if (ntg0 + nt < p.NTt)      // (output-channel tail of the last block: Co = 216 / 108 of the DCN predictors)
          wr[SET][u] = *reinterpret_cast<const u32x4*>(wg + ((long)((tap * p.KC + c) * p.NTt + ntg0 + nt)) * 1024 + l * 16);
      }
    }
  };
  if (!PC || producer) fetch(0, Set0());
  // XBN: per-channel scale / shift of the input's BatchNorm in LDS (behind the weight slab), computed while the first
  // chunk's loads are in flight; workgroup (0, 0) publishes mean / invstd / running statistics
  float* xsc = reinterpret_cast<float*>(wbuf + (S3 ? 3 : 1) * 9 * NT * WBLK + (PC ? p.patch_bytes + 3 * WPL : 0));      // (16-bit and split-product instances; PC: behind the second buffer)
  float* xsf = xsc + p.Ci;
  const bool xon = (SZ == 2 || S3) && p.xb.on;
  if (xon) {
    for (int ch = tid; ch < p.Ci; ch += THREADS) {
      float a, b;
      xbn_channel(p.xb, ch, bxl == 0 && byl == 0, a, b);
      xsc[ch] = a;
      xsf[ch] = b;
    }
    __syncthreads();
  }
#ifdef FAMI_T4_TRACE
  const bool trace = p.dbg && bxl == 300 && byl == 0 && lane == 0;
#define T4_STAMP(k) if (trace) p.dbg[(wave * 32 + c) * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime()
#else
#define T4_STAMP(k)
#endif
  // registers of chunk c -> LDS (patch at `patch`, weights at `wbuf`)
  auto store = [&](int c, char* patch, char* wbuf, auto setc) {
    constexpr int SET = decltype(setc)::value;
#pragma unroll
    for (int u = 0; u < PM; ++u) {
      const int i = stid + u * STH;
      if (i < npiece) {
        u32x4 v = pr[SET][u];
        if constexpr (SZ == 2) {
          const int ch0 = c * CHN + (i & 3) * PCN;      // border / outside pieces stay zero: the conv pads the NORMALISED tensor
          if (xon && goff[u] >= 0 && ch0 < p.Ci) v = xbn_piece<H>(v, xsc + ch0, xsf + ch0);
        } else if constexpr (S3) {
          const int ch0 = c * CHN + (i & 3) * PCN;
          if (xon && goff[u] >= 0 && ch0 < p.Ci) {      // f32 storage: exactly what the apply pass would have stored
            f32x4 t = __builtin_bit_cast(f32x4, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = fmaxf(__builtin_fmaf(t[j], xsc[ch0 + j], xsf[ch0 + j]), 0.f);
            v = __builtin_bit_cast(u32x4, t);
          }
        }
        if constexpr (S3) t4_split_store(patch + (i >> 2) * T4_S3_ROW + (i & 3) * 8, ppl, t4_split(v));
        else *reinterpret_cast<u32x4*>(patch + (i >> 2) * p.PS + (i & 3) * 16) = v;
      }
    }
#pragma unroll
    for (int u = 0; u < WR; ++u) {
      const int i = stid + u * STH;
      if (i < WPC) {
        // S3: row (i >> 2) & 15 of block i >> 6, channels (i & 3) * 4 .. + 3 of each plane (see fetch)
        if constexpr (S3) t4_split_store(wbuf + (i >> 2) * T4_S3_ROW + (i & 3) * 8, WPL, t4_split(wr[SET][u]));
        else *reinterpret_cast<u32x4*>(wbuf + i * 16) = wr[SET][u];
      }
    }
  };
  constexpr int BUFSZ_W = 3 * WPL;                 // PC: a buffer = patch planes + weight planes
  if constexpr (PC) {
    if (producer) {
      store(0, smem, smem + p.patch_bytes, Set0());
      if (nchunk > 1) fetch(1, Set1());
    }
    __syncthreads();
    if (producer) {
      // chunk c + 1 -> buffer (c + 1) & 1 while the multiplying waves are on chunk c; sets alternate: chunk k lives in set k & 1
      const int bsz = p.patch_bytes + BUFSZ_W;
      for (int c = 0; c < nchunk; c += 2) {
        if (c + 2 < nchunk) fetch(c + 2, Set0());
        if (c + 1 < nchunk) store(c + 1, smem + bsz, smem + bsz + p.patch_bytes, Set1());
        __syncthreads();
        if (c + 1 >= nchunk) break;
        if (c + 3 < nchunk) fetch(c + 3, Set1());
        if (c + 2 < nchunk) store(c + 2, smem, smem + p.patch_bytes, Set0());
        __syncthreads();
      }
    }
  }
  for (int c = 0; c < nchunk; ++c) {
    if constexpr (PC) {
      if (producer) break;      // (its loop is above; same number of barriers)
    } else {
      T4_STAMP(0);
      if (c > 0) __syncthreads();   // the previous chunk has been multiplied by every wave
      T4_STAMP(1);
      store(c, patch, wbuf, Set0());
      T4_STAMP(2);
      __syncthreads();
      T4_STAMP(3);
      if (c + 1 < nchunk) fetch(c + 1, Set0());   // in flight while this chunk is multiplied
    }
    char* const cbuf = PC ? smem + (c & 1) * (p.patch_bytes + BUFSZ_W) : smem;
    const char* const patch = cbuf;
    const char* const wbuf = cbuf + p.patch_bytes;
    // the nine taps of this chunk for a wave with MW live pixel tiles (compile-time: a per-tile "is it live" branch cut the
    // loop into 3-MFMA blocks, each behind its own LDS wait -- now a tap's fragments are requested while the previous
    // tap's MFMAs issue)
    auto taps = [&](auto mwc) {
      constexpr int MW = decltype(mwc)::value;
      constexpr bool DB = MTT <= 2;                    // two fragment sets (a tap ahead) only where the registers allow
      bf16x8 a3[S3 && DB ? 2 : 1][MW][3], w3[S3 && DB ? 2 : 1][NT][2];
      auto s3_load_w = [&](int tap, bf16x8 (&w)[NT][2]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const char* wb = wbuf + (tap * NT + nt) * WBLK;
          w[nt][0] = *reinterpret_cast<const bf16x8*>(wb + wo0);
          w[nt][1] = *reinterpret_cast<const bf16x8*>(wb + wo1);
        }
      };
      auto s3_load_a = [&](int tap, bf16x8 (&a)[MW][3]) {
        const int toff = p.sgn * dil * ((tap / 3 - 1) * p.PW + (tap % 3 - 1)) * p.PS;
#pragma unroll
        for (int mt = 0; mt < MW; ++mt) {
          const char* pb = patch + base[mt] + toff;
          a[mt][0] = *reinterpret_cast<const bf16x8*>(pb + xo0);
          a[mt][1] = *reinterpret_cast<const bf16x8*>(pb + xo1);
          a[mt][2] = *reinterpret_cast<const bf16x8*>(pb + xo2);
        }
      };
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int toff = p.sgn * dil * ((tap / 3 - 1) * p.PW + (tap % 3 - 1)) * p.PS;
        if constexpr (S3) {
          // fragments of tap t + 1 are requested before tap t is multiplied (two register sets): left to itself the
          // scheduler issued a tap's twelve LDS reads one to three MFMAs ahead of their use and the wave sat in
          // s_waitcnt lgkmcnt(0..1) before most MFMAs (2 waves per SIMD cannot cover that)
          if (tap == 0 || !DB) {
            s3_load_w(tap, w3[0]);
            s3_load_a(tap, a3[0]);
          }
          if (DB && tap < 8) {
            s3_load_w(tap + 1, w3[DB ? (tap + 1) & 1 : 0]);
            s3_load_a(tap + 1, a3[DB ? (tap + 1) & 1 : 0]);
          }
          const int cur = DB ? tap & 1 : 0;
#pragma unroll
          for (int m = 2; m >= 0; --m)          // low-order products first
#pragma unroll
            for (int mt = 0; mt < MW; ++mt)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
              {
                f32x4& dst = (T4_S3_ACC2 && m > 0) ? acc2[T4_S3_ACC2 ? mt : 0][T4_S3_ACC2 ? nt : 0] : acc[mt][nt];
                dst = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3[cur][nt][m == 2 ? 1 : 0], a3[cur][mt][m], dst, 0, 0, 0);
              }
          if (DB && tap < 8) {
            // one LDS read behind each of the first MFMAs, the rest of the MFMAs after them
#pragma unroll
            for (int k = 0; k < 3 * MW + 2 * NT; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * MW * NT - (3 * MW + 2 * NT), 0);
          }
          continue;
        }
        frag a[MW], w[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) w[nt] = *reinterpret_cast<const frag*>(wbuf + (tap * NT + nt) * 1024 + lane * 16);
#pragma unroll
        for (int mt = 0; mt < MW; ++mt) a[mt] = *reinterpret_cast<const frag*>(patch + base[mt] + toff);
#pragma unroll
        for (int mt = 0; mt < MW; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = T4Traits<H>::mma(w[nt], a[mt], acc[mt][nt]);
      }
    };
    if (mtw >= MTT) taps(std::integral_constant<int, MTT>());
    else if (MTT >= 4 && mtw == 3) taps(std::integral_constant<int, (MTT >= 4 ? 3 : 1)>());
    else if (MTT >= 3 && mtw == 2) taps(std::integral_constant<int, (MTT >= 3 ? 2 : 1)>());
    else if (mtw == 1) taps(std::integral_constant<int, 1>());
    T4_STAMP(4);
    if constexpr (PC) __syncthreads();   // chunk c is multiplied, chunk c + 1 is in the other buffer
  }

  if constexpr (S3 && T4_S3_ACC2) {
#pragma unroll
    for (int mt = 0; mt < MTT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] += acc2[mt][nt];
  }
  // ---- epilogue: D row = kq*4 + r (output channel), col = lane&15 (pixel)
  const long pix0 = (long)img * HW + p0;
  const int emode = p.emode;
  if (emode) {
    // EpiBN (see conv_igemm_h): channel tile by channel tile; the wave's tiles in registers, the 16 pixel lanes by DPP,
    // the four waves through the (now idle) LDS
    EpiPtr e = epi_late(__builtin_offsetof(ConvT4Args, e));
    __syncthreads();                                    // every wave is done with the last chunk's LDS
    float* ered = reinterpret_cast<float*>(smem);       // [waves][NT*32]
    const H* ez = reinterpret_cast<const H*>(e->z);
    const H* eyr = reinterpret_cast<const H*>(e->yr);
    const int erelu = e->relu, eC = e->C;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co0 = (ntg0 + nt) * 16 + kq * 4;
      f32x4 es = z4, eq = z4, ek = z4, emu = z4, eis = z4, esc = z4, esf = z4, bias4 = z4;
      if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + co0);
      if (emode == 1) {
        if (e->pivot_src) ek = *reinterpret_cast<const f32x4*>(e->pivot_src + co0);
      } else {
        emu = *reinterpret_cast<const f32x4*>(e->mean + co0);
        eis = *reinterpret_cast<const f32x4*>(e->invstd + co0);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(e->gamma + co0), be = *reinterpret_cast<const f32x4*>(e->beta + co0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a, b;
          epi_scale_shift(emu[r], eis[r], ga[r], be[r], a, b);
          esc[r] = a;
          esf[r] = b;
        }
      }
#pragma unroll
      for (int mt = 0; mt < MTT; ++mt) {
        const int j = (wave + CW * mt) * 16 + col;
        if (mt >= mtw || p0 + j >= p1) continue;
        f32x4 v = acc[mt][nt] + bias4;
        const long idx = (pix0 + j) * p.Co + co0;
        H* yp = reinterpret_cast<H*>(p.y) + idx;
        if (p.accumulate) v += ld4(yp);
        if (emode == 2) {
          const f32x4 zz = ld4(ez + idx);
          f32x4 yy = z4;
          if (erelu == 1) yy = ld4(eyr + idx);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            bool keep = true;
            if (erelu == 1) keep = yy[r] > 0.f;
            else if (erelu == 2) keep = __builtin_fmaf(zz[r], esc[r], esf[r]) > 0.f;
            v[r] = keep ? v[r] : 0.f;
          }
          st4(yp, v);
          const f32x4 g = ld4_round<H>(v);
          es += g;
          eq += g * ((zz - emu) * eis);
        } else {
          st4(yp, v);
          const f32x4 d = ld4_round<H>(v) - ek;
          es += d;
          eq += d * d;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        es[r] = row16_sum(es[r]);
        eq[r] = row16_sum(eq[r]);
      }
      if (col == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ered[wave * (NT * 32) + nt * 32 + kq * 4 + r] = es[r];
          ered[wave * (NT * 32) + nt * 32 + 16 + kq * 4 + r] = eq[r];
        }
      }
    }
    __syncthreads();
    if (tid < NT * 32) {
      const int nt = tid >> 5, st = (tid >> 4) & 1, c16 = tid & 15;
      const int co = (ntg0 + nt) * 16 + c16;
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < WV; ++wv) v += ered[wv * (NT * 32) + tid];
      double* srow = e->slots + (long)(bxl % e->ns) * 2 * eC;
      unsafeAtomicAdd(srow + st * eC + co, (double)v);
      if (emode == 1 && st == 0 && bxl == 0) bn_slots_pivot(e->slots, eC)[co] = e->pivot_src ? e->pivot_src[co] : 0.f;
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MTT; ++mt) {
    const int j = (wave + CW * mt) * 16 + col;
    if (mt >= mtw || p0 + j >= p1) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co0 = (ntg0 + nt) * 16 + kq * 4;
      if (co0 >= p.Co) continue;        // channel tail (Co % 4 == 0)
      f32x4 v = acc[mt][nt];
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + co0);
      if (p.relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      const long idx = (pix0 + j) * p.Co + co0;
      if (p.out_f32 || SZ == 4) {
        float* yp = reinterpret_cast<float*>(p.y) + idx;
        if (p.accumulate) v += ld4(yp);
        st4(yp, v);
      } else {
        H* yp = reinterpret_cast<H*>(p.y) + idx;
        if (p.accumulate) v += ld4(yp);
        st4(yp, v);
      }
    }
  }
}

// ---- register-blocked LDS kernel (16-bit): plan + launch.  Returns 1 if launched, 0 if not eligible, <0 on error.
// [fami_route_t] g_use_t4 (default 1)
// [fami_route_t] g_t4_dil (default 1)  // fami_conv_tune_lds(40 / 41): dilated 3x3 convolutions on the band kernels off / on
// [fami_route_t] g_use_t4_f32 (default 0)  // fami_conv_tune_lds(20 / 21): the f32 instance off / on.  Off by default: per launch it wins where
                               // the launch fills the chip (below), inside the f32 step it does not (61.2 vs 61.7 ms, and 61.4 vs 60.9
                               // when every shape takes it: noise) -- the exact-f32 MFMA step is bound by the matrix pipe itself   // fami_conv_tune_lds(10 / 11): off / on (default on for every eligible 16-bit 3x3)
// [fami_route_t] g_t4_bt (default 0)  // fami_conv_tune_lds(100 + bt): force the tiles per band (benchmarks)
static long long* g_t4_dbg = nullptr;   // fami_conv_t4_debug (FAMI_T4_TRACE builds)
// [fami_route_t] g_t4_s3_narrow (default 0)  // fami_conv_tune_lds(102030 / 102031): two channel tiles per workgroup on launches of < 200 workgroups off / on.
                                    // Per launch 55 -> 43 us (24x18 @192 ch) and 70 -> 57 (12x9 @384 ch); f32 step 49.0 -> 49.7 and 48.9 -> 49.6 ms: off.
// [fami_route_t] g_t4_s3_fill (default 2)  // fami_conv_tune_lds(102000 / 102001 / 102002): more, smaller bands on launches that leave CUs empty: off / all / tiny ones.
                                    // Per launch it wins (24x18 @192 ch 55 -> 45 us, 48x36 @96 ch 40.6 -> 35.6); inside the step other lanes
                                    // already fill those CUs and the smaller bands only add staging: 50.8 -> 51.4 and 49.5 -> 50.4 ms.  2 = only launches of
                                    // < 128 workgroups: the head's 4-frame convolutions (72 workgroups), which the trace shows running alone.
// [fami_route_t] g_t4_s3_pc (default 0)  // fami_conv_tune_lds(60 / 61): producer / consumer form of the split-product instance off / on
// [fami_route_t] g_t4_s3_mt_minft (default 64)  // ... only for frames of at least this many tiles (24x18 maps: 27 tiles = one band of 24 + one of 3)
// [fami_route_t] g_t4_s3_mt (default 3)  // fami_conv_tune_lds(52 / 53): pixel tiles per wave of the split-product instance.  3 (bands of <= 24 tiles, 15 LDS
                               // fragment reads per 27 MFMAs instead of 12 per 18; 256 VGPRs, 12-64 bytes of scratch): per launch 48 ch @96x72
                               // 54 -> 51.5 us, 96 ch @48x36 56 -> 40.6, but 192 ch @24x18 55 -> 71 (hence the frame-size rule); f32 step
                               // 53.5 -> 52.3 ms.  4 tiles per wave (and 3 with 64-wide channel blocks) spill hundreds of bytes: not built.
// [fami_route_t] g_t4_s3_minwg (default 0)  // fami_conv_tune_lds(2000 + n): the split-product instance only for launches of >= n workgroups (benchmarks)
// [fami_route_t] g_s3_default (default 1)  // fami_tune_defaults: what fami_conv_tune_lds(-1) restores (FAMI_F32_SPLIT=0 -> 0)
// [fami_route_t] g_use_t4_s3 (default 1)  // fami_conv_tune_lds(30 / 31): f32 storage on the bf16 matrix pipe (split products, see the kernel) off / on

// ---- the split-product f32 instance: plan + launch
static int try_conv3x3_t4_s3(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci, int Co,
                             int KC, int NTt, int sgn, int relu, int accumulate, hipStream_t s, const char* name,
                             const EpiBN& epi, const XBN& xbn, int dil = 1) {
  if (!g_use_t4 || !g_use_t4_s3 || (Ci % 4) != 0 || (x && (reinterpret_cast<uintptr_t>(x) & 15) != 0)) return 0;
  int NT = Co % 48 == 0 ? 3 : (Co % 64 == 0 ? 4 : 0);
  // dilated form (round 4: the DCN offset / mask predictors, 48 -> 216 / 108 and their input gradients): the last 48-wide
  // channel block may be partial (Co % 4 == 0); no BatchNorm hooks on that route
  // Measured (tools/bench_dil.py, B = 4 frames of 96x72): 48 -> 216 forward 54 vs 69 us for the exact-f32 implicit GEMM; 48 -> 108
  // 47 vs 40 and the input gradients 96 vs 61 / 48 vs 35 (one channel block = 96 workgroups, each restaging eleven patch rows
  // for four output rows): only wide forward launches take it (g_t4_dil = 2 forces every eligible one: tests)
  if (dil > 1) {
    if (!g_t4_dil || (Co % 4) != 0 || epi.slots || xbn.on) return 0;
    if (g_t4_dil == 1 && (sgn < 0 || Co < 160)) return 0;
    NT = 3;
  }
  if (!NT) return 0;
  const int HW = H * W, FT = (HW + 15) / 16;
  // The low-resolution branches: 2 bands x 20 frames x 4 channel blocks (24x18 @192 ch) or 1 x 20 x 8 (12x9 @384 ch) = 160
  // workgroups of 12 / 24 chunks each, and in the step's trace these launches run ALONE (the other lanes are waiting at the
  // module's fuse for exactly them) for 3.8 ms per f32 step on 160 of 256 CUs.  Two channel tiles per workgroup instead
  // of three: 240 workgroups, a third less MFMA and weight staging each -- more work in total, less on the critical path:
  // that was the idea; per launch it delivers, the step gets slower (see g_t4_s3_narrow), so it is off.
  if (g_t4_s3_narrow && NT == 3 && Co % 32 == 0 && (long)N * ((FT + 15) / 16) * (Co / 48) < 200 &&
      (long)N * ((FT + 15) / 16) * (Co / 32) <= 256)
    NT = 2;
  const int cblocks = (Co + 16 * NT - 1) / (16 * NT);
  auto positions = [&](int bt) { return (long)((bt * 16 + W - 2) / W + 1 + 2 * dil) * (W + 2 * dil); };
  const size_t wbytes = (size_t)3 * 9 * NT * 16 * T4_S3_ROW + (xbn.on ? (size_t)2 * Ci * sizeof(float) : 0);
  const int WVs = 8;
  const size_t lds_cap = 160 * 1024;
  // producer / consumer form (see the kernel): 4 multiplying + 4 staging waves, bands of <= 8 tiles, two LDS buffers
  if (g_t4_s3_pc && NT == 3 && dil == 1 && g_t4_bt == 0 && (g_t4_s3_pc == 2 || FT <= 8)) {   // 1: only frames of one band (12x9 maps), 2: always
    int BTp = 0;
    for (int bt = 8; bt >= 1 && !BTp; --bt)
      if (positions(bt) * 4 <= 6 * 256 && 2 * ((size_t)positions(bt) * 3 * T4_S3_ROW + (wbytes - (xbn.on ? (size_t)2 * Ci * sizeof(float) : 0))) + (xbn.on ? (size_t)2 * Ci * sizeof(float) : 0) <= lds_cap) BTp = bt;
    if (BTp) {
      if (BTp > FT) BTp = FT;
      if (!x) return 1;
      ConvT4Args a;
      a.e = epi; a.emode = epi.slots ? epi.mode : 0; a.xb = xbn;
      a.x = x; a.wp = wp; a.y = y; a.bias = bias;
      a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.BT = BTp; a.bands = (FT + BTp - 1) / BTp;
      a.PW = W + 2; a.PS = T4_S3_ROW; a.KC = KC; a.NTt = NTt; a.sgn = sgn; a.relu = relu; a.accumulate = accumulate; a.out_f32 = 1;
      a.dil = 1;
      const long npos = positions(BTp);
      a.patch_bytes = (int)(npos * 3 * a.PS);
      a.dbg = g_t4_dbg;
      const size_t lds = 2 * ((size_t)a.patch_bytes + (size_t)3 * 9 * NT * 16 * T4_S3_ROW) + (xbn.on ? (size_t)2 * Ci * sizeof(float) : 0);
      const dim3 grid(N * a.bands, cblocks);
      const int PMp = (int)((npos * 4 + 255) / 256);
      static bool attr4 = false, attr6 = false;
      if (PMp <= 4) {
        if (!attr4) { (void)hipFuncSetAttribute((const void*)conv3x3_t4_kernel<float, 3, 4, true, 8, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap); attr4 = true; }
        hipLaunchKernelGGL((conv3x3_t4_kernel<float, 3, 4, true, 8, 2, true>), grid, dim3(512), lds, s, a);
      } else {
        if (!attr6) { (void)hipFuncSetAttribute((const void*)conv3x3_t4_kernel<float, 3, 6, true, 8, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap); attr6 = true; }
        hipLaunchKernelGGL((conv3x3_t4_kernel<float, 3, 6, true, 8, 2, true>), grid, dim3(512), lds, s, a);
      }
      hipError_t err = hipGetLastError();
      if (err != hipSuccess) {
        fami_set_error(name, hipGetErrorString(err));
        return FAMI_EHIP;
      }
      return 1;
    }
  }
  // pixel tiles per wave: 2 (bands of <= 16 tiles), or 3 / 4 (<= 24 / 32) where the frame is large enough to use them
  int MTs = NT == 3 ? g_t4_s3_mt : 2;       // (3 tiles x 4 channel tiles spill; 4 x 3 too)
  if (MTs > 3) MTs = 3;
  if (MTs > 2 && FT < g_t4_s3_mt_minft) MTs = 2;
  const long pos_cap = (long)(MTs == 2 ? T4_PMAX : 7) * T4_THREADS / 4;
  int BT = 0;
  for (int bt = 8 * MTs; bt >= 1 && !BT; --bt)
    if (positions(bt) <= pos_cap && (size_t)positions(bt) * 3 * T4_S3_ROW + wbytes <= lds_cap) BT = bt;
  if (!BT) return 0;
  if (g_t4_bt == 0) {
    // bands of equal size; optionally (g_t4_s3_fill) more and smaller bands where the launch would leave CUs without a
    // workgroup (the low-resolution maps), up to one workgroup per CU
    long nb = (FT + BT - 1) / BT;
    const long per = (long)N * cblocks;
    if (dil == 1 && ((g_t4_s3_fill == 1 && nb * per < 256) || (g_t4_s3_fill == 2 && nb * per < 128))) {   // (dilated: the halo rows make small bands pure staging)
      long nb2 = 256 / per;
      if (nb2 > FT) nb2 = FT;
      if (nb2 > nb) nb = nb2;
    }
    BT = (int)((FT + nb - 1) / nb);
  }
  if (g_t4_bt > 0 && g_t4_bt <= BT) BT = g_t4_bt;
  if (BT > FT) BT = FT;
  if (BT <= 16) MTs = 2;
  if (!x) return 1;      // eligibility query (fami_conv_t4_eligible_s3)
  ConvT4Args a;
  a.e = epi; a.emode = epi.slots ? epi.mode : 0; a.xb = xbn;
  a.x = x; a.wp = wp; a.y = y; a.bias = bias;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.BT = BT; a.bands = (FT + BT - 1) / BT;
  a.PW = W + 2 * dil; a.PS = T4_S3_ROW; a.KC = KC; a.NTt = NTt; a.sgn = sgn; a.relu = relu; a.accumulate = accumulate; a.out_f32 = 1;
  a.dil = dil;
  const long npos = positions(BT);
  a.patch_bytes = (int)(npos * 3 * a.PS);     // three planes
  a.dbg = g_t4_dbg;
  size_t lds = (size_t)a.patch_bytes + wbytes;
  const dim3 grid(N * a.bands, cblocks);
  if ((long)N * a.bands * cblocks < g_t4_s3_minwg) return 0;
  const int PM = (int)((npos * 4 + WVs * 64 - 1) / (WVs * 64));
  bool ok = false;
#define FAMI_T4S3_CASE(nt, pm, mt)                                                                                        \
  if (NT == nt && MTs == mt && PM <= pm && !ok) {                                                                         \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv3x3_t4_kernel<float, nt, pm, true, 8, mt>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap); \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((conv3x3_t4_kernel<float, nt, pm, true, 8, mt>), grid, dim3(WVs * 64), lds, s, a);                 \
    ok = true;                                                                                                            \
  }
  FAMI_T4S3_CASE(3, 3, 2) FAMI_T4S3_CASE(3, 5, 2) FAMI_T4S3_CASE(4, 3, 2) FAMI_T4S3_CASE(4, 5, 2) FAMI_T4S3_CASE(2, 3, 2) FAMI_T4S3_CASE(2, 5, 2)
  FAMI_T4S3_CASE(3, 5, 3) FAMI_T4S3_CASE(3, 7, 3)
#undef FAMI_T4S3_CASE
  if (!ok) return 0;
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    fami_set_error(name, hipGetErrorString(err));
    return FAMI_EHIP;
  }
  return 1;
}

template <typename HT>
static int try_conv3x3_t4(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci, int Co,
                          int KC, int NTt, int sgn, int relu, int accumulate, int out_f32, hipStream_t s, const char* name,
                          const EpiBN& epi, const XBN& xbn, int dil = 1) {
  if (xbn.on && (sizeof(HT) != 2 || (Ci % 8) != 0)) return 0;
  if (!g_use_t4 || ((Ci * (int)sizeof(HT)) % 16) != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0) return 0;
  if (sizeof(HT) == 4 && !g_use_t4_f32) return 0;
  int NT = 0;
  if (Co % 48 == 0) NT = 3;
  else if (Co % 64 == 0) NT = 4;
  if (dil > 1) {                       // dilated form: see try_conv3x3_t4_s3
    if (!g_t4_dil || (Co % 4) != 0 || epi.slots || xbn.on || sizeof(HT) == 4) return 0;
    NT = 3;
  }
  if (!NT) return 0;
  const int HW = H * W, FT = (HW + 15) / 16, cblocks = (Co + 16 * NT - 1) / (16 * NT);
  // tiles per band.  Measured (tools/bench_t4.py, profiles/r03_bench_t4.txt): 12 tiles (192 pixels) is the best or equal
  // on every branch shape -- 16.1 / 13.0 / 13.7 us at 48 / 96 / 192 channels against 19-21 us with 6-8 tiles (the weight
  // slab is re-staged per band) and 23 us with 16 at 48 channels (staging registers spill); a frame smaller than that
  // (12x9 maps: 7 tiles) is one band: 19.3 us against 25-28 with 4-6 tiles (waves without a tile idle).  More workgroups
  // than that do not help even where the grid is below one workgroup per CU: other stream lanes fill the rest.
  const int cand[8] = {12, 10, 8, 6, 5, 4, 3, 2};
  auto positions = [&](int bt) { return (long)((bt * 16 + W - 2) / W + 1 + 2 * dil) * (W + 2 * dil); };
  const long pos_cap = (long)(NT == 3 ? 4 : T4_PMAX) * T4_THREADS / 4;   // staging registers (NT = 3: the 128-VGPR build)
  int BT = 0;
  for (int i = 0; i < 8 && !BT; ++i)
    if (positions(cand[i]) <= pos_cap) BT = cand[i];
  if (!BT) return 0;
  if (g_t4_bt > 0) BT = g_t4_bt;
  if (BT > FT) BT = FT;
  if (BT > 16 || positions(BT) > pos_cap) return 0;
  // f32: measured per launch (DT=f32 tools/bench_t4.py, profiles/r03_bench_t4_f32.txt) against the direct implicit GEMM:
  // 48->48 @96x72 57.0 vs 66.9 us, 192->48 199 vs 219, 96->48 equal, but 68.9 vs 57.6 (96 ch @48x36), 68.4 vs 56.1 (192
  // ch), 95 vs 60 (384 ch): with 16-channel chunks the low-resolution branches are 6-24 barrier-separated chunks on
  // fewer workgroups than CUs, and the exact-f32 MFMA has no second pipe to hide the gaps under.  The f32 instance
  // therefore takes only launches that fill the chip (>= 600 workgroups: the 96x72 maps) with 48-wide channel blocks.
  if (sizeof(HT) == 4 && g_t4_bt == 0 && !((long)N * ((FT + BT - 1) / BT) * cblocks >= 600 && NT == 3)) return 0;
  ConvT4Args a;
  a.e = epi; a.emode = epi.slots ? epi.mode : 0; a.xb = xbn;
  a.x = x; a.wp = wp; a.y = y; a.bias = bias;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.BT = BT; a.bands = (FT + BT - 1) / BT;
  a.PW = W + 2 * dil; a.PS = 80; a.KC = KC; a.NTt = NTt; a.sgn = sgn; a.relu = relu; a.accumulate = accumulate; a.out_f32 = out_f32;
  a.dil = dil;
  const long npos = positions(BT);
  a.patch_bytes = (int)(npos * a.PS);
  a.dbg = g_t4_dbg;
  const size_t wbytes = (size_t)9 * NT * 1024;
  size_t lds = (size_t)a.patch_bytes + wbytes + (xbn.on ? (size_t)2 * Ci * sizeof(float) : 0);
  if (lds < (size_t)T4_WAVES * NT * 32 * 4) lds = (size_t)T4_WAVES * NT * 32 * 4;
  if (lds > 100 * 1024) return 0;
  const dim3 grid(N * a.bands, cblocks);
  const int PM = (int)((npos * 4 + T4_THREADS - 1) / T4_THREADS);
  bool ok = false;
#define FAMI_T4_CASE(nt, pm)                                                                                              \
  if (NT == nt && PM <= pm && !ok) {                                                                                      \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv3x3_t4_kernel<HT, nt, pm>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((conv3x3_t4_kernel<HT, nt, pm>), grid, dim3(T4_THREADS), lds, s, a);                               \
    ok = true;                                                                                                            \
  }
  FAMI_T4_CASE(3, 3) FAMI_T4_CASE(3, 4) FAMI_T4_CASE(3, 5) FAMI_T4_CASE(4, 3) FAMI_T4_CASE(4, 4) FAMI_T4_CASE(4, 5)
#undef FAMI_T4_CASE
  if (!ok) return 0;
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    fami_set_error(name, hipGetErrorString(err));
    return FAMI_EHIP;
  }
  return 1;
}


// the dilated 3x3 stride-1 convolutions (padding = dilation): plain band kernels, no BatchNorm hooks
int fami_try_conv3x3_t4_dil(int half_kind, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci,
                            int Co, int KC, int NTt, int sgn, int relu, int accumulate, int out_f32, int dil, hipStream_t s,
                            const char* name) {
  const EpiBN epi = epi_none();
  const XBN xbn = xbn_none();
  if (half_kind == 2)
    return try_conv3x3_t4_s3(x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, s, name, epi, xbn, dil);
  if (half_kind == 1)
    return try_conv3x3_t4<f16_t>(x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, out_f32, s, name, epi, xbn, dil);
  return try_conv3x3_t4<bf16_t>(x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, out_f32, s, name, epi, xbn, dil);
}
int fami_try_conv3x3_t4(int half_kind, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci,
                        int Co, int KC, int NTt, int sgn, int relu, int accumulate, int out_f32, hipStream_t s,
                        const char* name, const EpiBN& epi, const XBN& xbn) {
  if (half_kind == 2 && g_use_t4 && g_use_t4_s3) {
    // round 4: the persistent, unit-pipelined form (conv_t5.hip) wherever it is eligible; bitwise the same results
    const int rc5 = fami_try_conv3x3_t5(2, x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, out_f32, s, name, epi, xbn);
    if (rc5 != 0) return rc5;
  }
  if (half_kind != 2 && g_use_t4) {
    // round 4: the weight-resident DMA-staged form (conv_t6.hip) where the whole weight image fits the LDS (48 input channels)
    const int rc6 = fami_try_conv3x3_t6(half_kind, x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, out_f32, s, name, epi, xbn);
    if (rc6 != 0) return rc6;
    const int rc5 = fami_try_conv3x3_t5(half_kind, x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, out_f32, s, name, epi, xbn);
    if (rc5 != 0) return rc5;
  }
  if (half_kind == 2) {
    const int rc = try_conv3x3_t4_s3(x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, s, name, epi, xbn);
    if (rc != 0) return rc;
    if (xbn.on) return 0;
  }
  if (half_kind == 2)
    return try_conv3x3_t4<float>(x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, out_f32, s, name, epi, xbn);
  if (half_kind == 1)
    return try_conv3x3_t4<f16_t>(x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, out_f32, s, name, epi, xbn);
  return try_conv3x3_t4<bf16_t>(x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, out_f32, s, name, epi, xbn);
}
// would a 16-bit 3x3 stride-1 pad-1 convolution [N,H,W,Ci] -> Co take this kernel (XBN callers ask before they decide not
// to materialise the normalised input)?
int fami_conv_t4_eligible16(int N, int H, int W, int Ci, int Co) {
  if (!g_use_t4 || (Ci % 8) != 0) return 0;
  if (fami_conv_t6_eligible(N, H, W, Ci, Co)) return 0;      // conv_t6.hip copies its patch by DMA: no transform on the way (its launch + the apply pass beat this kernel with the transform)
  const int NT = Co % 48 == 0 ? 3 : (Co % 64 == 0 ? 4 : 0);
  if (!NT) return 0;
  const int FT = (H * W + 15) / 16;
  auto positions = [&](int bt) { return (long)((bt * 16 + W - 2) / W + 3) * (W + 2); };
  const long pos_cap = (long)(NT == 3 ? 4 : T4_PMAX) * T4_THREADS / 4;
  const int cand[8] = {12, 10, 8, 6, 5, 4, 3, 2};
  int BT = 0;
  for (int i = 0; i < 8 && !BT; ++i)
    if (positions(cand[i]) <= pos_cap) BT = cand[i];
  if (!BT) return 0;
  if (g_t4_bt > 0) BT = g_t4_bt;
  if (BT > FT) BT = FT;
  if (BT > 16 || positions(BT) > pos_cap) return 0;
  const size_t lds = (size_t)positions(BT) * 80 + (size_t)9 * NT * 1024 + (size_t)2 * Ci * sizeof(float);
  return lds <= 100 * 1024 ? 1 : 0;
}
// ... and the same question for f32 storage (the split-product instance)
int fami_conv_t4_eligible_s3(int N, int H, int W, int Ci, int Co) {
  XBN xb = xbn_none();
  xb.on = 1;
  return try_conv3x3_t4_s3(nullptr, nullptr, nullptr, nullptr, N, H, W, Ci, Co, 0, 0, 1, 0, 0, nullptr, "", epi_none(), xb);
}
extern "C" void fami_conv_t4_debug(void* buf) { g_t4_dbg = reinterpret_cast<long long*>(buf); }
void fami_conv_t4_default_split(int on) { g_s3_default = on ? 1 : 0; }
void fami_conv_t4_tune(int on) {
  if (on < 0 || (on >= 8000 && on < 9000)) {   // conv_t6.hip: 8000 / 8001 off / on, 8100 + rows per band, 8400 + minimum jobs
    fami_conv_t6_tune(on);
    if (on >= 0) return;
  }
  if (on < 0 || (on >= 7000 && on < 8000)) {   // conv_t5.hip: 7000 / 7001 off / on, 7100 + rows per band, 7400 + minimum frame tiles, 7500 + workgroups
    fami_conv_t5_tune(on);
    if (on >= 0) return;
  }
  if (on < 0) { g_use_t4 = 1; g_t4_dil = 1; g_t4_bt = 0; g_use_t4_f32 = 0; g_use_t4_s3 = g_s3_default; g_t4_s3_minwg = 0; g_t4_s3_mt = 3; g_t4_s3_pc = 0; g_t4_s3_fill = 2; g_t4_s3_narrow = 0; }
  else if (on == 30 || on == 31) g_use_t4_s3 = on - 30;
  else if (on >= 40 && on <= 42) g_t4_dil = on - 40;
  else if (on == 102030 || on == 102031) g_t4_s3_narrow = on - 102030;
  else if (on >= 102000) g_t4_s3_fill = on - 102000;
  else if (on >= 2000) g_t4_s3_minwg = on - 2000;
  else if (on >= 52 && on <= 53) g_t4_s3_mt = on - 50;
  else if (on >= 60 && on <= 62) g_t4_s3_pc = on - 60;
  else if (on == 10 || on == 11) g_use_t4 = on - 10;
  else if (on == 20 || on == 21) g_use_t4_f32 = on - 20;
  else if (on >= 100) g_t4_bt = on - 100;
}
