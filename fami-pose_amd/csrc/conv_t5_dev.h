// Device code of conv_t5.hip (the persistent, unit-pipelined 3x3 kernel), shared with conv_pair.hip.
#pragma once
#include "conv_epi.h"
#include <type_traits>
typedef __bf16 t5_bf16x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ the kernel
struct ConvT5Args {
  EpiBN e;
  int emode;
  XBN xb;
  const void* x;      // [N,H,W,Ci]
  const void* wimg;   // S3: the split image; 16-bit: the packed fragment image [tap][KC][NTt][64][8]
  void* y;            // [N,H,W,Co]
  const float* bias;
  int N, H, W, Ci, Co;
  int R, bands, cblocks, njobs;   // output rows per band, bands per frame, output-channel blocks, N * bands * cblocks
  int PW, KC, NTt;
  int sgn, relu, accumulate, out_f32;
  int ppl;            // S3: bytes of one patch plane
  int patch_bytes;    // bytes of one patch buffer
  int nposmax;        // positions of a full band's patch ((R + 2) * PW)
  int abl_chunks;     // benchmarks (fami_conv_tune_lds(7700 + n)): walk only the first n channel chunks (WRONG results: an upper-bound experiment)
  long long* dbg;     // FAMI_T5_TRACE builds: s_memtime stamps of workgroup 9 (null otherwise)
};

#define T5_THREADS 512
#define T5_WAVES 8
#define T5_MTT 3      // pixel tiles per wave at most: bands of <= 18 tiles (waves 0, 1: three; the others two)
#ifndef T5_STAGGER
#define T5_STAGGER 1
#endif
#ifndef T5_FRAGPIPE
#define T5_FRAGPIPE 0   // 1: explicit fragment pipeline over (tap, tile) steps (measured: no gain, spills)
#endif
#ifndef T5_SBON
#define T5_SBON 1
#endif
#if T5_SBON
#define T5_SB __builtin_amdgcn_sched_barrier(0)
#else
#define T5_SB
#endif
#ifndef T5_WDB
#define T5_WDB 0     // weight fragments of the next tap in their own registers (1) or reloaded at the tap's first step (0)
#endif
#define T5_PM 4       // 16-byte patch pieces per thread and chunk (<= 512 positions)

template <typename H, bool S3> struct T5Frag { typedef typename H16<H>::x8 type; };
template <> struct T5Frag<float, true> { typedef bf16x8 type; };

// (the body is a device function of the block coordinates so that conv_pair.hip can run it beside a weight-gradient body in one launch)
template <typename H, int NT, bool S3>
__device__ __forceinline__ void conv3x3_t5_body(const ConvT5Args& p, const int bx, const int gx) {
  static_assert(S3 == (sizeof(H) == 4), "f32 storage runs the split-product form, 16-bit storage the plain one");
  typedef typename T5Frag<H, S3>::type frag;
  constexpr int SZ = (int)sizeof(H), CHN = 64 / SZ, PCN = 16 / SZ;   // a chunk is 64 bytes of a pixel: 16 f32 / 32 16-bit channels
  // Units: the split-product form cuts a chunk's nine taps into two units (taps 0-4, 5-8) with a weight region each
  // (41.5 KB together: region r is refilled by DMA while region 1 - r is multiplied); the 16-bit types take a whole
  // chunk per unit with two 27 KB regions.  (Three units of a tap row each, the first build, spent a third of every
  // unit at the barrier and in the DMA issue: s_memtime trace, tools/trace_t5.py.)
  constexpr int TRG = S3 ? 2 : 1;                                    // units per chunk
  constexpr int TP0 = S3 ? 5 : 9;                                    // taps of unit 0 (unit 1: the rest)
  constexpr int BLK = S3 ? 1536 : 1024;                              // bytes of one (tap, N tile) weight block
  constexpr int SLAB = TP0 * NT * BLK;                               // the larger unit's weights = offset of the second region
  constexpr int WBYTES = S3 ? 9 * NT * BLK : 2 * SLAB;               // both regions
  constexpr int NPIECE = (SLAB + 1023) / 1024, WPW = (NPIECE + T5_WAVES - 1) / T5_WAVES;
  constexpr int PS = S3 ? 32 : 80;                                   // LDS bytes per patch position (S3: per plane)
  constexpr int PST = S3 ? 1 : 0;                                    // the unit of a chunk in which the next chunk's patch is stored
  constexpr int MTT = T5_MTT, PM = T5_PM;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wbase = smem + 2 * p.patch_bytes;
  float* const ered = reinterpret_cast<float*>(wbase + WBYTES);   // [waves][NT * 32]: the EpiBN epilogue's exchange
  float* const xsc = ered + T5_WAVES * NT * 32;
  float* const xsf = xsc + p.Ci;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4;

  // ---- this workgroup's jobs: XCD x (= workgroup id % 8, observed placement; only speed depends on it) owns the x-th
  // contiguous eighth of the job list, so bands that share halo rows and the channel blocks of one band meet in one L2
  int job, jstride, jend;
  {
    const int G = gx, wg = bx;
    if ((G & 7) == 0) {
      const int x = wg & 7, q = p.njobs >> 3, r = p.njobs & 7;
      const int xs = x * q + min(x, r);
      job = xs + (wg >> 3);
      jstride = G >> 3;
      jend = xs + q + (x < r ? 1 : 0);
    } else {
      job = wg;
      jstride = G;
      jend = p.njobs;
    }
  }
  if (job >= jend) return;
#ifdef FAMI_T5_TRACE
  const bool trace = p.dbg && bx == 9 && lane == 0;
  int tru = 0;
#define T5_STAMP(k) if (trace && tru < 40) p.dbg[(wave * 40 + tru) * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime()
#define T5_NEXT() ++tru
#else
#define T5_STAMP(k)
#define T5_NEXT()
#endif
  const int HW = p.H * p.W;
  const int nchunk = p.abl_chunks > 0 ? min(p.abl_chunks, (p.Ci + CHN - 1) / CHN) : (p.Ci + CHN - 1) / CHN;
  struct Geo { int img, y0, rows, cb; };
  auto geo = [&](int j) {
    Geo g;
    g.cb = j % p.cblocks;
    const int t = j / p.cblocks;
    const int bnd = t % p.bands;
    g.img = t / p.bands;
    g.y0 = bnd * p.R;
    g.rows = min(p.R, p.H - g.y0);
    return g;
  };

  // ---- staging plan of the activation patch (job-invariant): piece i = tid + u * 512 -> position i >> 2, 16-byte piece i & 3
  const char* const xg = reinterpret_cast<const char*>(p.x);
  int prel[PM], prow[PM];
#pragma unroll
  for (int u = 0; u < PM; ++u) {
    const int i = tid + u * T5_THREADS;
    const int pos = i >> 2, pc = i & 3;
    const int r = pos / p.PW, c = pos - r * p.PW;
    const bool ok = pos < p.nposmax && c >= 1 && c <= p.W;          // the border columns are the zero padding
    prow[u] = ok ? r : 0x40000000;
    prel[u] = ((r * p.W + c - 1) * p.Ci + pc * PCN) * SZ;
  }
  u32x4 pr[PM];
  unsigned pvalid = 0;
  auto fetch = [&](const Geo& g, int c) {
    const char* xb = xg + ((long)(g.img * p.H + g.y0 - 1) * p.W) * p.Ci * SZ + c * 64;
#pragma unroll
    for (int u = 0; u < PM; ++u) {
      pr[u] = u32x4{0u, 0u, 0u, 0u};
      const int pc = (tid + u * T5_THREADS) & 3;
      const bool ld = prow[u] < g.rows + 2 && (unsigned)(g.y0 - 1 + prow[u]) < (unsigned)p.H && c * CHN + pc * PCN < p.Ci;
      if (ld) pr[u] = *reinterpret_cast<const u32x4*>(xb + prel[u]);
      pvalid = (pvalid & ~(1u << u)) | ((ld ? 1u : 0u) << u);
    }
  };
  const bool xon = p.xb.on;
  auto store = [&](char* buf, int c) {
#pragma unroll
    for (int u = 0; u < PM; ++u) {
      const int i = tid + u * T5_THREADS;
      if ((i >> 2) < p.nposmax) {
        u32x4 v = pr[u];
        const int ch0 = c * CHN + (i & 3) * PCN;
        if (xon && ((pvalid >> u) & 1u)) {       // border / outside pieces stay zero: the convolution pads the NORMALISED tensor
          if constexpr (S3) {
            f32x4 t = __builtin_bit_cast(f32x4, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = fmaxf(__builtin_fmaf(t[j], xsc[ch0 + j], xsf[ch0 + j]), 0.f);
            v = __builtin_bit_cast(u32x4, t);
          } else {
            v = xbn_piece<H>(v, xsc + ch0, xsf + ch0);
          }
        }
        if constexpr (S3) {
          const f32x4 f = __builtin_bit_cast(f32x4, v);
          const t5_bf16x4 h0 = __builtin_convertvector(f, t5_bf16x4);
          const f32x4 r1 = f - __builtin_convertvector(h0, f32x4);
          const t5_bf16x4 h1 = __builtin_convertvector(r1, t5_bf16x4);
          const f32x4 r2 = r1 - __builtin_convertvector(h1, f32x4);
          const t5_bf16x4 h2 = __builtin_convertvector(r2, t5_bf16x4);
          char* dst = buf + (i >> 2) * PS + (i & 3) * 8;
          *reinterpret_cast<t5_bf16x4*>(dst) = h0;
          *reinterpret_cast<t5_bf16x4*>(dst + p.ppl) = h1;
          *reinterpret_cast<t5_bf16x4*>(dst + 2 * p.ppl) = h2;
        } else {
          *reinterpret_cast<u32x4*>(buf + (i >> 2) * PS + (i & 3) * 16) = v;
        }
      }
    }
  };

  // ---- weight slab DMA plan: this wave copies 1 KiB pieces wave, wave + 8, ... of a slab; lane -> 16 bytes of a
  // (tap, N tile) block of the image (the LDS slab is the blocks of the unit's taps and this job's N tiles back to back)
  const char* const wg = reinterpret_cast<const char*>(p.wimg);
  int wsrc[WPW];
#pragma unroll
  for (int k = 0; k < WPW; ++k) {
    const int b = (wave + T5_WAVES * k) * 1024 + lane * 16;
    const int blk = b / BLK, within = b - blk * BLK;
    const int t3 = blk / NT, nt = blk - t3 * NT;
    wsrc[k] = b < SLAB ? (t3 * p.KC * p.NTt + nt) * BLK + within : -1;
  }
  auto dma_w = [&](int cb, int c, int r, char* slab) {
    const int t0 = r ? TP0 : 0, bytes = (r ? 9 - TP0 : TP0) * NT * BLK;
    const char* src = wg + ((long)(t0 * p.KC + c) * p.NTt + cb * NT) * BLK;
#pragma unroll
    for (int k = 0; k < WPW; ++k) {
      if ((wave + T5_WAVES * k) * 1024 < bytes) {         // wave-uniform
        if (wsrc[k] >= 0 && (wave + T5_WAVES * k) * 1024 + lane * 16 < bytes)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + wsrc[k]),
                                           (__attribute__((address_space(3))) void*)(slab + (wave + T5_WAVES * k) * 1024), 16, 0, 0);
      }
    }
  };

  // ---- XBN: per-channel scale / shift of the input's BatchNorm (one table per workgroup; workgroup 0 publishes the statistics)
  if (xon) {
    for (int ch = tid; ch < p.Ci; ch += T5_THREADS) {
      float a, b;
      xbn_channel(p.xb, ch, bx == 0, a, b);
      xsc[ch] = a;
      xsf[ch] = b;
    }
    __syncthreads();
  }

  // ---- per-lane fragment offsets
  const int s3h = kq >> 1, s3l = (kq & 1) * 16;
  const int xo0 = s3l, xo1 = p.ppl + s3l, xo2 = (s3h ? 2 : 0) * p.ppl + s3l;                       // X(0|0), X(1|1), X(0|2)
  const int wo0 = (s3h ? 1 : 0) * 512 + col * 32 + s3l, wo1 = (s3h ? 0 : 2) * 512 + col * 32 + s3l;   // W(0|1), W(2|0)

  // ---- prologue of the pipeline
  Geo gj = geo(job);
  fetch(gj, 0);
  store(smem, 0);
  // look-ahead state, advanced incrementally (a division per job, not per unit):
  //   the patch sequence -- (fjob, fc) is the chunk in the staging registers, gf its job's geometry
  //   the unit sequence -- (nj, nc, nr) is the next unit whose weights have to be requested, ncb its channel block
  int fjob = job, fc = 0;
  Geo gf = gj;
  auto advance_patch = [&]() {
    if (++fc == nchunk) {
      fc = 0;
      fjob += jstride;
      if (fjob < jend) gf = geo(fjob);
    }
  };
  advance_patch();
  bool pfull = fjob < jend;                   // the staging registers hold chunk (fjob, fc)
  if (pfull) fetch(gf, fc);
  dma_w(gj.cb, 0, 0, wbase);
  int nj = job, nc = 0, nr = 0, ncb = gj.cb;
  auto advance_unit = [&]() {
    if (++nr == TRG) {
      nr = 0;
      if (++nc == nchunk) {
        nc = 0;
        nj += jstride;
        if (nj < jend) ncb = nj % p.cblocks;
      }
    }
  };
  advance_unit();
  int u = 0, sq = 0;                          // unit counter, patch counter (patch buffer sq & 1)

  for (; job < jend; job += jstride) {
    gj = geo(job);
    const int npx = gj.rows * p.W;
    const int ntile = (npx + 15) >> 4;
    int mtw = (ntile - wave + T5_WAVES - 1) / T5_WAVES;     // tiles wave, wave + 8, wave + 16 (wave-uniform)
    mtw = mtw < 0 ? 0 : (mtw > MTT ? MTT : mtw);
    int base[MTT];
#pragma unroll
    for (int mt = 0; mt < MTT; ++mt) {
      const int pp = min((wave + T5_WAVES * mt) * 16 + col, npx - 1);   // lanes past the band re-read its last pixel (never stored)
      const int ry = pp / p.W, rx = pp - ry * p.W;
      base[mt] = ((ry + 1) * p.PW + rx + 1) * PS + (S3 ? 0 : kq * 16);
    }
    f32x4 acc[MTT][NT], acc2[S3 ? MTT : 1][S3 ? NT : 1];
#pragma unroll
    for (int mt = 0; mt < MTT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (S3) acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }

    for (int c = 0; c < nchunk; ++c, ++sq) {
      auto unit = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
        constexpr int TPU = r == 0 ? TP0 : 9 - TP0;        // taps of this unit
        constexpr int T0 = r == 0 ? 0 : TP0;
        // weight region of this unit / of the next one: the two regions alternate (16-bit: by unit parity)
        const int reg = TRG == 1 ? (u & 1) : r, nreg = TRG == 1 ? ((u + 1) & 1) : (r + 1) % TRG;
        T5_STAMP(0);
        __syncthreads();   // this unit's weights have landed (DMA), patch sq is stored, every wave has left unit u - 1
        T5_STAMP(1);
        // The next chunk's patch (in the staging registers since the previous chunk) -> the buffer chunk sq - 1 was read from.
        // Waves w and w + 4 share a SIMD: the lower half stores at the START of the unit, the upper half at its END, so
        // on every SIMD one wave splits and stores (VALU + LDS writes, ~1500 cycles) while the other multiplies.
        // Early form: store BEFORE the weight DMA below is issued -- the registers' loads are older than everything else in
        // flight; behind the DMA the compiler's vmcnt(0) made the store wait for the next unit's weights (3000-3800 cycles).
        const bool late = T5_STAGGER && wave >= T5_WAVES / 2;     // wave-uniform
        if (r == PST && !late && pfull) {
          store(smem + ((sq + 1) & 1) * p.patch_bytes, fc);
          advance_patch();
          pfull = false;
        }
        T5_STAMP(2);
        if (nj < jend) dma_w(ncb, nc, nr, wbase + nreg * SLAB);     // the next unit's weights -> the region unit u - 1 read
        advance_unit();
        if (!pfull && fjob < jend && (late ? r == 0 : r == PST)) {  // ... and request the patch after the one just stored
          fetch(gf, fc);
          pfull = true;
        }
        T5_STAMP(3);
        const char* const patch = smem + (sq & 1) * p.patch_bytes;
        const char* const wslab = wbase + reg * SLAB;
        auto taps = [&](auto mwc) {
          constexpr int MW = decltype(mwc)::value;
          if constexpr (S3 && !T5_FRAGPIPE) {
            // plain form: a tap's fifteen fragments, then its 27 MFMAs.  The s_memtime trace (tools/trace_t5.py) shows the tap
            // loop at ~800 cycles per tap on the SIMD that carries five tiles (720 cycles of MFMA): the two waves of a SIMD
            // cover each other's LDS waits; the explicit fragment pipeline below costs 20+ registers (spills) and gains nothing
#pragma unroll
            for (int t = 0; t < TPU; ++t) {
              const int tap = T0 + t;
              const int toff = p.sgn * ((tap / 3 - 1) * p.PW + (tap % 3 - 1)) * PS;
              frag a3[MW][3], w3[NT][2];
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                const char* wb = wslab + (t * NT + nt) * BLK;
                w3[nt][0] = *reinterpret_cast<const frag*>(wb + wo0);
                w3[nt][1] = *reinterpret_cast<const frag*>(wb + wo1);
              }
#pragma unroll
              for (int mt = 0; mt < MW; ++mt) {
                const char* pb = patch + base[mt] + toff;
                a3[mt][0] = *reinterpret_cast<const frag*>(pb + xo0);
                a3[mt][1] = *reinterpret_cast<const frag*>(pb + xo1);
                a3[mt][2] = *reinterpret_cast<const frag*>(pb + xo2);
              }
#pragma unroll
              for (int m = 2; m >= 0; --m)          // low-order products first (as conv_t4: bitwise the same sums)
#pragma unroll
                for (int mt = 0; mt < MW; ++mt)
#pragma unroll
                  for (int nt = 0; nt < NT; ++nt) {
                    f32x4& dst = m > 0 ? acc2[S3 ? mt : 0][S3 ? nt : 0] : acc[mt][nt];
                    dst = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3[nt][m == 2 ? 1 : 0], a3[mt][m], dst, 0, 0, 0);
                  }
            }
          } else if constexpr (S3) {
            // Software pipeline over STEPS (tap t, pixel tile mt): the fragments of step s + 1 -- three activation
            // fragments of its tile, and at a tap's first step the six weight fragments of the NEXT tap -- are requested
            // before the nine MFMAs of step s issue, in their own registers.  Left to itself the scheduler requested a
            // fragment one to three MFMAs ahead of its use and the wave sat in s_waitcnt lgkmcnt(0) fifteen times per tap
            // (the ISA of the first build).  The per-accumulator order (m = 2, 1 into the low-order accumulator, m = 0
            // into the high-order one; taps ascending) is conv_t4's: bitwise the same sums.
            constexpr int NS = TPU * MW;
            frag a3[2][3], w3[T5_WDB ? 2 : 1][NT][2];
            auto load_w = [&](int t, frag (&w)[NT][2]) {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                const char* wb = wslab + (t * NT + nt) * BLK;
                w[nt][0] = *reinterpret_cast<const frag*>(wb + wo0);
                w[nt][1] = *reinterpret_cast<const frag*>(wb + wo1);
              }
            };
            auto load_a = [&](int t, int mt, frag (&a)[3]) {
              const int tap = T0 + t;
              const int toff = p.sgn * ((tap / 3 - 1) * p.PW + (tap % 3 - 1)) * PS;
              const char* pb = patch + base[mt] + toff;
              a[0] = *reinterpret_cast<const frag*>(pb + xo0);
              a[1] = *reinterpret_cast<const frag*>(pb + xo1);
              a[2] = *reinterpret_cast<const frag*>(pb + xo2);
            };
            load_w(0, w3[0]);
            load_a(0, 0, a3[0]);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
              const int t = st / MW, mt = st % MW;
              if (T5_WDB ? (mt == 0 && t + 1 < TPU) : (mt == 0 && st > 0)) load_w(T5_WDB ? t + 1 : t, w3[T5_WDB ? (t + 1) & 1 : 0]);
              if (st + 1 < NS) load_a((st + 1) / MW, (st + 1) % MW, a3[(st + 1) & 1]);
              T5_SB;
#pragma unroll
              for (int m = 2; m >= 0; --m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                  f32x4& dst = m > 0 ? acc2[S3 ? mt : 0][S3 ? nt : 0] : acc[mt][nt];
                  dst = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3[T5_WDB ? t & 1 : 0][nt][m == 2 ? 1 : 0], a3[st & 1][m], dst, 0, 0, 0);
                }
              T5_SB;
            }
          } else {
#pragma unroll
            for (int t = 0; t < TPU; ++t) {
              const int tap = T0 + t;
              const int toff = p.sgn * ((tap / 3 - 1) * p.PW + (tap % 3 - 1)) * PS;
              frag a[MW], w[NT];
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) w[nt] = *reinterpret_cast<const frag*>(wslab + (t * NT + nt) * BLK + lane * 16);
#pragma unroll
              for (int mt = 0; mt < MW; ++mt) a[mt] = *reinterpret_cast<const frag*>(patch + base[mt] + toff);
#pragma unroll
              for (int mt = 0; mt < MW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = H16<typename std::conditional<S3, bf16_t, H>::type>::mfma(w[nt], a[mt], acc[mt][nt]);
            }
          }
        };
        if (mtw == 3) taps(std::integral_constant<int, 3>());
        else if (mtw == 2) taps(std::integral_constant<int, 2>());
        else if (mtw == 1) taps(std::integral_constant<int, 1>());
        if (r == PST && late && pfull) {
          store(smem + ((sq + 1) & 1) * p.patch_bytes, fc);
          advance_patch();
          pfull = false;
        }
        T5_STAMP(4);
        T5_NEXT();
        ++u;
      };
      unit(std::integral_constant<int, 0>());
      if constexpr (TRG > 1) unit(std::integral_constant<int, 1>());
    }

    if constexpr (S3) {
#pragma unroll
      for (int mt = 0; mt < MTT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] += acc2[mt][nt];
    }
    // ---- epilogue of the job: D row = kq*4 + r (output channel), col = lane & 15 (pixel)
    const long pix0 = (long)gj.img * HW + gj.y0 * p.W;
    const int ntg0 = gj.cb * NT;
    const int emode = p.emode;
    if (emode) {
      // EpiBN (conv_epi.h): the wave's tiles in registers, the 16 pixel lanes by DPP, the eight waves through LDS
      EpiPtr e = epi_late(__builtin_offsetof(ConvT5Args, e));
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
          const int j = (wave + T5_WAVES * mt) * 16 + col;
          if (mt >= mtw || j >= npx) continue;
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
        for (int wv = 0; wv < T5_WAVES; ++wv) v += ered[wv * (NT * 32) + tid];
        const int slot = (gj.img * p.bands + gj.y0 / p.R) % e->ns;
        double* srow = e->slots + (long)slot * 2 * eC;
        unsafeAtomicAdd(srow + st * eC + co, (double)v);
        if (emode == 1 && st == 0 && gj.img == 0 && gj.y0 == 0) bn_slots_pivot(e->slots, eC)[co] = e->pivot_src ? e->pivot_src[co] : 0.f;
      }
      // (the next epilogue writes `ered` at least nine barriers from here)
      continue;
    }
#pragma unroll
    for (int mt = 0; mt < MTT; ++mt) {
      const int j = (wave + T5_WAVES * mt) * 16 + col;
      if (mt >= mtw || j >= npx) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co0 = (ntg0 + nt) * 16 + kq * 4;
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
}

template <typename H, int NT, bool S3>
__global__ __launch_bounds__(T5_THREADS, 2) void conv3x3_t5_kernel(ConvT5Args p) {
  conv3x3_t5_body<H, NT, S3>(p, blockIdx.x, gridDim.x);
}
