// Device code of conv_wg16.hip's DMA-staged weight-gradient kernel ("wg6") and the fragment helpers of that file, shared with conv_pair.hip.
#pragma once
#include "conv_epi.h"
#include <type_traits>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// two transposing reads (4 K-values each, one 64-bit register pair) -> one 8-value MFMA fragment.  Assembled from DWORDS:
// built element by element from the 16-bit lanes the compiler emitted a v_perm / shift-or per half-word -- with 14 reads
// per K step the kernel was instruction-issue bound (SQ_ACTIVE_INST_ANY 45 % of the wave cycles, profiles/r03_pmc_wgrad16)
template <typename X8>
__device__ __forceinline__ X8 frag_of(s16x4 lo, s16x4 hi) {
  const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
  const u32x4 t = {a.x, a.y, b.x, b.y};
  return __builtin_bit_cast(X8, t);
}

#define WG16_THREADS 512
#define WG16_WAVES 8
#define WG16_XSWEEPS 8   // patch positions per thread and run (512 / (2*CIT) positions per sweep: >= 85)

// ------------------------------------------------------------------ DMA-staged form for the 48-channel branch ("wg6", round 4)
// The 48 -> 48 3x3 weight gradient @96x72 is the most numerous weight-gradient launch of the 16-bit step, and tools/abl_wg16.py
// says where its 28 us go: launch + prologue + first staging 6.6, K loop 7.7, restaging 3.8, partial-slab store 4.7, slab reduce
// 7.0 -- the MFMA work is 8 % of it, and 128-256 workgroups each pay prologue, staging through registers and an 83 KB slab.
// Here a workgroup owns a band of RB whole rows of ONE frame (default 16: 120 workgroups, half as many slabs per pixel as
// before at twice the pixels each) and walks it in units of four rows (288 pixels = nine K steps of 32):
//   * the unit's X patch (six rows, zero border columns, 96-byte positions) and its dY rows (contiguous in HBM and in LDS) are
//     copied by LDS DMA (buffer loads: border / out-of-image granules carry an out-of-range offset = zeros) -- no registers,
//     no stash phase, no zeroing; two buffers: unit u + 1 is requested right after the barrier that opens unit u;
//   * one barrier per unit (288 pixels x 81 MFMA tiles), every pixel of a unit is a whole K step (no ragged tail);
//   * fragments, the pixel <-> K-slot map and the (input tile, tap) pairs are conv_wgrad16_kernel's.
// LDS DMA issued from inline assembly: hipcc's wait-count pass puts an s_waitcnt vmcnt(0) in front of the K loop's transposing
// LDS reads while an LDS-DMA builtin is outstanding (it cannot tell the two buffers apart) -- unit u + 1's copy then never
// overlaps unit u's MFMAs (found in the ISA; the kernel ran 7.5 k cycles per unit against 3 k of MFMA).  The asm form is
// invisible to that pass; the kernel waits with its own s_waitcnt vmcnt(0) at the top of every unit.
typedef int wg6_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ wg6_i32x4 wg6_rsrc(const void* base, int bytes) {
  const unsigned long a = (unsigned long)base;
  const wg6_i32x4 r = {(int)(unsigned)a, (int)((a >> 32) & 0xffff), bytes, 0x00020000};
  return r;
}
__device__ __forceinline__ void wg6_dma16(wg6_i32x4 r, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(voff), "s"(r) : "memory");
}
struct Wg6Args {
  const void* x;    // [N,H,W,Ci]
  const void* dy;   // [N,Ho,Wo,Co]
  float* part;      // [G][9][Ci][Co]
  int N, H, W, Ci, Co;
  int st, Ho, Wo;   // stride (1 | 2), output map
  int UR, upf;      // OUTPUT rows per unit, units per frame (Ho / UR)
  int PR;           // patch rows of a unit (st (UR - 1) + 2 dil + 1)
  int dil;          // dilation = padding (1; 3: the DCN predictors of the head, stride 1)
  int M;            // output pixels of a unit (UR * Wo)
  int nunits;       // units per workgroup (consecutive, frame-major)
  int NU;           // units in total (N * upf)
  int coBlocks;
  int PW, RG;       // W + 2, 16-byte granules per patch row (2 CIT PW: the block's channel slice)
  int q512, r512;   // 512 / RG, 512 % RG
  int dyq, dxr;     // 32 / Wo, 32 % Wo
  int XI, YI;       // DMA instructions (1 KiB) of a unit's patch / dY rows
  long long* dbg;   // FAMI_WG6_TRACE builds: s_memtime stamps of one workgroup
};
#define WG6_XJ 7      // most patch DMA instructions per wave and unit (XI <= 56)
#define WG6_YJ 4      // ... dY (YI <= 32)

// KS: K steps of 32 pixels per unit (ceil(M / 32): the dY rows past M are zeros); CIT x COT: 16-channel tiles of the workgroup's
// channel block (3 x 3: the HRNet branches; 4 x 4 / 4 x 3: the 64-channel 3x3 convolutions of stage 1 and the 256 -> 48 transition)
// XJ: most patch DMA instructions per wave and unit (8 for the dilated launches: a unit of two output rows reads 2 + 2 dil patch rows)
// (the body is a device function of the block coordinates so that conv_pair.hip can run it beside an input-gradient body in one launch)
template <typename H, int KS, int CIT, int COT, int XJ = WG6_XJ>
__device__ __forceinline__ void conv_wgrad6_body(const Wg6Args& p, const int bx, const int by, const int gx) {
  typedef typename H16<H>::x8 hx8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TAPS = 9, NPW = (CIT * TAPS + WG16_WAVES - 1) / WG16_WAVES, PS = 32 * CIT, PSY = 32 * COT, GX = 2 * CIT, GY = 2 * COT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kq = lane >> 4;
  const int rsel = l16 >> 2, piece = l16 & 3;
  int job;
  {   // XCD x owns the x-th contiguous eighth of the job list
    const int n = gx, lin = bx;
    const int q = n >> 3, r = n & 7, xc = lin & 7, l = lin >> 3;
    job = xc * q + (xc < r ? xc : r) + l;
  }
  const int cob = by % p.coBlocks, cib = by / p.coBlocks;
  const int u0g = job * p.nunits;                    // first unit (global index) of this workgroup
  const int nunits = min(p.nunits, p.NU - u0g);
  const int W = p.W, PW = p.PW, Wo = p.Wo;
  const int XB = p.XI * 1024, BUFSZ = XB + p.YI * 1024;
#ifdef FAMI_WG6_TRACE
  const bool trace = p.dbg && job == 5 && by == 0 && lane == 0;
  int tslot = 0;
#define WG6_STAMP() if (trace) p.dbg[wave * 64 + tslot++] = (long long)__builtin_amdgcn_s_memtime()
#else
#define WG6_STAMP()
#endif
  WG6_STAMP();

  // ---- DMA plan of this lane (unit-invariant): patch granule -> (patch row, byte offset from the first patch row's pixel 0,
  // channel slice included); dY granule -> byte offset from the unit's first pixel
  const long xfb = (long)p.H * W * p.Ci * 2, yfb = (long)p.Ho * Wo * p.Co * 2;     // one frame
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
  int xrow[XJ], xoff[XJ], yoff[WG6_YJ];
  {
    const int q0 = wave * 64 + lane;
    int r = q0 / p.RG, wi = q0 - r * p.RG;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int pos = wi / GX, c = wi - pos * GX;
      const bool ok = r < p.PR && pos >= p.dil && pos < W + p.dil;      // (dil = pad border positions on either side)
      xrow[j] = ok ? r : 0x40000000;                  // never a valid image row
      xoff[j] = ((r * W + pos - p.dil) * p.Ci + cib * (16 * CIT) + c * 8) * 2;
      r += p.q512;
      wi += p.r512;
      if (wi >= p.RG) {
        wi -= p.RG;
        ++r;
      }
    }
#pragma unroll
    for (int j = 0; j < WG6_YJ; ++j) {
      const int q = (wave + WG16_WAVES * j) * 64 + lane;
      const int pix = q / GY, c = q - pix * GY;
      // (a channel tail -- Co not a multiple of the block -- reads zeros; a granule that straddles Co picks up the next pixel's first
      //  channels: they only reach accumulator columns >= Co, which are not stored)
      yoff[j] = (pix < p.M && cob * (16 * COT) + c * 8 < p.Co) ? (pix * p.Co + cob * (16 * COT) + c * 8) * 2 : (int)0x80000000;
    }
  }
  // a unit's scalars (frame, rows, buffer descriptors) once per unit: the pieces of its copy are issued one per K step, and the
  // divisions behind them were ~40 scalar instructions in front of every K step's MFMAs
  struct UnitGeo { int yt, ui; wg6_i32x4 rx, ry; };
  auto unit_geo = [&](int ug) {
    UnitGeo g;
    const int img = ug / p.upf;
    g.ui = ug - img * p.upf;
    g.yt = p.st * g.ui * p.UR - p.dil;                 // image row of the unit's first patch row
    g.rx = wg6_rsrc(reinterpret_cast<const char*>(p.x) + (long)img * xfb, (int)xfb);
    g.ry = wg6_rsrc(reinterpret_cast<const char*>(p.dy) + (long)img * yfb, (int)yfb);
    return g;
  };
  // piece k of a unit's copy (k < XJ: patch, else dY rows); wave-uniform guards
  auto dma_piece = [&](const UnitGeo& g, unsigned buf, int k) {
    if (k < XJ) {
      const int i = wave + WG16_WAVES * k;             // (wave-uniform)
      if (i < p.XI) {
        unsigned off = (unsigned)(g.yt * W * p.Ci * 2 + xoff[k]);
        if ((unsigned)(g.yt + xrow[k]) >= (unsigned)p.H) off = 0x80000000u;
        wg6_dma16(g.rx, off, buf + i * 1024);
      }
    } else {
      const int i = wave + WG16_WAVES * (k - XJ);
      if (i < p.YI) {
        const int yo = yoff[k - XJ];
        wg6_dma16(g.ry, yo < 0 ? 0x80000000u : (unsigned)(g.ui * p.UR * Wo * p.Co * 2 + yo), buf + XB + i * 1024);
      }
    }
  };
  if (nunits > 0) {
    const UnitGeo g0 = unit_geo(u0g);
#pragma unroll
    for (int k = 0; k < XJ + WG6_YJ; ++k) dma_piece(g0, lds0, k);
  }
  WG6_STAMP();

  // ---- pairs of this wave: q = wave + 8 i -> (ci tile, tap)
  int poff[NPW], ptap[NPW], pci[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int q = wave + WG16_WAVES * i;
    const bool ok = q < CIT * TAPS;
    pci[i] = ok ? q / TAPS : 0;
    ptap[i] = ok ? q - pci[i] * TAPS : -1;
    const int t = ok ? ptap[i] : 0;
    poff[i] = ((t / 3) * PW + (t % 3)) * p.dil * PS + pci[i] * 32 + piece * 8;
  }
  f32x4 acc[NPW][COT];
#pragma unroll
  for (int i = 0; i < NPW; ++i)
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool full = ptap[NPW - 1] >= 0;   // wave-uniform: does this wave use its last pair slot?
  // this lane's two pixels of K step 0 (local index pl = ks*32 + kq*4 + h*16 + rsel, conv_wgrad16_kernel's map)
  int py0[2], px0[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int pl = kq * 4 + h * 16 + rsel;
    py0[h] = pl / Wo;
    px0[h] = pl - py0[h] * Wo;
  }

  for (int u = 0; u < nunits; ++u) {
    WG6_STAMP();
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's share of unit u has landed
    __builtin_amdgcn_s_barrier();         // ... everybody's, and every wave has left unit u - 1 (the other buffer is free)
    asm volatile("" ::: "memory");
    WG6_STAMP();
    const bool more = u + 1 < nunits;
    const unsigned nbuf = lds0 + ((u + 1) & 1) * BUFSZ;
    const char* xt = smem + (u & 1) * BUFSZ;
    const UnitGeo gn = unit_geo(more ? u0g + u + 1 : u0g + u);
    int py[2], pxx[2], pl[2], ya[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      pl[h] = kq * 4 + h * 16 + rsel;
      py[h] = py0[h];
      pxx[h] = px0[h];
      ya[h] = XB + pl[h] * PSY + piece * 8;
    }
    // Two fragment sets: step ks + 1 is requested (and one piece of unit u + 1's copy issued) before step ks is multiplied;
    // the loop is unrolled and the scheduler fenced, so the order below is the order in the ISA.
    auto body = [&](auto npc) {
      constexpr int NP = decltype(npc)::value;
      hx8 bfr[2][COT], afr[2][NP];
      auto load = [&](int set) {
        int xb[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          // pixels past the unit meet a zero dY row; their X address only has to stay inside the buffer
          xb[h] = pl[h] < p.M ? p.st * (py[h] * PW + pxx[h]) * PS : 0;
          pl[h] += 32;
          pxx[h] += p.dxr;
          py[h] += p.dyq;
          if (pxx[h] >= Wo) {
            pxx[h] -= Wo;
            py[h] += 1;
          }
        }
#pragma unroll
        for (int c = 0; c < COT; ++c) {
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xt + ya[0] + c * 32));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xt + ya[1] + c * 32));
          bfr[set][c] = frag_of<hx8>(lo, hi);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xt + xb[0] + poff[i]));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xt + xb[1] + poff[i]));
          afr[set][i] = frag_of<hx8>(lo, hi);
        }
        ya[0] += 32 * PSY;
        ya[1] += 32 * PSY;
      };
      load(0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) load((ks + 1) & 1);
        if (more) {
          if (ks < XJ + WG6_YJ) dma_piece(gn, nbuf, ks);
          if (ks == KS - 1) {
#pragma unroll
            for (int k = KS; k < XJ + WG6_YJ; ++k) dma_piece(gn, nbuf, k);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
          for (int c = 0; c < COT; ++c) acc[i][c] = H16<H>::mfma(afr[ks & 1][i], bfr[ks & 1][c], acc[i][c]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (full) body(std::integral_constant<int, NPW>());
    else body(std::integral_constant<int, NPW - 1>());
  }
  WG6_STAMP();

  // D row = kq*4 + r (ci), col = l16 (co)  ->  slab [job][tap][ci][co]
  float* slab = p.part + (long)job * TAPS * p.Ci * p.Co;
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    if (ptap[i] < 0) continue;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
      const int co = cob * (16 * COT) + c * 16 + l16;
      if (co >= p.Co) continue;                        // (channel tail)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = cib * (16 * CIT) + pci[i] * 16 + kq * 4 + r;
        slab[((long)ptap[i] * p.Ci + ci) * p.Co + co] = acc[i][c][r];
      }
    }
  }
  WG6_STAMP();
}

template <typename H, int KS, int CIT, int COT, int XJ = WG6_XJ>
__global__ __launch_bounds__(WG16_THREADS, 1) void conv_wgrad6_kernel(Wg6Args p) {
  conv_wgrad6_body<H, KS, CIT, COT, XJ>(p, blockIdx.x, blockIdx.y, gridDim.x);
}
