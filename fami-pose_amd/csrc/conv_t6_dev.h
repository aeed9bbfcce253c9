// Device code of conv_t6.hip (the weight-resident / phased DMA-staged 3x3 kernels), shared with conv_pair.hip.
#pragma once
#include "conv_epi.h"
#include <type_traits>

struct ConvT6Args {
  EpiBN e;
  int emode;
  const void* x;      // [N,H,W,Ci]
  const void* wimg;   // packed fragment image [tap][KC][NTt][64][8]
  void* y;            // [N,H,W,Co]
  const float* bias;
  int N, H, W, Ci, Co;
  int KC, NTt;
  int sgn, relu, accumulate, out_f32;
  int RB, bands;      // output rows per band (even), bands per frame
  int PW, RG;         // W + 2, 16-byte granules per patch row (PW * Ci / 8)
  int REMP;           // (tile, channel tile) pairs of the tiles past the eighth: (2 W / 16 - 8) * NT
  int pj;             // patch DMA instructions per wave
  int q512, r512;     // 512 / RG, 512 % RG
  long long* dbg;     // FAMI_T6_TRACE builds: s_memtime stamps of one workgroup (null otherwise)
  XBN xb;             // XB instances (round 6): x is the INPUT z of a train-mode BatchNorm + ReLU; the workgroup applies it to its patch in LDS ...
  void* xout;         // ... and writes the normalised rows it owns here ([N,H,W,Ci]: the tensor the reference materialises, the backward pass reads it)
};

#define T6_THREADS 512
#define T6_WAVES 8

// EpiBN mode 2 on one lane's four output channels (conv_epi.h; conv_t4.hip's epilogue): v = dL/d(BN output) complete -> ReLU mask
// (from the BN output, or recomputed from its input z exactly as the forward apply pass computes it), the masked value rounded to
// the storage type is what gets stored; sum g and sum g * xhat are taken from the rounded values.
// Every operand is on chip already: zz = the BN input's four values, yy = the BN output's (rmode 1; both
// requested a unit ahead), ct = this lane's rows of the workgroup's channel table in LDS ([mean | invstd | scale | shift][NT * 16]).
template <typename H>
__device__ __forceinline__ f32x4 t6_epi2p(f32x4 v, f32x4 zz, f32x4 yy, const float* ct, int cstride, int rmode, f32x4& s, f32x4& q) {
  const f32x4 mu = *reinterpret_cast<const f32x4*>(ct), is = *reinterpret_cast<const f32x4*>(ct + cstride);
  if (rmode == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = yy[r] > 0.f ? v[r] : 0.f;
  } else if (rmode == 2) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(ct + 2 * cstride), b = *reinterpret_cast<const f32x4*>(ct + 3 * cstride);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(zz[r], a[r], b[r]) > 0.f ? v[r] : 0.f;
  }
  const f32x4 g = ld4_round<H>(v);
  s += g;
  q += g * ((zz - mu) * is);
  return v;
}

// G: 16-byte granules per pixel (Ci / 8); NT: channel tiles per workgroup; MT: own pixel tiles per wave (a unit is 2 MT rows);
// EX: 1 if the unit has tiles past the 8 MT-th; ACC: y += result; EM: EpiBN mode (0 | 1 | 2)
// (the body is a device function of the block coordinates so that conv_pair.hip can run it beside a weight-gradient body in one launch)
// XB (round 6, forward only): the BatchNorm + ReLU in front of this convolution (conv1 -> bn1 -> ReLU -> conv2 of a BasicBlock,
// basic_model.py:34-63) runs INSIDE this launch.  x is the BatchNorm's input z; its statistics sit in the slot rows the producing
// convolution's epilogue filled (p.xb: folded by every workgroup in its prologue, published by the first).  Every patch row is
// transformed once, in LDS, right after it has landed -- y = relu(fma(z, sc, sf)), the arithmetic of norm.hip's apply pass, so the
// values are bit for bit what bn_apply2_kernel would have written; the zero border and the rows outside the image stay zero -- and
// the workgroup stores the rows it OWNS to p.xout.  What it replaces: one launch per BasicBlock and the second read of z.
template <typename H, int G, int NT, int MT, int EX, bool ACC, int EM, bool XB = false>
__device__ __forceinline__ void conv3x3_t6_body(const ConvT6Args& p, const int bx, const int by, const int gx) {
  typedef typename H16<H>::x8 frag;
  constexpr int NK = (9 * G + 3) / 4;              // 32-wide K chunks over (tap, granule)
  constexpr int WJ = (NK * NT + T6_WAVES - 1) / T6_WAVES;   // weight DMA instructions per wave
  constexpr int PSB = G * 16;                      // bytes per patch position
  constexpr int PF = 2;                            // fragment sets in flight
  constexpr int UR = 2 * MT;                       // rows per unit
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wslab = smem;                        // [NK][NT][64][16 B]
  char* const patch = smem + WJ * T6_WAVES * 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4;
  int job;
  {   // XCD x owns the x-th contiguous eighth of the job list: neighbouring bands (shared halo rows) in one L2
    const int n = gx, lin = bx;
    const int q = n >> 3, r = n & 7, xc = lin & 7, l = lin >> 3;
    job = xc * q + (xc < r ? xc : r) + l;
  }
  const int img = job / p.bands, bnd = job - img * p.bands;
  const int y0 = bnd * p.RB;
  const int nrows = min(p.RB, p.H - y0);
  const int nunits = nrows / UR;
  const int ntg0 = by * NT;
  const int W = p.W, PW = p.PW;
#ifdef FAMI_T6_TRACE
  const bool trace = p.dbg && job == 100 && lane == 0;
  int tslot = 0;
#define T6_STAMP() if (trace) p.dbg[wave * 64 + tslot++] = (long long)__builtin_amdgcn_s_memtime()
#else
#define T6_STAMP()
#endif
  T6_STAMP();

  // ---- DMA.  Buffer loads: a lane whose granule is a border / out-of-image / padding granule carries an out-of-range
  // offset and the buffer unit writes zeros.  The weight slab and the rows of unit 0 are requested here; the rows of unit
  // u + 1 after the barrier of unit u (a CU's vector memory path moves 64 B / clk: the 114 KB of a job are 1.8 k cycles of
  // it, and only 84 KB of them stand in front of the first MFMA).
  {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wimg), 0, 9 * p.KC * p.NTt * 1024, 0x00020000);
    const int lane_w = col * 16;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int i = wave + T6_WAVES * j;           // (wave-uniform)
      const int k = i / NT, nt = i - k * NT;
      const int kg = 4 * k + kq;
      const int tap = kg / G, c8 = kg - tap * G;
      unsigned off = (unsigned)(((tap * p.KC + (c8 >> 2)) * p.NTt + ntg0 + nt) * 1024 + ((c8 & 3) << 8) + lane_w);
      if (kg >= 9 * G || k >= NK || ntg0 + nt >= p.NTt) off = 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(wslab + i * 1024), 16, off, 0, 0, 0);
    }
  }
  T6_STAMP();
  const long fbytes = (long)p.H * W * PSB;       // one frame
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.x)) + (long)img * fbytes, 0, (int)fbytes, 0x00020000);
  int dr, dwi, dj = 0;                             // patch DMA cursor of this lane: row, granule in the row, instruction of the wave
  {
    const int q0 = wave * 64 + lane;
    dr = q0 / p.RG;
    dwi = q0 - dr * p.RG;
  }
  auto dma_rows = [&](int rows) {                  // request the patch up to (not including) row `rows`
    int jn = ((((rows * p.RG + 63) >> 6) + T6_WAVES - 1) / T6_WAVES);
    if (jn > p.pj) jn = p.pj;
    for (; dj < jn; ++dj) {
      const int pos = dwi / G, c = dwi - pos * G;
      const int yy = y0 - 1 + dr, xx = pos - 1;
      unsigned off = (unsigned)(((yy * W + xx) * G + c) * 16);
      if (dr >= nrows + 2 || (unsigned)yy >= (unsigned)p.H || (unsigned)xx >= (unsigned)W) off = 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(patch + (wave + T6_WAVES * dj) * 1024), 16, off, 0, 0, 0);
      dr += p.q512;                                // 512 granules on: 512 = q512 * RG + r512
      dwi += p.r512;
      if (dwi >= p.RG) {
        dwi -= p.RG;
        ++dr;
      }
    }
  };
  dma_rows(UR + 2);
  T6_STAMP();

  // ---- EpiBN mode 2: the workgroup's channel table (behind the patch) is written before the first barrier; the BN input's
  // values of a unit are requested before the unit's MFMA loop and used in its (deferred) epilogue
  typedef H hx4 __attribute__((ext_vector_type(4)));
  float* const ctab = reinterpret_cast<float*>(patch + p.pj * (T6_WAVES * 1024));    // [4][NT * 16]
  const H* zsrc = nullptr;
  const H* yrsrc = nullptr;
  int rmode = 0;
  if (EM == 2) {
    EpiPtr e = epi_late(__builtin_offsetof(ConvT6Args, e));
    zsrc = reinterpret_cast<const H*>(e->z);
    yrsrc = reinterpret_cast<const H*>(e->yr);
    rmode = e->relu;
    if (tid < NT * 16) {
      const int co = ntg0 * 16 + tid;
      const float mu = e->mean[co], is = e->invstd[co];
      float a = 0.f, b = 0.f;
      if (rmode == 2) epi_scale_shift(mu, is, e->gamma[co], e->beta[co], a, b);
      ctab[tid] = mu;
      ctab[NT * 16 + tid] = is;
      ctab[2 * NT * 16 + tid] = a;
      ctab[3 * NT * 16 + tid] = b;
    }
  }

  // ---- XB: scale / shift of the input BatchNorm, [2][Ci] floats behind the channel table (written before the first barrier)
  float* const xtab = ctab + 4 * NT * 16;
  constexpr int XST = 510 / G * G;                    // sweeping threads: a multiple of G, so a thread's channel granule is fixed
  int xs_row = 0, xs_wi = 0;
  if constexpr (XB) {
    static_assert(!ACC && EM != 2, "the input BatchNorm is a forward-pass form");
    if (tid < p.Ci) {
      float a, b;
      xbn_channel(p.xb, tid, job == 0 && by == 0, a, b);
      xtab[tid] = a;
      xtab[p.Ci + tid] = b;
    }
    xs_row = tid / p.RG;
    xs_wi = tid - xs_row * p.RG;
  }
  // rows [ra, rb) of the patch (all landed, none transformed yet): granule q = row * RG + wi, position wi / G, channel granule wi % G
  auto xb_sweep = [&](int ra, int rb) {
    if (tid >= XST) return;
    const int c8 = xs_wi % G;                         // (RG is a multiple of G)
    float sc[8], sf[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      sc[t] = xtab[c8 * 8 + t];
      sf[t] = xtab[p.Ci + c8 * 8 + t];
    }
    const int drow = XST / p.RG, dwi2 = XST - drow * p.RG;
    H* const xo = reinterpret_cast<H*>(p.xout);
    int row = ra + xs_row, wi = xs_wi;
    auto step = [&](int& r_, int& w_) {
      r_ += drow;
      w_ += dwi2;
      if (w_ >= p.RG) {
        w_ -= p.RG;
        ++r_;
      }
    };
    // two granules per trip: their LDS reads are issued together (a trip is one dependent chain read -> 30 VALU -> write -> store)
    while (row < rb) {
      int row2 = row, wi2 = wi;
      step(row2, wi2);
      const int pos = wi / G, yy = y0 - 1 + row, pos2 = wi2 / G, yy2 = y0 - 1 + row2;
      const bool in1 = pos >= 1 && pos <= W && (unsigned)yy < (unsigned)p.H;
      const bool in2 = row2 < rb && pos2 >= 1 && pos2 <= W && (unsigned)yy2 < (unsigned)p.H;
      char* a1 = patch + (row * p.RG + wi) * 16;
      char* a2 = patch + (row2 * p.RG + wi2) * 16;
      u32x4 v1, v2;
      if (in1) v1 = *reinterpret_cast<const u32x4*>(a1);
      if (in2) v2 = *reinterpret_cast<const u32x4*>(a2);
      if (in1) {
        v1 = xbn_piece<H>(v1, sc, sf);
        *reinterpret_cast<u32x4*>(a1) = v1;
        if (row >= 1 && row <= nrows)
          *reinterpret_cast<u32x4*>(xo + ((long)(img * p.H + yy) * W + (pos - 1)) * p.Ci + c8 * 8) = v1;
      }
      if (in2) {
        v2 = xbn_piece<H>(v2, sc, sf);
        *reinterpret_cast<u32x4*>(a2) = v2;
        if (row2 >= 1 && row2 <= nrows)
          *reinterpret_cast<u32x4*>(xo + ((long)(img * p.H + yy2) * W + (pos2 - 1)) * p.Ci + c8 * 8) = v2;
      }
      row = row2;
      wi = wi2;
      step(row, wi);
    }
  };

  // ---- epilogue constants (loaded after the first barrier)
  const int nte = wave % NT;                          // channel tile of the extra pair (waves 0 .. REMP-1: tile 8 MT + wave / NT)
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 es[NT], eq[NT], ek[NT], bias4[NT], ese = z4, eqe = z4, eke = z4, biase = z4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) es[nt] = eq[nt] = ek[nt] = bias4[nt] = z4;
  auto load_consts = [&]() {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co0 = (ntg0 + nt) * 16 + kq * 4;
      if (p.bias) bias4[nt] = *reinterpret_cast<const f32x4*>(p.bias + co0);
      if (EM == 1 && p.e.pivot_src) ek[nt] = *reinterpret_cast<const f32x4*>(p.e.pivot_src + co0);
    }
    if (EX) {
      const int co0 = (ntg0 + nte) * 16 + kq * 4;
      if (p.bias) biase = *reinterpret_cast<const f32x4*>(p.bias + co0);
      if (EM == 1 && p.e.pivot_src) eke = *reinterpret_cast<const f32x4*>(p.e.pivot_src + co0);
    }
  };

  // ---- per-lane constants: K chunk -> LDS byte offset of this lane quarter's (tap, granule) relative to the pixel's own position
  int koff[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    int kg = 4 * k + kq;
    if (kg >= 9 * G) kg = 4 * G;      // zero weights there: any finite in-receptive-field value (the pixel's own granule 0)
    const int tap = kg / G, c8 = kg - tap * G;
    koff[k] = (p.sgn * ((tap / 3 - 1) * PW + (tap % 3 - 1)) * G + c8) * 16;
  }
  // own tiles wave, wave + 8, ...; the extra pair of waves 0 .. REMP-1: tile 8 MT + wave / NT, channel tile wave % NT
  const bool has_e = EX && wave < p.REMP;
  int base[MT], oown[MT], basee = 0, oex = 0;      // LDS byte offset of the lane's pixel (unit 0); output element offset in the unit
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int j = (wave + T6_WAVES * m) * 16 + col;
    const int rr = j / W, xx = j - rr * W;
    base[m] = ((rr + 1) * PW + xx + 1) * PSB;
    oown[m] = j * p.Co + ntg0 * 16 + kq * 4;
  }
  if (EX) {
    const int je = (T6_WAVES * MT + wave / NT) * 16 + col;
    const int rre = je / W, xxe = je - rre * W;
    basee = ((rre + 1) * PW + xxe + 1) * PSB;
    oex = je * p.Co + (ntg0 + nte) * 16 + kq * 4;
  }
  const int wl = lane * 16, wle = lane * 16 + nte * 1024;

  f32x4 sv[MT][NT], sve = z4;
  // EM == 2: what the epilogue of the unit in flight reads at the lane's outputs, requested before the unit's MFMA loop:
  // the BN input, the BN output (rmode 1: zero registers otherwise), the gradient so far (ACC)
  hx4 zp[MT][NT], zpe, rp[MT][NT], rpe, ap[MT][NT], ape;
  auto prefetch = [&](int u) {
    const long ub = (long)(img * p.H + y0 + UR * u) * W * p.Co;
    const H* zb = zsrc + ub;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) zp[m][nt] = *reinterpret_cast<const hx4*>(zb + oown[m] + nt * 16);
    if (has_e) zpe = *reinterpret_cast<const hx4*>(zb + oex);
    if (rmode == 1) {
      const H* rb = yrsrc + ub;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) rp[m][nt] = *reinterpret_cast<const hx4*>(rb + oown[m] + nt * 16);
      if (has_e) rpe = *reinterpret_cast<const hx4*>(rb + oex);
    }
    if (ACC) {
      const H* ab = reinterpret_cast<const H*>(p.y) + ub;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) ap[m][nt] = *reinterpret_cast<const hx4*>(ab + oown[m] + nt * 16);
      if (has_e) ape = *reinterpret_cast<const hx4*>(ab + oex);
    }
  };
  auto emit1 = [&](f32x4 v, H* yp, int ctl, const hx4& zq, const hx4& rq, const hx4& aq, const f32x4& b4, const f32x4& k4, f32x4& s, f32x4& q) {
    v += b4;
    if (EM == 2) {
      if (ACC) v += __builtin_convertvector(aq, f32x4);
      v = t6_epi2p<H>(v, __builtin_convertvector(zq, f32x4), __builtin_convertvector(rq, f32x4), ctab + ctl, NT * 16, rmode, s, q);
    } else if (ACC) {
      v += ld4(yp);
    }
    st4(yp, v);
    if (EM == 1) {
      const f32x4 d = ld4_round<H>(v) - k4;
      s += d;
      q += d * d;
    }
  };
  auto emit = [&](int u) {
    H* yb = reinterpret_cast<H*>(p.y) + (long)(img * p.H + y0 + UR * u) * W * p.Co;     // (wave-uniform)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        emit1(sv[m][nt], yb + oown[m] + nt * 16, nt * 16 + kq * 4, zp[m][nt], rp[m][nt], ap[m][nt], bias4[nt], ek[nt], es[nt], eq[nt]);
    if (has_e) emit1(sve, yb + oex, nte * 16 + kq * 4, zpe, rpe, ape, biase, eke, ese, eqe);
  };

  // ---- units
  const int ustep = UR * PW * PSB;
  T6_STAMP();
  for (int u = 0; u < nunits; ++u) {
    T6_STAMP();
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): this wave's share of the unit's rows (and of the weights) has landed; the stores of unit u - 2 are long done (lgkmcnt(0): the channel table's writes)
    __builtin_amdgcn_s_barrier();         // ... and everybody else's
    asm volatile("" ::: "memory");        // no LDS read of the unit may move (or be hoisted out of the loop) above the wait
    T6_STAMP();
    if (EM == 2) {                        // the epilogue reads what was requested a unit ago: ahead of this unit's requests, so its wait is the one above
      if (u > 0) emit(u - 1);
      if (u + 1 < nunits) dma_rows(UR * (u + 2) + 2);
      if (u == 0) load_consts();
      prefetch(u);
    } else {
      if (u + 1 < nunits) dma_rows(UR * (u + 2) + 2);
      if (u == 0) load_consts();
      if (u > 0) emit(u - 1);
    }
    if constexpr (XB) {                   // the rows that landed for this unit: transformed once, visible to every wave behind the barrier
      xb_sweep(u == 0 ? 0 : UR * u + 2, UR * (u + 1) + 2);
      __syncthreads();
    }
    T6_STAMP();
    f32x4 acc[MT][NT], acce = z4;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[m][nt] = z4;
    const char* pbe = patch + basee + u * ustep;
    auto body = [&](auto ec) {
      constexpr bool E = decltype(ec)::value;
      // PF register sets: the fragments of chunk k + PF - 1 are requested before chunk k is multiplied
      frag a[PF][NT], b[PF][MT], ae[PF], be[PF];
      auto ld = [&](int k, int s) {
#pragma unroll
        for (int m = 0; m < MT; ++m) b[s][m] = *reinterpret_cast<const frag*>(patch + base[m] + u * ustep + koff[k]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a[s][nt] = *reinterpret_cast<const frag*>(wslab + (k * NT + nt) * 1024 + wl);
        if constexpr (E) {
          be[s] = *reinterpret_cast<const frag*>(pbe + koff[k]);
          ae[s] = *reinterpret_cast<const frag*>(wslab + k * NT * 1024 + wle);
        }
      };
#pragma unroll
      for (int k = 0; k < PF - 1; ++k) ld(k, k);
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int s = k % PF;
        if (k + PF - 1 < NK) ld(k + PF - 1, (k + PF - 1) % PF);
        __builtin_amdgcn_sched_barrier(0);      // as written: left alone, the scheduler requests a fragment one to two MFMAs ahead of its use
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[m][nt] = H16<H>::mfma(a[s][nt], b[s][m], acc[m][nt]);
        if constexpr (E) acce = H16<H>::mfma(ae[s], be[s], acce);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (has_e) body(std::integral_constant<bool, EX != 0>());
    else body(std::integral_constant<bool, false>());
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) sv[m][nt] = acc[m][nt];
    sve = acce;
    T6_STAMP();
  }
  if (nunits > 0) emit(nunits - 1);
  T6_STAMP();

  // ---- EpiBN: per-channel sums of the workgroup -> fp64 slot rows
  if (EM != 0) {
    EpiPtr e = epi_late(__builtin_offsetof(ConvT6Args, e));
    __syncthreads();                                   // every wave is done with the patch
    float* ered = reinterpret_cast<float*>(patch);     // [waves][NT*32] own tiles, then [waves][32] extra pairs
    float* erex = ered + T6_WAVES * NT * 32;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = row16_sum(es[nt][r]), q = row16_sum(eq[nt][r]);
        if (col == 0) {
          ered[wave * (NT * 32) + nt * 32 + kq * 4 + r] = s;
          ered[wave * (NT * 32) + nt * 32 + 16 + kq * 4 + r] = q;
        }
      }
    }
    if (EX) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = row16_sum(ese[r]), q = row16_sum(eqe[r]);
        if (col == 0) {
          erex[wave * 32 + kq * 4 + r] = has_e ? s : 0.f;
          erex[wave * 32 + 16 + kq * 4 + r] = has_e ? q : 0.f;
        }
      }
    }
    __syncthreads();
    if (tid < NT * 32) {
      const int nt = tid >> 5, st = (tid >> 4) & 1, c16 = tid & 15;
      const int co = (ntg0 + nt) * 16 + c16;
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < T6_WAVES; ++wv) v += ered[wv * (NT * 32) + tid];
      if (EX) {
        for (int wv = nt; wv < p.REMP; wv += NT) v += erex[wv * 32 + (tid & 31)];   // the extra pairs with this channel tile
      }
      const int eC = e->C;
      double* srow = e->slots + (long)(job % e->ns) * 2 * eC;
      unsafeAtomicAdd(srow + st * eC + co, (double)v);
      if (EM == 1 && st == 0 && job == 0) bn_slots_pivot(e->slots, eC)[co] = e->pivot_src ? e->pivot_src[co] : 0.f;
    }
  }
}

template <typename H, int G, int NT, int MT, int EX, bool ACC, int EM, bool XB = false>
__global__ __launch_bounds__(T6_THREADS, 1) void conv3x3_t6_kernel(ConvT6Args p) {
  conv3x3_t6_body<H, G, NT, MT, EX, ACC, EM, XB>(p, blockIdx.x, blockIdx.y, gridDim.x);
}

// ------------------------------------------------------------------ the same kernel for 96 / 192 / 384 input channels ("t7")
// The weight image of a 48-channel SLICE of the input (42 KiB for 48 output channels) is what fits the LDS, so the wider layers
// walk their input channels in PHASES of 48: a workgroup owns a band of whole rows of one frame (all of its pixels' accumulators
// stay in registers: 18 tiles = 2 per wave + two shared ones on the 48x36 maps, 9 = 1 + a shared one on the 24x18 maps, the whole
// 7-tile frame on the 12x9 maps) and, per phase, the dense-K weight slab of the slice and the slice of the patch (96 of a
// position's 2 Ci bytes) are copied by LDS DMA into one of two buffers while the previous phase is multiplied out of the other --
// conv_wgrad6_kernel's pipeline (one wait + barrier per phase, the copy issued from inline assembly so that hipcc's wait-count pass
// does not serialise it against the LDS reads).  Replaces conv3x3_t4_kernel on these shapes: its 32-channel chunks meet twice per
// chunk, stage through registers, and re-stage the 27 KB weight slab of a chunk for every 192 pixels.
typedef int t7_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ t7_i32x4 t7_rsrc(const void* base, int bytes) {
  const unsigned long a = (unsigned long)base;
  const t7_i32x4 r = {(int)(unsigned)a, (int)((a >> 32) & 0xffff), bytes, 0x00020000};
  return r;
}
__device__ __forceinline__ void t7_dma16(t7_i32x4 r, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(voff), "s"(r) : "memory");
}
struct ConvT7Args {
  EpiBN e;
  int emode;
  const void* x;      // [N,H,W,Ci]
  const void* wimg;   // packed fragment image [tap][KC][NTt][64][8]
  void* y;            // [N,H,W,Co]
  const float* bias;
  int N, H, W, Ci, Co;
  int KC, NTt;
  int sgn, accumulate;
  int RB, bands;      // output rows per band (H % RB == 0), bands per frame
  int PW, RG;         // W + 2, 16-byte granules per patch row of a slice (G PW)
  int q512, r512;     // 512 / RG, 512 % RG
  int nph;            // phases (Ci / (8 G))
  int TU, npix;       // 16-pixel tiles of a band (the last may be ragged), pixels of a band (RB * W)
  int REMP;           // (tile, channel tile) pairs of the tiles past the 8 MT-th
  int PI;             // patch DMA instructions (1 KiB) of a phase
  int jpw;            // jobs (bands) per workgroup, consecutive
};
#define T7_PJ 6       // most patch DMA instructions per wave and phase (PI <= 48)

// G: 16-byte granules of a phase's channel slice (6: 48 channels, NT = 3 -- the W48 branches; 4: 32 channels, NT = 4 -- the 64 /
// 128 / 256 / 512-channel layers of HRNet-W64 and the 64 -> 64 convolutions of stage 1, round 5).  G = 4: a K chunk is one tap
// (lane quarter kq = the slice's granule kq), a weight-slab block is a block of the packed image as it stands, and the 64-byte
// positions would put the 16 lanes of a ds_read_b128 group on 8 bank quads (2-way conflicts at any padding, tools/probes note in
// DESIGN 4) -- so granule c of linear patch position P sits at slot c ^ ((P >> 1) & 2): the four positions of a group that share
// a quad block (P mod 4 equal) then take four different slots.  The copy un-swizzles at the source (a lane's LDS slot is fixed,
// its source granule is slot ^ key), the reads compute the slot from P (4 VALU per fragment).
template <typename H, int G, int NT, int MT, int EX, bool ACC, int EM>
__device__ __forceinline__ void conv3x3_t7_body(const ConvT7Args& p, const int bx, const int by, const int gx) {
  typedef typename H16<H>::x8 frag;
  constexpr int NK = (9 * G + 3) / 4, PSB = G * 16, PF = 2;
  constexpr bool SWZ = G == 4;
  constexpr int WI = NK * NT;                      // weight DMA instructions of a phase (42 | 36)
  constexpr int WJ = (WI + T6_WAVES - 1) / T6_WAVES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4;
  int wg;
  {   // XCD x owns the x-th contiguous eighth of the workgroup list (consecutive jobs = neighbouring bands)
    const int n = gx, lin = bx;
    const int q = n >> 3, r = n & 7, xc = lin & 7, l = lin >> 3;
    wg = xc * q + (xc < r ? xc : r) + l;
  }
  const int job0 = wg * p.jpw, job1 = min(job0 + p.jpw, p.N * p.bands);     // this workgroup's jobs (bands), consecutive
  const int ntg0 = by * NT;
  const int W = p.W, PW = p.PW;
  const int BUFSZ = (WI + p.PI) * 1024;
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;

  // ---- DMA plan of this lane (job- and phase-invariant parts): patch granule -> (patch row, byte offset from the pixel 0 of
  // the band's first row at phase 0, or < 0: a border-column granule)
  const long fbytes = (long)p.H * W * p.Ci * 2;
  const t7_i32x4 rw = t7_rsrc(p.wimg, 9 * p.KC * p.NTt * 1024);
  int xrow[T7_PJ], xoff[T7_PJ];
  {
    const int q0 = wave * 64 + lane;
    int r = q0 / p.RG, wi = q0 - r * p.RG;
#pragma unroll
    for (int j = 0; j < T7_PJ; ++j) {
      const int pos = wi / G;
      int c = wi - pos * G;
      if (SWZ) c ^= ((r * PW + pos) >> 1) & 2;      // (the slot is fixed by the copy's LDS address: fetch the granule that belongs there)
      const bool ok = r < p.RB + 2 && pos >= 1 && pos <= W;
      xrow[j] = ok ? r - 1 : 0x40000000;            // image row relative to the band's first row; never valid for a border column
      xoff[j] = (((r - 1) * W + pos - 1) * p.Ci + c * 8) * 2;
      r += p.q512;
      wi += p.r512;
      if (wi >= p.RG) {
        wi -= p.RG;
        ++r;
      }
    }
  }
  // a job's scalars (first input row, frame descriptor) once per step instead of once per DMA piece
  struct JobGeo { int y0; t7_i32x4 rx; };
  auto job_geo = [&](int jb) {
    JobGeo g;
    const int img = jb / p.bands;
    g.y0 = (jb - img * p.bands) * p.RB;
    g.rx = t7_rsrc(reinterpret_cast<const char*>(p.x) + (long)img * fbytes, (int)fbytes);
    return g;
  };
  // piece k of the copy of (job, phase) (k < WJ: weight slab, else patch slice); wave-uniform guards
  auto dma_piece = [&](const JobGeo& g, int ph, unsigned buf, int k) {
    if (k < WJ) {
      const int i = wave + T6_WAVES * k;
      if (i < WI) {
        const int kc = i / NT, nt = i - kc * NT;
        const int kg = 4 * kc + kq;
        const int tap = kg / G, c8 = ph * G + (kg - tap * G);      // 8-channel granule of the whole input
        unsigned off = (unsigned)(((tap * p.KC + (c8 >> 2)) * p.NTt + ntg0 + nt) * 1024 + ((c8 & 3) << 8) + col * 16);
        if (kg >= 9 * G || ntg0 + nt >= p.NTt) off = 0x80000000u;
        t7_dma16(rw, off, buf + i * 1024);
      }
    } else {
      const int j = k - WJ;
      const int i = wave + T6_WAVES * j;
      if (i < p.PI) {
        unsigned off = (unsigned)(g.y0 * W * p.Ci * 2 + xoff[j] + ph * PSB);
        if ((unsigned)(g.y0 + xrow[j]) >= (unsigned)p.H) off = 0x80000000u;
        t7_dma16(g.rx, off, buf + (WI + i) * 1024);
      }
    }
  };
  if (job0 < job1) {
    const JobGeo g0 = job_geo(job0);
#pragma unroll
    for (int k = 0; k < WJ + T7_PJ; ++k) dma_piece(g0, 0, lds0, k);
  }

  // ---- per-lane constants
  int koff[NK];                                    // byte offset of the lane quarter's (tap, granule) from the pixel's own position; SWZ: the tap's shift in POSITIONS
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    int kg = 4 * k + kq;
    if (kg >= 9 * G) kg = 4 * G;
    const int tap = kg / G, c8 = kg - tap * G;
    const int sh = p.sgn * ((tap / 3 - 1) * PW + (tap % 3 - 1));
    koff[k] = SWZ ? sh : (sh * G + c8) * 16;
  }
  // LDS byte address of a fragment: base = the pixel's own position (SWZ: its linear position index)
  auto faddr = [&](int base_, int k) {
    if constexpr (SWZ) {
      int P = base_ + koff[k];
      asm volatile("" : "+v"(P));      // (computed where it is used: hoisted out of the phase loop the 27 addresses cost 27 registers and the EpiBN instances spill)
      return WI * 1024 + P * PSB + ((kq ^ ((P >> 1) & 2)) << 4);
    } else {
      return base_ + koff[k];
    }
  };
  const int nte = wave % NT;
  const bool has_e = EX && wave < p.REMP;
  bool own[MT], pvalid[MT], pvalide = false;       // own: the wave has this tile (wave-uniform); pvalid: the lane's pixel is inside the band
  int base[MT], oown[MT], basee = 0, oex = 0;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int t = wave + T6_WAVES * m;
    own[m] = t < p.TU;
    int j = t * 16 + col;
    pvalid[m] = own[m] && j < p.npix;
    if (j >= p.npix) j = p.npix - 1;                // ragged last tile: re-read the band's last pixel (never stored)
    const int rr = j / W, xx = j - rr * W;
    base[m] = SWZ ? (rr + 1) * PW + xx + 1 : WI * 1024 + ((rr + 1) * PW + xx + 1) * PSB;
    oown[m] = j * p.Co + ntg0 * 16 + kq * 4;
  }
  if (EX) {
    int je = (T6_WAVES * MT + wave / NT) * 16 + col;
    pvalide = has_e && je < p.npix;
    if (je >= p.npix) je = p.npix - 1;
    const int rre = je / W, xxe = je - rre * W;
    basee = SWZ ? (rre + 1) * PW + xxe + 1 : WI * 1024 + ((rre + 1) * PW + xxe + 1) * PSB;
    oex = je * p.Co + (ntg0 + nte) * 16 + kq * 4;
  }
  const int wl = lane * 16, wle = lane * 16 + nte * 1024;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  // a finished job's results wait one step for their stores: bias added, rounded to the storage type and packed (8 bytes per tile
  // and lane) unless the launch accumulates into y (then fp32, the old value is added when they are written)
  typedef H hx4 __attribute__((ext_vector_type(4)));
  typedef typename std::conditional<ACC, f32x4, hx4>::type SV;
  f32x4 acc[MT][NT], acce = z4;
  SV sv[MT][NT], sve;
  // EpiBN mode 2 (as in conv3x3_t6_kernel): the channel table behind the two buffers, written before the first barrier; what a
  // job's epilogue reads at the lane's outputs is requested in front of the job's last phase
  float* const ctab = reinterpret_cast<float*>(smem + 2 * BUFSZ);     // [4][NT * 16]
  const H* zsrc = nullptr;
  const H* yrsrc = nullptr;
  int rmode = 0;
  if (EM == 2) {
    EpiPtr e = epi_late(__builtin_offsetof(ConvT7Args, e));
    zsrc = reinterpret_cast<const H*>(e->z);
    yrsrc = reinterpret_cast<const H*>(e->yr);
    rmode = e->relu;
    if (tid < NT * 16) {
      const int co = ntg0 * 16 + tid;
      const float mu = e->mean[co], is = e->invstd[co];
      float a = 0.f, b = 0.f;
      if (rmode == 2) epi_scale_shift(mu, is, e->gamma[co], e->beta[co], a, b);
      ctab[tid] = mu;
      ctab[NT * 16 + tid] = is;
      ctab[2 * NT * 16 + tid] = a;
      ctab[3 * NT * 16 + tid] = b;
    }
  }
  hx4 zp[MT][NT], zpe, ap[MT][NT], ape;        // (the BN output of rmode 1 is read in the epilogue: no registers left for it)
  auto prefetch = [&](int jq) {
    const int img = jq / p.bands, y0 = (jq - img * p.bands) * p.RB;
    const long ub = (long)(img * p.H + y0) * W * p.Co;
    const H* zb = zsrc + ub;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) zp[m][nt] = *reinterpret_cast<const hx4*>(zb + oown[m] + nt * 16);
    if (EX) zpe = *reinterpret_cast<const hx4*>(zb + oex);
    if (ACC) {
      const H* ab = reinterpret_cast<const H*>(p.y) + ub;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) ap[m][nt] = *reinterpret_cast<const hx4*>(ab + oown[m] + nt * 16);
      if (EX) ape = *reinterpret_cast<const hx4*>(ab + oex);
    }
  };
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[m][nt] = z4;
  // EpiBN sums of the workgroup: an LDS table [NT][sum | sum-of-products][16] behind the channel table, added to with ds_add_f32 once per
  // job and wave (round 5: the per-lane accumulators -- 8 NT + 8 registers live across every phase of every job -- spilled in the
  // MT = 2 instances: 160-400 bytes of scratch; inside the W64 step the spilling instances cost more than the kernel saved)
  float* const lstat = ctab + 4 * NT * 16;
  if (EM != 0 && tid < NT * 32) lstat[tid] = 0.f;     // (ordered before the first add by the first step's barrier)
  auto stat_add = [&](int nt, const f32x4& s4, const f32x4& q4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = row16_sum(s4[r]), q = row16_sum(q4[r]);
      if (col == 0) {
        __hip_atomic_fetch_add(lstat + nt * 32 + kq * 4 + r, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(lstat + nt * 32 + 16 + kq * 4 + r, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  };
  auto save1 = [&](f32x4 v, SV& out, int co0, bool valid, f32x4& s, f32x4& q) {
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + co0);
    if constexpr (ACC) out = v;
    else {
      out = __builtin_convertvector(v, hx4);
      if (EM == 1 && valid) {
        f32x4 k4 = z4;
        if (p.e.pivot_src) k4 = *reinterpret_cast<const f32x4*>(p.e.pivot_src + co0);
        const f32x4 d = __builtin_convertvector(out, f32x4) - k4;
        s += d;
        q += d * d;
      }
    }
  };
  auto emit1 = [&](const SV& v, H* yp, int ctl, const hx4& zq, const hx4& aq, f32x4& s, f32x4& q) {
    if constexpr (EM == 2) {
      f32x4 t;
      if constexpr (ACC) t = v + __builtin_convertvector(aq, f32x4);
      else t = __builtin_convertvector(v, f32x4);          // (rounded once already: masking commutes with the rounding)
      f32x4 yy = z4;
      if (rmode == 1) yy = ld4(yrsrc + (yp - reinterpret_cast<H*>(p.y)));
      st4(yp, t6_epi2p<H>(t, __builtin_convertvector(zq, f32x4), yy, ctab + ctl, NT * 16, rmode, s, q));
    } else if constexpr (ACC) st4(yp, v + ld4(yp));
    else *reinterpret_cast<hx4*>(yp) = v;
  };
  auto emit = [&](int jb) {                          // the saved results of job jb
    const int img = jb / p.bands, y0 = (jb - img * p.bands) * p.RB;
    H* yb = reinterpret_cast<H*>(p.y) + (long)(img * p.H + y0) * W * p.Co;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 s4 = z4, q4 = z4;
#pragma unroll
      for (int m = 0; m < MT; ++m)
        if (pvalid[m]) emit1(sv[m][nt], yb + oown[m] + nt * 16, nt * 16 + kq * 4, zp[m][nt], ap[m][nt], s4, q4);
      if (EM == 2) stat_add(nt, s4, q4);
    }
    if (EX) {
      f32x4 s4 = z4, q4 = z4;
      if (pvalide) emit1(sve, yb + oex, nte * 16 + kq * 4, zpe, ape, s4, q4);
      if (EM == 2 && has_e) stat_add(nte, s4, q4);       // (has_e is wave-uniform; lanes of a ragged tile add zeros)
    }
  };

  // ---- (job, phase) steps: one wait + barrier each; the next step's copy is issued while this one is multiplied, across job
  // boundaries; a job's results are written after the NEXT step's barrier (no store in front of a wait)
  int jb = job0, ph = 0, step = 0, pending = -1;
  while (jb < job1) {
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): this wave's share of the step has landed (lgkmcnt(0): the channel table's writes)
    __builtin_amdgcn_s_barrier();         // ... everybody's, and every wave has left the previous step (the other buffer is free)
    asm volatile("" ::: "memory");
    int nj = jb, nph = ph + 1;
    if (nph == p.nph) {
      nph = 0;
      ++nj;
    }
    const bool more = nj < job1;
    const JobGeo gn = job_geo(more ? nj : jb);
    const unsigned nbuf = lds0 + ((step + 1) & 1) * BUFSZ;
    const char* cb = smem + (step & 1) * BUFSZ;
    if (pending >= 0) {
      emit(pending);
      pending = -1;
    }
    if (EM == 2 && nph == 0) prefetch(jb);
    auto body = [&](auto ec, auto oc) {
      constexpr bool E = decltype(ec)::value;
      constexpr bool O = decltype(oc)::value;          // the wave has own tiles (only the 12x9 maps leave a wave without)
      // (the extra pair's weight fragment is the wave's own a[s][nte] when the wave has own tiles: selected, not loaded again)
      constexpr bool AE = E && !O;
      frag a[PF][NT], b[PF][MT], ae[PF], be[PF];
      auto ld = [&](int k, int s) {
        if constexpr (O) {
#pragma unroll
          for (int m = 0; m < MT; ++m) b[s][m] = *reinterpret_cast<const frag*>(cb + faddr(base[m], k));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) a[s][nt] = *reinterpret_cast<const frag*>(cb + (k * NT + nt) * 1024 + wl);
        }
        if constexpr (E) {
          be[s] = *reinterpret_cast<const frag*>(cb + faddr(basee, k));
          if constexpr (AE) ae[s] = *reinterpret_cast<const frag*>(cb + k * NT * 1024 + wle);
        }
      };
      ld(0, 0);
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int s = k % PF;
        if (k + 1 < NK) ld(k + 1, (k + 1) % PF);
        if (more && k < WJ + T7_PJ) dma_piece(gn, nph, nbuf, k);      // (G = 6: NK = 14 >= WJ + T7_PJ = 12; G = 4: the last two pieces go with the last chunk)
        if constexpr (NK < WJ + T7_PJ) {
          if (more && k == NK - 1) {
#pragma unroll
            for (int kk = NK; kk < WJ + T7_PJ; ++kk) dma_piece(gn, nph, nbuf, kk);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (O) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = H16<H>::mfma(a[s][nt], b[s][m], acc[m][nt]);
        }
        if constexpr (E) {
          if constexpr (AE) acce = H16<H>::mfma(ae[s], be[s], acce);
          else {
            frag aw = a[s][0];
#pragma unroll
            for (int nt = 1; nt < NT; ++nt) aw = nte == nt ? a[s][nt] : aw;      // (wave-uniform)
            acce = H16<H>::mfma(aw, be[s], acce);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    typedef std::integral_constant<bool, true> Yes;
    typedef std::integral_constant<bool, false> No;
    if (own[MT - 1]) {
      if (has_e) body(std::integral_constant<bool, EX != 0>(), Yes());
      else body(No(), Yes());
    } else if (more) {                               // a wave without tiles still issues its share of the next step's copy
#pragma unroll
      for (int k = 0; k < WJ + T7_PJ; ++k) dma_piece(gn, nph, nbuf, k);
    }
    if (nph == 0) {                                  // the job is complete: keep its results for the next step, clear the accumulators
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 s4 = z4, q4 = z4;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          save1(acc[m][nt], sv[m][nt], (ntg0 + nt) * 16 + kq * 4, pvalid[m], s4, q4);
          acc[m][nt] = z4;
        }
        if (EM == 1 && !ACC && own[0]) stat_add(nt, s4, q4);      // (own[0] is wave-uniform)
      }
      if (EX) {
        f32x4 s4 = z4, q4 = z4;
        save1(acce, sve, (ntg0 + nte) * 16 + kq * 4, pvalide, s4, q4);
        if (EM == 1 && !ACC && has_e) stat_add(nte, s4, q4);
      }
      acce = z4;
      pending = jb;
    }
    jb = nj;
    ph = nph;
    ++step;
  }
  if (pending >= 0) emit(pending);

  if (EM != 0) {
    EpiPtr e = epi_late(__builtin_offsetof(ConvT7Args, e));
    __syncthreads();                                   // every wave's adds are in the table
    if (tid < NT * 32) {
      const int nt = tid >> 5, st = (tid >> 4) & 1, c16 = tid & 15;
      const int co = (ntg0 + nt) * 16 + c16;
      const float v = lstat[tid];
      const int eC = e->C;
      double* srow = e->slots + (long)(wg % e->ns) * 2 * eC;
      unsafeAtomicAdd(srow + st * eC + co, (double)v);
      if (EM == 1 && st == 0 && wg == 0) bn_slots_pivot(e->slots, eC)[co] = e->pivot_src ? e->pivot_src[co] : 0.f;
    }
  }
}

template <typename H, int G, int NT, int MT, int EX, bool ACC, int EM>
__global__ __launch_bounds__(T6_THREADS, 1) void conv3x3_t7_kernel(ConvT7Args p) {
  conv3x3_t7_body<H, G, NT, MT, EX, ACC, EM>(p, blockIdx.x, blockIdx.y, gridDim.x);
}
