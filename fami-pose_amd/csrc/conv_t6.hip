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
template <typename H, int G, int NT, int MT, int EX, bool ACC, int EM>
__global__ __launch_bounds__(T6_THREADS, 1) void conv3x3_t6_kernel(ConvT6Args p) {
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
    const int n = gridDim.x, lin = blockIdx.x;
    const int q = n >> 3, r = n & 7, xc = lin & 7, l = lin >> 3;
    job = xc * q + (xc < r ? xc : r) + l;
  }
  const int img = job / p.bands, bnd = job - img * p.bands;
  const int y0 = bnd * p.RB;
  const int nrows = min(p.RB, p.H - y0);
  const int nunits = nrows / UR;
  const int ntg0 = blockIdx.y * NT;
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
__global__ __launch_bounds__(T6_THREADS, 1) void conv3x3_t7_kernel(ConvT7Args p) {
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
    const int n = gridDim.x, lin = blockIdx.x;
    const int q = n >> 3, r = n & 7, xc = lin & 7, l = lin >> 3;
    wg = xc * q + (xc < r ? xc : r) + l;
  }
  const int job0 = wg * p.jpw, job1 = min(job0 + p.jpw, p.N * p.bands);     // this workgroup's jobs (bands), consecutive
  const int ntg0 = blockIdx.y * NT;
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

template <typename HT>
static int t6_launch(const T6Plan& q, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci,
                     int Co, int KC, int NTt, int sgn, int relu, int accumulate, int out_f32, hipStream_t s, const EpiBN& epi) {
  ConvT6Args a;
  a.e = epi; a.emode = epi.slots ? epi.mode : 0;
  a.x = x; a.wimg = wp; a.y = y; a.bias = bias;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.KC = KC; a.NTt = NTt;
  a.sgn = sgn; a.relu = relu; a.accumulate = accumulate; a.out_f32 = out_f32;
  a.RB = q.RB; a.bands = q.bands; a.PW = W + 2; a.RG = (W + 2) * 6; a.REMP = (q.TU - 8 * q.MT) * 3; a.pj = q.pj; a.dbg = g_t6_dbg;
  a.q512 = 512 / a.RG; a.r512 = 512 % a.RG;
  const dim3 grid(N * q.bands, Co / 48);
  bool ok = false;
  const bool acc_ = accumulate != 0;
  const int ex_ = q.TU > 8 * q.MT ? 1 : 0;
#define FAMI_T6_CASE(mt, ex, ac, em)                                                                                      \
  if (!ok && q.MT == mt && ex_ == ex && acc_ == ac && a.emode == em) {                                                 \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv3x3_t6_kernel<HT, 6, 3, mt, ex, ac, em>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((conv3x3_t6_kernel<HT, 6, 3, mt, ex, ac, em>), grid, dim3(T6_THREADS), q.lds, s, a);               \
    ok = true;                                                                                                            \
  }
  FAMI_T6_CASE(1, 1, false, 0) FAMI_T6_CASE(1, 1, false, 1) FAMI_T6_CASE(1, 1, true, 0) FAMI_T6_CASE(1, 1, false, 2) FAMI_T6_CASE(1, 1, true, 2)
  FAMI_T6_CASE(1, 0, false, 0) FAMI_T6_CASE(1, 0, false, 1) FAMI_T6_CASE(1, 0, true, 0) FAMI_T6_CASE(1, 0, false, 2) FAMI_T6_CASE(1, 0, true, 2)
  FAMI_T6_CASE(2, 1, false, 0) FAMI_T6_CASE(2, 1, false, 1) FAMI_T6_CASE(2, 1, true, 0) FAMI_T6_CASE(2, 1, false, 2) FAMI_T6_CASE(2, 1, true, 2)
  FAMI_T6_CASE(2, 0, false, 0) FAMI_T6_CASE(2, 0, false, 1) FAMI_T6_CASE(2, 0, true, 0) FAMI_T6_CASE(2, 0, false, 2) FAMI_T6_CASE(2, 0, true, 2)
#undef FAMI_T6_CASE
  if (!ok) return 0;
  return 1;
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
static int t7_launch(const T7Plan& q, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci,
                     int Co, int KC, int NTt, int sgn, int accumulate, hipStream_t s, const EpiBN& epi) {
  ConvT7Args a;
  a.e = epi; a.emode = epi.slots ? epi.mode : 0;
  a.x = x; a.wimg = wp; a.y = y; a.bias = bias;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.KC = KC; a.NTt = NTt; a.sgn = sgn; a.accumulate = accumulate;
  a.RB = q.RB; a.bands = q.bands; a.PW = W + 2; a.RG = (W + 2) * q.SG; a.q512 = 512 / a.RG; a.r512 = 512 % a.RG;
  a.nph = Ci / (8 * q.SG); a.TU = q.TU; a.npix = q.RB * W; a.REMP = q.TU > 8 * q.MT ? (q.TU - 8 * q.MT) * q.NT : 0; a.PI = q.PI; a.jpw = q.jpw;
  const dim3 grid(q.G, Co / (16 * q.NT));
  bool ok = false;
  const bool acc_ = accumulate != 0;
#define FAMI_T7_CASE_G(sg, nt, mt, ex, ac, em)                                                                            \
  if (!ok && q.SG == sg && q.MT == mt && q.EX == ex && acc_ == ac && a.emode == em) {                                  \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv3x3_t7_kernel<HT, sg, nt, mt, ex, ac, em>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((conv3x3_t7_kernel<HT, sg, nt, mt, ex, ac, em>), grid, dim3(T6_THREADS), q.lds, s, a);             \
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

// Returns 1 if launched, 0 if the shape is not eligible, < 0 on error.  half_kind: 0 bf16, 1 fp16.
int fami_try_conv3x3_t6(int half_kind, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci,
                        int Co, int KC, int NTt, int sgn, int relu, int accumulate, int out_f32, hipStream_t s,
                        const char* name, const EpiBN& epi, const XBN& xbn) {
  if (half_kind > 1 || xbn.on || (epi.slots && epi.mode != 1 && epi.mode != 2)) return 0;
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
  if (half_kind == 1) rc = t6_launch<f16_t>(q, x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, out_f32, s, epi);
  else rc = t6_launch<bf16_t>(q, x, wp, bias, y, N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, out_f32, s, epi);
  if (!rc) return 0;
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    fami_set_error(name, hipGetErrorString(err));
    return FAMI_EHIP;
  }
  return 1;
}
// 1: the 48-channel kernel takes it, 2: the phased kernel (48-channel phases), 3: the phased kernel with 32-channel phases, 0: neither
extern "C" int fami_conv_t6_eligible(int N, int H, int W, int Ci, int Co) {
  const T7Plan q7 = t7_plan(N, H, W, Ci, Co);
  if (q7.ok) return q7.SG == 4 ? 3 : 2;
  return t6_plan(N, H, W, Ci, Co).ok ? 1 : 0;
}
void fami_conv_t6_tune(int on) {
  if (on < 0) { g_use_t6 = 1; g_t6_rows = 0; g_t6_min_jobs = 96; g_t6_mt = 0; g_use_t7 = 1; g_t7_rows = 0; g_t7_target = 240; g_t7_c64 = 1; }
  else if (on == 8502 || on == 8503) g_t7_c64 = on - 8502;
  else if (on >= 8700 && on < 8999) g_t7_target = on - 8700;
  else if (on == 8500 || on == 8501) g_use_t7 = on - 8500;
  else if (on >= 8600 && on < 8700) g_t7_rows = on - 8600;
  else if (on >= 8200 && on <= 8202) g_t6_mt = on - 8200;
  else if (on == 8000 || on == 8001) g_use_t6 = on - 8000;
  else if (on >= 8400 && on < 8500) g_t6_min_jobs = on - 8400;
  else if (on >= 8100 && on < 8200) g_t6_rows = on - 8100;
}
