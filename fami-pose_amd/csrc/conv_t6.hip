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
//     (global_load_lds_dwordx4, per-lane source addresses into the packed fragment image) in the dense order;
//   * a workgroup owns a band of RB whole rows of one frame; its patch (RB + 2 rows, zero border columns, 96-byte
//     positions, unpadded: conflict-free for ds_read_b128, see below) is ONE contiguous LDS region filled by DMA too --
//     border and out-of-image granules are copied from a 16-byte zero constant, so there is no store phase, no VGPR
//     staging and no zeroing; every DMA of the job is issued up front, in the order the units need it;
//   * the band is multiplied in UNITS of two rows (144 pixels = 9 tiles at W = 72): unit 0 waits for rows 0-3 + the
//     weights (s_waitcnt vmcnt(n) on the wave's own queue, then a barrier), unit 1 for rows 4-5, unit 2 for the rest;
//     later units need no barrier at all;
//   * wave w owns tile w (all three channel tiles); the ninth tile's three channel tiles go to waves 0-2: per SIMD
//     (waves s, s + 4) 7 / 7 / 7 / 6 (tile, channel tile) pairs;
//   * a unit's results are written while the NEXT unit is multiplied (after its barrier), so no store sits in front of
//     a vmcnt wait.
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

__device__ __forceinline__ void t6_wait_vm(int n) {   // wave-uniform n: s_waitcnt vmcnt(n) (lgkmcnt / expcnt untouched)
  switch (n) {
    case 0: __builtin_amdgcn_s_waitcnt(0x0f70); break;
    case 1: __builtin_amdgcn_s_waitcnt(0x0f71); break;
    case 2: __builtin_amdgcn_s_waitcnt(0x0f72); break;
    case 3: __builtin_amdgcn_s_waitcnt(0x0f73); break;
    case 4: __builtin_amdgcn_s_waitcnt(0x0f74); break;
    case 5: __builtin_amdgcn_s_waitcnt(0x0f75); break;
    case 6: __builtin_amdgcn_s_waitcnt(0x0f76); break;
    case 7: __builtin_amdgcn_s_waitcnt(0x0f77); break;
    case 8: __builtin_amdgcn_s_waitcnt(0x0f78); break;
    case 9: __builtin_amdgcn_s_waitcnt(0x0f79); break;
    case 10: __builtin_amdgcn_s_waitcnt(0x0f7a); break;
    case 11: __builtin_amdgcn_s_waitcnt(0x0f7b); break;
    case 12: __builtin_amdgcn_s_waitcnt(0x0f7c); break;
    default: __builtin_amdgcn_s_waitcnt(0x0f70); break;
  }
}

// G: 16-byte granules per pixel (Ci / 8); NT: channel tiles per workgroup; MT: own pixel tiles per wave (a unit is 2 MT rows);
// EX: 1 if the unit has tiles past the 8 MT-th; ACC: y += result; EM: EpiBN mode (0 | 1)
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
  auto emit1 = [&](f32x4 v, H* yp, const f32x4& b4, const f32x4& k4, f32x4& s, f32x4& q) {
    v += b4;
    if (ACC) v += ld4(yp);
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
      for (int nt = 0; nt < NT; ++nt) emit1(sv[m][nt], yb + oown[m] + nt * 16, bias4[nt], ek[nt], es[nt], eq[nt]);
    if (has_e) emit1(sve, yb + oex, biase, eke, ese, eqe);
  };

  // ---- units
  const int ustep = UR * PW * PSB;
  T6_STAMP();
  for (int u = 0; u < nunits; ++u) {
    T6_STAMP();
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's share of the unit's rows (and of the weights) has landed; the stores of unit u - 2 are long done
    __builtin_amdgcn_s_barrier();         // ... and everybody else's
    asm volatile("" ::: "memory");        // no LDS read of the unit may move (or be hoisted out of the loop) above the wait
    T6_STAMP();
    if (u + 1 < nunits) dma_rows(UR * (u + 2) + 2);
    if (u == 0) load_consts();
    if (u > 0) emit(u - 1);
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

  // ---- EpiBN mode 1: per-channel sums of the workgroup -> fp64 slot rows
  if (EM == 1) {
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
      if (st == 0 && job == 0) bn_slots_pivot(e->slots, eC)[co] = e->pivot_src ? e->pivot_src[co] : 0.f;
    }
  }
}

// ---- plan + launch
static int g_use_t6 = 1;        // fami_conv_tune_lds(8000 / 8001): off / on
static int g_t6_rows = 0;       // fami_conv_tune_lds(8100 + RB): force the rows per band (benchmarks)
static int g_t6_min_jobs = 96;
static int g_t6_mt = 0;         // fami_conv_tune_lds(8201 / 8202): units of two / four rows (0: four where the band allows)
static long long* g_t6_dbg = nullptr;
extern "C" void fami_conv_t6_debug(void* buf) { g_t6_dbg = reinterpret_cast<long long*>(buf); }
         // fami_conv_tune_lds(8200 + n): fragment sets in flight (2 .. 4)  // fami_conv_tune_lds(8400 + n): only launches of >= n jobs

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
    if (wbytes + (size_t)pj * 8192 + 8 * (NT + 1) * 32 * 4 > lds_cap) break;
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
  const bool acc_ = accumulate != 0, em1 = a.emode == 1;
  const int ex_ = q.TU > 8 * q.MT ? 1 : 0;
#define FAMI_T6_CASE(mt, ex, ac, em)                                                                                      \
  if (!ok && q.MT == mt && ex_ == ex && acc_ == ac && em1 == (em == 1)) {                                                 \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv3x3_t6_kernel<HT, 6, 3, mt, ex, ac, em>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((conv3x3_t6_kernel<HT, 6, 3, mt, ex, ac, em>), grid, dim3(T6_THREADS), q.lds, s, a);               \
    ok = true;                                                                                                            \
  }
  FAMI_T6_CASE(1, 1, false, 0) FAMI_T6_CASE(1, 1, false, 1) FAMI_T6_CASE(1, 1, true, 0)
  FAMI_T6_CASE(1, 0, false, 0) FAMI_T6_CASE(1, 0, false, 1) FAMI_T6_CASE(1, 0, true, 0)
  FAMI_T6_CASE(2, 1, false, 0) FAMI_T6_CASE(2, 1, false, 1) FAMI_T6_CASE(2, 1, true, 0)
  FAMI_T6_CASE(2, 0, false, 0) FAMI_T6_CASE(2, 0, false, 1) FAMI_T6_CASE(2, 0, true, 0)
#undef FAMI_T6_CASE
  if (!ok) return 0;
  return 1;
}

// Returns 1 if launched, 0 if the shape is not eligible, < 0 on error.  half_kind: 0 bf16, 1 fp16.
int fami_try_conv3x3_t6(int half_kind, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci,
                        int Co, int KC, int NTt, int sgn, int relu, int accumulate, int out_f32, hipStream_t s,
                        const char* name, const EpiBN& epi, const XBN& xbn) {
  if (half_kind > 1 || xbn.on || (epi.slots && epi.mode != 1)) return 0;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (reinterpret_cast<uintptr_t>(wp) & 15) != 0) return 0;
  if (out_f32 || relu || (accumulate && epi.slots)) return 0;                      // (forward-only fused ReLU / fp32 heatmap outputs stay on conv_t4)
  if ((long)H * W * Ci * 2 >= (1L << 31) || (long)9 * KC * NTt * 1024 >= (1L << 31)) return 0;
  const T6Plan q = t6_plan(N, H, W, Ci, Co);
  if (!q.ok || KC * 32 < Ci || NTt * 16 < Co) return 0;
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
extern "C" int fami_conv_t6_eligible(int N, int H, int W, int Ci, int Co) { return t6_plan(N, H, W, Ci, Co).ok; }
void fami_conv_t6_tune(int on) {
  if (on < 0) { g_use_t6 = 1; g_t6_rows = 0; g_t6_min_jobs = 96; g_t6_mt = 0; }
  else if (on >= 8200 && on <= 8202) g_t6_mt = on - 8200;
  else if (on == 8000 || on == 8001) g_use_t6 = on - 8000;
  else if (on >= 8400 && on < 8900) g_t6_min_jobs = on - 8400;
  else if (on >= 8100 && on < 8200) g_t6_rows = on - 8100;
}
