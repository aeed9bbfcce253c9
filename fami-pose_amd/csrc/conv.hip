// Convolution family for the HRNet branches and the alignment head, fp32,
// NHWC activations, implicit GEMM on the CDNA4 f32-input matrix core
// (v_mfma_f32_16x16x4_f32: exact f32, one rounding per product).
//
// Replaces: every nn.Conv2d on the hot path of the reference
//   posetimation/backbones/hrnet.py:569-629 (293 convs of HRNet-W48),
//   posetimation/layers/basic_model.py:21-23,25-63,66-113 and
//   posetimation/layers/basic_layer.py:18-19 (conv_bn_relu.conv), plus their
//   autograd (dgrad / wgrad).
//
// Mapping (forward): GEMM M = N*Ho*Wo output pixels (rows), N = Cout (cols),
// K = kh*kw*Cin walked tap-major.  A wave owns MT 16-pixel row tiles x NT
// 16-channel column tiles.  A-fragments are read straight from the NHWC
// activation (each lane: one pixel, 4 consecutive channels = one 16-byte load,
// zero outside the image); B-fragments come from a weight image pre-packed in
// fragment order (pack_w_kernel) so a wave-load is one contiguous 1 KiB line
// that stays L2-resident across the grid.  The K order inside a 16-channel
// chunk is permuted identically on both operands so the 4 values of a lane's
// 16-byte load feed 4 consecutive MFMAs.  No LDS, no barriers: at the f32 MFMA
// rate (32 cycles / instruction / SIMD) operand traffic is ~25 B/clk/CU.
// dgrad is the same kernel with the transposed position map (MODE 1).
#include "common.h"
#include <type_traits>
#include <string.h>

#include "conv_epi.h"
extern "C" int fami_conv2d_fwd_bnin_ok(int N, int H, int W, int Ci, int Co);      // conv_t6.hip
// conv_pair.hip
#include "conv_pair.h"
int fami_pair_launch(const PairCapture& c, hipStream_t s);


struct ConvArgs {
  EpiBN e;              // read late (epi_late); `emode` below is the one field the top of the kernel looks at
  int emode;            // 0 | e.mode when e.slots is set
  const float* x;       // GEMM input activation  [N,Hi,Wi,Ci]
  const float* wp;      // packed weights [taps][KC][NTt][64][4]
  float* y;             // GEMM output activation [N,Ho,Wo,Co]
  const float* bias;    // [Co] or null
  const float* addend;  // [N,Ho,Wo,Co] or null (added before relu)
  int N, Hi, Wi, Ci, Ho, Wo, Co;
  int kh, kw, sh, pad, dil;  // sh = log2(stride)
  int KC, NTt, relu, accumulate, P;
  unsigned x_bytes, wp_bytes;  // buffer-descriptor extents (out-of-range lanes read 0)
  int xcd;                     // 1: XCD-contiguous tile order (xcd_tile)
  int prio;                    // > 0: raise the waves' issue priority (s_setprio): f32 MFMA kernels against concurrent memory-bound lanes
  int par;                     // stride-2 dgrad (f32): waves own pixels of ONE parity class and walk only its taps
};

// f32 weight image = [16x16-tile image][32x32-tile image]:
//   first  packed[tap][kc16][nt16][lane][t] : K index kc16*16 + (lane>>4)*4 + t, N index nt16*16 + (lane&15)
//   second packed[tap][kc8 ][nt32][lane][t] : K index kc8*8   + (lane>>5)*4 + t, N index nt32*32 + (lane&31)
//          (operand layout of v_mfma_f32_32x32x2_f32; present for 1x1 convs with N >= 64 and K % 4 == 0 -- the only
//          shapes where the 32x32 tile measured faster than the 16x16 one, tools/bench_c32.py)
// mode 0: K = Cin, N = Cout ; mode 1 (dgrad): K = Cout, N = Cin.
__host__ __device__ static inline long pack16_elems(int kd, int nd, int taps) {
  return (long)taps * ((kd + 15) / 16) * ((nd + 15) / 16) * 256;
}
__host__ __device__ static inline long pack32_elems(int kd, int nd, int taps) {
  return (taps == 1 && nd >= 64 && kd % 4 == 0) ? (long)taps * ((kd + 7) / 8) * ((nd + 31) / 32) * 256 : 0;
}
__device__ __forceinline__ float pack_f32_elem(const float* __restrict__ w, long i, int Co, int Ci, int taps, int mode) {
  const int kd = mode == 0 ? Ci : Co, nd = mode == 0 ? Co : Ci;
  const long n16 = pack16_elems(kd, nd, taps);
  int kidx, nidx, tap;
  if (i < n16) {
    const int KC = (kd + 15) / 16, NTt = (nd + 15) / 16;
    const int t = (int)(i & 3), lane = (int)((i >> 2) & 63);
    long r = i >> 8;
    const int nt = (int)(r % NTt);
    r /= NTt;
    const int kc = (int)(r % KC);
    tap = (int)(r / KC);
    kidx = kc * 16 + (lane >> 4) * 4 + t;
    nidx = nt * 16 + (lane & 15);
  } else {
    const long j = i - n16;
    const int KC = (kd + 7) / 8, NTt = (nd + 31) / 32;
    const int t = (int)(j & 3), lane = (int)((j >> 2) & 63);
    long r = j >> 8;
    const int nt = (int)(r % NTt);
    r /= NTt;
    const int kc = (int)(r % KC);
    tap = (int)(r / KC);
    kidx = kc * 8 + (lane >> 5) * 4 + t;
    nidx = nt * 32 + (lane & 31);
  }
  const int co = mode == 0 ? nidx : kidx, ci = mode == 0 ? kidx : nidx;
  return (co < Co && ci < Ci) ? w[((long)co * Ci + ci) * taps + tap] : 0.f;
}
__global__ void pack_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int taps, int mode) {
  const int kd = mode == 0 ? Ci : Co, nd = mode == 0 ? Co : Ci;
  const long total = pack16_elems(kd, nd, taps) + pack32_elems(kd, nd, taps);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    wp[i] = pack_f32_elem(w, i, Co, Ci, taps, mode);
}

#define FAMI_OOB 0x80000000u  // byte offset beyond any tensor: the buffer unit returns 0 for it

// KS = number of K partitions among the 4 waves of a workgroup (split-K for the low-resolution
// branches, whose pixel count alone cannot fill 1024 SIMDs); partial accumulators meet in LDS.
// LIN = 1 (stride 1, Ci % 16 == 0, <= 25 taps): the K-loop bookkeeping leaves the vector pipe, which the f32-input MFMA
// shares with the VALU on gfx950.  A lane's pixel never changes and the input address of tap (dy, dx) is the pixel's
// own address plus the wave-uniform (dy*Wi + dx)*Ci, so a tap switch is one v_add plus a select on a per-lane tap
// validity mask built once in the prologue (was: two bounds tests and two quarter-rate 64-bit multiply-adds), and the
// per-iteration K offsets (kc*64 into the pixel, the weight block) travel in the buffer instructions' SGPR offset
// instead of one v_add per load and a channel-tail test.
template <int MT, int NT, int MODE, int VEC, int KS, int ST, int LIN = 0>
__global__ __launch_bounds__(256) void conv_igemm_f32(ConvArgs p) {
  // The exact-f32 MFMA issues on the vector ALUs (gfx950): a BatchNorm / elementwise kernel of another stream lane that is
  // resident on the same SIMD takes its VALU slots out of this kernel's matrix rate (in-step 77 us against 67 us alone).
  // With a raised priority the arbiter serves these waves first; the memory-bound lanes fill the remaining slots.
  if (p.prio) __builtin_amdgcn_s_setprio(3);
  __shared__ float red[KS > 1 ? (4 - 4 / KS) * MT * NT * 256 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps the K-loop bookkeeping scalar
  const int row = lane & 15, kq = lane >> 4;
  const int kpart = wave % KS, mgrp = wave / KS;
  int bx, by;
  xcd_tile(p.xcd, bx, by);
  // The input gradient of a stride-2 convolution: an output pixel (y, x) only receives the taps with (y + pad - ky*dil)
  // and (x + pad - kx*dil) even -- for a 3x3 kernel 1, 2, 2 or 4 of the 9 taps depending on the parities of y and x.  Walking
  // all taps with zero operands (the general path) spends 3/4 of the MFMAs on zeros.  With p.par the pixel tiles are
  // dealt per parity class (class-major tile order: 4 runs of tiles), a wave's pixels share one class, and its K loop
  // walks that class's taps only.
  const bool par = MODE == 1 && p.par;
  const int tix = bx * (4 / KS) + mgrp;   // pixel-tile index of this wave
  int m0 = tix * (MT * 16);
  bool active = m0 < p.P;  // wave-uniform
  int ca = 0, cb = 0, cHa = 0, cWb = 0, cP = 0, crem = 0;   // par: row / column parity, class extent, tile within the class
  if (par) {
    const int H0 = (p.Ho + 1) >> 1, H1 = p.Ho >> 1, W0 = (p.Wo + 1) >> 1, W1 = p.Wo >> 1;
    int rem = tix, c = 0;
    for (; c < 4; ++c) {
      cHa = (c >> 1) ? H1 : H0;
      cWb = (c & 1) ? W1 : W0;
      cP = p.N * cHa * cWb;
      const int tl = (cP + MT * 16 - 1) / (MT * 16);
      if (rem < tl) break;
      rem -= tl;
    }
    active = c < 4;
    ca = c >> 1;
    cb = c & 1;
    crem = rem;
  }
  const int emode = p.emode;   // kernel argument: uniform
  if (KS == 1 && !active && !emode) return;      // (with statistics the wave still meets the others at the barrier)
  const int ntg0 = by * NT;
  const int HoWo = p.Ho * p.Wo;

  int pn[MT], py[MT], px[MT];
  bool pv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (par) {
      const int ml = (crem * MT + mt) * 16 + row;
      pv[mt] = active && ml < cP;
      const int mm = pv[mt] ? ml : 0;
      const int hw = cHa * cWb;
      const int n = mm / hw, r = mm - n * hw;
      const int yy = r / cWb;
      pn[mt] = n;
      py[mt] = 2 * yy + ca;
      px[mt] = 2 * (r - yy * cWb) + cb;
      continue;
    }
    const int m = m0 + mt * 16 + row;
    pv[mt] = m < p.P;
    const int mm = pv[mt] ? m : 0;
    const int n = mm / HoWo, r = mm - n * HoWo;
    const int oy = r / p.Wo;
    pn[mt] = n;
    py[mt] = oy;
    px[mt] = r - oy * p.Wo;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // branch-free operand fetch through buffer descriptors: a lane outside the image (padding), past the
  // tensor or in a channel tail carries the out-of-range offset and the hardware returns 0 for it
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.wp_bytes, 0x00020000);
  unsigned aoff[MT];
  const int taps = p.kh * p.kw;
  int tap = 0, kc = kpart;
  int lky = 0, lkx = 0;   // LIN: (ky, kx) of `tap`, kept in step with it (no division per tap switch)
  // par: is tap tp one of this wave's parity class?  (wave-uniform)
  auto tap_ok = [&](int tp) {
    const int ky = tp / p.kw, kx = tp - ky * p.kw;
    return (((ca + p.pad - ky * p.dil) | (cb + p.pad - kx * p.dil)) & 1) == 0;
  };
  auto next_tap = [&]() {
    ++tap;
    if (par) {
      while (tap < taps && !tap_ok(tap)) ++tap;
    }
    if (LIN) {
      if (++lkx == p.kw) {
        lkx = 0;
        ++lky;
      }
    }
  };
  int ntap_mine = taps;
  if (par) {
    ntap_mine = 0;
    for (int t = 0; t < taps; ++t) ntap_mine += tap_ok(t) ? 1 : 0;
    while (tap < taps && !tap_ok(tap)) ++tap;   // first tap of the class
  }
  while (kc >= p.KC) {
    kc -= p.KC;
    next_tap();
  }
  unsigned boff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) boff[nt] = (unsigned)(min(ntg0 + nt, p.NTt - 1) * 256 + lane * 4) * 4u;

  // LIN: per-lane pixel address and tap validity (bit t set = tap t reads padding; bits >= taps stay set: drain)
  unsigned lbase[MT], linv[MT];
  if (LIN) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      lbase[mt] = (unsigned)((((pn[mt] * p.Hi + py[mt]) * p.Wi + px[mt]) * p.Ci + kq * 4) * 4);
      linv[mt] = 0xffffffffu;
    }
    int ky = 0, kx = 0;
    for (int t = 0; t < taps; ++t) {
      const int dy = MODE == 0 ? ky * p.dil - p.pad : p.pad - ky * p.dil;
      const int dx = MODE == 0 ? kx * p.dil - p.pad : p.pad - kx * p.dil;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bool v = pv[mt] && (unsigned)(py[mt] + dy) < (unsigned)p.Hi && (unsigned)(px[mt] + dx) < (unsigned)p.Wi;
        linv[mt] &= ~((v ? 1u : 0u) << t);
      }
      if (++kx == p.kw) {
        kx = 0;
        ++ky;
      }
    }
  }

  auto tap_setup = [&](int tp) {
    if (LIN) {
      const int dy = MODE == 0 ? lky * p.dil - p.pad : p.pad - lky * p.dil;
      const int dx = MODE == 0 ? lkx * p.dil - p.pad : p.pad - lkx * p.dil;
      const unsigned delta = (unsigned)((dy * p.Wi + dx) * p.Ci * 4);
      const int tb = tp < 31 ? tp : 31;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const unsigned m = (unsigned)((int)(linv[mt] << (31 - tb)) >> 31);   // all ones: padding / drain
        aoff[mt] = ((lbase[mt] + delta) & ~m) | (FAMI_OOB & m);
      }
      return;
    }
    const int ky = tp / p.kw, kx = tp - ky * p.kw;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      bool v = pv[mt] && tp < taps;  // past the last tap (pipeline drain): everything reads 0
      int iy, ix;
      if (MODE == 0) {
        iy = (py[mt] << p.sh) - p.pad + ky * p.dil;
        ix = (px[mt] << p.sh) - p.pad + kx * p.dil;
        v = v && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      } else {
        const int ty = py[mt] + p.pad - ky * p.dil, tx = px[mt] + p.pad - kx * p.dil;
        iy = ty >> p.sh;
        ix = tx >> p.sh;
        v = v && ty >= 0 && tx >= 0 && (iy << p.sh) == ty && (ix << p.sh) == tx && iy < p.Hi && ix < p.Wi;
      }
      aoff[mt] = v ? (unsigned)((((pn[mt] * p.Hi + iy) * p.Wi + ix) * p.Ci + kq * 4) * 4) : FAMI_OOB;
    }
  };

  auto load = [&](f32x4(&a)[MT], f32x4(&b)[NT]) {
    if (LIN) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        a[mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, aoff[mt], kc * 64, 0));
      const int wbs = (tap * p.KC + kc) * p.NTt * 1024;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        b[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, boff[nt], wbs, 0));
      kc += KS;
      if (kc >= p.KC) {
        do {
          kc -= p.KC;
          next_tap();
        } while (kc >= p.KC);
        tap_setup(tap);
      }
      return;
    }
    const int cbase = kc * 16 + kq * 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (VEC) {
        const unsigned o = cbase < p.Ci ? aoff[mt] + kc * 64 : FAMI_OOB;
        a[mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, o, 0, 0));
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned o = cbase + q < p.Ci ? aoff[mt] + kc * 64 + q * 4 : FAMI_OOB;
          a[mt][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, o, 0, 0));
        }
      }
    }
    const unsigned wb = (unsigned)((tap * p.KC + kc) * p.NTt) * 1024u;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      b[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wb + boff[nt], 0, 0));
    kc += KS;
    if (kc >= p.KC) {
      do {
        kc -= p.KC;
        next_tap();
      } while (kc >= p.KC);
      tap_setup(tap);
    }
  };

  // LIN: a second accumulator set takes the odd K quarters, so an accumulator is reused every 2*MT*NT MFMAs instead of
  // every MT*NT: register-only loops issue v_mfma_f32_16x16x4_f32 at 126 TFLOP/s with 4 and 139 with 8 independent
  // accumulators per wave at 8 waves per SIMD (tools/probes/mfma_peak2.hip)
  f32x4 accB[LIN ? MT : 1][LIN ? NT : 1];
  if (LIN) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) accB[LIN ? mt : 0][LIN ? nt : 0] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto mma = [&](const f32x4(&a)[MT], const f32x4(&b)[NT]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if (LIN && (t & 1))
            accB[LIN ? mt : 0][LIN ? nt : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][t], b[nt][t], accB[LIN ? mt : 0][LIN ? nt : 0], 0, 0, 0);
          else
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][t], b[nt][t], acc[mt][nt], 0, 0, 0);
        }
  };

  f32x4 af[ST][MT], bf[ST][NT];
  const int Tall = ntap_mine * p.KC;
  const int T = active ? (Tall - kpart + KS - 1) / KS : 0;  // iterations owned by this wave
  // ST-stage register pipeline (ST-1 operand sets in flight: the loop is bound by memory latency x concurrency, not
  // by MFMA issue) with an unconditional body: loads issued past the last iteration carry out-of-range offsets
  // (zeros), so a T that is not a multiple of ST costs MFMA groups on zeros instead of branches in the loop
  if (T > 0) {
    tap_setup(tap);
#pragma unroll
    for (int st = 0; st < ST - 1; ++st) load(af[st], bf[st]);
    for (int it = 0; it < T; it += ST) {
#pragma unroll
      for (int st = 0; st < ST; ++st) {
        load(af[(st + ST - 1) % ST], bf[(st + ST - 1) % ST]);
        mma(af[st], bf[st]);
      }
    }
  }

  if (LIN) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] += accB[LIN ? mt : 0][LIN ? nt : 0];
  }
  if (KS > 1) {
    // partial sums of k-parts 1..KS-1 go through LDS to the k-part-0 wave of the same pixel group
    if (kpart > 0) {
      float* dst = red + ((mgrp * (KS - 1)) + (kpart - 1)) * (MT * NT * 256);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[((mt * NT + nt) * 4 + r) * 64 + lane] = acc[mt][nt][r];
    }
    __syncthreads();
    if (kpart > 0 || !active) return;
#pragma unroll
    for (int k = 0; k < KS - 1; ++k) {
      const float* src = red + ((mgrp * (KS - 1)) + k) * (MT * NT * 256);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mt][nt][r] += src[((mt * NT + nt) * 4 + r) * 64 + lane];
    }
  }

  // epilogue: D row = kq*4 + r (pixel), col = row (channel)
  float bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = (ntg0 + nt) * 16 + row;
    bv[nt] = (p.bias && co < p.Co) ? p.bias[co] : 0.f;
  }
  if (emode) {
    // EpiBN: a lane owns ONE channel per tile here, so the per-channel sums are register accumulations over its pixels;
    // channel tile by channel tile, so one tile's parameters are live at a time
    EpiPtr e = epi_late(__builtin_offsetof(ConvArgs, e));
    __shared__ float ered[4][NT * 32];
    const float* ez = reinterpret_cast<const float*>(e->z);
    const float* eyr = reinterpret_cast<const float*>(e->yr);
    const int erelu = e->relu, eC = e->C;
    double* srow = e->slots + (long)((KS == 1 ? bx : tix) % e->ns) * 2 * eC;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = (ntg0 + nt) * 16 + row;
      const int cc = min(co, p.Co - 1);
      float es = 0.f, eq = 0.f, ek = 0.f, emu = 0.f, eis = 0.f, esc = 0.f, esf = 0.f;
      if (emode == 1) {
        ek = e->pivot_src ? e->pivot_src[cc] : 0.f;
      } else {
        emu = e->mean[cc];
        eis = e->invstd[cc];
        epi_scale_shift(emu, eis, e->gamma[cc], e->beta[cc], esc, esf);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int pm_mine = pv[mt] ? (pn[mt] * p.Ho + py[mt]) * p.Wo + px[mt] : -1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int m = m0 + mt * 16 + kq * 4 + r;
          if (par) m = __shfl(pm_mine, kq * 4 + r);
          if (m < 0 || m >= p.P || !active || co >= p.Co) continue;
          const long idx = (long)m * p.Co + co;
          float v = acc[mt][nt][r] + bv[nt];
          if (p.accumulate) v += p.y[idx];
          if (emode == 1) {
            const float d = v - ek;
            es += d;
            eq = __builtin_fmaf(d, d, eq);
          } else {
            const float zz = ez[idx];
            bool keep = true;
            if (erelu == 1) keep = eyr[idx] > 0.f;
            else if (erelu == 2) keep = __builtin_fmaf(zz, esc, esf) > 0.f;
            v = keep ? v : 0.f;
            es += v;
            eq = __builtin_fmaf(v, (zz - emu) * eis, eq);
          }
          p.y[idx] = v;
        }
      }
      // lanes l, l^16, l^32, l^48 hold the same channel: fold them, then the waves of the workgroup through LDS
      es += __shfl_xor(es, 16, 64);
      es += __shfl_xor(es, 32, 64);
      eq += __shfl_xor(eq, 16, 64);
      eq += __shfl_xor(eq, 32, 64);
      if (KS == 1) {
        if (kq == 0) {
          ered[wave][nt * 32 + row] = es;
          ered[wave][nt * 32 + 16 + row] = eq;
        }
      } else if (kq == 0 && co < p.Co) {   // KS > 1: only the k-part-0 waves get here, each adds its own sums
        unsafeAtomicAdd(srow + co, (double)es);
        unsafeAtomicAdd(srow + eC + co, (double)eq);
        if (emode == 1 && tix == 0) bn_slots_pivot(e->slots, eC)[co] = ek;
      }
    }
    if (KS == 1) {
      __syncthreads();
      const int t = threadIdx.x;
      if (t < NT * 32) {
        const int nt = t >> 5, st = (t >> 4) & 1;
        const int co = (ntg0 + nt) * 16 + (t & 15);
        if (co < p.Co) {
          const float v = (ered[0][t] + ered[1][t]) + (ered[2][t] + ered[3][t]);
          unsafeAtomicAdd(srow + st * eC + co, (double)v);
          if (emode == 1 && st == 0 && bx == 0) bn_slots_pivot(e->slots, eC)[co] = e->pivot_src ? e->pivot_src[co] : 0.f;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    // par: the flat output pixel of tile row j lives in lane j (any kq): fetch it from there
    const int pm_mine = pv[mt] ? (pn[mt] * p.Ho + py[mt]) * p.Wo + px[mt] : -1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int m = m0 + mt * 16 + kq * 4 + r;
      if (par) {
        m = __shfl(pm_mine, kq * 4 + r);
        if (m < 0) continue;
      }
      if (m >= p.P) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = (ntg0 + nt) * 16 + row;
        if (co >= p.Co) continue;
        const long idx = (long)m * p.Co + co;
        float v = acc[mt][nt][r] + bv[nt];
        if (p.addend) v += p.addend[idx];
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.accumulate) v += p.y[idx];
        p.y[idx] = v;
      }
    }
  }
}

// ------------------------------------------------------------------ f32 implicit GEMM on 32x32 tiles
// Same direct (global -> register) scheme on v_mfma_f32_32x32x2_f32: measured on this part the 32x32x2 form issues at
// 99 % of the 157 TFLOP/s f32 MFMA peak from a single wave, the 16x16x4 form at 60-90 % depending on occupancy
// (tools/probes/mfma_peak2.hip), and a 32x32 tile needs half the operand bytes per FLOP.  A wave owns 32 pixels x NT
// 32-channel tiles; MFMA A = weights (rows = output channels), B = activations (cols = pixels), so a lane ends with
// groups of 4 consecutive output channels of one pixel (16-byte stores).  A lane's fragment is 4 consecutive channels
// (one 16-byte load): lanes 0-31 carry channels kc*8+0..3, lanes 32-63 channels kc*8+4..7 of the 8-channel K step.
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NT, int MODE, int KS>
__global__ __launch_bounds__(256) void conv_igemm32_f32(ConvArgs p) {
  if (p.prio) __builtin_amdgcn_s_setprio(3);   // see conv_igemm_f32
  __shared__ float red[KS > 1 ? (4 - 4 / KS) * NT * 1024 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, kh2 = lane >> 5;
  const int kpart = wave % KS, mgrp = wave / KS;
  int bx, by;
  xcd_tile(p.xcd, bx, by);
  const int m0 = (bx * (4 / KS) + mgrp) * 32;
  const bool active = m0 < p.P;  // wave-uniform
  if (KS == 1 && !active) return;
  const int ntg0 = by * NT;
  const int HoWo = p.Ho * p.Wo;
  const int m = m0 + col;
  const bool pv = m < p.P;
  const int mm = pv ? m : 0;
  const int pn = mm / HoWo, rr = mm - pn * HoWo;
  const int py = rr / p.Wo, px = rr - py * p.Wo;

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.wp_bytes, 0x00020000);
  const int taps = p.kh * p.kw;
  int tap = 0, kc = kpart;  // p.KC = 8-channel K steps, p.NTt = 32-channel tiles (set by the host for this kernel)
  while (kc >= p.KC) {
    kc -= p.KC;
    ++tap;
  }
  unsigned boff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) boff[nt] = (unsigned)(min(ntg0 + nt, p.NTt - 1) * 256 + lane * 4) * 4u;
  unsigned aoff = FAMI_OOB;
  auto tap_setup = [&](int tp) {
    const int ky = tp / p.kw, kx = tp - ky * p.kw;
    bool v = pv && tp < taps;
    int iy, ix;
    if (MODE == 0) {
      iy = (py << p.sh) - p.pad + ky * p.dil;
      ix = (px << p.sh) - p.pad + kx * p.dil;
      v = v && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
    } else {
      const int ty = py + p.pad - ky * p.dil, tx = px + p.pad - kx * p.dil;
      iy = ty >> p.sh;
      ix = tx >> p.sh;
      v = v && ty >= 0 && tx >= 0 && (iy << p.sh) == ty && (ix << p.sh) == tx && iy < p.Hi && ix < p.Wi;
    }
    aoff = v ? (unsigned)((((pn * p.Hi + iy) * p.Wi + ix) * p.Ci + kh2 * 4) * 4) : FAMI_OOB;
  };
  auto load = [&](f32x4& a, f32x4(&b)[NT]) {
    const unsigned o = (kc * 8 + kh2 * 4) < p.Ci ? aoff + kc * 32 : FAMI_OOB;
    a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, o, 0, 0));
    const unsigned wb = (unsigned)((tap * p.KC + kc) * p.NTt) * 1024u;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      b[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wb + boff[nt], 0, 0));
    kc += KS;
    if (kc >= p.KC) {
      do {
        kc -= p.KC;
        ++tap;
      } while (kc >= p.KC);
      tap_setup(tap);
    }
  };
  auto mma = [&](const f32x4& a, const f32x4(&b)[NT]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[nt][t], a[t], acc[nt], 0, 0, 0);
  };
  f32x4 a0, a1, b0[NT], b1[NT];
  const int Tall = taps * p.KC;
  const int T = active ? (Tall - kpart + KS - 1) / KS : 0;
  if (T > 0) {
    tap_setup(tap);
    load(a0, b0);
    for (int it = 0; it < T; it += 2) {
      load(a1, b1);
      mma(a0, b0);
      load(a0, b0);
      mma(a1, b1);
    }
  }
  if (KS > 1) {
    if (kpart > 0) {
      float* dst = red + ((mgrp * (KS - 1)) + (kpart - 1)) * (NT * 1024);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(nt * 16 + r) * 64 + lane] = acc[nt][r];
    }
    __syncthreads();
    if (kpart > 0 || !active) return;
#pragma unroll
    for (int k = 0; k < KS - 1; ++k) {
      const float* src = red + ((mgrp * (KS - 1)) + k) * (NT * 1024);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] += src[(nt * 16 + r) * 64 + lane];
    }
  }
  // D row (output channel) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31 (pixel)
  if (!pv) return;
  const bool cvec = (p.Co & 3) == 0;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co0 = (ntg0 + nt) * 32 + 8 * q + 4 * kh2;
      if (co0 >= p.Co) continue;
      f32x4 v = {acc[nt][4 * q], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]};
      const long idx = (long)m * p.Co + co0;
      if (cvec) {
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + co0);
        if (p.addend) v += *reinterpret_cast<const f32x4*>(p.addend + idx);
        if (p.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (p.accumulate) v += *reinterpret_cast<const f32x4*>(p.y + idx);
        *reinterpret_cast<f32x4*>(p.y + idx) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (co0 + r >= p.Co) continue;
          float u = v[r] + (p.bias ? p.bias[co0 + r] : 0.f);
          if (p.addend) u += p.addend[idx + r];
          if (p.relu) u = fmaxf(u, 0.f);
          if (p.accumulate) u += p.y[idx + r];
          p.y[idx + r] = u;
        }
      }
    }
  }
}

// ------------------------------------------------------------------ 16-bit implicit GEMM (fwd / dgrad)
// bf16 or fp16 activations and weights (template H), fp32 accumulation on v_mfma_f32_16x16x32_{bf16,f16}.  Same mapping as the f32 kernel with
// the operand roles swapped: the MFMA A operand is the weight fragment (rows = 16 output channels) and the B operand
// the activation fragment (cols = 16 pixels), so a lane ends up with 4 CONSECUTIVE output channels of one pixel and
// the epilogue is one 8-byte (bf16) or 16-byte (f32) store per tile instead of four scattered scalars.
// K is walked tap-major in chunks of 32 channels; a lane's fragment is 8 consecutive channels = one 16-byte load.
struct ConvArgsH {
  EpiBN e;
  int emode;
  const void* x;     // GEMM input activation  [N,Hi,Wi,Ci] (16-bit storage type H)
  const void* wp;    // packed weights [taps][KC][NTt][64][8] (H)
  void* y;           // GEMM output activation [N,Ho,Wo,Co], H or f32 (out_f32)
  const float* bias; // [Co] or null
  int N, Hi, Wi, Ci, Ho, Wo, Co;
  int kh, kw, sh, pad, dil;
  int KC, NTt, relu, accumulate, P, out_f32;
  unsigned x_bytes, wp_bytes;
  int xcd;
  int par;           // stride-2 dgrad by parity class (see ConvArgs)
};

// packed[tap][kc][nt][lane][j] ; mode 0: K = Cin, N = Cout ; mode 1 (dgrad): K = Cout, N = Cin
template <typename H>
__global__ void pack_w_h_kernel(const float* __restrict__ w, H* __restrict__ wp, int Co, int Ci, int taps,
                                int KC, int NTt, int mode) {
  const long total = (long)taps * KC * NTt * 512;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    long r = i >> 9;
    const int nt = (int)(r % NTt);
    r /= NTt;
    const int kc = (int)(r % KC), tap = (int)(r / KC);
    const int kidx = kc * 32 + (lane >> 4) * 8 + j, nidx = nt * 16 + (lane & 15);
    const int co = mode == 0 ? nidx : kidx, ci = mode == 0 ? kidx : nidx;
    float v = 0.f;
    if (co < Co && ci < Ci) v = w[((long)co * Ci + ci) * taps + tap];
    wp[i] = (H)v;
  }
}

// All weight images of a step in ONE launch: descriptor i = {src element offset (fp32 parameter arena), dst element
// offset (packed arena), Co, Ci, taps, mode}; blockIdx.y walks the descriptors.  Replaces ~600 pack launches per step.
struct PackDesc { long src, dst; int Co, Ci, taps, mode; };
template <typename T>
__global__ __launch_bounds__(256) void pack_w_batch_kernel(const float* __restrict__ params, T* __restrict__ packed,
                                                           const PackDesc* __restrict__ desc) {
  const PackDesc d = desc[blockIdx.y];
  const float* w = params + d.src;
  T* wp = packed + d.dst;
  const int kd = d.mode == 0 ? d.Ci : d.Co, nd = d.mode == 0 ? d.Co : d.Ci;
  constexpr int KG = sizeof(T) == 4 ? 16 : 32, FR = sizeof(T) == 4 ? 4 : 8;  // channels per K group / elements per lane fragment
  if (d.taps > 1 && d.taps <= 9) {
    // 3x3 images (every byte but ~2 % of a step's packing): OIHW keeps the taps of one (co, ci) pair adjacent, the
    // fragment image wants a tap's [K group][channel tile] blocks adjacent -- a stride-`taps` gather if done element by
    // element (1.5 TB/s).  Per (K group, channel tile) block the source is 16 (or KG) runs of KG*taps (16*taps)
    // contiguous floats: copy them coalesced into LDS, write each tap's 1 KiB fragment block as one contiguous run.
    __shared__ float tile[32 * (16 * 9 + 1) + 16];
    const int taps = d.taps;
    const int KC = (kd + KG - 1) / KG, NTt = (nd + 15) / 16;
    const int rows = d.mode == 0 ? 16 : KG, cols = d.mode == 0 ? KG : 16;   // co-local x ci-local extent of a block
    const int run = cols * taps, pitch = run + 1;
    for (int b = blockIdx.x; b < KC * NTt; b += gridDim.x) {
      const int kc = b / NTt, nt = b - kc * NTt;
      const int co0 = d.mode == 0 ? nt * 16 : kc * KG, ci0 = d.mode == 0 ? kc * KG : nt * 16;
      __syncthreads();
      for (int i = threadIdx.x; i < rows * run; i += 256) {
        const int r = i / run, c = i - r * run;
        const int co = co0 + r, ci = ci0 + c / taps;
        tile[r * pitch + c] = (co < d.Co && ci < d.Ci) ? w[((long)co * d.Ci + ci0) * taps + c] : 0.f;
      }
      __syncthreads();
      for (int o = threadIdx.x; o < taps * 64 * FR; o += 256) {
        const int tap = o / (64 * FR), within = o - tap * (64 * FR);
        const int lane = within / FR, j = within - lane * FR;
        const int kl = (lane >> 4) * FR + j, nl = lane & 15;
        const int col = d.mode == 0 ? nl : kl, cil = d.mode == 0 ? kl : nl;
        st1(wp + ((long)(tap * KC + kc) * NTt + nt) * (64 * FR) + within, tile[col * pitch + cil * taps + tap]);
      }
    }
    return;
  }
  if constexpr (sizeof(T) == 4) {
    const long total = pack16_elems(kd, nd, d.taps) + pack32_elems(kd, nd, d.taps);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
      st1(wp + i, pack_f32_elem(w, i, d.Co, d.Ci, d.taps, d.mode));
  } else {
    const int KC = (kd + KG - 1) / KG, NTt = (nd + 15) / 16;
    const long total = (long)d.taps * KC * NTt * 64 * FR;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const int j = (int)(i % FR), lane = (int)((i / FR) & 63);
      long r = i / (FR * 64);
      const int nt = (int)(r % NTt);
      r /= NTt;
      const int kc = (int)(r % KC), tap = (int)(r / KC);
      const int kidx = kc * KG + (lane >> 4) * FR + j, nidx = nt * 16 + (lane & 15);
      const int co = d.mode == 0 ? nidx : kidx, ci = d.mode == 0 ? kidx : nidx;
      float v = 0.f;
      if (co < d.Co && ci < d.Ci) v = w[((long)co * d.Ci + ci) * d.taps + tap];
      st1(wp + i, v);
    }
  }
}

typedef short s16x8 __attribute__((ext_vector_type(8)));

template <typename H, int MT, int NT, int MODE, int VEC, int KS, int ST>
__global__ __launch_bounds__(256) void conv_igemm_h(ConvArgsH p) {
  typedef typename H16<H>::x8 hx8;
  __shared__ float red[KS > 1 ? (4 - 4 / KS) * MT * NT * 256 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps the K-loop bookkeeping scalar
  const int col = lane & 15, kq = lane >> 4;
  const int kpart = wave % KS, mgrp = wave / KS;
  int bx, by;
  xcd_tile(p.xcd, bx, by);
  // stride-2 input gradient by parity class (see conv_igemm_f32)
  const bool par = MODE == 1 && p.par;
  const int tix = bx * (4 / KS) + mgrp;
  const int m0 = tix * (MT * 16);
  bool active = m0 < p.P;  // wave-uniform
  int ca = 0, cb = 0, cHa = 0, cWb = 0, cP = 0, crem = 0;
  if (par) {
    const int H0 = (p.Ho + 1) >> 1, H1 = p.Ho >> 1, W0 = (p.Wo + 1) >> 1, W1 = p.Wo >> 1;
    int rem = tix, c = 0;
    for (; c < 4; ++c) {
      cHa = (c >> 1) ? H1 : H0;
      cWb = (c & 1) ? W1 : W0;
      cP = p.N * cHa * cWb;
      const int tl = (cP + MT * 16 - 1) / (MT * 16);
      if (rem < tl) break;
      rem -= tl;
    }
    active = c < 4;
    ca = c >> 1;
    cb = c & 1;
    crem = rem;
  }
  const int emode = p.emode;   // kernel argument: uniform
  if (KS == 1 && !active && !emode) return;      // (with statistics the wave still meets the others at the barrier)
  const int ntg0 = by * NT;
  const int HoWo = p.Ho * p.Wo;

  int pn[MT], py[MT], px[MT];
  bool pv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (par) {
      const int ml = (crem * MT + mt) * 16 + col;
      pv[mt] = active && ml < cP;
      const int mm = pv[mt] ? ml : 0;
      const int hw = cHa * cWb;
      const int n = mm / hw, r = mm - n * hw;
      const int yy = r / cWb;
      pn[mt] = n;
      py[mt] = 2 * yy + ca;
      px[mt] = 2 * (r - yy * cWb) + cb;
      continue;
    }
    const int m = m0 + mt * 16 + col;
    pv[mt] = m < p.P;
    const int mm = pv[mt] ? m : 0;
    const int n = mm / HoWo, r = mm - n * HoWo;
    const int oy = r / p.Wo;
    pn[mt] = n;
    py[mt] = oy;
    px[mt] = r - oy * p.Wo;
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.wp_bytes, 0x00020000);
  unsigned aoff[MT];
  const int taps = p.kh * p.kw;
  int tap = 0, kc = kpart;
  auto tap_ok = [&](int tp) {   // par: is tap tp one of this wave's parity class?  (wave-uniform)
    const int ky = tp / p.kw, kx = tp - ky * p.kw;
    return (((ca + p.pad - ky * p.dil) | (cb + p.pad - kx * p.dil)) & 1) == 0;
  };
  auto next_tap = [&]() {
    ++tap;
    if (par) {
      while (tap < taps && !tap_ok(tap)) ++tap;
    }
  };
  int ntap_mine = taps;
  if (par) {
    ntap_mine = 0;
    for (int t = 0; t < taps; ++t) ntap_mine += tap_ok(t) ? 1 : 0;
    while (tap < taps && !tap_ok(tap)) ++tap;
  }
  while (kc >= p.KC) {
    kc -= p.KC;
    next_tap();
  }
  unsigned boff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) boff[nt] = (unsigned)(min(ntg0 + nt, p.NTt - 1) * 512 + lane * 8) * 2u;

  auto tap_setup = [&](int tp) {
    const int ky = tp / p.kw, kx = tp - ky * p.kw;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      bool v = pv[mt] && tp < taps;
      int iy, ix;
      if (MODE == 0) {
        iy = (py[mt] << p.sh) - p.pad + ky * p.dil;
        ix = (px[mt] << p.sh) - p.pad + kx * p.dil;
        v = v && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      } else {
        const int ty = py[mt] + p.pad - ky * p.dil, tx = px[mt] + p.pad - kx * p.dil;
        iy = ty >> p.sh;
        ix = tx >> p.sh;
        v = v && ty >= 0 && tx >= 0 && (iy << p.sh) == ty && (ix << p.sh) == tx && iy < p.Hi && ix < p.Wi;
      }
      aoff[mt] = v ? (unsigned)((((pn[mt] * p.Hi + iy) * p.Wi + ix) * p.Ci + kq * 8) * 2) : FAMI_OOB;
    }
  };

  auto load = [&](hx8(&a)[MT], hx8(&b)[NT]) {
    const int cbase = kc * 32 + kq * 8;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (VEC) {
        const unsigned o = cbase < p.Ci ? aoff[mt] + kc * 64 : FAMI_OOB;
        a[mt] = __builtin_bit_cast(hx8, __builtin_amdgcn_raw_buffer_load_b128(rx, o, 0, 0));
      } else if ((p.Ci & 3) == 0) {
        // channel counts that are multiples of 4 but not of 8 (the 108 mask channels of the DCN predictors: their input gradient is on
        // the head's serial chain): two 8-byte loads per fragment instead of eight 2-byte ones
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const unsigned o0 = cbase < p.Ci ? aoff[mt] + kc * 64 : FAMI_OOB;
        const unsigned o1 = cbase + 4 < p.Ci ? aoff[mt] + kc * 64 + 8 : FAMI_OOB;
        const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, o0, 0, 0));
        const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, o1, 0, 0));
        const u32x4 t = {lo[0], lo[1], hi[0], hi[1]};
        a[mt] = __builtin_bit_cast(hx8, t);
      } else {
        s16x8 t;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const unsigned o = cbase + q < p.Ci ? aoff[mt] + kc * 64 + q * 2 : FAMI_OOB;
          t[q] = (short)__builtin_amdgcn_raw_buffer_load_b16(rx, o, 0, 0);
        }
        a[mt] = __builtin_bit_cast(hx8, t);
      }
    }
    const unsigned wb = (unsigned)((tap * p.KC + kc) * p.NTt) * 1024u;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      b[nt] = __builtin_bit_cast(hx8, __builtin_amdgcn_raw_buffer_load_b128(rw, wb + boff[nt], 0, 0));
    kc += KS;
    if (kc >= p.KC) {
      do {
        kc -= p.KC;
        next_tap();
      } while (kc >= p.KC);
      tap_setup(tap);
    }
  };

  auto mma = [&](const hx8(&a)[MT], const hx8(&b)[NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = H16<H>::mfma(b[nt], a[mt], acc[mt][nt]);
  };

  hx8 af[ST][MT], bf[ST][NT];
  const int Tall = ntap_mine * p.KC;
  const int T = active ? (Tall - kpart + KS - 1) / KS : 0;
  if (T > 0) {
    tap_setup(tap);
#pragma unroll
    for (int st = 0; st < ST - 1; ++st) load(af[st], bf[st]);
    for (int it = 0; it < T; it += ST) {
#pragma unroll
      for (int st = 0; st < ST; ++st) {
        load(af[(st + ST - 1) % ST], bf[(st + ST - 1) % ST]);
        mma(af[st], bf[st]);
      }
    }
  }

  if (KS > 1) {
    if (kpart > 0) {
      float* dst = red + ((mgrp * (KS - 1)) + (kpart - 1)) * (MT * NT * 256);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[((mt * NT + nt) * 4 + r) * 64 + lane] = acc[mt][nt][r];
    }
    __syncthreads();
    if (kpart > 0 || !active) return;
#pragma unroll
    for (int k = 0; k < KS - 1; ++k) {
      const float* src = red + ((mgrp * (KS - 1)) + k) * (MT * NT * 256);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mt][nt][r] += src[((mt * NT + nt) * 4 + r) * 64 + lane];
    }
  }

  // epilogue: D row = kq*4 + r (output channel), col = lane&15 (pixel): 4 consecutive channels per lane
  const bool cvec = (p.Co & 3) == 0;
  if (emode) {
    // EpiBN (host guarantees Co % 4 == 0 and 16-bit output): channel tile by channel tile so only one tile's
    // parameters are live; a lane holds 4 channels of one pixel, the 16 pixels of a tile row are one DPP row
    EpiPtr e = epi_late(__builtin_offsetof(ConvArgsH, e));
    __shared__ float ered[4][NT * 32];
    const H* ez = reinterpret_cast<const H*>(e->z);
    const H* eyr = reinterpret_cast<const H*>(e->yr);
    const int erelu = e->relu, eC = e->C;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co0 = (ntg0 + nt) * 16 + kq * 4;
      const bool cok = co0 < p.Co;
      const int cc = cok ? co0 : 0;
      f32x4 es = z4, eq = z4, ek = z4, emu = z4, eis = z4, esc = z4, esf = z4, bias4 = z4;
      if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + cc);
      if (emode == 1) {
        if (e->pivot_src) ek = *reinterpret_cast<const f32x4*>(e->pivot_src + cc);
      } else {
        emu = *reinterpret_cast<const f32x4*>(e->mean + cc);
        eis = *reinterpret_cast<const f32x4*>(e->invstd + cc);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(e->gamma + cc), be = *reinterpret_cast<const f32x4*>(e->beta + cc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a, b;
          epi_scale_shift(emu[r], eis[r], ga[r], be[r], a, b);
          esc[r] = a;
          esf[r] = b;
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = par ? (pv[mt] ? (pn[mt] * p.Ho + py[mt]) * p.Wo + px[mt] : p.P) : m0 + mt * 16 + col;
        if (m >= p.P || !cok || !active) continue;
        f32x4 v = acc[mt][nt] + bias4;
        const long idx = (long)m * p.Co + co0;
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
          const f32x4 g = ld4_round<H>(v);        // the sums see the gradient as stored
          es += g;
          eq += g * ((zz - emu) * eis);
        } else {
          st4(yp, v);
          const f32x4 d = ld4_round<H>(v) - ek;   // statistics of the tensor as stored
          es += d;
          eq += d * d;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        es[r] = row16_sum(es[r]);
        eq[r] = row16_sum(eq[r]);
      }
      if (KS == 1) {
        if (col == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            ered[wave][nt * 32 + kq * 4 + r] = es[r];
            ered[wave][nt * 32 + 16 + kq * 4 + r] = eq[r];
          }
        }
      } else if (col == 0 && cok) {
        double* srow = e->slots + (long)(tix % e->ns) * 2 * eC;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          unsafeAtomicAdd(srow + co0 + r, (double)es[r]);
          unsafeAtomicAdd(srow + eC + co0 + r, (double)eq[r]);
          if (emode == 1 && tix == 0) bn_slots_pivot(e->slots, eC)[co0 + r] = ek[r];
        }
      }
    }
    if (KS == 1) {
      __syncthreads();
      const int t = threadIdx.x;
      if (t < NT * 32) {
        const int nt = t >> 5, st = (t >> 4) & 1, c16 = t & 15;
        const int co = (ntg0 + nt) * 16 + c16;
        if (co < p.Co) {
          const float v = (ered[0][t] + ered[1][t]) + (ered[2][t] + ered[3][t]);
          double* srow = e->slots + (long)(bx % e->ns) * 2 * eC;
          unsafeAtomicAdd(srow + st * eC + co, (double)v);
          if (emode == 1 && st == 0 && bx == 0) bn_slots_pivot(e->slots, eC)[co] = e->pivot_src ? e->pivot_src[co] : 0.f;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = par ? (pv[mt] ? (pn[mt] * p.Ho + py[mt]) * p.Wo + px[mt] : p.P) : m0 + mt * 16 + col;
    if (m >= p.P) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co0 = (ntg0 + nt) * 16 + kq * 4;
      if (co0 >= p.Co) continue;
      f32x4 v = acc[mt][nt];
      const long idx = (long)m * p.Co + co0;
      if (cvec) {
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + co0);
        if (p.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (p.out_f32) {
          float* yp = reinterpret_cast<float*>(p.y) + idx;
          if (p.accumulate) v += ld4(yp);
          st4(yp, v);
        } else {
          H* yp = reinterpret_cast<H*>(p.y) + idx;
          if (p.accumulate) v += ld4(yp);
          st4(yp, v);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (co0 + r >= p.Co) continue;
          float u = v[r] + (p.bias ? p.bias[co0 + r] : 0.f);
          if (p.relu) u = fmaxf(u, 0.f);
          if (p.out_f32) {
            float* yp = reinterpret_cast<float*>(p.y) + idx + r;
            *yp = p.accumulate ? *yp + u : u;
          } else {
            H* yp = reinterpret_cast<H*>(p.y) + idx + r;
            st1(yp, p.accumulate ? ld1(yp) + u : u);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ LDS-staged 3x3 (stride 1, pad 1, dilation 1)
// The HRNet branch convolutions (90 % of the FLOPs).  The direct kernels above re-fetch every activation 9 times and
// every weight fragment once per wave from L2 and are bound by L2 requests (TCC busy 85 %, profiles/r01_pmc_conv_f32).
// Here a workgroup owns a band of R full output rows of one frame (R*W <= 144 pixels = up to 9 row tiles) and
//   * stages the input band + one halo row above and below, one channel chunk at a time, ONCE into LDS
//     (contiguous copy; no halo columns: out-of-image taps are masked to zero per lane at fragment-read time);
//   * stages the weight fragments of one (chunk, tap) into a double-buffered LDS slot shared by the 4 waves,
//     prefetching the next tap's fragments into registers while the current tap is multiplied;
//   * distributes the (row tile, channel tile) pairs round-robin over the 4 waves; both MFMA operands are 16-byte
//     LDS reads (activation fragment = 4 f32 / 8 bf16 channels of one pixel, tap shift = a wave-uniform LDS offset).
// MFMA A = weights (rows = output channels), B = activations (cols = pixels): a lane ends with 4 consecutive output
// channels of one pixel -> one 16-byte (f32) / 8-byte (bf16) store.  dgrad = the same kernel with the tap shift
// negated and the mode-1 weight image.
struct ConvLdsArgs {
  EpiBN e;
  int emode;
  const void* x;      // [N,H,W,Ci]
  const void* wp;     // packed weights (pack_w_kernel / pack_w_bf16_kernel layout)
  void* y;            // [N,H,W,Co]
  const float* bias;  // [Co] or null
  int N, H, W, Ci, Co;
  int R, bands;       // rows per band, bands per frame
  int CH, CHP;        // real / LDS-padded channels per chunk
  int PSTRIDE;        // LDS bytes per pixel (odd multiple of 16)
  int KC, NTt;        // packed-weight geometry: K groups over all of Ci, channel tiles over all of Co
  int sgn;            // +1 forward, -1 dgrad
  int relu, accumulate, out_f32;
  int patch_bytes;
  int simz;           // 0 (cost simulation of the 3-plane split kernel: an offset the compiler cannot fold)
};

template <typename T> struct LdsTraits;
template <> struct LdsTraits<float> {
  typedef f32x4 frag;
  static constexpr int KSTEP = 16;
  __device__ static __forceinline__ f32x4 mma(const frag& w, const frag& a, f32x4 acc) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t], a[t], acc, 0, 0, 0);
    return acc;
  }
  __device__ static __forceinline__ frag zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
};
template <typename H> struct LdsTraits16 {
  typedef typename H16<H>::x8 frag;
  static constexpr int KSTEP = 32;
  __device__ static __forceinline__ f32x4 mma(const frag& w, const frag& a, f32x4 acc) { return H16<H>::mfma(w, a, acc); }
  __device__ static __forceinline__ frag zero() { return __builtin_bit_cast(frag, u32x4{0u, 0u, 0u, 0u}); }
};
template <> struct LdsTraits<bf16_t> : LdsTraits16<bf16_t> {};
template <> struct LdsTraits<f16_t> : LdsTraits16<f16_t> {};

template <typename T, int NT, int KSC, int SIM = 0>
__global__ __launch_bounds__(256) void conv3x3_lds_kernel(ConvLdsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef LdsTraits<T> TR;
  typedef typename TR::frag frag;
  constexpr int SZ = (int)sizeof(T);
  // taps per weight slab: one barrier per slab.  bf16 MFMAs are 8x shorter, so a slab carries a whole tap row there.
  constexpr int TPS = SZ == 2 ? 3 : 1;
  constexpr int NSLAB = 9 / TPS;
  constexpr int WSLAB = TPS * KSC * NT * 1024;             // bytes of one slab
  constexpr int WPIECES = WSLAB / 16;                      // 16-byte pieces of one slab
  constexpr int WR = (WPIECES + 255) / 256;                // prefetch registers (u32x4) per thread
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4;
  int bxl, byl;
  xcd_tile(1, bxl, byl);  // neighbouring row bands (shared halo rows) on one XCD's L2
  const int img = bxl / p.bands, bnd = bxl - img * p.bands;
  const int y0 = bnd * p.R;
  const int rows = min(p.R, p.H - y0);
  const int npix = rows * p.W;
  const int ntile = (npix + 15) >> 4;
  const int ntg0 = byl * NT;
  char* patch = smem;
  char* wbuf = smem + p.patch_bytes;

  // Row-tile slots of this wave: slots 0/1 = tiles wave, wave+4 with all NT channel tiles; slot 2 = the ninth
  // tile, whose NT channel tiles are dealt to waves 0..NT-1 one each (7+7+7+6 MFMA groups: balanced).
  int base[3], vmask[3];
  const int nt2 = wave < NT ? wave : 0;
  const bool has2 = ntile > 8 && wave < NT;
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) {
    const int mt = sl < 2 ? wave + 4 * sl : 8;
    const int j = mt * 16 + col;
    base[sl] = 0;
    vmask[sl] = 0;
    if (mt < ntile && j < npix && (sl < 2 || has2)) {
      const int ry = j / p.W, rx = j - ry * p.W;
      base[sl] = ((ry + 1) * p.W + rx) * p.PSTRIDE + kq * 16;
      int m = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y0 + ry + p.sgn * (t / 3 - 1), xx = rx + p.sgn * (t % 3 - 1);
        if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) m |= 1 << t;
      }
      vmask[sl] = m;
    }
  }
  f32x4 acc[2][NT], acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int sl = 0; sl < 2; ++sl)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[sl][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const char* xg = reinterpret_cast<const char*>(p.x);
  const char* wg = reinterpret_cast<const char*>(p.wp);
  const int pcs_real = p.CH * SZ / 16, pcs_all = p.CHP * SZ / 16;  // 16-byte pieces per pixel (real / padded)
  const int npx_patch = (p.R + 2) * p.W;
  const int nchunk = p.Ci / p.CH;

  // global byte offset of 16-byte piece i of the weight slab holding taps tp0 .. tp0+TPS-1 of chunk kc0
  auto wsrc = [&](int i, int tp0, int kc0) -> long {
    const int blk = i >> 6, l = i & 63;        // blk = (tt*KSC + ks)*NT + nt
    const int nt = blk % NT, r = blk / NT;
    const int ks = r % KSC, tt = r / KSC;
    return ((long)(((tp0 + tt) * p.KC + kc0 + ks) * p.NTt + ntg0 + nt)) * 1024 + l * 16;
  };

  for (int c = 0; c < nchunk; ++c) {
    if (c > 0) __syncthreads();  // the previous chunk's last slab has been consumed
    {                            // ---- stage the activation patch of this chunk
      const long img_base = (long)img * p.H * p.W * p.Ci * SZ + (long)c * p.CH * SZ;
      const int total = npx_patch * pcs_all;
      for (int i0 = tid; i0 < total; i0 += 256 * 4) {
        u32x4 v[4];
        int dsto[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * 256;
          v[u] = u32x4{0u, 0u, 0u, 0u};
          dsto[u] = -1;
          if (i < total) {
            const int px = i / pcs_all, pc = i - px * pcs_all;
            const int pr = px / p.W, xc = px - pr * p.W;
            const int yy = y0 - 1 + pr;
            dsto[u] = px * p.PSTRIDE + pc * 16;
            if ((unsigned)yy < (unsigned)p.H && pc < pcs_real)
              v[u] = *reinterpret_cast<const u32x4*>(xg + img_base + ((long)yy * p.W + xc) * p.Ci * SZ + pc * 16);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (dsto[u] >= 0) *reinterpret_cast<u32x4*>(patch + dsto[u]) = v[u];
      }
    }
    const int kc0 = c * KSC;
#pragma unroll
    for (int u = 0; u < WR; ++u) {  // slab 0 straight into buffer 0
      const int i = tid + u * 256;
      if (i < WPIECES) *reinterpret_cast<u32x4*>(wbuf + i * 16) = *reinterpret_cast<const u32x4*>(wg + wsrc(i, 0, kc0));
    }
    __syncthreads();

    for (int g = 0; g < NSLAB; ++g) {
      u32x4 wr[WR];  // next slab -> registers while this one is multiplied
      if (g + 1 < NSLAB) {
#pragma unroll
        for (int u = 0; u < WR; ++u) {
          const int i = tid + u * 256;
          wr[u] = u32x4{0u, 0u, 0u, 0u};
          if (i < WPIECES) wr[u] = *reinterpret_cast<const u32x4*>(wg + wsrc(i, (g + 1) * TPS, kc0));
        }
      }
#pragma unroll
      for (int tt = 0; tt < TPS; ++tt) {
        const int t = g * TPS + tt;
        const char* wb = wbuf + (g & 1) * WSLAB + tt * (KSC * NT * 1024) + lane * 16;
        const int toff = p.sgn * ((t / 3 - 1) * p.W + (t % 3 - 1)) * p.PSTRIDE;
        if constexpr (SIM == 1 && SZ == 2) {
          // cost model of the split-operand kernel: 3 planes of each operand (same bytes re-read through offsets the
          // compiler cannot fold), 6 products per tile pair
#pragma unroll
          for (int ks = 0; ks < KSC; ++ks) {
            frag a3[3][3], w3[NT][3], wx3[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
              for (int sl = 0; sl < 3; ++sl) {
                a3[sl][pl] = TR::zero();
                if ((vmask[sl] >> t) & 1) a3[sl][pl] = *reinterpret_cast<const frag*>(patch + base[sl] + toff + ks * 64 + pl * p.simz);
              }
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) w3[nt][pl] = *reinterpret_cast<const frag*>(wb + (ks * NT + nt) * 1024 + pl * p.simz);
              wx3[pl] = *reinterpret_cast<const frag*>(wb + (ks * NT + nt2) * 1024 + pl * p.simz);
            }
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PW[6] = {0, 1, 2, 0, 1, 0};   // small terms first
#pragma unroll
            for (int q = 0; q < 6; ++q) {
#pragma unroll
              for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[sl][nt] = TR::mma(w3[nt][PW[q]], a3[sl][PA[q]], acc[sl][nt]);
              acc2 = TR::mma(wx3[PW[q]], a3[2][PA[q]], acc2);
            }
          }
          continue;
        }
        frag a[KSC][3], w[KSC][NT], wx[KSC];
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) {  // every fragment of the tap is requested before the first MFMA
#pragma unroll
          for (int sl = 0; sl < 3; ++sl) {
            a[ks][sl] = TR::zero();
            if ((vmask[sl] >> t) & 1) a[ks][sl] = *reinterpret_cast<const frag*>(patch + base[sl] + toff + ks * 64);
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) w[ks][nt] = *reinterpret_cast<const frag*>(wb + (ks * NT + nt) * 1024);
          wx[ks] = *reinterpret_cast<const frag*>(wb + (ks * NT + nt2) * 1024);
        }
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) {
          if constexpr (SZ == 4) {  // element-major order: consecutive MFMAs hit different accumulators
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
              for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                  acc[sl][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ks][nt][e], a[ks][sl][e], acc[sl][nt], 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wx[ks][e], a[ks][2][e], acc2, 0, 0, 0);
            }
          } else {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[sl][nt] = TR::mma(w[ks][nt], a[ks][sl], acc[sl][nt]);
            acc2 = TR::mma(wx[ks], a[ks][2], acc2);
          }
        }
      }
      if (g + 1 < NSLAB) {
        char* wn = wbuf + ((g + 1) & 1) * WSLAB;
#pragma unroll
        for (int u = 0; u < WR; ++u) {
          const int i = tid + u * 256;
          if (i < WPIECES) *reinterpret_cast<u32x4*>(wn + i * 16) = wr[u];
        }
      }
      __syncthreads();
    }
  }

  // ---- epilogue: D row = kq*4 + r (output channel), col = lane&15 (pixel)
  const long pix0 = ((long)img * p.H + y0) * p.W;
  const int emode = p.emode;
  if (emode) {
    // EpiBN (see conv_igemm_h): per channel tile, the wave's row tiles are summed in registers, the 16 pixel lanes by
    // DPP, the four waves through the (now idle) LDS; 16-bit or f32 storage output (not the f32 heatmap form)
    EpiPtr e = epi_late(__builtin_offsetof(ConvLdsArgs, e));
    float* ered = reinterpret_cast<float*>(smem);   // [4][NT*32]; every wave is past the last slab's barrier
    const T* ez = reinterpret_cast<const T*>(e->z);
    const T* eyr = reinterpret_cast<const T*>(e->yr);
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
      for (int sl = 0; sl < 3; ++sl) {
        const int mt = sl < 2 ? wave + 4 * sl : 8;
        const int j = mt * 16 + col;
        if (sl == 2 && !(has2 && nt == nt2)) continue;
        if (mt >= ntile || j >= npix) continue;
        f32x4 v = (sl < 2 ? acc[sl < 2 ? sl : 0][nt] : acc2) + bias4;
        const long idx = (pix0 + j) * p.Co + co0;
        T* yp = reinterpret_cast<T*>(p.y) + idx;
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
          const f32x4 g = ld4_round<T>(v);
          es += g;
          eq += g * ((zz - emu) * eis);
        } else {
          st4(yp, v);
          const f32x4 d = ld4_round<T>(v) - ek;
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
      const float v = (ered[tid] + ered[NT * 32 + tid]) + (ered[2 * NT * 32 + tid] + ered[3 * NT * 32 + tid]);
      double* srow = e->slots + (long)(bxl % e->ns) * 2 * eC;
      unsafeAtomicAdd(srow + st * eC + co, (double)v);
      if (emode == 1 && st == 0 && bxl == 0) bn_slots_pivot(e->slots, eC)[co] = e->pivot_src ? e->pivot_src[co] : 0.f;
    }
    return;
  }
  auto store = [&](int mt, int nt, f32x4 v) {
    const int j = mt * 16 + col;
    if (mt >= ntile || j >= npix) return;
    const int co0 = (ntg0 + nt) * 16 + kq * 4;
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
      T* yp = reinterpret_cast<T*>(p.y) + idx;
      if (p.accumulate) v += ld4(yp);
      st4(yp, v);
    }
  };
#pragma unroll
  for (int sl = 0; sl < 2; ++sl)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) store(wave + 4 * sl, nt, acc[sl][nt]);
  if (has2) store(8, nt2, acc2);
}

// ------------------------------------------------------------------ wgrad
// one scalar through a buffer descriptor, converted to f32 (out-of-range offsets read 0)
template <typename T>
__device__ __forceinline__ float ldbuf(const __amdgpu_buffer_rsrc_t r, unsigned off);
template <>
__device__ __forceinline__ float ldbuf<float>(const __amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
template <>
__device__ __forceinline__ float ldbuf<bf16_t>(const __amdgpu_buffer_rsrc_t r, unsigned off) {
  return H16<bf16_t>::from_bits((unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0));
}
template <>
__device__ __forceinline__ float ldbuf<f16_t>(const __amdgpu_buffer_rsrc_t r, unsigned off) {
  return H16<f16_t>::from_bits((unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0));
}

template <typename T>
struct WgradArgs {
  const T* x;       // [N,H,W,Ci]
  const T* dy;      // [N,Ho,Wo,Co]
  float* part;      // [psplit][Co][Ci][taps]
  int N, H, W, Ci, Ho, Wo, Co, kh, kw, sh, pad, dil;
  int P, chunk, ciBlocks, coBlocks;
  unsigned x_bytes, dy_bytes;
  int xcd;
  int psplit, w8;   // linear-address kernel: number of pixel chunks; 8-wave workgroups (3x3 only)
  int prio;         // > 0: s_setprio (see ConvArgs)
};

// dW[tap][ci][co] = sum_pixels X[pix@tap][ci] * dY[pix][co]; MFMA rows = ci, cols = co, K = pixels.
template <typename T, int MT, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_f32(WgradArgs<T> p) {
  __shared__ float red[4 * MT * NT * 256];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps the K-loop bookkeeping scalar
  const int c16 = lane & 15, kq = lane >> 4;
  const int ps = blockIdx.x;
  int by = blockIdx.y;
  const int cob = by % p.coBlocks;
  by /= p.coBlocks;
  const int cib = by % p.ciBlocks;
  const int tap = by / p.ciBlocks;
  const int ky = tap / p.kw, kx = tap - ky * p.kw;
  const int taps = p.kh * p.kw;

  const int p_lo = ps * p.chunk;
  const int p_hi = min(p.P, p_lo + p.chunk);
  const int sub = p.chunk >> 2;  // chunk is a multiple of 16
  const int w_lo = p_lo + wave * sub;
  const int w_hi = min(p_hi, w_lo + sub);

  // operands come through buffer descriptors: a lane whose pixel is outside the image / past the wave's range
  // or whose channel is past Ci / Co carries the out-of-range offset and reads 0 (no branches in the loop)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);
  unsigned cio[MT], coo[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int c = (cib * MT + mt) * 16 + c16;
    cio[mt] = c < p.Ci ? (unsigned)c * (unsigned)sizeof(T) : FAMI_OOB;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = (cob * NT + nt) * 16 + c16;
    coo[nt] = c < p.Co ? (unsigned)c * (unsigned)sizeof(T) : FAMI_OOB;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this lane walks pixels w_lo + kq, +4, +8, ...
  int pix = w_lo + kq;
  int n, oy, ox;
  {
    const int HoWo = p.Ho * p.Wo;
    const int pp = pix < p.P ? pix : 0;
    n = pp / HoWo;
    const int r = pp - n * HoWo;
    oy = r / p.Wo;
    ox = r - oy * p.Wo;
  }
  const int wrap_x = p.Wo >= 4 ? 1 : 4;  // how many row wraps a +4 step can cross (Wo >= 1)

  auto load = [&](float(&a)[MT], float(&b)[NT]) {
    const bool pvalid = pix < w_hi;
    const int iy = (oy << p.sh) - p.pad + ky * p.dil, ix = (ox << p.sh) - p.pad + kx * p.dil;
    const bool xin = pvalid && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const unsigned xoff = xin ? (unsigned)(((n * p.H + iy) * p.W + ix) * p.Ci) * (unsigned)sizeof(T) : FAMI_OOB;
    const unsigned yoff = pvalid ? (unsigned)(pix * p.Co) * (unsigned)sizeof(T) : FAMI_OOB;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      a[mt] = ldbuf<T>(rx, xoff + cio[mt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      b[nt] = ldbuf<T>(ry, yoff + coo[nt]);
    pix += 4;
    ox += 4;
    for (int k = 0; k < wrap_x; ++k) {
      const bool w = ox >= p.Wo;
      ox -= w ? p.Wo : 0;
      oy += w ? 1 : 0;
    }
    const bool wy = oy >= p.Ho;
    oy -= wy ? p.Ho : 0;
    n += wy ? 1 : 0;
  };
  auto mma = [&](const float(&a)[MT], const float(&b)[NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
  };

  // unconditional two-stage pipeline; steps past the wave's range load zeros (pvalid false)
  const int Tn = (w_hi > w_lo) ? (w_hi - w_lo + 3) / 4 : 0;
  float a0[MT], b0[NT], a1[MT], b1[NT];
  if (Tn > 0) {
    load(a0, b0);
    for (int it = 0; it < Tn; it += 2) {
      load(a1, b1);
      mma(a0, b0);
      load(a0, b0);
      mma(a1, b1);
    }
  }

  // cross-wave reduction through LDS, then scatter into the OIHW-ordered partial slab
  float* mine = red + wave * (MT * NT * 256);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[((mt * NT + nt) * 4 + r) * 64 + lane] = acc[mt][nt][r];
  __syncthreads();
  float* slab = p.part + (long)ps * p.Co * p.Ci * taps;
  for (int e = threadIdx.x; e < MT * NT * 256; e += 256) {
    const float v = red[e] + red[MT * NT * 256 + e] + red[2 * MT * NT * 256 + e] + red[3 * MT * NT * 256 + e];
    const int l = e & 63, r = (e >> 6) & 3, tile = e >> 8;
    const int mt = tile / NT, nt = tile - mt * NT;
    const int cci = (cib * MT + mt) * 16 + (l >> 4) * 4 + r;
    const int cco = (cob * NT + nt) * 16 + (l & 15);
    if (cci < p.Ci && cco < p.Co) slab[((long)cco * p.Ci + cci) * taps + tap] = v;
  }
}

// 3x3 (any kh*kw > 1) weight gradient, one WAVE PER TAP: the kh*kw waves of a workgroup walk the same pixel
// chunk, so dY and the shifted X rows are fetched from HBM/MALL once and served to the other taps by L1/L2
// (the one-tap-per-workgroup layout above re-read both tensors kh*kw times: 478 MB per 48->48 conv).
// Slab layout [psplit][tap][ci][co] (co contiguous => 64-byte stores); the reduce kernel transposes to OIHW.
template <typename T, int MT, int NT>
__global__ __launch_bounds__(576) void conv_wgrad_taps_f32(WgradArgs<T> p) {
  const int lane = threadIdx.x & 63;
  const int tap = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c16 = lane & 15, kq = lane >> 4;
  int ps, byl;
  xcd_tile(p.xcd, ps, byl);  // neighbouring pixel chunks (shared halo rows) on one XCD's L2
  const int cob = byl % p.coBlocks, cib = byl / p.coBlocks;
  const int ky = tap / p.kw, kx = tap - ky * p.kw;
  const int taps = p.kh * p.kw;
  const int w_lo = ps * p.chunk;
  const int w_hi = min(p.P, w_lo + p.chunk);

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);
  unsigned cio[MT], coo[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int c = (cib * MT + mt) * 16 + c16;
    cio[mt] = c < p.Ci ? (unsigned)c * (unsigned)sizeof(T) : FAMI_OOB;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = (cob * NT + nt) * 16 + c16;
    coo[nt] = c < p.Co ? (unsigned)c * (unsigned)sizeof(T) : FAMI_OOB;
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  int pix = w_lo + kq;
  int n, oy, ox;
  {
    const int HoWo = p.Ho * p.Wo;
    const int pp = pix < p.P ? pix : 0;
    n = pp / HoWo;
    const int r = pp - n * HoWo;
    oy = r / p.Wo;
    ox = r - oy * p.Wo;
  }
  const int wrap_x = p.Wo >= 4 ? 1 : 4;
  auto load = [&](float(&a)[MT], float(&b)[NT]) {
    const bool pvalid = pix < w_hi;
    const int iy = (oy << p.sh) - p.pad + ky * p.dil, ix = (ox << p.sh) - p.pad + kx * p.dil;
    const bool xin = pvalid && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const unsigned xoff = xin ? (unsigned)(((n * p.H + iy) * p.W + ix) * p.Ci) * (unsigned)sizeof(T) : FAMI_OOB;
    const unsigned yoff = pvalid ? (unsigned)(pix * p.Co) * (unsigned)sizeof(T) : FAMI_OOB;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      a[mt] = ldbuf<T>(rx, xoff + cio[mt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      b[nt] = ldbuf<T>(ry, yoff + coo[nt]);
    pix += 4;
    ox += 4;
    for (int k = 0; k < wrap_x; ++k) {
      const bool w = ox >= p.Wo;
      ox -= w ? p.Wo : 0;
      oy += w ? 1 : 0;
    }
    const bool wy = oy >= p.Ho;
    oy -= wy ? p.Ho : 0;
    n += wy ? 1 : 0;
  };
  auto mma = [&](const float(&a)[MT], const float(&b)[NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
  };
  const int Tn = (w_hi > w_lo) ? (w_hi - w_lo + 3) / 4 : 0;
  float a0[MT], b0[NT], a1[MT], b1[NT];
  if (Tn > 0) {
    load(a0, b0);
    for (int it = 0; it < Tn; it += 2) {
      load(a1, b1);
      mma(a0, b0);
      load(a0, b0);
      mma(a1, b1);
    }
  }
  // D row = ci (kq*4 + r), col = co (c16)
  float* slab = p.part + ((long)ps * taps + tap) * p.Ci * p.Co;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cci = (cib * MT + mt) * 16 + kq * 4 + r;
      if (cci >= p.Ci) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int cco = (cob * NT + nt) * 16 + c16;
        if (cco < p.Co) slab[(long)cci * p.Co + cco] = acc[mt][nt][r];
      }
    }
}

// The same kernel for stride-1 "same"-size convolutions (Ho = H, Wo = W; Ci, Co multiples of 16; W >= 4) with the
// per-iteration bookkeeping taken off the vector pipe.  f32-input MFMA and VALU share one pipe on gfx950
// (tools/probes/mfma_valu_overlap.hip), and the general kernel above spends ~35 VALU instructions per 9 MFMAs on pixel
// -> (n, y, x) wrap logic, bounds tests and two quarter-rate 64-bit multiply-adds: ~170 of every ~460 pipe cycles.  Here
//  * the input address of output pixel `pix` at tap (dy, dx) is LINEAR in NHWC: (pix + dy*W + dx)*Ci -- image and row
//    wraps only matter for VALIDITY, so the per-lane offsets advance by one v_add each per iteration;
//  * the position of the wave's 4-pixel group (ox0, oy0) is tracked in SGPRs; a group that lies inside one row, away from
//    the border this tap looks across and inside the chunk takes the fast path (no per-lane test at all: 94 % of the
//    groups at W = 72); the others take a wave-uniform branch to the per-lane test.
template <typename T, int MT, int NT>
__global__ __launch_bounds__(1024) void conv_wgrad_taps_lin_f32(WgradArgs<T> p) {
  if (p.prio) __builtin_amdgcn_s_setprio(3);   // see conv_igemm_f32
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c16 = lane & 15, kq = lane >> 4;
  int bx, byl;
  xcd_tile(p.xcd, bx, byl);
  // 9 waves on 4 SIMDs leave one SIMD of the CU with a third wave's worth of extra MFMA work.  3x3 kernels therefore run
  // workgroups of 8 (w8 = 1) or 16 (w8 = 2) waves: workgroup c < psplit takes taps 0..7 of slab c, the workgroups behind
  // them take tap 8 of eight slabs each (their operands are L2 hits: the first kind reads the same rows at the same
  // time).  With 16 waves a slab is the sum of TWO pixel chunks: the upper eight waves hand their accumulators to the
  // lower eight through LDS, which halves the slab traffic and the reduce pass behind it.
  int tap = wave, slab = bx, ps = bx, half = 0;
  bool active = true;
  if (p.w8 == 1) {
    if (bx >= p.psplit) {
      slab = ps = (bx - p.psplit) * 8 + wave;
      tap = 8;
      active = slab < p.psplit;
    }
  } else if (p.w8 == 2) {
    half = wave >> 3;
    const int w = wave & 7;
    if (bx < p.psplit) {
      tap = w;
    } else {
      slab = (bx - p.psplit) * 8 + w;
      tap = 8;
      active = slab < p.psplit;
    }
    ps = 2 * slab + half;
  }
  const int cob = byl % p.coBlocks, cib = byl / p.coBlocks;
  const int ky = tap / p.kw, kx = tap - ky * p.kw;
  const int taps = p.kh * p.kw;
  const int dky = ky * p.dil - p.pad, dkx = kx * p.dil - p.pad;
  const int w_lo = active ? min(ps * p.chunk, p.P) : 0;   // multiple of 16 (or the end)
  const int w_hi = active ? min(p.P, w_lo + p.chunk) : 0;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);
  // per-lane byte offsets of this lane's pixel (pix0 + kq) at tile 0; tiles are +64 B immediates
  unsigned vx = (unsigned)(((w_lo + kq + dky * p.W + dkx) * p.Ci + cib * MT * 16 + c16) * (int)sizeof(T));
  unsigned vy = (unsigned)(((w_lo + kq) * p.Co + cob * NT * 16 + c16) * (int)sizeof(T));
  const unsigned sx = 4u * (unsigned)p.Ci * (unsigned)sizeof(T), sy = 4u * (unsigned)p.Co * (unsigned)sizeof(T);

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // wave-uniform position of the group's first pixel
  int pix0 = w_lo;
  int oy0, ox0;
  {
    const int r = w_lo % (p.H * p.W);
    oy0 = r / p.W;
    ox0 = r - oy0 * p.W;
  }
  auto load = [&](float(&a)[MT], float(&b)[NT]) {
    const bool fast = (ox0 + 3 < p.W) && (ox0 + dkx >= 0) && (ox0 + 3 + dkx < p.W) && ((unsigned)(oy0 + dky) < (unsigned)p.H) &&
                      (pix0 + 3 < w_hi);
    unsigned xo = vx, yo = vy;
    if (!fast) {
      asm volatile("" ::: "memory");   // keep this a branch: if-converted it would run for every group
      int ox = ox0 + kq, oy = oy0;
      if (ox >= p.W) {
        ox -= p.W;
        oy = oy + 1 >= p.H ? 0 : oy + 1;
      }
      const bool pv = pix0 + kq < w_hi;
      const bool xin = pv && (unsigned)(ox + dkx) < (unsigned)p.W && (unsigned)(oy + dky) < (unsigned)p.H;
      xo = xin ? vx : FAMI_OOB;
      yo = pv ? vy : FAMI_OOB;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = ldbuf<T>(rx, xo + (unsigned)(mt * 16 * (int)sizeof(T)));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = ldbuf<T>(ry, yo + (unsigned)(nt * 16 * (int)sizeof(T)));
    vx += sx;
    vy += sy;
    pix0 += 4;
    ox0 += 4;
    if (ox0 >= p.W) {
      ox0 -= p.W;
      oy0 = oy0 + 1 >= p.H ? 0 : oy0 + 1;
    }
  };
  auto mma = [&](const float(&a)[MT], const float(&b)[NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
  };
  const int Tn = (w_hi > w_lo) ? (w_hi - w_lo + 3) / 4 : 0;
  // two operand sets.  The scheduler would sink each load group below the MFMAs that precede its use (one buffer
  // instead of two), hence the barriers.  Steps past the chunk load zeros.  (A third set in flight measured slower:
  // 72.4 vs 67.0 us on the 48-channel launch.)
  float a0[MT], b0[NT], a1[MT], b1[NT];
  if (Tn > 0) {
    load(a0, b0);
    for (int it = 0; it < Tn; it += 2) {
      load(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0);
      load(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1);
    }
  }
  if (p.w8 == 2) {
    extern __shared__ float wg_sm[];   // [8 waves][MT*NT tiles][4][64 lanes]
    float* mine = wg_sm + (wave & 7) * (MT * NT * 256) + lane;
    if (half) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mine[((mt * NT + nt) * 4 + r) * 64] = acc[mt][nt][r];
    }
    __syncthreads();
    if (half) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mt][nt][r] += mine[((mt * NT + nt) * 4 + r) * 64];
  }
  if (!active) return;
  float* slabp = p.part + ((long)slab * taps + tap) * p.Ci * p.Co;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cci = (cib * MT + mt) * 16 + kq * 4 + r;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int cco = (cob * NT + nt) * 16 + c16;
        slabp[(long)cci * p.Co + cco] = acc[mt][nt][r];
      }
    }
}

// ---- slab reduces.  The bodies are device functions shared by the one-reduce kernels and the batched kernel
// (wgrad_reduce_batch_kernel: up to 16 reduces of a backward pass in ONE launch -- 303 launches per step were 4.6-4.9 %
// of the kernel time and sat between every weight gradient and the next kernel of its stream lane), so a deferred reduce
// sums in exactly the order the immediate one does.
//
// slabs [psplit][tap][ci][co] -> dw OIHW (=|+=).  64 outputs x 16 slab groups per block: the reduction is
// latency-bound (each output owns a strided column), so parallelism comes from splitting the slab axis.
__device__ __forceinline__ void reduce_taps_body(const float* __restrict__ part, float* __restrict__ dw, int Co, int Ci,
                                                 int taps, int psplit, int accumulate, int bid, float* sm /* [1024] */) {
  const long n = (long)taps * Ci * Co;
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long i = (long)bid * 64 + o;
  float s0 = 0.f, s1 = 0.f;
  if (i < n) {
    int k = g;
    for (; k + 16 < psplit; k += 32) {
      s0 += part[(long)k * n + i];
      s1 += part[(long)(k + 16) * n + i];
    }
    if (k < psplit) s0 += part[(long)k * n + i];
  }
  sm[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (g == 0 && i < n) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += sm[q * 64 + o];
    const int co = (int)(i % Co);
    const long r = i / Co;
    const int ci = (int)(r % Ci), tap = (int)(r / Ci);
    float* d = dw + ((long)co * Ci + ci) * taps + tap;
    *d = accumulate ? *d + s : s;
  }
}
__global__ __launch_bounds__(1024) void wgrad_reduce_taps_kernel(const float* __restrict__ part,
                                                                 float* __restrict__ dw, int Co, int Ci, int taps,
                                                                 int psplit, int accumulate) {
  __shared__ float sm[1024];
  reduce_taps_body(part, dw, Co, Ci, taps, psplit, accumulate, blockIdx.x, sm);
}

// same reduction with 16-byte loads: a thread owns 4 consecutive output channels of one (tap, ci), 2^sg slab groups per
// block walk the slab axis with four independent partial sums (Co % 4 == 0; summation order is fixed)
__device__ __forceinline__ void reduce_taps4_body(const float* __restrict__ part, float* __restrict__ dw, int Co, int Ci,
                                                  int taps, int psplit, int accumulate, int sg, int bid, f32x4* sm /* [256] */) {
  const int SG = 1 << sg, cols = 256 >> sg;
  const int o = threadIdx.x & (cols - 1), g = threadIdx.x >> (8 - sg);
  const long n4 = (long)taps * Ci * Co / 4;
  const long i = (long)bid * cols + o;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 s0 = z, s1 = z, s2 = z, s3 = z;
  if (i < n4) {
    const f32x4* p4 = reinterpret_cast<const f32x4*>(part) + i;
    int k = g;
    // eight slabs per thread in flight: the 48-channel layers give this kernel only ~1.3 workgroups per CU (324 blocks),
    // so the bytes in flight per CU -- not the bandwidth -- set its time (42 MB in 16.7 us with four loads in flight)
    for (; k + 7 * SG < psplit; k += 8 * SG) {
      const f32x4 v0 = p4[(long)k * n4], v1 = p4[(long)(k + SG) * n4], v2 = p4[(long)(k + 2 * SG) * n4],
                  v3 = p4[(long)(k + 3 * SG) * n4], v4 = p4[(long)(k + 4 * SG) * n4], v5 = p4[(long)(k + 5 * SG) * n4],
                  v6 = p4[(long)(k + 6 * SG) * n4], v7 = p4[(long)(k + 7 * SG) * n4];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
      s0 += v4; s1 += v5; s2 += v6; s3 += v7;
    }
    for (; k + 3 * SG < psplit; k += 4 * SG) {
      s0 += p4[(long)k * n4];
      s1 += p4[(long)(k + SG) * n4];
      s2 += p4[(long)(k + 2 * SG) * n4];
      s3 += p4[(long)(k + 3 * SG) * n4];
    }
    for (; k < psplit; k += SG) s0 += p4[(long)k * n4];
  }
  sm[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && i < n4) {
    f32x4 t = sm[o];
    for (int q = 1; q < SG; ++q) t += sm[q * cols + o];
    const long e = i * 4;
    const int co = (int)(e % Co);
    const long r = e / Co;
    const int ci = (int)(r % Ci), tap = (int)(r / Ci);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float* d = dw + ((long)(co + u) * Ci + ci) * taps + tap;
      *d = accumulate ? *d + t[u] : t[u];
    }
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_taps4_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                 int Co, int Ci, int taps, int psplit, int accumulate,
                                                                 int sg) {
  __shared__ f32x4 sm[256];
  reduce_taps4_body(part, dw, Co, Ci, taps, psplit, accumulate, sg, blockIdx.x, sm);
}

static void launch_reduce_taps(const float* part, float* dw, int Co, int Ci, int taps, int psplit, int accumulate,
                               hipStream_t s);

// dw[i] (=|+=) sum_k part[k][i]; 64 outputs x 4 slab groups per block so the serial chain is psplit/4 long
__device__ __forceinline__ void reduce_plain_body(const float* __restrict__ part, float* __restrict__ dw, long n,
                                                  int psplit, int accumulate, int bid, float* sm /* [256] */) {
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long i = (long)bid * 64 + o;
  float s = 0.f;
  if (i < n)
    for (int k = g; k < psplit; k += 4) s += part[(long)k * n + i];
  sm[threadIdx.x] = s;
  __syncthreads();
  if (g == 0 && i < n) {
    s = sm[o] + sm[64 + o] + sm[128 + o] + sm[192 + o];
    dw[i] = accumulate ? dw[i] + s : s;
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                           long n, int psplit, int accumulate) {
  __shared__ float sm[256];
  reduce_plain_body(part, dw, n, psplit, accumulate, blockIdx.x, sm);
}

// One deferred reduce.  kind 0: plain [psplit][n] -> dw[n] ; 1: taps layout, 16-byte loads (sg) ; 2: taps layout, scalar.
struct ReduceDesc {
  const float* part;
  float* dw;
  int Co, Ci, taps, psplit, accumulate, kind, sg, blocks;
};
#define FAMI_REDUCE_BATCH 16
struct ReduceBatch { ReduceDesc d[FAMI_REDUCE_BATCH]; };
// blockIdx.y = entry, blockIdx.x = block of that entry's own grid (the launch takes the largest); 1024 threads: the
// scalar taps form uses all of them, the other two forms the first 256
__global__ __launch_bounds__(1024) void wgrad_reduce_batch_kernel(ReduceBatch b) {
  __shared__ f32x4 sm[256];   // 4 KiB: [1024] floats for the scalar taps form
  const ReduceDesc& d = b.d[blockIdx.y];
  if ((int)blockIdx.x >= d.blocks) return;
  if (d.kind == 2) {
    reduce_taps_body(d.part, d.dw, d.Co, d.Ci, d.taps, d.psplit, d.accumulate, blockIdx.x, reinterpret_cast<float*>(sm));
    return;
  }
  if (threadIdx.x >= 256) return;   // (no barrier is shared with the upper waves: they leave before any)
  if (d.kind == 1) reduce_taps4_body(d.part, d.dw, d.Co, d.Ci, d.taps, d.psplit, d.accumulate, d.sg, blockIdx.x, sm);
  else reduce_plain_body(d.part, d.dw, (long)d.Co * d.Ci * d.taps, d.psplit, d.accumulate, blockIdx.x, reinterpret_cast<float*>(sm));
}

// ------------------------------------------------------------------ bf16 wgrad on the bf16 matrix core
// dW[tap][ci][co] = sum_p X[p + tap][ci] * dY[p][co]: the reduction runs over PIXELS, but NHWC keeps channels, not
// pixels, contiguous -- both MFMA operands need a transpose.  gfx950's ds_read_b64_tr_b16 does it on the way out of
// LDS: the 16 lanes of a group hand in 16 row pieces (4 pixels x 16 channels) and each lane receives its channel's
// 4 pixel values.  A workgroup stages a run of pixels ONCE (X with a +-(W+1)-pixel halo so all 9 taps are LDS address
// shifts, dY without), every (channel tile, tap) pair is an accumulator group in registers (fp32), out-of-image taps
// are redirected per pixel to a zero row, and the 32-pixel K steps feed v_mfma_f32_16x16x32_bf16.
// 3x3, stride 1, pad 1, dilation 1; everything else takes the scalar-operand kernels above.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct WgradLdsArgs {
  const void* x;     // [N,H,W,Ci] (16-bit storage type)
  const void* dy;    // [N,H,W,Co]
  float* part;       // [G][9][Ci][Co]
  int N, H, W, Ci, Co, P;
  int chunk;         // pixels per staged sub-chunk (multiple of 32)
  int nsub;          // sub-chunks walked by one workgroup (accumulators persist, LDS is restaged)
  int ciBlocks, coBlocks;
  int CiB, CoB;      // channels per block (16 * CIT, 16 * COT)
  int xrow, yrow;    // LDS bytes per pixel of the X / dY tiles
  int xbytes;        // size of the X tile
  int xcd;
};

template <typename H, int CIT, int COT>
__global__ __launch_bounds__(256) void conv_wgrad_h_kernel(WgradLdsArgs p) {
  typedef typename H16<H>::x8 hx8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPW = (CIT * 9 + 3) / 4;  // (ci tile, tap) pairs per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kq = lane >> 4;
  const int rsel = l16 >> 2, piece = l16 & 3;  // this lane feeds pixel row `rsel` (of 4), channels piece*4..+3
  int g, byl;
  xcd_tile(p.xcd, g, byl);
  const int cob = byl % p.coBlocks, cib = byl / p.coBlocks;
  const int halo = p.W + 1;
  char* xt = smem;                             // [(chunk + 2*halo)][xrow]
  char* yt = smem + p.xbytes;                  // [chunk][yrow]
  char* zrow = yt + p.chunk * p.yrow;          // 32 zero bytes

  // pairs of this wave: q = wave + 4*i -> (ci tile, tap)
  int pci[NPW], ptap[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int q = wave + 4 * i;
    const bool ok = q < CIT * 9;
    pci[i] = ok ? q / 9 : 0;
    ptap[i] = ok ? q % 9 : -1;
  }
  f32x4 acc[NPW][COT];
#pragma unroll
  for (int i = 0; i < NPW; ++i)
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int HW = p.H * p.W;

  for (int sub = 0; sub < p.nsub; ++sub) {
  const int p0 = (g * p.nsub + sub) * p.chunk;
  if (p0 >= p.P) break;
  const int M = min(p.chunk, p.P - p0);       // pixels of this sub-chunk
  if (sub > 0) __syncthreads();               // the previous sub-chunk has been consumed
  // ---- stage X (pixels p0-halo .. p0+chunk+halo) and dY (p0 .. p0+chunk); out-of-tensor pixels are zero-filled
  {
    const int xpcs = p.CiB / 8, ypcs = p.CoB / 8;  // 16-byte pieces per pixel
    const int nx = (p.chunk + 2 * halo) * xpcs, ny = p.chunk * ypcs;
    const char* xg = reinterpret_cast<const char*>(p.x) + (long)cib * p.CiB * 2;
    const char* yg = reinterpret_cast<const char*>(p.dy) + (long)cob * p.CoB * 2;
    for (int i0 = tid; i0 < nx + ny; i0 += 256 * 4) {
      u32x4 v[4];
      int dsto[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256;
        v[u] = u32x4{0u, 0u, 0u, 0u};
        dsto[u] = -1;
        if (i < nx) {
          const int px = i / xpcs, pc = i - px * xpcs;
          const long pg = (long)p0 - halo + px;
          dsto[u] = px * p.xrow + pc * 16;
          if (pg >= 0 && pg < p.P) v[u] = *reinterpret_cast<const u32x4*>(xg + pg * p.Ci * 2 + pc * 16);
        } else if (i < nx + ny) {
          const int k = i - nx;
          const int px = k / ypcs, pc = k - px * ypcs;
          const long pg = (long)p0 + px;
          dsto[u] = p.xbytes + px * p.yrow + pc * 16;
          if (pg < p.P) v[u] = *reinterpret_cast<const u32x4*>(yg + pg * p.Co * 2 + pc * 16);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (dsto[u] >= 0) *reinterpret_cast<u32x4*>(smem + dsto[u]) = v[u];
    }
    if (tid < 2) *reinterpret_cast<u32x4*>(zrow + tid * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  __syncthreads();

  const int ksteps = (M + 31) >> 5;
  for (int ks = 0; ks < ksteps; ++ks) {
    // the two pixels (of the 8 this lane-group covers) whose rows THIS lane hands to the transposing reads
    int pl[2], py[2], pxx[2];
    bool pin[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      pl[h] = ks * 32 + kq * 8 + h * 4 + rsel;  // local pixel index
      const int pg = p0 + pl[h];
      pin[h] = pl[h] < M;
      const int r = pg % HW;
      py[h] = r / p.W;
      pxx[h] = r - py[h] * p.W;
    }
    // B operand (dY): 8 pixels x 16 output channels per tile
    hx8 bfr[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) {
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4*)(pin[0] ? yt + pl[0] * p.yrow + c * 32 + piece * 8 : zrow + piece * 8));
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4*)(pin[1] ? yt + pl[1] * p.yrow + c * 32 + piece * 8 : zrow + piece * 8));
      s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      bfr[c] = __builtin_bit_cast(hx8, t);
    }
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int tap = ptap[i];
      if (tap < 0) continue;  // wave-uniform
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const char* a0 = zrow + piece * 8;
      const char* a1 = zrow + piece * 8;
      if (pin[0] && (unsigned)(py[0] + dy) < (unsigned)p.H && (unsigned)(pxx[0] + dx) < (unsigned)p.W)
        a0 = xt + (pl[0] + halo + dy * p.W + dx) * p.xrow + pci[i] * 32 + piece * 8;
      if (pin[1] && (unsigned)(py[1] + dy) < (unsigned)p.H && (unsigned)(pxx[1] + dx) < (unsigned)p.W)
        a1 = xt + (pl[1] + halo + dy * p.W + dx) * p.xrow + pci[i] * 32 + piece * 8;
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
      s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      const hx8 afr = __builtin_bit_cast(hx8, t);
#pragma unroll
      for (int c = 0; c < COT; ++c) acc[i][c] = H16<H>::mfma(afr, bfr[c], acc[i][c]);
    }
  }

  }  // sub-chunks

  // D row = kq*4 + r (ci), col = l16 (co)  ->  slab [g][tap][ci][co]
  float* slab = p.part + (long)g * 9 * p.Ci * p.Co;
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    if (ptap[i] < 0) continue;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
      const int co = cob * p.CoB + c * 16 + l16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = cib * p.CiB + pci[i] * 16 + kq * 4 + r;
        slab[((long)ptap[i] * p.Ci + ci) * p.Co + co] = acc[i][c][r];
      }
    }
  }
}

// ------------------------------------------------------------------ f32 wgrad staged through LDS
// Same decomposition as the bf16 kernel above, on the exact f32 MFMA (16x16x4: K = 4 pixels).  f32 fragments need no
// transposing read: lane (channel l&15, pixel l>>4) reads one float; rows are padded so the 4 pixel rows of a read land
// in distinct bank groups.  A workgroup of 8 waves stages X (+-(W+1)-pixel halo) and dY of a pixel run once; the
// (input-channel tile, tap) pairs are dealt round-robin to the waves, each pair holding COT accumulator tiles.
struct WgradLdsArgsF {
  const float* x;   // [N,H,W,Ci]
  const float* dy;  // [N,H,W,Co]
  float* part;      // [G][9][Ci][Co]
  int N, H, W, Ci, Co, P;
  int chunk;        // pixels per workgroup (multiple of 8)
  int ciBlocks, coBlocks;
  int CiB, CoB;     // channels per block (16 * CIT, 16 * COT)
  int xrow, yrow;   // LDS floats per pixel of the X / dY tiles (padded)
  int xfloats;      // size of the X tile in floats
  int xcd;
};

template <int CIT, int COT>
__global__ __launch_bounds__(512) void conv_wgrad_lds_f32_kernel(WgradLdsArgsF p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPW = (CIT * 9 + 7) / 8;  // (ci tile, tap) pairs per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, kq = lane >> 4;
  int g, byl;
  xcd_tile(p.xcd, g, byl);
  const int cob = byl % p.coBlocks, cib = byl / p.coBlocks;
  const int halo = p.W + 1;
  float* xt = reinterpret_cast<float*>(smem);  // [(chunk + 2*halo)][xrow]
  float* yt = xt + p.xfloats;                  // [chunk][yrow]
  const int zoff = p.xfloats + p.chunk * p.yrow;  // 16 zero floats

  const int p0 = g * p.chunk;
  const int M = min(p.chunk, p.P - p0);
  // ---- stage X (pixels p0-halo .. p0+chunk+halo) and dY (p0 .. p0+chunk); pixels outside the tensor are zero
  {
    const int xpcs = p.CiB / 4, ypcs = p.CoB / 4;  // 16-byte pieces per pixel
    const int nx = (p.chunk + 2 * halo) * xpcs, ny = p.chunk * ypcs;
    const float* xg = p.x + (long)cib * p.CiB;
    const float* yg = p.dy + (long)cob * p.CoB;
    for (int i0 = tid; i0 < nx + ny; i0 += 512 * 4) {
      f32x4 v[4];
      int dsto[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 512;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        dsto[u] = -1;
        if (i < nx) {
          const int px = i / xpcs, pc = i - px * xpcs;
          const long pg = (long)p0 - halo + px;
          dsto[u] = px * p.xrow + pc * 4;
          if (pg >= 0 && pg < p.P) v[u] = *reinterpret_cast<const f32x4*>(xg + pg * p.Ci + pc * 4);
        } else if (i < nx + ny) {
          const int k = i - nx;
          const int px = k / ypcs, pc = k - px * ypcs;
          const long pg = (long)p0 + px;
          dsto[u] = p.xfloats + px * p.yrow + pc * 4;
          if (pg < p.P) v[u] = *reinterpret_cast<const f32x4*>(yg + pg * p.Co + pc * 4);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (dsto[u] >= 0) *reinterpret_cast<f32x4*>(xt + dsto[u]) = v[u];
    }
    if (tid < 4) *reinterpret_cast<f32x4*>(xt + zoff + tid * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // pairs of this wave: q = wave + 8*i -> (ci tile, tap); per-pair LDS shift of the tap
  int pci[NPW], pdy[NPW], pdx[NPW], psh[NPW];
  bool pok[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int q = wave + 8 * i;
    pok[i] = q < CIT * 9;
    const int qq = pok[i] ? q : 0;
    pci[i] = qq / 9;
    const int tap = qq - pci[i] * 9;
    pdy[i] = tap / 3 - 1;
    pdx[i] = tap - (tap / 3) * 3 - 1;
    psh[i] = (halo + pdy[i] * p.W + pdx[i]) * p.xrow + pci[i] * 16 + c16;
  }
  f32x4 acc[NPW][COT];
#pragma unroll
  for (int i = 0; i < NPW; ++i)
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // image coordinates of this lane's pixel of the current K step (pixel p0 + ks*4 + kq)
  int oy, ox;
  {
    const int r = (p0 + kq) % (p.H * p.W);
    oy = r / p.W;
    ox = r - oy * p.W;
  }
  __syncthreads();

  const int ksteps = (M + 3) >> 2;
  int pl = kq;
  auto step = [&]() {
    float bfr[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) bfr[c] = yt[pl * p.yrow + c * 16 + c16];
    const int xb = pl * p.xrow;
    float afr[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const bool in = (unsigned)(oy + pdy[i]) < (unsigned)p.H && (unsigned)(ox + pdx[i]) < (unsigned)p.W;
      afr[i] = xt[in ? xb + psh[i] : zoff + c16];
    }
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      if (!pok[i]) continue;  // wave-uniform
#pragma unroll
      for (int c = 0; c < COT; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[i], bfr[c], acc[i][c], 0, 0, 0);
    }
    pl += 4;
    ox += 4;
    const bool w = ox >= p.W;  // W >= 4 (plan): at most one row wrap per step
    ox -= w ? p.W : 0;
    oy += w ? 1 : 0;
    oy -= oy >= p.H ? p.H : 0;
  };
  int ks = 0;
  for (; ks + 2 <= ksteps; ks += 2) {
    step();
    step();
  }
  if (ks < ksteps) step();

  // D row = kq*4 + r (ci), col = c16 (co)  ->  slab [g][tap][ci][co]
  float* slab = p.part + (long)g * 9 * p.Ci * p.Co;
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    if (!pok[i]) continue;
    const int tap = (pdy[i] + 1) * 3 + pdx[i] + 1;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
      const int co = cob * p.CoB + c * 16 + c16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = cib * p.CiB + pci[i] * 16 + kq * 4 + r;
        slab[((long)tap * p.Ci + ci) * p.Co + co] = acc[i][c][r];
      }
    }
  }
}

// ------------------------------------------------------------------ host side
static int pick_nt(int tiles) {
  const int cand[5] = {6, 4, 3, 2, 1};
  int best = 1;
  double bs = 1e30;
  for (int i = 0; i < 5; ++i) {
    const int c = cand[i];
    const int padded = ((tiles + c - 1) / c) * c;
    const double s = padded * (1.0 + 0.5 / c);
    if (s < bs - 1e-9) {
      bs = s;
      best = c;
    }
  }
  return best;
}

static int pick_small(int tiles) {  // wgrad tile counts in {4,3,2,1}
  int best = 1;
  double bs = 1e30;
  for (int c = 4; c >= 1; --c) {
    const int padded = ((tiles + c - 1) / c) * c;
    const double s = padded * (1.0 + 0.3 / c);
    if (s < bs - 1e-9) {
      bs = s;
      best = c;
    }
  }
  return best;
}

// A weight-gradient call made through fami_conv2d_wgrad_defer_* records its reduce here instead of launching it
static thread_local ReduceDesc* g_defer = nullptr;
static ReduceDesc reduce_desc_taps(const float* part, float* dw, int Co, int Ci, int taps, int psplit, int accumulate) {
  ReduceDesc d;
  d.part = part; d.dw = dw; d.Co = Co; d.Ci = Ci; d.taps = taps; d.psplit = psplit; d.accumulate = accumulate;
  const long n = (long)Co * Ci * taps;
  if (Co % 4 == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0) {
    int sg = 0;
    while (sg < 4 && (8 << sg) <= psplit) ++sg;  // up to 16 slab groups, each at least 4 slabs deep
    d.kind = 1; d.sg = sg; d.blocks = fami_cdiv(n / 4, 256 >> sg);
  } else {
    d.kind = 2; d.sg = 0; d.blocks = fami_cdiv(n, 64);
  }
  return d;
}
static void launch_reduce_taps(const float* part, float* dw, int Co, int Ci, int taps, int psplit, int accumulate,
                               hipStream_t s) {
  const ReduceDesc d = reduce_desc_taps(part, dw, Co, Ci, taps, psplit, accumulate);
  if (g_defer) {
    *g_defer = d;
    return;
  }
  if (d.kind == 1)
    hipLaunchKernelGGL(wgrad_reduce_taps4_kernel, dim3(d.blocks), dim3(256), 0, s, part, dw, Co, Ci, taps, psplit, accumulate, d.sg);
  else
    hipLaunchKernelGGL(wgrad_reduce_taps_kernel, dim3(d.blocks), dim3(1024), 0, s, part, dw, Co, Ci, taps, psplit, accumulate);
}
static void launch_reduce_plain(const float* part, float* dw, int Co, int Ci, int taps, int psplit, int accumulate,
                                hipStream_t s) {
  const long n = (long)Co * Ci * taps;
  if (g_defer) {
    ReduceDesc d;
    d.part = part; d.dw = dw; d.Co = Co; d.Ci = Ci; d.taps = taps; d.psplit = psplit; d.accumulate = accumulate;
    d.kind = 0; d.sg = 0; d.blocks = fami_cdiv(n, 64);
    *g_defer = d;
    return;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(fami_cdiv(n, 64)), dim3(256), 0, s, part, dw, n, psplit, accumulate);
}

// [fami_route_t] g_prio (default 0)  // fami_conv_tune_stages(120 / 121): s_setprio in the f32 MFMA kernels off / on
// [fami_route_t] g_lin_conv (default 1)  // fami_conv_tune_stages(100 / 101): linear-address form of the f32 implicit GEMM off / on
// [fami_route_t] g_par (default 1)  // fami_conv_tune_stages(110 / 111): parity-class stride-2 input gradient off / on
// pixel tiles of MT*16 pixels: all of them, or (stride-2 dgrad by parity class) the sum over the four classes
static long igemm_tiles(const ConvArgs& a, int MT) {
  if (!a.par) return fami_cdiv(a.P, MT * 16);
  long t = 0;
  for (int c = 0; c < 4; ++c) {
    const long Ha = (c >> 1) ? a.Ho / 2 : (a.Ho + 1) / 2, Wb = (c & 1) ? a.Wo / 2 : (a.Wo + 1) / 2;
    t += fami_cdiv(a.N * Ha * Wb, MT * 16);
  }
  return t;
}
template <int MODE, int VEC>
static int launch_igemm(const ConvArgs& a, int MT, int NT, int KS, int ST, hipStream_t s) {
  const dim3 grid(fami_cdiv(igemm_tiles(a, MT), 4 / KS), fami_cdiv(a.NTt, NT));
  if constexpr (VEC) {
    if (g_lin_conv && MT == 1 && ST == 2 && a.sh == 0 && a.Ci % 16 == 0 && a.kh * a.kw <= 25 && (NT == 3 || NT == 4)) {
#define FAMI_LIN(nt, ks)                                                                            \
  if (NT == nt && KS == ks) {                                                                       \
    hipLaunchKernelGGL((conv_igemm_f32<1, nt, MODE, 1, ks, 2, 1>), grid, dim3(256), 0, s, a);       \
    return 0;                                                                                       \
  }
      FAMI_LIN(3, 1) FAMI_LIN(3, 2) FAMI_LIN(3, 4) FAMI_LIN(4, 1) FAMI_LIN(4, 2) FAMI_LIN(4, 4)
#undef FAMI_LIN
    }
  }
#define FAMI_CASE(mt, nt, ks)                                                                       \
  if (MT == mt && NT == nt && KS == ks) {                                                           \
    if (ST == 3 && mt <= 2) hipLaunchKernelGGL((conv_igemm_f32<mt, nt, MODE, VEC, ks, 3>), grid, dim3(256), 0, s, a); \
    else if (ST == 4 && mt == 1) hipLaunchKernelGGL((conv_igemm_f32<mt, nt, MODE, VEC, ks, 4>), grid, dim3(256), 0, s, a); \
    else hipLaunchKernelGGL((conv_igemm_f32<mt, nt, MODE, VEC, ks, 2>), grid, dim3(256), 0, s, a);   \
    return 0;                                                                                       \
  }
  if constexpr (VEC) {
#define FAMI_ROW(nt) FAMI_CASE(1, nt, 1) FAMI_CASE(1, nt, 2) FAMI_CASE(1, nt, 4) FAMI_CASE(2, nt, 1) FAMI_CASE(2, nt, 2) FAMI_CASE(2, nt, 4)
    FAMI_ROW(1) FAMI_ROW(2) FAMI_ROW(3) FAMI_ROW(4) FAMI_ROW(6)
    FAMI_CASE(4, 3, 1) FAMI_CASE(4, 4, 1)
#undef FAMI_ROW
  } else {
    FAMI_CASE(2, 1, 1) FAMI_CASE(2, 2, 1) FAMI_CASE(2, 3, 1) FAMI_CASE(2, 4, 1)
  }
#undef FAMI_CASE
  return -1;
}

// [fami_route_t] g_force_mt (default 0), g_force_nt (default 0), g_force_ks (default 0)  // tuning overrides (fami_conv_tune)
// [fami_route_t] g_stages (default 0)  // pipeline depth override (fami_conv_tune_stages)

// [fami_route_t] g_use32 (default 1)  // fami_conv_tune(-1, ...) disables the 32x32-tile f32 kernel (benchmarks / tests)

// 32x32x2 path: eligible when the weight image carries the 32-tile section (N >= 32, K % 4 == 0)
// [fami_route_t] g_xcd_w (default 1)  // same switch for the weight-gradient kernels (fami_conv_tune_xcd bit 1)
// [fami_route_t] g_xcd (default -1)  // fami_conv_tune_xcd: 0 natural tile order, 1 XCD-contiguous, -1 default (= 1: PMC FETCH_SIZE of the
                        // 48->48 3x3 @96x72 N=20 launch drops from 46.6 MB to 13.7 MB, time -1..-2 %; tools/bench_xcd.py)

static int run_igemm32(ConvArgs a, int mode, hipStream_t s, const char* name) {
  a.xcd = g_xcd < 0 ? 1 : g_xcd;
  a.prio = g_prio;
  const long n16 = pack16_elems(a.Ci, a.Co, a.kh * a.kw);
  a.wp = a.wp + n16;
  a.KC = fami_cdiv(a.Ci, 8);
  a.NTt = fami_cdiv(a.Co, 32);
  a.wp_bytes = (unsigned)((long)a.kh * a.kw * a.KC * a.NTt * 1024);
  // tile choice: up to 3 channel tiles per wave (48 accumulator registers); fewer tiles / split-K when the pixel
  // count alone cannot fill the 1024 SIMDs
  int NT = a.NTt >= 3 && a.NTt % 3 == 0 ? 3 : (a.NTt % 2 == 0 ? 2 : (a.NTt >= 3 ? 3 : a.NTt));
  const long mt = fami_cdiv(a.P, 32);
  int KS = 1;
  const long iters = (long)a.kh * a.kw * a.KC;
  if (mt * fami_cdiv(a.NTt, NT) < 2048 && NT > 1) NT = a.NTt % 2 == 0 && NT == 3 ? 2 : 1;
  if (mt * fami_cdiv(a.NTt, NT) < 2048 && NT > 1) NT = 1;
  const long waves = mt * fami_cdiv(a.NTt, NT);
  if (iters >= 16) {
    if (waves < 1536) KS = 4;
    else if (waves < 3072) KS = 2;
  }
  if (g_force_nt > 0 && g_force_mt == 32) { NT = g_force_nt; KS = g_force_ks ? g_force_ks : 1; }
  const dim3 grid(fami_cdiv(a.P, (4 / KS) * 32), fami_cdiv(a.NTt, NT));
  bool ok = false;
#define FAMI_C32(nt, ks)                                                                          \
  if (NT == nt && KS == ks) {                                                                     \
    if (mode == 0) hipLaunchKernelGGL((conv_igemm32_f32<nt, 0, ks>), grid, dim3(256), 0, s, a);   \
    else hipLaunchKernelGGL((conv_igemm32_f32<nt, 1, ks>), grid, dim3(256), 0, s, a);             \
    ok = true;                                                                                    \
  }
  FAMI_C32(1, 1) FAMI_C32(1, 2) FAMI_C32(1, 4) FAMI_C32(2, 1) FAMI_C32(2, 2) FAMI_C32(2, 4) FAMI_C32(3, 1) FAMI_C32(3, 2) FAMI_C32(3, 4)
#undef FAMI_C32
  if (!ok) {
    fami_set_error(name, "no 32x32 kernel instance");
    return FAMI_ESHAPE;
  }
  FAMI_CHECK_LAUNCH(name);
  return FAMI_OK;
}

static int run_igemm(ConvArgs a, int mode, hipStream_t s, const char* name) {
  a.xcd = g_xcd < 0 ? 1 : g_xcd;
  a.prio = g_prio;
  a.par = (mode == 1 && a.sh == 1 && g_par) ? 1 : 0;
  const int vec = (a.Ci % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
  // (the 32x32-tile kernel has no EpiBN epilogue: a fused call takes the 16x16 kernels)
  if (vec && !a.e.slots && g_use32 && (g_force_mt == 0 || g_force_mt == 32) && pack32_elems(a.Ci, a.Co, a.kh * a.kw) > 0 &&
      (long)a.kh * a.kw * fami_cdiv(a.Ci, 8) * fami_cdiv(a.Co, 32) * 1024 < (1L << 31))
    return run_igemm32(a, mode, s, name);
  int NT = pick_nt(a.NTt);
  int MT, KS = 1;
  if (!vec) {
    MT = 2;
    if (NT > 4) NT = 4;
  } else {
    // Measured on MI355X (tests/bench_conv.py): the kernel is latency-bound, so the smallest register tile
    // (highest occupancy) wins; split-K restores parallelism on the low-resolution branches.
    if (NT == 6) NT = 3;
    const long nblk = fami_cdiv(a.NTt, NT);
    const long waves = (long)fami_cdiv(a.P, 16) * nblk;
    MT = 1;
    const long iters = (long)a.kh * a.kw * a.KC;
    if (iters >= 16) {
      if (waves < 3000) KS = 4;
      else if (waves < 6000) KS = 2;
    }
  }
  if (vec && g_force_mt && g_force_mt != 16) { MT = g_force_mt; NT = g_force_nt ? g_force_nt : NT; KS = g_force_ks ? g_force_ks : 1; }
  const int ST = g_stages ? g_stages : 2;
  int rc;
  if (mode == 0)
    rc = vec ? launch_igemm<0, 1>(a, MT, NT, KS, ST, s) : launch_igemm<0, 0>(a, MT, NT, KS, 2, s);
  else
    rc = vec ? launch_igemm<1, 1>(a, MT, NT, KS, ST, s) : launch_igemm<1, 0>(a, MT, NT, KS, 2, s);
  if (rc != 0) {
    fami_set_error(name, "no kernel instance for tile shape");
    return FAMI_ESHAPE;
  }
  FAMI_CHECK_LAUNCH(name);
  return FAMI_OK;
}

// The LDS-staged 3x3 kernel cuts L2 requests ~5x.  Measured on MI355X: launch by launch it does not beat the direct
// kernels on the HRNet shapes (f32 48ch: 75 vs 67 us; bf16: 24.7 vs 21.5 us, tools/bench_dvfs.py), but inside the
// training step -- where the branch lanes run several convs at once and share the L2s -- the bf16 step is 4 % faster
// with it (42.3 -> 40.6 ms) and the f32 step 4 % slower (77.3 -> 80.7 ms; interleaved A/B, tools/ab_step.py).
// Default (-1): bf16 staged, f32 direct.  fami_conv_tune_lds(0/1) forces one path for both (tests exercise both).
// [fami_route_t] g_use_lds (default -1)
// [fami_route_t] g_lds_sim (default 0)  // fami_conv_tune_lds(2): LDS kernel in its split-operand cost-simulation form (benchmarks)
// [fami_route_t] g_wgrad_nsub (default 2)  // sub-chunks per workgroup of the 16-bit LDS wgrad (fewer, larger partial slabs): 2 = half the slab
                             // traffic and reduce pass, bf16 step 34.6 -> 33.3 ms in both A/B orders; 3 and 4 equal to 2
// [fami_route_t] g_wgrad_ps (default 0)  // fami_conv_tune_wgrad_lds(1000 + n): pixel-split target of the per-tap f32 wgrad (benchmarks)
// [fami_route_t] g_wgrad_mt (default 0)  // fami_conv_tune_wgrad_lds(100 + mt): cap on input-channel tiles per f32 wgrad workgroup
// [fami_route_t] g_wgrad_lds (default 1)  // fami_conv_tune_wgrad_lds(0): bf16 weight gradients on the scalar-operand kernels
// [fami_route_t] g_wgrad_lds_f32 (default 2)  // f32 LDS weight gradient: 0 never, 1 whenever eligible, 2 only where it measured faster
// [fami_route_t] g_wgrad_lin (default 1)  // fami_conv_tune_wgrad_lds(50 / 51): linear-address per-tap f32 kernel off / on

// LDS-staged path: plan + launch.  Returns 1 if launched, 0 if the shape is not eligible, <0 on error.
template <typename T>
static int try_conv3x3_lds(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci, int Co,
                           int KC, int NTt, int sgn, int relu, int accumulate, int out_f32, hipStream_t s,
                           const char* name, const EpiBN& epi = epi_none(), const XBN& xbn = xbn_none()) {
  constexpr int SZ = (int)sizeof(T), KSTEP = LdsTraits<T>::KSTEP;
  if (g_use_lds != 0) {   // fami_conv_tune_lds(0) still forces the direct kernels
    // register-blocked LDS kernel (conv_t4.hip): every storage type since round 3
    const int rc = fami_try_conv3x3_t4(std::is_same<T, float>::value ? 2 : (std::is_same<T, f16_t>::value ? 1 : 0), x, wp, bias, y,
                                       N, H, W, Ci, Co, KC, NTt, sgn, relu, accumulate, out_f32, s, name, epi, xbn);
    if (rc != 0) return rc;
  }
  if (xbn.on) return 0;   // only the register-blocked kernel applies a BatchNorm to its input
  // default: bf16 from 96 input channels up (per launch: 192 ch 18.8 vs 30.3 us, 384 ch 27.8 vs 32.4, 96 ch equal,
  // 48 ch 23.9 vs 21.8 -> direct; tools/bench_xcd.py with KNOB=lds), f32 never (slower on every shape)
  const int use = g_use_lds < 0 ? (SZ == 2 && Ci >= 96 ? 1 : 0) : g_use_lds;
  if (!use || W > 144 || (Ci * SZ) % 16 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0) return 0;
  int NT = 0;
  if (Co % 48 == 0) NT = 3;
  else if (Co % 64 == 0) NT = 4;
  if (!NT) return 0;
  // channel chunk staged per pass
  int CH = 0;
  const int cand_f32[4] = {48, 64, 32, 16}, cand_bf16[3] = {96, 64, 32};
  if (SZ == 4) { for (int i = 0; i < 4 && !CH; ++i) if (Ci % cand_f32[i] == 0) CH = cand_f32[i]; }
  else { for (int i = 0; i < 3 && !CH; ++i) if (Ci % cand_bf16[i] == 0) CH = cand_bf16[i]; }
  if (!CH && Ci <= (SZ == 4 ? 64 : 96)) CH = Ci;  // single chunk, padded to a whole K group in LDS
  if (!CH) return 0;
  ConvLdsArgs a;
  a.CH = CH;
  a.CHP = ((CH + KSTEP - 1) / KSTEP) * KSTEP;
  if (a.CHP / KSTEP > 4) return 0;
  a.PSTRIDE = a.CHP * SZ;
  if (((a.PSTRIDE / 16) & 1) == 0) a.PSTRIDE += 16;
  a.R = 144 / W;
  if (a.R < 1) a.R = 1;
  if (a.R > H) a.R = H;
  a.bands = fami_cdiv(H, a.R);
  a.patch_bytes = (a.R + 2) * W * a.PSTRIDE;
  const size_t lds = (size_t)a.patch_bytes + 2 * (size_t)(SZ == 2 ? 3 : 1) * (a.CHP / KSTEP) * NT * 1024;
  if (lds > 156 * 1024) return 0;
  a.x = x; a.wp = wp; a.y = y; a.bias = bias;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.KC = KC; a.NTt = NTt; a.sgn = sgn;
  a.relu = relu; a.accumulate = accumulate; a.out_f32 = out_f32; a.simz = 0;
  a.e = epi; a.emode = epi.slots ? epi.mode : 0;
  const dim3 grid(N * a.bands, Co / (16 * NT));
  const int KSC = a.CHP / KSTEP;
  bool launched = false;
#define FAMI_LDS_CASE(nt, ksc)                                                                                     \
  if (NT == nt && KSC == ksc) {                                                                                    \
    static bool attr = false;                                                                                      \
    if (!attr) {                                                                                                   \
      (void)hipFuncSetAttribute((const void*)conv3x3_lds_kernel<T, nt, ksc>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024); \
      attr = true;                                                                                                 \
    }                                                                                                              \
    if (g_lds_sim && SZ == 2) {                                                                                    \
      static bool attr2 = false;                                                                                   \
      if (!attr2) {                                                                                                \
        (void)hipFuncSetAttribute((const void*)conv3x3_lds_kernel<T, nt, ksc, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024); \
        attr2 = true;                                                                                              \
      }                                                                                                            \
      hipLaunchKernelGGL((conv3x3_lds_kernel<T, nt, ksc, 1>), grid, dim3(256), lds, s, a);                         \
    } else                                                                                                         \
    hipLaunchKernelGGL((conv3x3_lds_kernel<T, nt, ksc>), grid, dim3(256), lds, s, a);                              \
    launched = true;                                                                                               \
  }
  FAMI_LDS_CASE(3, 1) FAMI_LDS_CASE(3, 2) FAMI_LDS_CASE(3, 3) FAMI_LDS_CASE(3, 4)
  FAMI_LDS_CASE(4, 1) FAMI_LDS_CASE(4, 2) FAMI_LDS_CASE(4, 3) FAMI_LDS_CASE(4, 4)
#undef FAMI_LDS_CASE
  if (!launched) return 0;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fami_set_error(name, hipGetErrorString(e));
    return FAMI_EHIP;
  }
  return 1;
}

static bool wgrad_lin_ok(int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil) {
  return stride == 1 && 2 * pad == dil * (kh - 1) && 2 * pad == dil * (kw - 1) && kh * kw > 1 && kh * kw <= 9 && W >= 4 &&
         Ci % 16 == 0 && Co % 16 == 0;
}
static bool geom_ok(int kh, int kw, int stride, int pad, int dil) {
  return kh >= 1 && kw >= 1 && kh <= 7 && kw <= 7 && (stride == 1 || stride == 2) && pad >= 0 && dil >= 1;
}
static inline int out_dim(int i, int k, int stride, int pad, int dil) {
  return (i + 2 * pad - dil * (k - 1) - 1) / stride + 1;
}

extern "C" {
int fami_dcn_tune(int mode);          // align.hip
int fami_bn_tune_small(long elems);   // norm.hip

// tuning hook (benchmarks only): force the implicit-GEMM tile (0 = heuristic)
int fami_conv_tune(int mt, int nt, int ks) {
  // mt = 16: force the 16x16-tile f32 kernel with its heuristics (nt = ks = 0) ; mt = 32: force (nt, ks) of the 32x32-tile
  // kernel ; other mt > 0: force the (mt, nt, ks) tile of the 16x16 kernels ; 0: heuristics
  g_force_mt = mt; g_force_nt = nt; g_force_ks = ks;
  return FAMI_OK;
}
// 1 routes eligible 3x3 stride-1 convolutions through the LDS-staged kernel, 0 through the direct kernels,
// -1 = default (bf16: staged, f32: direct)
void fami_conv_stem_tune(int on);
int fami_conv_tune_lds(int on) {
  if (on == 9000 || on == 9001) {            // conv_stem.hip (the stem's first convolution, 16-bit forward): off / on
    fami_conv_stem_tune(on - 9000);
    return FAMI_OK;
  }
  if (on < 0) fami_conv_stem_tune(-1);
  if (on == 10 || on == 11 || on == 20 || on == 21 || on == 30 || on == 31 || (on >= 40 && on <= 42) || (on >= 52 && on <= 54) || (on >= 60 && on <= 62) || on >= 100) {   // (30 / 31: split-product f32 instance) register-blocked kernel (conv_t4.hip): 10 / 11 off / on (20 / 21: its f32 instance);
    fami_conv_t4_tune(on);                   // 100 + tiles per band (100 = heuristic)
    return FAMI_OK;
  }
  g_use_lds = on < 0 ? -1 : (on ? 1 : 0);
  g_lds_sim = on == 2;
  if (on < 0) fami_conv_t4_tune(-1);
  return FAMI_OK;
}
// 0 routes bf16 weight gradients through the scalar-operand f32-MFMA kernels (benchmarks / tests)
int fami_conv_tune_wgrad_lds(int on) {
  if (on >= 30000) {         // split-product f32 kernel (conv_wgs3.hip): 30000 / 30001 off / on, 30100 + tiles per run,
    fami_wgrad_s3_tune(on - 30000);  // 31000 + workgroup target
    return FAMI_OK;
  }
  if (on >= 20000) {         // round-3 16-bit kernel (conv_wg16.hip): 20000 / 20001 off / on, 20100 + tiles per run,
    fami_wgrad16_tune(on - 20000);   // 21000 + workgroup target
    return FAMI_OK;
  }
  if (on >= 1000) {
    g_wgrad_ps = on - 1000;
    return FAMI_OK;
  }
  if (on >= 50 && on <= 54) {  // benchmarks / tests: general (50) / linear-address per-tap f32 kernel (51: default mix; 52: 9-wave workgroups; 53: 8-wave; 54: 16-wave, two chunks per slab)
    g_wgrad_lin = on - 50;
    return FAMI_OK;
  }
  if (on >= 100) {  // benchmarks: 100 + mt caps the input-channel tiles per workgroup of the f32 kernels
    g_wgrad_mt = on - 100;
    return FAMI_OK;
  }
  if (on < 0) {  // defaults
    fami_wgrad16_tune(-1);
    fami_wgrad_s3_tune(-1);
    g_wgrad_lds = 1;
    g_wgrad_lds_f32 = 2;
    g_wgrad_nsub = 2;
    g_wgrad_lin = 1;
    return FAMI_OK;
  }
  g_wgrad_lds = on ? 1 : 0;
  g_wgrad_lds_f32 = on ? 1 : 0;
  if (on > 1) g_wgrad_nsub = on - 1;  // benchmarks: on = 1 + sub-chunks per workgroup
  return FAMI_OK;
}
// tuning hook (benchmarks only): register-pipeline depth of the implicit GEMM (2..4; 0 = default)
int fami_conv_tune_xcd(int mode) {
  g_xcd = mode < 0 ? -1 : (mode & 1);
  g_xcd_w = mode < 0 ? 1 : ((mode >> 1) & 1);
  return FAMI_OK;
}
int fami_conv_tune_stages(int stages) {
  if (stages == 120 || stages == 121) {   // s_setprio in the f32 MFMA kernels off / on
    g_prio = stages - 120;
    return FAMI_OK;
  }
  if (stages == 110 || stages == 111) {   // benchmarks / tests: parity-class stride-2 input gradient off / on
    g_par = stages - 110;
    return FAMI_OK;
  }
  if (stages == 100 || stages == 101) {   // benchmarks / tests: linear-address f32 implicit GEMM off / on
    g_lin_conv = stages - 100;
    return FAMI_OK;
  }
  g_stages = (stages >= 2 && stages <= 4) ? stages : 0;
  return FAMI_OK;
}
// Every benchmark / test knob of the library back to its default.  The knobs are process-wide (they select between
// kernels that compute the same function); tests call this from an autouse fixture, so a test that forgets its
// `finally` cannot poison the rest of the suite.
int fami_tune_reset(void) {
  fami_conv_tune(0, 0, 0);
  fami_conv_tune_lds(-1);
  fami_conv_tune_wgrad_lds(-1);
  fami_conv_tune_xcd(-1);
  g_stages = 0; g_prio = 0; g_lin_conv = 1; g_par = 1;
  g_wgrad_ps = 0; g_wgrad_mt = 0;
  fami_dcn_tune(-1);
  fami_bn_tune_small(-1);
  return FAMI_OK;
}
// The library's DEFAULT f32 arithmetic for 3x3 stride-1 convolutions: 1 = split products on the bf16 matrix pipe,
// 0 = exact-f32 MFMA (FAMI_F32_SPLIT=0).  Stored as the default state, so fami_tune_reset / fami_conv_tune_lds(-1)
// restore it instead of silently re-enabling the split kernels.
int fami_tune_defaults(int f32_split) {
  if (f32_split >= 0) {
    fami_conv_t4_default_split(f32_split);
    fami_wgrad_s3_default(f32_split);
  }
  return fami_tune_reset();
}

// f32 image = [16x16 fragment image][32x32 fragment image (wide 1x1 only)][split image (3x3, K % 16 == 0: three bf16 planes of
// every weight, the layout conv_t5.hip copies straight into LDS; fami_split_image_elems)]
long fami_packed_weight_elems(int Co, int Ci, int kh, int kw, int mode) {
  const int kd = mode == 0 ? Ci : Co, nd = mode == 0 ? Co : Ci;
  return pack16_elems(kd, nd, kh * kw) + pack32_elems(kd, nd, kh * kw) + fami_split_image_elems(kd, nd, kh * kw);
}

int fami_pack_conv_weight_f32(const float* w_oihw, float* wp, int Co, int Ci, int kh, int kw, int mode,
                              hipStream_t s) {
  FAMI_REQUIRE(w_oihw && wp && Co > 0 && Ci > 0 && (mode == 0 || mode == 1), "fami_pack_conv_weight_f32", "bad argument");
  const int kd = mode == 0 ? Ci : Co, nd = mode == 0 ? Co : Ci;
  const long total = pack16_elems(kd, nd, kh * kw) + pack32_elems(kd, nd, kh * kw);
  hipLaunchKernelGGL(pack_w_kernel, dim3(fami_ew_grid(total)), dim3(256), 0, s, w_oihw, wp, Co, Ci, kh * kw, mode);
  FAMI_CHECK_LAUNCH("fami_pack_conv_weight_f32");
  if (fami_split_image_elems(kd, nd, kh * kw) > 0) {
    fami_pack_split_single(w_oihw, wp + total, Co, Ci, kh * kw, mode, s);
    FAMI_CHECK_LAUNCH("fami_pack_conv_weight_f32/split");
  }
  return FAMI_OK;
}

// y[N,Ho,Wo,Co] = conv(x[N,H,W,Ci], W) (+bias) (+addend) (relu) ; wp packed with mode 0
static int conv_fwd_f32_impl(const char* nm, const float* x, const float* wp, const float* bias, const float* addend,
                             float* y, int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil,
                             int relu, int accumulate, const EpiBN& e, hipStream_t s, const XBN& xbn = xbn_none());
int fami_conv2d_fwd_f32(const float* x, const float* wp, const float* bias, const float* addend, float* y, int N,
                        int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int relu,
                        int accumulate, hipStream_t s) {
  return conv_fwd_f32_impl("fami_conv2d_fwd_f32", x, wp, bias, addend, y, N, H, W, Ci, Co, kh, kw, stride, pad, dil, relu,
                           accumulate, epi_none(), s);
}
// the same (no addend / relu / accumulate: a BatchNorm follows) with the BatchNorm statistics of y taken in the
// epilogue: `slots` = fami_bn_slots_bytes(Co) bytes, zero on entry, handed to fami_bn_apply_slots_f32 /
// fami_bn_finalize_slots_f32 afterwards; pivot_src [Co] (the running mean) or null
int fami_conv2d_fwd_stats_f32(const float* x, const float* wp, const float* bias, float* y, int N, int H, int W, int Ci,
                              int Co, int kh, int kw, int stride, int pad, int dil, void* slots, const float* pivot_src,
                              hipStream_t s) {
  FAMI_REQUIRE(slots, "fami_conv2d_fwd_stats_f32", "null slots");
  EpiBN e = epi_none();
  e.slots = reinterpret_cast<double*>(slots); e.ns = bn_slots(Co); e.mode = 1; e.C = Co; e.pivot_src = pivot_src;
  return conv_fwd_f32_impl("fami_conv2d_fwd_stats_f32", x, wp, bias, nullptr, y, N, H, W, Ci, Co, kh, kw, stride, pad, dil,
                           0, 0, e, s);
}
// f32 form of fami_conv2d_fwd_xbn_* (see the 16-bit entry points below): z is the PRE-normalisation input; the
// split-product kernel applies BatchNorm + ReLU while it stages z
int fami_conv2d_fwd_xbn_f32(const float* z, const float* wp, const float* bias, float* y, int N, int H, int W, int Ci, int Co,
                            void* slots, const float* pivot_src, const void* xslots, long xP, const float* xgamma,
                            const float* xbeta, float* xmean, float* xinvstd, float* xrunning_mean, float* xrunning_var,
                            float xmomentum, float xeps, hipStream_t s) {
  FAMI_REQUIRE(xslots && xgamma && xbeta && xmean && xinvstd && xP > 0, "fami_conv2d_fwd_xbn_f32", "bad argument");
  EpiBN e = epi_none();
  if (slots) {
    e.slots = reinterpret_cast<double*>(slots); e.ns = bn_slots(Co); e.mode = 1; e.C = Co; e.pivot_src = pivot_src;
  }
  XBN xb = xbn_none();
  xb.on = 1; xb.slots = reinterpret_cast<const double*>(xslots); xb.ns = bn_slots(Ci); xb.C = Ci; xb.P = xP;
  xb.gamma = xgamma; xb.beta = xbeta; xb.mean = xmean; xb.invstd = xinvstd; xb.running_mean = xrunning_mean;
  xb.running_var = xrunning_var; xb.momentum = xmomentum; xb.eps = xeps;
  return conv_fwd_f32_impl("fami_conv2d_fwd_xbn_f32", z, wp, bias, nullptr, y, N, H, W, Ci, Co, 3, 3, 1, 1, 1, 0, 0, e, s, xb);
}
}  // extern "C"
int fami_try_conv_stem1(int half_kind, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci, int Co,
                        int kh, int kw, int stride, int pad, int dil, int NTt, int relu, int accumulate, int out_f32, hipStream_t s,
                        const char* name, const EpiBN& epi);      // conv_stem.hip
static int conv_fwd_f32_impl(const char* nm, const float* x, const float* wp, const float* bias, const float* addend,
                             float* y, int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil,
                             int relu, int accumulate, const EpiBN& e, hipStream_t s, const XBN& xbn) {
  FAMI_REQUIRE(x && wp && y && N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0, nm, "bad argument");
  if (!geom_ok(kh, kw, stride, pad, dil)) {
    fami_set_error(nm, "unsupported geometry");
    return FAMI_ESHAPE;
  }
  ConvArgs a;
  a.e = e; a.emode = e.slots ? e.mode : 0;
  a.x = x; a.wp = wp; a.y = y; a.bias = bias; a.addend = addend;
  a.N = N; a.Hi = H; a.Wi = W; a.Ci = Ci;
  a.Ho = out_dim(H, kh, stride, pad, dil); a.Wo = out_dim(W, kw, stride, pad, dil); a.Co = Co;
  a.kh = kh; a.kw = kw; a.sh = stride == 2 ? 1 : 0; a.pad = pad; a.dil = dil;
  a.KC = fami_cdiv(Ci, 16); a.NTt = fami_cdiv(Co, 16); a.relu = relu; a.accumulate = accumulate;
  const long P = (long)N * a.Ho * a.Wo;
  const long xb = (long)N * H * W * Ci * 4, wb = (long)kh * kw * a.KC * a.NTt * 1024;
  FAMI_REQUIRE(P > 0 && P < (1L << 31) && xb < (1L << 31) && wb < (1L << 31), nm, "tensor >= 2 GiB");
  a.P = (int)P; a.x_bytes = (unsigned)xb; a.wp_bytes = (unsigned)wb;
  if (Ci == 3 && !addend && !xbn.on) {
    const int rc = fami_try_conv_stem1(2, x, wp, bias, y, N, H, W, Ci, Co, kh, kw, stride, pad, dil, a.NTt, relu, accumulate, 1, s, nm, e);
    if (rc != 0) return rc < 0 ? rc : FAMI_OK;
  }
  if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1 && !addend) {
    const int rc = try_conv3x3_lds<float>(x, wp, bias, y, N, H, W, Ci, Co, a.KC, a.NTt, +1, relu, accumulate, 1, s, nm, e, xbn);
    if (rc != 0) return rc < 0 ? rc : FAMI_OK;
  }
  if (kh == 3 && kw == 3 && stride == 1 && pad == dil && dil > 1 && !addend && !e.slots && !xbn.on && g_use_lds != 0) {
    const int rc = fami_try_conv3x3_t4_dil(2, x, wp, bias, y, N, H, W, Ci, Co, a.KC, a.NTt, +1, relu, accumulate, 1, dil, s, nm);
    if (rc != 0) return rc < 0 ? rc : FAMI_OK;
  }
  if (xbn.on) {
    fami_set_error(nm, "input BatchNorm needs the split-product 3x3 kernel (ask fami_conv2d_xbn_ok_f32 first)");
    return FAMI_ESHAPE;
  }
  return run_igemm(a, 0, s, nm);
}
extern "C" {

// dx[N,H,W,Ci] = conv^T(dy[N,Ho,Wo,Co], W) (+addend) ; wp packed with mode 1
static int conv_dgrad_f32_impl(const char* nm, const float* dy, const float* wp, const float* addend, float* dx, int N,
                               int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                               const EpiBN& e, hipStream_t s);
int fami_conv2d_dgrad_f32(const float* dy, const float* wp, const float* addend, float* dx, int N, int H, int W,
                          int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                          hipStream_t s) {
  return conv_dgrad_f32_impl("fami_conv2d_dgrad_f32", dy, wp, addend, dx, N, H, W, Ci, Co, kh, kw, stride, pad, dil,
                             accumulate, epi_none(), s);
}
// the same as the LAST contribution to dx = dL/d(output of a train-mode BatchNorm(+ReLU) with input z): stores
// dz = relu-mask(dx) instead of dx and adds sum dz, sum dz*xhat into `slots` (zero on entry) for
// fami_bn_bwd_apply_slots_f32.  relu: 0 none, 1 mask from the BatchNorm output yrelu, 2 recomputed from z.
int fami_conv2d_dgrad_bnstats_f32(const float* dy, const float* wp, float* dx, int N, int H, int W, int Ci, int Co,
                                  int kh, int kw, int stride, int pad, int dil, int accumulate, const float* z,
                                  const float* yrelu, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, int relu, void* slots, hipStream_t s) {
  FAMI_REQUIRE(slots && z && mean && invstd && gamma && beta && (relu != 1 || yrelu) && relu >= 0 && relu <= 2,
               "fami_conv2d_dgrad_bnstats_f32", "bad argument");
  EpiBN e = epi_none();
  e.slots = reinterpret_cast<double*>(slots); e.ns = bn_slots(Ci); e.mode = 2; e.relu = relu; e.C = Ci;
  e.z = z; e.yr = yrelu; e.mean = mean; e.invstd = invstd; e.gamma = gamma; e.beta = beta;
  return conv_dgrad_f32_impl("fami_conv2d_dgrad_bnstats_f32", dy, wp, nullptr, dx, N, H, W, Ci, Co, kh, kw, stride, pad,
                             dil, accumulate, e, s);
}
}  // extern "C"
static int conv_dgrad_f32_impl(const char* nm, const float* dy, const float* wp, const float* addend, float* dx, int N,
                               int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                               const EpiBN& e, hipStream_t s) {
  FAMI_REQUIRE(dy && wp && dx && N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0, nm, "bad argument");
  if (!geom_ok(kh, kw, stride, pad, dil)) {
    fami_set_error(nm, "unsupported geometry");
    return FAMI_ESHAPE;
  }
  ConvArgs a;
  a.e = e; a.emode = e.slots ? e.mode : 0;
  a.x = dy; a.wp = wp; a.y = dx; a.bias = nullptr; a.addend = addend;
  a.N = N; a.Hi = out_dim(H, kh, stride, pad, dil); a.Wi = out_dim(W, kw, stride, pad, dil); a.Ci = Co;
  a.Ho = H; a.Wo = W; a.Co = Ci;
  a.kh = kh; a.kw = kw; a.sh = stride == 2 ? 1 : 0; a.pad = pad; a.dil = dil;
  a.KC = fami_cdiv(Co, 16); a.NTt = fami_cdiv(Ci, 16); a.relu = 0; a.accumulate = accumulate;
  const long P = (long)N * H * W;
  const long xb = (long)N * a.Hi * a.Wi * Co * 4, wb = (long)kh * kw * a.KC * a.NTt * 1024;
  FAMI_REQUIRE(P > 0 && P < (1L << 31) && xb < (1L << 31) && wb < (1L << 31), nm, "tensor >= 2 GiB");
  a.P = (int)P; a.x_bytes = (unsigned)xb; a.wp_bytes = (unsigned)wb;
  if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1 && !addend) {
    // dgrad of a stride-1 "same" conv is the same conv on dy with the taps mirrored: GEMM K = Co, N = Ci
    const int rc = try_conv3x3_lds<float>(dy, wp, nullptr, dx, N, H, W, Co, Ci, a.KC, a.NTt, -1, 0, accumulate, 1, s, nm, e);
    if (rc != 0) return rc < 0 ? rc : FAMI_OK;
  }
  if (kh == 3 && kw == 3 && stride == 1 && pad == dil && dil > 1 && !addend && !e.slots && g_use_lds != 0) {
    const int rc = fami_try_conv3x3_t4_dil(2, dy, wp, nullptr, dx, N, H, W, Co, Ci, a.KC, a.NTt, -1, 0, accumulate, 1, dil, s, nm);
    if (rc != 0) return rc < 0 ? rc : FAMI_OK;
  }
  return run_igemm(a, 1, s, nm);
}
extern "C" {

struct WgradPlan { int MT, NT, ciBlocks, coBlocks, psplit, chunk, pertap, w8; long P; };
static bool wgrad_lin_ok(int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil);
static WgradPlan wgrad_plan(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int w8 = 0) {
  WgradPlan q;
  const int Ho = out_dim(H, kh, stride, pad, dil), Wo = out_dim(W, kw, stride, pad, dil);
  q.P = (long)N * Ho * Wo;
  q.MT = pick_small(fami_cdiv(Ci, 16));
  q.NT = pick_small(fami_cdiv(Co, 16));
  if (g_wgrad_mt > 0 && g_wgrad_mt <= q.MT) q.MT = g_wgrad_mt;  // benchmarks: fewer input-channel tiles per workgroup
  q.ciBlocks = fami_cdiv(fami_cdiv(Ci, 16), q.MT);
  q.coBlocks = fami_cdiv(fami_cdiv(Co, 16), q.NT);
  q.pertap = (kh * kw > 1 && kh * kw <= 9) ? 1 : 0;   // one wave per tap (3x3): workgroups of kh*kw waves
  const long by = q.pertap ? (long)q.ciBlocks * q.coBlocks : (long)kh * kw * q.ciBlocks * q.coBlocks;
  long ps = ((q.pertap ? (g_wgrad_ps ? g_wgrad_ps : 512) : 1024) + by - 1) / by;
  q.w8 = 0;
  // w8 = 3: 16-wave workgroups while a column of the grid still has >= 16 of them (48 / 96 / 192 channels: 66.9 vs 69.2,
  // 64.3 vs 67.7, 67.1 vs 70.2 us, and half the slabs), 8-wave ones beyond (384 channels, 7 per column: 120 vs 78 us)
  if (w8 == 3) w8 = 256 / by >= 16 ? 2 : 1;
  if (w8 && q.pertap && kh * kw == 9 && ps >= 9) {
    // 8-/16-wave workgroups: c slabs need c + ceil(c/8) workgroups.  Never more workgroups than the target (one
    // workgroup beyond the resident set costs a whole extra round: 192 channels, 9 columns x 57 = 513 workgroups ran
    // 90 us against 80 us for 9 x 43)
    q.w8 = w8;
    const long T = (g_wgrad_ps ? g_wgrad_ps : (w8 == 2 ? 256 : 512)) / by;
    ps = T * 8 / 9;
    while (ps > 1 && ps + (ps + 7) / 8 > T) --ps;
    if (ps < 1) ps = 1;
    if (w8 == 2) ps *= 2;   // pixel chunks: two per slab
  }
  const long maxps = q.pertap ? (q.P + 63) / 64 : (q.P + 255) / 256;
  if (ps > maxps) ps = maxps;
  if (ps < 1) ps = 1;
  long chunk = (q.P + ps - 1) / ps;
  chunk = ((chunk + 15) / 16) * 16;
  ps = (q.P + chunk - 1) / chunk;
  if (q.w8 == 2) ps = (ps + 1) / 2;   // slabs
  q.psplit = (int)ps;
  q.chunk = (int)chunk;
  return q;
}

// bf16 transposing-LDS wgrad plan (3x3 stride-1 pad-1): pixel groups G, chunk pixels, channel blocks
struct WgradLdsPlan { int ok, CIT, COT, ciBlocks, coBlocks, G, chunk, nsub; size_t lds; int xrow, yrow, xbytes; };
static WgradLdsPlan wgrad_lds_plan(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil) {
  WgradLdsPlan q;
  q.ok = 0;
  if (!(kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1) || (Ci % 16) || (Co % 16) || W > 160) return q;
  q.CIT = Ci % 48 == 0 ? 3 : (Ci % 32 == 0 ? 2 : 1);
  q.COT = Co % 48 == 0 ? 3 : (Co % 32 == 0 ? 2 : 1);
  q.ciBlocks = Ci / (16 * q.CIT);
  q.coBlocks = Co / (16 * q.COT);
  const long P = (long)N * H * W;
  const long blocks = (long)q.ciBlocks * q.coBlocks;
  long G = (512 + blocks - 1) / blocks;
  long chunk = (P + G - 1) / G;
  chunk = ((chunk + 31) / 32) * 32;
  if (chunk < 32) chunk = 32;
  if (chunk > 320) chunk = 320;
  q.nsub = g_wgrad_nsub;
  G = (P + chunk * q.nsub - 1) / (chunk * q.nsub);
  q.G = (int)G;
  q.chunk = (int)chunk;
  // LDS rows are unpadded: the 4 pixel rows a 16-lane group hands to one transposing read are 32/64/96 bytes apart
  // (distinct bank groups); padding would only separate the kq groups and costs the second resident workgroup
  q.xrow = 32 * q.CIT;
  q.yrow = 32 * q.COT;
  q.xbytes = (q.chunk + 2 * (W + 1)) * q.xrow;
  q.lds = (size_t)q.xbytes + (size_t)q.chunk * q.yrow + 32;
  q.ok = q.lds <= 150 * 1024 && P < (1L << 31) && G < 65536;
  return q;
}

// f32 LDS wgrad plan (3x3 stride-1 pad-1): one 8-wave workgroup per CU (the tile takes most of the 160 KB), about two
// workgroups per CU in total so the partial-slab traffic stays at ~512 slabs
static int lds_row_floats(int cb) {  // row stride with (stride mod 64) in {16, 48}: conflict-free 4-row reads
  const int m = cb % 64;
  return (m == 16 || m == 48) ? cb : cb + 16;  // cb is a multiple of 16: m in {0, 32} -> +16
}
static WgradLdsPlan wgrad_lds_plan_f32(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil) {
  WgradLdsPlan q;
  q.ok = 0;
  if (!(kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1) || (Ci % 16) || (Co % 16) || W < 4) return q;
  q.CIT = Ci % 48 == 0 ? 3 : (Ci % 64 == 0 ? 4 : (Ci % 32 == 0 ? 2 : 1));
  q.COT = Co % 48 == 0 ? 3 : (Co % 64 == 0 ? 4 : (Co % 32 == 0 ? 2 : 1));
  q.ciBlocks = Ci / (16 * q.CIT);
  q.coBlocks = Co / (16 * q.COT);
  q.nsub = 1;
  q.xrow = lds_row_floats(16 * q.CIT);
  q.yrow = lds_row_floats(16 * q.COT);
  const long P = (long)N * H * W;
  const long blocks = (long)q.ciBlocks * q.coBlocks;
  const long budget = 156 * 1024 / 4 - 16 - 2L * (W + 1) * q.xrow;  // floats left for the chunk rows
  if (budget < 8L * (q.xrow + q.yrow)) return q;
  long cmax = budget / (q.xrow + q.yrow);
  cmax -= cmax % 8;
  long G = (512 + blocks - 1) / blocks;
  long chunk = (P + G - 1) / G;
  chunk = ((chunk + 7) / 8) * 8;
  if (chunk > cmax) chunk = cmax;
  G = (P + chunk - 1) / chunk;
  q.G = (int)G;
  q.chunk = (int)chunk;
  q.xbytes = (int)((chunk + 2 * (W + 1)) * q.xrow);  // floats
  q.lds = ((size_t)q.xbytes + (size_t)chunk * q.yrow + 16) * 4;
  q.ok = P < (1L << 31) && G < 65536;
  return q;
}

// 1 if a 16-bit 3x3 stride-1 pad-1 convolution of this shape can take its input un-normalised (fami_conv2d_fwd_xbn_* and
// fami_conv2d_wgrad_defer_xbn_* would both run)
int fami_conv2d_xbn_ok(int N, int H, int W, int Ci, int Co) {
  return (g_use_lds != 0 && g_wgrad_lds && fami_conv_t4_eligible16(N, H, W, Ci, Co) &&
          fami_wgrad16_slabs(N, H, W, Ci, Co, 3, 1, 1, 1) > 0) ? 1 : 0;
}
long fami_conv2d_wgrad_workspace(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil) {
  if (!geom_ok(kh, kw, stride, pad, dil)) return -1;
  const WgradPlan q = wgrad_plan(N, H, W, Ci, Co, kh, kw, stride, pad, dil);
  long need = (long)q.psplit * Co * Ci * kh * kw * (long)sizeof(float);
  const WgradLdsPlan l = wgrad_lds_plan(N, H, W, Ci, Co, kh, kw, stride, pad, dil);
  if (l.ok) {
    const long nl = (long)l.G * Co * Ci * 9 * (long)sizeof(float);
    if (nl > need) need = nl;
  }
  const WgradLdsPlan f = wgrad_lds_plan_f32(N, H, W, Ci, Co, kh, kw, stride, pad, dil);
  if (f.ok) {
    const long nl = (long)f.G * Co * Ci * 9 * (long)sizeof(float);
    if (nl > need) need = nl;
  }
  if (kh == kw) {
    const long nl = fami_wgrad16_slabs(N, H, W, Ci, Co, kh, stride, pad, dil) * Co * Ci * kh * kw * (long)sizeof(float);
    if (nl > need) need = nl;
  }
  if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1) {
    const long nl = fami_wgrad_s3_slabs(N, H, W, Ci, Co) * Co * Ci * 9 * (long)sizeof(float);
    if (nl > need) need = nl;
  }
  return need;
}

}  // extern "C"

// dw[Co,Ci,kh,kw] fp32 (=|+=) sum_pixels x (*) dy ; workspace from fami_conv2d_wgrad_workspace
template <typename T>
static int wgrad_impl(const T* x, const T* dy, float* dw, float* workspace, long ws_bytes, int N, int H, int W, int Ci,
                      int Co, int kh, int kw, int stride, int pad, int dil, int accumulate, hipStream_t s,
                      const char* nm) {
  FAMI_REQUIRE(x && dy && dw && workspace && N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0, nm, "bad argument");
  if (!geom_ok(kh, kw, stride, pad, dil)) {
    fami_set_error(nm, "unsupported geometry");
    return FAMI_ESHAPE;
  }
  // stride-1 same-size convolutions of whole 16-channel tiles: linear-address kernel (f32 storage only)
  const bool lin = std::is_same<T, float>::value && g_wgrad_lin && wgrad_lin_ok(H, W, Ci, Co, kh, kw, stride, pad, dil);
  const WgradPlan q = wgrad_plan(N, H, W, Ci, Co, kh, kw, stride, pad, dil, !lin ? 0 : (g_wgrad_lin == 1 ? 3 : (g_wgrad_lin == 3 ? 1 : (g_wgrad_lin == 4 ? 2 : 0))));
  const long need = (long)q.psplit * Co * Ci * kh * kw * (long)sizeof(float);
  FAMI_REQUIRE(ws_bytes >= need, nm, "workspace too small");
  FAMI_REQUIRE(q.P < (1L << 31), nm, "size out of range");
  WgradArgs<T> a;
  a.x = x; a.dy = dy; a.part = workspace;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci;
  a.Ho = out_dim(H, kh, stride, pad, dil); a.Wo = out_dim(W, kw, stride, pad, dil); a.Co = Co;
  a.kh = kh; a.kw = kw; a.sh = stride == 2 ? 1 : 0; a.pad = pad; a.dil = dil;
  a.P = (int)q.P; a.chunk = q.chunk; a.ciBlocks = q.ciBlocks; a.coBlocks = q.coBlocks; a.xcd = g_xcd_w; a.prio = g_prio;
  const long xb = (long)N * H * W * Ci * (long)sizeof(T), yb = q.P * Co * (long)sizeof(T);
  FAMI_REQUIRE(xb < (1L << 31) && yb < (1L << 31), nm, "tensor >= 2 GiB");
  a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)yb;
  if (q.pertap) {
    const dim3 grid(q.w8 ? q.psplit + (q.psplit + 7) / 8 : q.psplit, q.ciBlocks * q.coBlocks), block(q.w8 ? 512 * q.w8 : kh * kw * 64);
    const size_t lin_lds = q.w8 == 2 ? (size_t)8 * q.MT * q.NT * 1024 : 0;
    a.psplit = q.psplit; a.w8 = q.w8;
    bool ok = false;
#define FAMI_TCASE(mt, nt)                                                                  \
  if (q.MT == mt && q.NT == nt) {                                                           \
    if constexpr (std::is_same<T, float>::value) {                                          \
      if (lin) {                                                                            \
        static bool attr = false;                                                           \
        if (!attr) {                                                                        \
          (void)hipFuncSetAttribute((const void*)conv_wgrad_taps_lin_f32<T, mt, nt>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * mt * nt * 1024); \
          attr = true;                                                                      \
        }                                                                                   \
        hipLaunchKernelGGL((conv_wgrad_taps_lin_f32<T, mt, nt>), grid, block, lin_lds, s, a); \
      }                                                                                     \
      else hipLaunchKernelGGL((conv_wgrad_taps_f32<T, mt, nt>), grid, block, 0, s, a);      \
    } else {                                                                                \
      hipLaunchKernelGGL((conv_wgrad_taps_f32<T, mt, nt>), grid, block, 0, s, a);           \
    }                                                                                       \
    ok = true;                                                                              \
  }
    FAMI_TCASE(1, 1) FAMI_TCASE(1, 2) FAMI_TCASE(1, 3) FAMI_TCASE(1, 4)
    FAMI_TCASE(2, 1) FAMI_TCASE(2, 2) FAMI_TCASE(2, 3) FAMI_TCASE(2, 4)
    FAMI_TCASE(3, 1) FAMI_TCASE(3, 2) FAMI_TCASE(3, 3) FAMI_TCASE(3, 4)
    FAMI_TCASE(4, 1) FAMI_TCASE(4, 2) FAMI_TCASE(4, 3) FAMI_TCASE(4, 4)
#undef FAMI_TCASE
    if (!ok) {
      fami_set_error(nm, "no kernel instance");
      return FAMI_ESHAPE;
    }
    FAMI_CHECK_LAUNCH(nm);
    launch_reduce_taps(workspace, dw, Co, Ci, kh * kw, q.psplit, accumulate, s);
    FAMI_CHECK_LAUNCH(nm);
    return FAMI_OK;
  }
  const dim3 grid(q.psplit, kh * kw * q.ciBlocks * q.coBlocks);
  bool done = false;
#define FAMI_WCASE(mt, nt)                                                                  \
  if (q.MT == mt && q.NT == nt) {                                                           \
    hipLaunchKernelGGL((conv_wgrad_f32<T, mt, nt>), grid, dim3(256), 0, s, a);              \
    done = true;                                                                            \
  }
  FAMI_WCASE(1, 1) FAMI_WCASE(1, 2) FAMI_WCASE(1, 3) FAMI_WCASE(1, 4)
  FAMI_WCASE(2, 1) FAMI_WCASE(2, 2) FAMI_WCASE(2, 3) FAMI_WCASE(2, 4)
  FAMI_WCASE(3, 1) FAMI_WCASE(3, 2) FAMI_WCASE(3, 3) FAMI_WCASE(3, 4)
  FAMI_WCASE(4, 1) FAMI_WCASE(4, 2) FAMI_WCASE(4, 3) FAMI_WCASE(4, 4)
#undef FAMI_WCASE
  if (!done) {
    fami_set_error(nm, "no kernel instance");
    return FAMI_ESHAPE;
  }
  FAMI_CHECK_LAUNCH(nm);
  launch_reduce_plain(workspace, dw, Co, Ci, kh * kw, q.psplit, accumulate, s);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}

// ---- bf16 implicit GEMM launch
template <typename H, int MODE, int VEC>
static int launch_igemm_h(const ConvArgsH& a, int MT, int NT, int KS, int ST, hipStream_t s) {
  long tiles = fami_cdiv(a.P, MT * 16);
  if (a.par) {
    tiles = 0;
    for (int c = 0; c < 4; ++c) {
      const long Ha = (c >> 1) ? a.Ho / 2 : (a.Ho + 1) / 2, Wb = (c & 1) ? a.Wo / 2 : (a.Wo + 1) / 2;
      tiles += fami_cdiv(a.N * Ha * Wb, MT * 16);
    }
  }
  const dim3 grid(fami_cdiv(tiles, 4 / KS), fami_cdiv(a.NTt, NT));
#define FAMI_CASE(mt, nt, ks)                                                                        \
  if (MT == mt && NT == nt && KS == ks) {                                                            \
    if (ST == 3 && mt <= 2) hipLaunchKernelGGL((conv_igemm_h<H, mt, nt, MODE, VEC, ks, 3>), grid, dim3(256), 0, s, a); \
    else if (ST == 4 && mt <= 2) hipLaunchKernelGGL((conv_igemm_h<H, mt, nt, MODE, VEC, ks, 4>), grid, dim3(256), 0, s, a); \
    else hipLaunchKernelGGL((conv_igemm_h<H, mt, nt, MODE, VEC, ks, 2>), grid, dim3(256), 0, s, a);   \
    return 0;                                                                                        \
  }
  if constexpr (VEC) {
#define FAMI_ROW(nt) FAMI_CASE(1, nt, 1) FAMI_CASE(1, nt, 2) FAMI_CASE(1, nt, 4) FAMI_CASE(2, nt, 1) FAMI_CASE(2, nt, 2) FAMI_CASE(2, nt, 4) FAMI_CASE(4, nt, 1)
    FAMI_ROW(1) FAMI_ROW(2) FAMI_ROW(3) FAMI_ROW(4)
#undef FAMI_ROW
  } else {
    FAMI_CASE(2, 1, 1) FAMI_CASE(2, 2, 1) FAMI_CASE(2, 3, 1) FAMI_CASE(2, 4, 1)
  }
#undef FAMI_CASE
  return -1;
}

template <typename H>
static int run_igemm_h(ConvArgsH a, int mode, hipStream_t s, const char* name) {
  a.xcd = g_xcd < 0 ? 1 : g_xcd;
  a.par = (mode == 1 && a.sh == 1 && g_par) ? 1 : 0;
  const int vec = (a.Ci % 8 == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
  int NT = pick_nt(a.NTt);
  if (NT == 6) NT = 3;
  if (NT > 4) NT = 4;
  int MT = 2, KS = 1;
  if (vec) {
    // the bf16 kernel is operand-fetch bound (an MFMA needs 2 KiB of fragments): larger register tiles raise the
    // MFMAs per fetched KiB, split-K restores parallelism on the low-resolution branches
    const long nblk = fami_cdiv(a.NTt, NT);
    const long tiles = (long)fami_cdiv(a.P, 16) * nblk;
    const long iters = (long)a.kh * a.kw * a.KC;
    // (a two-chunk reduction -- the 64 -> 256 1x1 of stage 1 and the input gradient of its 256 -> 64 twin, 20 frames @96x72 -- is all
    //  prologue and epilogue: 64-pixel waves instead of 128: forward 38.1 -> 32.6 us, input gradient 38.6 -> 32.2, accumulating 58.4 -> 52.0,
    //  tools/probes/layer1_1x1_tiles.py)
    if (tiles >= 16384) MT = iters <= 2 ? 2 : 4;
    else if (tiles >= 4096) MT = 2;
    else {
      MT = 1;
      if (iters >= 8) KS = tiles < 1024 ? 4 : 2;
    }
  }
  if (vec && g_force_mt) { MT = g_force_mt; NT = g_force_nt ? g_force_nt : NT; KS = g_force_ks ? g_force_ks : 1; }
  const int ST = g_stages ? g_stages : 2;
  int rc;
  if (mode == 0)
    rc = vec ? launch_igemm_h<H, 0, 1>(a, MT, NT, KS, ST, s) : launch_igemm_h<H, 0, 0>(a, MT, NT, KS, 2, s);
  else
    rc = vec ? launch_igemm_h<H, 1, 1>(a, MT, NT, KS, ST, s) : launch_igemm_h<H, 1, 0>(a, MT, NT, KS, 2, s);
  if (rc != 0) {
    fami_set_error(name, "no kernel instance for tile shape");
    return FAMI_ESHAPE;
  }
  FAMI_CHECK_LAUNCH(name);
  return FAMI_OK;
}

extern "C" {

static int wgrad_f32_entry(const float* x, const float* dy, float* dw, float* workspace, long ws_bytes, int N, int H,
                           int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                           hipStream_t s, const XBN& xbn);
int fami_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, float* workspace, long ws_bytes, int N, int H,
                          int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                          hipStream_t s) {
  return wgrad_f32_entry(x, dy, dw, workspace, ws_bytes, N, H, W, Ci, Co, kh, kw, stride, pad, dil, accumulate, s, xbn_none());
}
static int wgrad_f32_entry(const float* x, const float* dy, float* dw, float* workspace, long ws_bytes, int N, int H,
                           int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                           hipStream_t s, const XBN& xbn) {
  if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1 && x && dy && dw && workspace) {
    // split-product kernel on the bf16 matrix pipe (conv_wgs3.hip)
    const int G = fami_try_wgrad_s3(x, dy, workspace, ws_bytes, N, H, W, Ci, Co, s, "fami_conv2d_wgrad_f32", xbn);
    if (G < 0) return G;
    if (G > 0) {
      launch_reduce_taps(workspace, dw, Co, Ci, 9, G, accumulate, s);
      FAMI_CHECK_LAUNCH("fami_conv2d_wgrad_f32/reduce");
      return FAMI_OK;
    }
  }
  if (xbn.on) {
    fami_set_error("fami_conv2d_wgrad_defer_xbn_f32", "input BatchNorm needs the split-product kernel (ask fami_conv2d_xbn_ok_f32 first)");
    return FAMI_ESHAPE;
  }
  WgradLdsPlan l = g_wgrad_lds_f32 ? wgrad_lds_plan_f32(N, H, W, Ci, Co, kh, kw, stride, pad, dil) : WgradLdsPlan{0};
  // measured (tools/bench_wgrad.py f32): the staged kernel wins once the channel blocks alone give >= 64 workgroup
  // columns (384 channels: 136 vs 151 us); below that the per-tap scalar-operand kernel is faster (85 vs 105 us)
  // ... and the linear-address per-tap kernel beats both wherever it applies (384 channels: 78 vs 103 us)
  if (g_wgrad_lds_f32 == 2 && l.ok &&
      (l.ciBlocks * l.coBlocks < 64 || (g_wgrad_lin && wgrad_lin_ok(H, W, Ci, Co, kh, kw, stride, pad, dil))))
    l.ok = 0;
  if (l.ok && x && dy && dw && workspace && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0) {
    const long need = (long)l.G * Co * Ci * 9 * (long)sizeof(float);
    FAMI_REQUIRE(ws_bytes >= need, "fami_conv2d_wgrad_f32", "workspace too small");
    WgradLdsArgsF a;
    a.x = x; a.dy = dy; a.part = workspace;
    a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.P = N * H * W;
    a.chunk = l.chunk; a.ciBlocks = l.ciBlocks; a.coBlocks = l.coBlocks; a.xcd = g_xcd_w;
    a.CiB = 16 * l.CIT; a.CoB = 16 * l.COT; a.xrow = l.xrow; a.yrow = l.yrow; a.xfloats = l.xbytes;
    const dim3 grid(l.G, l.ciBlocks * l.coBlocks);
    bool ok = false;
#define FAMI_WF_CASE(cit, cot)                                                                                        \
  if (l.CIT == cit && l.COT == cot) {                                                                                 \
    static bool attr = false;                                                                                         \
    if (!attr) {                                                                                                      \
      (void)hipFuncSetAttribute((const void*)conv_wgrad_lds_f32_kernel<cit, cot>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr = true;                                                                                                    \
    }                                                                                                                 \
    hipLaunchKernelGGL((conv_wgrad_lds_f32_kernel<cit, cot>), grid, dim3(512), l.lds, s, a);                          \
    ok = true;                                                                                                        \
  }
    FAMI_WF_CASE(1, 1) FAMI_WF_CASE(1, 2) FAMI_WF_CASE(1, 3) FAMI_WF_CASE(1, 4) FAMI_WF_CASE(2, 1) FAMI_WF_CASE(2, 2)
    FAMI_WF_CASE(2, 3) FAMI_WF_CASE(2, 4) FAMI_WF_CASE(3, 1) FAMI_WF_CASE(3, 2) FAMI_WF_CASE(3, 3) FAMI_WF_CASE(3, 4)
    FAMI_WF_CASE(4, 1) FAMI_WF_CASE(4, 2) FAMI_WF_CASE(4, 3) FAMI_WF_CASE(4, 4)
#undef FAMI_WF_CASE
    if (ok) {
      FAMI_CHECK_LAUNCH("fami_conv2d_wgrad_f32/lds");
      launch_reduce_taps(workspace, dw, Co, Ci, 9, l.G, accumulate, s);
      FAMI_CHECK_LAUNCH("fami_conv2d_wgrad_f32/reduce");
      return FAMI_OK;
    }
  }
  return wgrad_impl<float>(x, dy, dw, workspace, ws_bytes, N, H, W, Ci, Co, kh, kw, stride, pad, dil, accumulate, s,
                           "fami_conv2d_wgrad_f32");
}
}  // extern "C"

extern "C" {
// Deferred slab reduce: the weight-gradient kernel is launched now, its reduce is described in desc_out (a HOST buffer of
// FAMI_REDUCE_DESC_LONGS longs, opaque) and launched later -- up to 16 at a time -- by fami_wgrad_reduce_batch.  The
// workspace must stay untouched until then.  Two deferred reduces into the same dw must not share a batch when the
// second accumulates (the caller flushes in between).
int fami_conv2d_wgrad_defer_f32(const float* x, const float* dy, float* dw, float* workspace, long ws_bytes, int N, int H,
                                int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                                long* desc_out, hipStream_t s) {
  FAMI_REQUIRE(desc_out, "fami_conv2d_wgrad_defer_f32", "null descriptor");
  ReduceDesc d;
  d.part = nullptr;
  g_defer = &d;
  const int rc = fami_conv2d_wgrad_f32(x, dy, dw, workspace, ws_bytes, N, H, W, Ci, Co, kh, kw, stride, pad, dil, accumulate, s);
  g_defer = nullptr;
  if (rc != FAMI_OK) return rc;
  FAMI_REQUIRE(d.part, "fami_conv2d_wgrad_defer_f32", "no reduce recorded");
  memcpy(desc_out, &d, sizeof(d));
  return FAMI_OK;
}
int fami_conv2d_wgrad_defer_xbn_f32(const float* z, const float* dy, float* dw, float* workspace, long ws_bytes, int N, int H,
                                    int W, int Ci, int Co, int accumulate, long* desc_out, const float* xmean,
                                    const float* xinvstd, const float* xgamma, const float* xbeta, hipStream_t s) {
  FAMI_REQUIRE(desc_out && xmean && xinvstd && xgamma && xbeta, "fami_conv2d_wgrad_defer_xbn_f32", "bad argument");
  XBN xb = xbn_none();
  xb.on = 1; xb.C = Ci; xb.gamma = xgamma; xb.beta = xbeta; xb.mean = const_cast<float*>(xmean);
  xb.invstd = const_cast<float*>(xinvstd);
  ReduceDesc d;
  d.part = nullptr;
  g_defer = &d;
  const int rc = wgrad_f32_entry(z, dy, dw, workspace, ws_bytes, N, H, W, Ci, Co, 3, 3, 1, 1, 1, accumulate, s, xb);
  g_defer = nullptr;
  if (rc != FAMI_OK) return rc;
  FAMI_REQUIRE(d.part, "fami_conv2d_wgrad_defer_xbn_f32", "no reduce recorded");
  memcpy(desc_out, &d, sizeof(d));
  return FAMI_OK;
}
// conv_pair.h, f32 storage: the split-product input gradient (conv_t5.hip's persistent kernel) and the deferred split-product weight
// gradient (conv_wgs3.hip) of a 3x3 stride-1 convolution as ONE launch where a combined instance exists (ask
// fami_conv2d_bwd_pair_ok_f32), as the two single launches otherwise.  xmean != NULL: x is the input z of a BatchNorm + ReLU nobody
// materialised (the weight gradient applies it while staging, as fami_conv2d_wgrad_defer_xbn_f32).  Bitwise the two-call form.
int fami_conv2d_bwd_pair_f32(const float* x, const float* dy, const float* wpd, float* dx, float* dw, float* workspace,
                             long ws_bytes, int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil,
                             int acc_dx, int acc_dw, long* desc_out, const float* xmean, const float* xinvstd,
                             const float* xgamma, const float* xbeta, hipStream_t s) {
  static const char* nm = "fami_conv2d_bwd_pair_f32";
  FAMI_REQUIRE(desc_out, nm, "null descriptor");
  XBN xb = xbn_none();
  if (xmean) {
    FAMI_REQUIRE(xinvstd && xgamma && xbeta, nm, "bad argument");
    xb.on = 1; xb.C = Ci; xb.gamma = xgamma; xb.beta = xbeta; xb.mean = const_cast<float*>(xmean);
    xb.invstd = const_cast<float*>(xinvstd);
  }
  PairCapture pc;
  pc.a.kind = pc.b.kind = 0;
  ReduceDesc d;
  d.part = nullptr;
  fami_pair_capture() = &pc;
  int rc = conv_dgrad_f32_impl(nm, dy, wpd, nullptr, dx, N, H, W, Ci, Co, kh, kw, stride, pad, dil, acc_dx, epi_none(), s);
  if (rc == FAMI_OK) {
    g_defer = &d;
    rc = wgrad_f32_entry(x, dy, dw, workspace, ws_bytes, N, H, W, Ci, Co, kh, kw, stride, pad, dil, acc_dw, s, xb);
    g_defer = nullptr;
  }
  fami_pair_capture() = nullptr;
  if (rc != FAMI_OK) return rc;
  FAMI_REQUIRE(d.part, nm, "no reduce recorded");
  memcpy(desc_out, &d, sizeof(d));
  const int pr = fami_pair_launch(pc, s);
  FAMI_REQUIRE(pr >= 0, nm, "a recorded launch has no kernel instance");
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}
// f32 storage: both split-product kernels (conv_t4.hip S3, conv_wgs3.hip) would take the shape
int fami_conv2d_xbn_ok_f32(int N, int H, int W, int Ci, int Co) {
  return (g_use_lds != 0 && fami_conv_t4_eligible_s3(N, H, W, Ci, Co) && fami_wgrad_s3_slabs(N, H, W, Ci, Co) > 0) ? 1 : 0;
}
int fami_wgrad_reduce_desc_longs(void) { return (int)((sizeof(ReduceDesc) + sizeof(long) - 1) / sizeof(long)); }
int fami_wgrad_reduce_batch(const long* descs, int n, hipStream_t s) {
  FAMI_REQUIRE(descs && n > 0, "fami_wgrad_reduce_batch", "bad argument");
  const int stride = fami_wgrad_reduce_desc_longs();
  for (int i0 = 0; i0 < n; i0 += FAMI_REDUCE_BATCH) {
    const int m = n - i0 < FAMI_REDUCE_BATCH ? n - i0 : FAMI_REDUCE_BATCH;
    ReduceBatch b;
    int maxb = 1;
    for (int i = 0; i < m; ++i) {
      memcpy(&b.d[i], descs + (long)(i0 + i) * stride, sizeof(ReduceDesc));
      if (b.d[i].blocks > maxb) maxb = b.d[i].blocks;
    }
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(maxb, m), dim3(1024), 0, s, b);
    FAMI_CHECK_LAUNCH("fami_wgrad_reduce_batch");
  }
  return FAMI_OK;
}
}  // extern "C"

// 16-bit activations / gradients, fp32 weight gradient
template <typename HT>
static int wgrad_h_impl(const char* nm, const HT* x, const HT* dy, float* dw, float* workspace, long ws_bytes, int N,
                        int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                        hipStream_t s, const XBN& xbn = xbn_none()) {
  if (g_wgrad_lds && x && dy && dw && workspace && kh == kw) {
    // round-3 kernel (conv_wg16.hip): pipelined staging, padded patch, 8 waves; 1x1 / 3x3, stride 1 / 2, any dilation
    const int G = fami_try_wgrad16(std::is_same<HT, f16_t>::value ? 1 : 0, x, dy, workspace, ws_bytes, N, H, W, Ci, Co, kh, stride,
                                   pad, dil, s, nm, xbn);
    if (G < 0) return G;
    if (G > 0) {
      launch_reduce_taps(workspace, dw, Co, Ci, kh * kw, G, accumulate, s);
      FAMI_CHECK_LAUNCH(nm);
      return FAMI_OK;
    }
  }
  if (xbn.on) {
    fami_set_error(nm, "input BatchNorm needs the conv_wg16 kernel (ask fami_conv2d_xbn_ok first)");
    return FAMI_ESHAPE;
  }
  const WgradLdsPlan l = g_wgrad_lds ? wgrad_lds_plan(N, H, W, Ci, Co, kh, kw, stride, pad, dil) : WgradLdsPlan{0};
  if (l.ok && x && dy && dw && workspace && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0) {
    const long need = (long)l.G * Co * Ci * 9 * (long)sizeof(float);
    FAMI_REQUIRE(ws_bytes >= need, nm, "workspace too small");
    WgradLdsArgs a;
    a.x = x; a.dy = dy; a.part = workspace;
    a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.P = N * H * W;
    a.chunk = l.chunk; a.nsub = l.nsub; a.ciBlocks = l.ciBlocks; a.coBlocks = l.coBlocks; a.xcd = g_xcd_w;
    a.CiB = 16 * l.CIT; a.CoB = 16 * l.COT; a.xrow = l.xrow; a.yrow = l.yrow; a.xbytes = l.xbytes;
    const dim3 grid(l.G, l.ciBlocks * l.coBlocks);
    bool ok = false;
#define FAMI_WL_CASE(cit, cot)                                                                                     \
  if (l.CIT == cit && l.COT == cot) {                                                                              \
    static bool attr = false;                                                                                      \
    if (!attr) {                                                                                                   \
      (void)hipFuncSetAttribute((const void*)conv_wgrad_h_kernel<HT, cit, cot>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
      attr = true;                                                                                                 \
    }                                                                                                              \
    hipLaunchKernelGGL((conv_wgrad_h_kernel<HT, cit, cot>), grid, dim3(256), l.lds, s, a);                         \
    ok = true;                                                                                                     \
  }
    FAMI_WL_CASE(1, 1) FAMI_WL_CASE(1, 2) FAMI_WL_CASE(1, 3) FAMI_WL_CASE(2, 1) FAMI_WL_CASE(2, 2) FAMI_WL_CASE(2, 3)
    FAMI_WL_CASE(3, 1) FAMI_WL_CASE(3, 2) FAMI_WL_CASE(3, 3)
#undef FAMI_WL_CASE
    if (ok) {
      FAMI_CHECK_LAUNCH(nm);
      launch_reduce_taps(workspace, dw, Co, Ci, 9, l.G, accumulate, s);
      FAMI_CHECK_LAUNCH(nm);
      return FAMI_OK;
    }
  }
  return wgrad_impl<HT>(x, dy, dw, workspace, ws_bytes, N, H, W, Ci, Co, kh, kw, stride, pad, dil, accumulate, s, nm);
}

template <typename HT>
static int pack_conv_weight_h_impl(const char* nm, const float* w_oihw, HT* wp, int Co, int Ci, int kh, int kw, int mode,
                                   hipStream_t s) {
  FAMI_REQUIRE(w_oihw && wp && Co > 0 && Ci > 0 && (mode == 0 || mode == 1), nm, "bad argument");
  const int kd = mode == 0 ? Ci : Co, nd = mode == 0 ? Co : Ci;
  const int KC = fami_cdiv(kd, 32), NTt = fami_cdiv(nd, 16);
  const long total = (long)kh * kw * KC * NTt * 512;
  hipLaunchKernelGGL(pack_w_h_kernel<HT>, dim3(fami_ew_grid(total)), dim3(256), 0, s, w_oihw, wp, Co, Ci, kh * kw, KC, NTt, mode);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}

// conv_stem.hip: the stem's 3 -> 64 stride-2 convolution with K dense over (tap, channel)
int fami_try_conv_stem1(int half_kind, const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci, int Co,
                        int kh, int kw, int stride, int pad, int dil, int NTt, int relu, int accumulate, int out_f32, hipStream_t s,
                        const char* name, const EpiBN& epi);
void fami_conv_stem_tune(int on);

// y[N,Ho,Wo,Co] (16-bit, or f32 when out_f32) (=|+=) relu?( conv(x[N,H,W,Ci]) + bias ) ; wp packed with mode 0
template <typename HT>
static int conv_fwd_h_impl(const char* nm, const HT* x, const HT* wp, const float* bias, void* y, int N, int H, int W,
                           int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int relu, int accumulate,
                           int out_f32, hipStream_t s, const EpiBN& e = epi_none(), const XBN& xbn = xbn_none()) {
  FAMI_REQUIRE(x && wp && y && N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0, nm, "bad argument");
  if (!geom_ok(kh, kw, stride, pad, dil)) {
    fami_set_error(nm, "unsupported geometry");
    return FAMI_ESHAPE;
  }
  if (e.slots && (out_f32 || relu || (Co & 3))) {
    fami_set_error(nm, "fused BatchNorm statistics need a 16-bit output, no ReLU and Co % 4 == 0");
    return FAMI_ESHAPE;
  }
  ConvArgsH a;
  a.e = e; a.emode = e.slots ? e.mode : 0;
  a.x = x; a.wp = wp; a.y = y; a.bias = bias;
  a.N = N; a.Hi = H; a.Wi = W; a.Ci = Ci;
  a.Ho = out_dim(H, kh, stride, pad, dil); a.Wo = out_dim(W, kw, stride, pad, dil); a.Co = Co;
  a.kh = kh; a.kw = kw; a.sh = stride == 2 ? 1 : 0; a.pad = pad; a.dil = dil;
  a.KC = fami_cdiv(Ci, 32); a.NTt = fami_cdiv(Co, 16); a.relu = relu; a.accumulate = accumulate; a.out_f32 = out_f32;
  const long P = (long)N * a.Ho * a.Wo;
  const long xb = (long)N * H * W * Ci * 2, wb = (long)kh * kw * a.KC * a.NTt * 1024;
  FAMI_REQUIRE(P > 0 && P < (1L << 31) && xb < (1L << 31) && wb < (1L << 31), nm, "tensor >= 2 GiB");
  a.P = (int)P; a.x_bytes = (unsigned)xb; a.wp_bytes = (unsigned)wb;
  if (Ci == 3 && !xbn.on) {
    const int rc = fami_try_conv_stem1(std::is_same<HT, f16_t>::value ? 1 : 0, x, wp, bias, y, N, H, W, Ci, Co, kh, kw, stride, pad, dil,
                                       a.NTt, relu, accumulate, out_f32, s, nm, e);
    if (rc != 0) return rc < 0 ? rc : FAMI_OK;
  }
  if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1) {
    const int rc = try_conv3x3_lds<HT>(x, wp, bias, y, N, H, W, Ci, Co, a.KC, a.NTt, +1, relu, accumulate, out_f32, s, nm, e, xbn);
    if (rc != 0) return rc < 0 ? rc : FAMI_OK;
  }
  if (kh == 3 && kw == 3 && stride == 1 && pad == dil && dil > 1 && !e.slots && !xbn.on && g_use_lds != 0) {
    const int rc = fami_try_conv3x3_t4_dil(std::is_same<HT, f16_t>::value ? 1 : 0, x, wp, bias, y, N, H, W, Ci, Co, a.KC, a.NTt, +1, relu,
                                           accumulate, out_f32, dil, s, nm);
    if (rc != 0) return rc < 0 ? rc : FAMI_OK;
  }
  if (xbn.on) {
    fami_set_error(nm, "input BatchNorm needs the register-blocked 3x3 kernel (ask fami_conv2d_xbn_ok first)");
    return FAMI_ESHAPE;
  }
  return run_igemm_h<HT>(a, 0, s, nm);
}

// dx[N,H,W,Ci] (=|+=) conv_transpose(dy[N,Ho,Wo,Co]) ; wp packed with mode 1
template <typename HT>
static int conv_dgrad_h_impl(const char* nm, const HT* dy, const HT* wp, HT* dx, int N, int H, int W, int Ci, int Co,
                             int kh, int kw, int stride, int pad, int dil, int accumulate, hipStream_t s,
                             const EpiBN& e = epi_none()) {
  FAMI_REQUIRE(dy && wp && dx && N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0, nm, "bad argument");
  if (!geom_ok(kh, kw, stride, pad, dil)) {
    fami_set_error(nm, "unsupported geometry");
    return FAMI_ESHAPE;
  }
  if (e.slots && (Ci & 3)) {
    fami_set_error(nm, "fused BatchNorm statistics need Ci % 4 == 0");
    return FAMI_ESHAPE;
  }
  ConvArgsH a;
  a.e = e; a.emode = e.slots ? e.mode : 0;
  a.x = dy; a.wp = wp; a.y = dx; a.bias = nullptr;
  a.N = N; a.Hi = out_dim(H, kh, stride, pad, dil); a.Wi = out_dim(W, kw, stride, pad, dil); a.Ci = Co;
  a.Ho = H; a.Wo = W; a.Co = Ci;
  a.kh = kh; a.kw = kw; a.sh = stride == 2 ? 1 : 0; a.pad = pad; a.dil = dil;
  a.KC = fami_cdiv(Co, 32); a.NTt = fami_cdiv(Ci, 16); a.relu = 0; a.accumulate = accumulate; a.out_f32 = 0;
  const long P = (long)N * H * W;
  const long xb = (long)N * a.Hi * a.Wi * Co * 2, wb = (long)kh * kw * a.KC * a.NTt * 1024;
  FAMI_REQUIRE(P > 0 && P < (1L << 31) && xb < (1L << 31) && wb < (1L << 31), nm, "tensor >= 2 GiB");
  a.P = (int)P; a.x_bytes = (unsigned)xb; a.wp_bytes = (unsigned)wb;
  if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1) {
    const int rc = try_conv3x3_lds<HT>(dy, wp, nullptr, dx, N, H, W, Co, Ci, a.KC, a.NTt, -1, 0, accumulate, 0, s, nm, e);
    if (rc != 0) return rc < 0 ? rc : FAMI_OK;
  }
  if (kh == 3 && kw == 3 && stride == 1 && pad == dil && dil > 1 && !e.slots && g_use_lds != 0) {
    const int rc = fami_try_conv3x3_t4_dil(std::is_same<HT, f16_t>::value ? 1 : 0, dy, wp, nullptr, dx, N, H, W, Co, Ci, a.KC, a.NTt, -1, 0,
                                           accumulate, 0, dil, s, nm);
    if (rc != 0) return rc < 0 ? rc : FAMI_OK;
  }
  return run_igemm_h<HT>(a, 1, s, nm);
}

extern "C" {

// every weight image of a step in one launch.  desc: device array of n records {long src_elem_off, long dst_elem_off,
// int Co, int Ci, int taps, int mode} (32 bytes each); params = fp32 parameter arena, packed = destination arena.
int fami_pack_conv_weights_batch_f32(const float* params, float* packed, const void* desc, int n, hipStream_t s) {
  FAMI_REQUIRE(params && packed && desc && n > 0 && n < 65536, "fami_pack_conv_weights_batch_f32", "bad argument");
  hipLaunchKernelGGL(pack_w_batch_kernel<float>, dim3(48, n), dim3(256), 0, s, params, packed, reinterpret_cast<const PackDesc*>(desc));
  FAMI_CHECK_LAUNCH("fami_pack_conv_weights_batch_f32");
  fami_pack_split_batch(params, packed, desc, n, s);      // the split images behind the 3x3 fragment images (conv_t5.hip)
  FAMI_CHECK_LAUNCH("fami_pack_conv_weights_batch_f32/split");
  return FAMI_OK;
}

// the packed 16-bit image has the same geometry for bf16 and fp16
long fami_packed_weight_elems_bf16(int Co, int Ci, int kh, int kw, int mode) {
  const int kd = mode == 0 ? Ci : Co, nd = mode == 0 ? Co : Ci;
  return (long)kh * kw * fami_cdiv(kd, 32) * fami_cdiv(nd, 16) * 512;
}
long fami_packed_weight_elems_f16(int Co, int Ci, int kh, int kw, int mode) {
  return fami_packed_weight_elems_bf16(Co, Ci, kh, kw, mode);
}

#define FAMI_CONV_H_ABI(sfx, HT)                                                                                       \
  int fami_conv2d_wgrad_##sfx(const HT* x, const HT* dy, float* dw, float* workspace, long ws_bytes, int N, int H,     \
                              int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,     \
                              hipStream_t s) {                                                                         \
    return wgrad_h_impl<HT>("fami_conv2d_wgrad_" #sfx, x, dy, dw, workspace, ws_bytes, N, H, W, Ci, Co, kh, kw,        \
                            stride, pad, dil, accumulate, s);                                                          \
  }                                                                                                                    \
  int fami_conv2d_wgrad_defer_##sfx(const HT* x, const HT* dy, float* dw, float* workspace, long ws_bytes, int N,      \
                                    int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil,        \
                                    int accumulate, long* desc_out, hipStream_t s) {                                   \
    FAMI_REQUIRE(desc_out, "fami_conv2d_wgrad_defer_" #sfx, "null descriptor");                                        \
    ReduceDesc d;                                                                                                      \
    d.part = nullptr;                                                                                                  \
    g_defer = &d;                                                                                                      \
    const int rc = wgrad_h_impl<HT>("fami_conv2d_wgrad_defer_" #sfx, x, dy, dw, workspace, ws_bytes, N, H, W, Ci, Co,  \
                                    kh, kw, stride, pad, dil, accumulate, s);                                          \
    g_defer = nullptr;                                                                                                 \
    if (rc != FAMI_OK) return rc;                                                                                      \
    FAMI_REQUIRE(d.part, "fami_conv2d_wgrad_defer_" #sfx, "no reduce recorded");                                       \
    memcpy(desc_out, &d, sizeof(d));                                                                                   \
    return FAMI_OK;                                                                                                    \
  }                                                                                                                    \
  int fami_pack_conv_weights_batch_##sfx(const float* params, HT* packed, const void* desc, int n, hipStream_t s) {    \
    FAMI_REQUIRE(params && packed && desc && n > 0 && n < 65536, "fami_pack_conv_weights_batch_" #sfx, "bad argument");\
    hipLaunchKernelGGL(pack_w_batch_kernel<HT>, dim3(48, n), dim3(256), 0, s, params, packed,                          \
                       reinterpret_cast<const PackDesc*>(desc));                                                       \
    FAMI_CHECK_LAUNCH("fami_pack_conv_weights_batch_" #sfx);                                                           \
    return FAMI_OK;                                                                                                    \
  }                                                                                                                    \
  int fami_pack_conv_weight_##sfx(const float* w_oihw, HT* wp, int Co, int Ci, int kh, int kw, int mode,               \
                                  hipStream_t s) {                                                                     \
    return pack_conv_weight_h_impl<HT>("fami_pack_conv_weight_" #sfx, w_oihw, wp, Co, Ci, kh, kw, mode, s);            \
  }                                                                                                                    \
  int fami_conv2d_fwd_##sfx(const HT* x, const HT* wp, const float* bias, void* y, int N, int H, int W, int Ci,        \
                            int Co, int kh, int kw, int stride, int pad, int dil, int relu, int accumulate,            \
                            int out_f32, hipStream_t s) {                                                              \
    return conv_fwd_h_impl<HT>("fami_conv2d_fwd_" #sfx, x, wp, bias, y, N, H, W, Ci, Co, kh, kw, stride, pad, dil,     \
                               relu, accumulate, out_f32, s);                                                          \
  }                                                                                                                    \
  int fami_conv2d_dgrad_##sfx(const HT* dy, const HT* wp, HT* dx, int N, int H, int W, int Ci, int Co, int kh,         \
                              int kw, int stride, int pad, int dil, int accumulate, hipStream_t s) {                   \
    return conv_dgrad_h_impl<HT>("fami_conv2d_dgrad_" #sfx, dy, wp, dx, N, H, W, Ci, Co, kh, kw, stride, pad, dil,     \
                                 accumulate, s);                                                                       \
  }                                                                                                                    \
  /* BatchNorm (+ReLU) of the INPUT applied while it is staged (XBN, conv_epi.h): x is the pre-normalisation tensor z.  */ \
  /* fwd: statistics of z come from xslots (filled by fami_conv2d_fwd_stats_*), mean / invstd / running statistics are */ \
  /* published; slots / pivot_src (optional) request the statistics of y for the BatchNorm that follows.               */ \
  int fami_conv2d_fwd_xbn_##sfx(const HT* z, const HT* wp, const float* bias, HT* y, int N, int H, int W, int Ci,      \
                                int Co, void* slots, const float* pivot_src, const void* xslots, long xP,             \
                                const float* xgamma, const float* xbeta, float* xmean, float* xinvstd,                 \
                                float* xrunning_mean, float* xrunning_var, float xmomentum, float xeps,                \
                                hipStream_t s) {                                                                       \
    FAMI_REQUIRE(xslots && xgamma && xbeta && xmean && xinvstd && xP > 0, "fami_conv2d_fwd_xbn_" #sfx, "bad argument");\
    EpiBN e = epi_none();                                                                                              \
    if (slots) {                                                                                                       \
      e.slots = reinterpret_cast<double*>(slots); e.ns = bn_slots(Co); e.mode = 1; e.C = Co; e.pivot_src = pivot_src;  \
    }                                                                                                                  \
    XBN xb = xbn_none();                                                                                               \
    xb.on = 1; xb.slots = reinterpret_cast<const double*>(xslots); xb.ns = bn_slots(Ci); xb.C = Ci; xb.P = xP;         \
    xb.gamma = xgamma; xb.beta = xbeta; xb.mean = xmean; xb.invstd = xinvstd; xb.running_mean = xrunning_mean;         \
    xb.running_var = xrunning_var; xb.momentum = xmomentum; xb.eps = xeps;                                             \
    return conv_fwd_h_impl<HT>("fami_conv2d_fwd_xbn_" #sfx, z, wp, bias, y, N, H, W, Ci, Co, 3, 3, 1, 1, 1, 0, 0, 0,   \
                               s, e, xb);                                                                              \
  }                                                                                                                    \
  /* Round 6: the same BatchNorm + ReLU of the input INSIDE the convolution's launch, with the normalised tensor written to   */ \
  /* a_out as well (the weight-resident 48-channel kernel transforms its patch in LDS and stores the rows it owns): one launch */ \
  /* and one read of z per BasicBlock less than fami_bn_apply_slots_* + fami_conv2d_fwd_stats_*, bit for bit their results.     */ \
  /* Ask fami_conv2d_fwd_bnin_ok first.                                                                                          */ \
  int fami_conv2d_fwd_bnin_##sfx(const HT* z, const HT* wp, const float* bias, HT* y, HT* a_out, int N, int H, int W,  \
                                 int Ci, int Co, void* slots, const float* pivot_src, const void* xslots, long xP,     \
                                 const float* xgamma, const float* xbeta, float* xmean, float* xinvstd,                \
                                 float* xrunning_mean, float* xrunning_var, float xmomentum, float xeps,               \
                                 hipStream_t s) {                                                                      \
    static const char* nm = "fami_conv2d_fwd_bnin_" #sfx;                                                              \
    FAMI_REQUIRE(a_out && xslots && xgamma && xbeta && xmean && xinvstd && xP > 0, nm, "bad argument");                \
    if (!fami_conv2d_fwd_bnin_ok(N, H, W, Ci, Co)) {                                                                   \
      fami_set_error(nm, "no kernel applies the input BatchNorm in its launch for this shape (ask fami_conv2d_fwd_bnin_ok)"); \
      return FAMI_ESHAPE;                                                                                              \
    }                                                                                                                  \
    EpiBN e = epi_none();                                                                                              \
    if (slots) {                                                                                                       \
      e.slots = reinterpret_cast<double*>(slots); e.ns = bn_slots(Co); e.mode = 1; e.C = Co; e.pivot_src = pivot_src;  \
    }                                                                                                                  \
    XBN xb = xbn_none();                                                                                               \
    xb.on = 1; xb.slots = reinterpret_cast<const double*>(xslots); xb.ns = bn_slots(Ci); xb.C = Ci; xb.P = xP;         \
    xb.gamma = xgamma; xb.beta = xbeta; xb.mean = xmean; xb.invstd = xinvstd; xb.running_mean = xrunning_mean;         \
    xb.running_var = xrunning_var; xb.momentum = xmomentum; xb.eps = xeps; xb.out = a_out;                             \
    return conv_fwd_h_impl<HT>(nm, z, wp, bias, y, N, H, W, Ci, Co, 3, 3, 1, 1, 1, 0, 0, 0, s, e, xb);                 \
  }                                                                                                                    \
  int fami_conv2d_wgrad_defer_xbn_##sfx(const HT* z, const HT* dy, float* dw, float* workspace, long ws_bytes, int N,  \
                                        int H, int W, int Ci, int Co, int accumulate, long* desc_out,                  \
                                        const float* xmean, const float* xinvstd, const float* xgamma,                 \
                                        const float* xbeta, hipStream_t s) {                                           \
    FAMI_REQUIRE(desc_out && xmean && xinvstd && xgamma && xbeta, "fami_conv2d_wgrad_defer_xbn_" #sfx, "bad argument");\
    XBN xb = xbn_none();                                                                                               \
    xb.on = 1; xb.C = Ci; xb.gamma = xgamma; xb.beta = xbeta; xb.mean = const_cast<float*>(xmean);                     \
    xb.invstd = const_cast<float*>(xinvstd);                                                                           \
    ReduceDesc d;                                                                                                      \
    d.part = nullptr;                                                                                                  \
    g_defer = &d;                                                                                                      \
    const int rc = wgrad_h_impl<HT>("fami_conv2d_wgrad_defer_xbn_" #sfx, z, dy, dw, workspace, ws_bytes, N, H, W, Ci,  \
                                    Co, 3, 3, 1, 1, 1, accumulate, s, xb);                                             \
    g_defer = nullptr;                                                                                                 \
    if (rc != FAMI_OK) return rc;                                                                                      \
    FAMI_REQUIRE(d.part, "fami_conv2d_wgrad_defer_xbn_" #sfx, "no reduce recorded");                                   \
    memcpy(desc_out, &d, sizeof(d));                                                                                   \
    return FAMI_OK;                                                                                                    \
  }                                                                                                                    \
  /* fused BatchNorm statistics, see the _f32 forms */                                                                \
  int fami_conv2d_fwd_stats_##sfx(const HT* x, const HT* wp, const float* bias, HT* y, int N, int H, int W, int Ci,    \
                                  int Co, int kh, int kw, int stride, int pad, int dil, void* slots,                   \
                                  const float* pivot_src, hipStream_t s) {                                             \
    FAMI_REQUIRE(slots, "fami_conv2d_fwd_stats_" #sfx, "null slots");                                                  \
    EpiBN e = epi_none();                                                                                              \
    e.slots = reinterpret_cast<double*>(slots); e.ns = bn_slots(Co); e.mode = 1; e.C = Co; e.pivot_src = pivot_src;    \
    return conv_fwd_h_impl<HT>("fami_conv2d_fwd_stats_" #sfx, x, wp, bias, y, N, H, W, Ci, Co, kh, kw, stride, pad,    \
                               dil, 0, 0, 0, s, e);                                                                    \
  }                                                                                                                    \
  int fami_conv2d_dgrad_bnstats_##sfx(const HT* dy, const HT* wp, HT* dx, int N, int H, int W, int Ci, int Co,         \
                                      int kh, int kw, int stride, int pad, int dil, int accumulate, const HT* z,       \
                                      const HT* yrelu, const float* mean, const float* invstd, const float* gamma,     \
                                      const float* beta, int relu, void* slots, hipStream_t s) {                       \
    FAMI_REQUIRE(slots && z && mean && invstd && gamma && beta && (relu != 1 || yrelu) && relu >= 0 && relu <= 2,      \
                 "fami_conv2d_dgrad_bnstats_" #sfx, "bad argument");                                                   \
    EpiBN e = epi_none();                                                                                              \
    e.slots = reinterpret_cast<double*>(slots); e.ns = bn_slots(Ci); e.mode = 2; e.relu = relu; e.C = Ci;              \
    e.z = z; e.yr = yrelu; e.mean = mean; e.invstd = invstd; e.gamma = gamma; e.beta = beta;                           \
    return conv_dgrad_h_impl<HT>("fami_conv2d_dgrad_bnstats_" #sfx, dy, wp, dx, N, H, W, Ci, Co, kh, kw, stride, pad,  \
                                 dil, accumulate, s, e);                                                               \
  }
/* conv_pair.h: input gradient (optionally with the backward-statistics epilogue of fami_conv2d_dgrad_bnstats_*: slots != NULL)   */
/* and deferred weight gradient of one 3x3 stride-1 convolution as ONE launch where a combined instance exists (ask            */
/* fami_conv2d_bwd_pair_ok), as the two single launches otherwise.  Results are bitwise those of the two-call form.            */
#define FAMI_CONV_PAIR_ABI(sfx, HT)                                                                                    \
  int fami_conv2d_bwd_pair_##sfx(const HT* x, const HT* dy, const HT* wpd, HT* dx, float* dw, float* workspace,        \
                                 long ws_bytes, int N, int H, int W, int Ci, int Co, int kh, int kw, int stride,       \
                                 int pad, int dil, int acc_dx, int acc_dw, long* desc_out, const HT* z,                \
                                 const HT* yrelu, const float* mean, const float* invstd, const float* gamma,          \
                                 const float* beta, int relu, void* slots, hipStream_t s) {                            \
    static const char* nm = "fami_conv2d_bwd_pair_" #sfx;                                                              \
    FAMI_REQUIRE(desc_out, nm, "null descriptor");                                                                     \
    EpiBN e = epi_none();                                                                                              \
    if (slots) {                                                                                                       \
      FAMI_REQUIRE(z && mean && invstd && gamma && beta && (relu != 1 || yrelu) && relu >= 0 && relu <= 2, nm,         \
                   "bad argument");                                                                                    \
      e.slots = reinterpret_cast<double*>(slots); e.ns = bn_slots(Ci); e.mode = 2; e.relu = relu; e.C = Ci;            \
      e.z = z; e.yr = yrelu; e.mean = mean; e.invstd = invstd; e.gamma = gamma; e.beta = beta;                         \
    }                                                                                                                  \
    PairCapture pc;                                                                                                    \
    pc.a.kind = pc.b.kind = 0;                                                                                         \
    ReduceDesc d;                                                                                                      \
    d.part = nullptr;                                                                                                  \
    fami_pair_capture() = &pc;                                                                                         \
    int rc = conv_dgrad_h_impl<HT>(nm, dy, wpd, dx, N, H, W, Ci, Co, kh, kw, stride, pad, dil, acc_dx, s, e);          \
    if (rc == FAMI_OK) {                                                                                               \
      g_defer = &d;                                                                                                    \
      rc = wgrad_h_impl<HT>(nm, x, dy, dw, workspace, ws_bytes, N, H, W, Ci, Co, kh, kw, stride, pad, dil, acc_dw, s); \
      g_defer = nullptr;                                                                                               \
    }                                                                                                                  \
    fami_pair_capture() = nullptr;                                                                                     \
    if (rc != FAMI_OK) return rc;                                                                                      \
    FAMI_REQUIRE(d.part, nm, "no reduce recorded");                                                                    \
    memcpy(desc_out, &d, sizeof(d));                                                                                   \
    const int pr = fami_pair_launch(pc, s);                                                                            \
    FAMI_REQUIRE(pr >= 0, nm, "a recorded launch has no kernel instance");                                             \
    FAMI_CHECK_LAUNCH(nm);                                                                                             \
    return FAMI_OK;                                                                                                    \
  }
FAMI_CONV_H_ABI(bf16, bf16_t)
FAMI_CONV_H_ABI(f16, f16_t)
FAMI_CONV_PAIR_ABI(bf16, bf16_t)
FAMI_CONV_PAIR_ABI(f16, f16_t)
#undef FAMI_CONV_H_ABI
#undef FAMI_CONV_PAIR_ABI

}  // extern "C"
