// BatchNorm (train-mode batch statistics + running-stat update, eval-mode
// apply) and its backward, fused with ReLU and the residual add, for NHWC fp32
// activations viewed as a [P = N*H*W][C] matrix.  HBM-bound streaming kernels:
// each thread owns one 4-channel vector column and strides over pixels, so a
// wave reads whole 16-byte-per-lane contiguous rows.
//
// Replaces nn.BatchNorm2d (+ nn.ReLU, `out += residual`) as used by
//   posetimation/layers/basic_model.py:25-63 (BasicBlock), :66-113 (Bottleneck),
//   posetimation/layers/basic_layer.py:25-26 (conv_bn_relu.bn),
//   posetimation/backbones/hrnet.py:99-143 (fuse layers), :724-762 (transitions)
// with momentum 0.1, eps 1e-5, biased variance for normalisation and unbiased
// variance for running_var (torch.nn.BatchNorm2d semantics).
#include "common.h"

#define BN_MAXG 512
#ifndef BN_U
#define BN_U 4  // rows per thread in flight in the statistics passes
#endif

struct ColMap {
  int cv, prow, rows, CV;
  bool active;
};
__device__ __forceinline__ ColMap col_map(int C) {
  ColMap m;
  m.CV = C >> 2;
  m.rows = 256 / m.CV;
  m.cv = threadIdx.x % m.CV;
  m.prow = threadIdx.x / m.CV;
  m.active = m.prow < m.rows;
  return m;
}

// sum of one double per thread over a 256-thread block (result valid in thread 0)
__device__ __forceinline__ double block_sum_d(double v, double* sm4) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm4[wave] = v;
  __syncthreads();
  return (sm4[0] + sm4[1]) + (sm4[2] + sm4[3]);
}

// partial[g][0][c] = sum (x - K), partial[g][1][c] = sum (x - K)^2 over this block's pixels, with the pivot
// K[c] = x[0][c] (the first pixel): shifted sums keep E[d^2] - E[d]^2 free of the cancellation that the raw
// moments suffer when |mean| >> std.
template <typename T>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ x, float* __restrict__ partial,
                                                         long P, int C) {
  extern __shared__ float sm[];  // [rows][2][C]
  const ColMap m = col_map(C);
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
  if (m.active) {
    const f32x4 piv = ld4(x + m.cv * 4);
    // BN_U rows per thread in flight: with one load per thread and <= 512 workgroups the pass is bound by memory
    // latency (8 KB in flight per CU), not bandwidth.  The sums are taken in the same row order as a rolled loop.
    const long step = (long)gridDim.x * m.rows;
    long p = (long)blockIdx.x * m.rows + m.prow;
    for (; p + (BN_U - 1) * step < P; p += BN_U * step) {
      f32x4 v[BN_U];
#pragma unroll
      for (int u = 0; u < BN_U; ++u) v[u] = ld4(x + (p + u * step) * C + m.cv * 4);
#pragma unroll
      for (int u = 0; u < BN_U; ++u) {
        const f32x4 d = v[u] - piv;
        s += d;
        q += d * d;
      }
    }
    for (; p < P; p += step) {
      const f32x4 v = ld4(x + p * C + m.cv * 4) - piv;
      s += v;
      q += v * v;
    }
    float* d = sm + (long)m.prow * 2 * C;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      d[m.cv * 4 + t] = s[t];
      d[C + m.cv * 4 + t] = q[t];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * C; e += 256) {
    float tot = 0.f;
    for (int r = 0; r < m.rows; ++r) tot += sm[(long)r * 2 * C + e];
    partial[(long)blockIdx.x * 2 * C + e] = tot;
  }
}

template <typename T>
__global__ void bn_finalize_kernel(const float* __restrict__ partial, const T* __restrict__ x, int G, long P,
                                   int C, float* mean, float* invstd, float* running_mean, float* running_var,
                                   float momentum, float eps) {
  __shared__ double sm4[4];
  const int c = blockIdx.x;  // one 256-thread block per channel: G <= 512 partials are two loads per thread
  double s = 0.0, q = 0.0;
  for (int g = threadIdx.x; g < G; g += 256) {
    s += (double)partial[(long)g * 2 * C + c];
    q += (double)partial[(long)g * 2 * C + C + c];
  }
  s = block_sum_d(s, sm4);
  q = block_sum_d(q, sm4);
  if (threadIdx.x != 0) return;
  const double dm = s / (double)P;              // mean of (x - pivot)
  double var = q / (double)P - dm * dm;
  if (var < 0.0) var = 0.0;
  const double mu = (double)ld1(x + c) + dm;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unb = P > 1 ? var * (double)P / (double)(P - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
  }
}

// mean / invstd (+ running statistics) from slot rows a convolution epilogue filled (fuse terms: the cross-resolution sum
// kernel applies the normalisation itself and only needs the two vectors)
__global__ void bn_finalize_slots_kernel(const double* __restrict__ slots, int NS, const float* __restrict__ pivot, long P,
                                         int C, float* mean, float* invstd, float* running_mean, float* running_var,
                                         float momentum, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < NS; ++k) {
    s += slots[(long)k * 2 * C + c];
    q += slots[(long)k * 2 * C + C + c];
  }
  const double invP = 1.0 / (double)P;
  const double dm = s * invP;
  double var = q * invP - dm * dm;
  if (var < 0.0) var = 0.0;
  const double mu = (double)pivot[c] + dm;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unb = P > 1 ? var * (double)P / (double)(P - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
  }
}

// running statistics from an already finalised (mean, invstd): lets BatchNorm calls that SHARE a module run concurrently
// on several streams without touching the running buffers, and applies their updates afterwards in call order
// (Alignment_V15.py:125-135 applies one regressor to every supporting frame, updating frame by frame).
__global__ void bn_running_update_kernel(float* running_mean, float* running_var, const float* __restrict__ mean,
                                         const float* __restrict__ invstd, int C, long P, float momentum, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mu = (double)mean[c], is = (double)invstd[c];
  double var = 1.0 / (is * is) - (double)eps;
  if (var < 0.0) var = 0.0;
  const double unb = P > 1 ? var * (double)P / (double)(P - 1) : var;
  running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
  running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
}


// Batched form: the deferred updates of a forked region (32 per step for the translation regressors) in ONE launch, in
// call order -- thread c walks the entries sequentially for its channel, so several updates of one module's buffers
// apply in the order the reference applies them (frame by frame).
#define FAMI_BNRU_BATCH 32
struct BnRunBatch { float* rm[FAMI_BNRU_BATCH]; float* rv[FAMI_BNRU_BATCH]; const float* mean[FAMI_BNRU_BATCH]; const float* invstd[FAMI_BNRU_BATCH];
                    int C[FAMI_BNRU_BATCH]; float P[FAMI_BNRU_BATCH], mom[FAMI_BNRU_BATCH], eps[FAMI_BNRU_BATCH]; int n; };
__global__ __launch_bounds__(256) void bn_running_update_batch_kernel(BnRunBatch b) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < b.n; ++k) {
    if (c >= b.C[k]) continue;
    const double mu = (double)b.mean[k][c], is = (double)b.invstd[k][c];
    double var = 1.0 / (is * is) - (double)b.eps[k];
    if (var < 0.0) var = 0.0;
    const double P = (double)b.P[k];
    const double unb = P > 1.0 ? var * P / (P - 1.0) : var;
    const float momentum = b.mom[k];
    b.rm[k][c] = (float)((1.0 - momentum) * b.rm[k][c] + momentum * mu);
    b.rv[k][c] = (float)((1.0 - momentum) * b.rv[k][c] + momentum * unb);
  }
}

__global__ void bn_eval_stats_kernel(const float* running_mean, const float* running_var, float* mean, float* invstd,
                                     int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = running_mean[c];
  invstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// y = [relu]( (x-mean)*invstd*gamma + beta [+ residual] )
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const T* __restrict__ residual, T* __restrict__ y,
                                                       long P, int C, int relu) {
  const ColMap m = col_map(C);
  if (!m.active) return;
  f32x4 sc, sf;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = m.cv * 4 + t;
    sc[t] = invstd[c] * gamma[c];
    sf[t] = beta[c] - mean[c] * sc[t];
  }
  for (long p = (long)blockIdx.x * m.rows + m.prow; p < P; p += (long)gridDim.x * m.rows) {
    const long o = p * C + m.cv * 4;
    f32x4 v = ld4(x + o) * sc + sf;
    if (residual) v += ld4(residual + o);
    if (relu) {
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
    }
    st4(y + o, v);
  }
}

// backward pass 1: partial[g][0][c] = sum dz, partial[g][1][c] = sum dz*xhat ; dz = relu ? dy*(y>0) : dy
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const T* __restrict__ dy,
                                                             const T* __restrict__ x,
                                                             const T* __restrict__ y,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             float* __restrict__ partial, long P, int C, int relu) {
  extern __shared__ float sm[];
  const ColMap m = col_map(C);
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
  if (m.active) {
    f32x4 mu, is;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      mu[t] = mean[m.cv * 4 + t];
      is[t] = invstd[m.cv * 4 + t];
    }
    const long step = (long)gridDim.x * m.rows;
    // Rows past the end re-read row p (a cache hit) and contribute an exact 0: the 3-7 rows per thread of the
    // low-resolution branches then also go out BN_U at a time instead of through a rolled tail (backward statistics
    // of the 48x36 / 24x18 maps -3...-11 %; the same form measured +7 % on the forward statistics of the 96x72 map,
    // which keeps the main loop + rolled tail).
    for (long p = (long)blockIdx.x * m.rows + m.prow; p < P; p += BN_U * step) {
      f32x4 g[BN_U], yy[BN_U], xx[BN_U];
      long o[BN_U];
#pragma unroll
      for (int u = 0; u < BN_U; ++u) {
        const long pu = p + u * step;
        o[u] = (pu < P ? pu : p) * C + m.cv * 4;
        g[u] = ld4(dy + o[u]);
      }
      if (relu) {
#pragma unroll
        for (int u = 0; u < BN_U; ++u) yy[u] = ld4(y + o[u]);
      }
#pragma unroll
      for (int u = 0; u < BN_U; ++u) xx[u] = ld4(x + o[u]);
#pragma unroll
      for (int u = 0; u < BN_U; ++u) {
        if (relu) {
#pragma unroll
          for (int t = 0; t < 4; ++t) g[u][t] = yy[u][t] > 0.f ? g[u][t] : 0.f;
        }
        if (p + u * step >= P) g[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 xh = (xx[u] - mu) * is;
        s += g[u];
        q += g[u] * xh;
      }
    }
    float* d = sm + (long)m.prow * 2 * C;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      d[m.cv * 4 + t] = s[t];
      d[C + m.cv * 4 + t] = q[t];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * C; e += 256) {
    float tot = 0.f;
    for (int r = 0; r < m.rows; ++r) tot += sm[(long)r * 2 * C + e];
    partial[(long)blockIdx.x * 2 * C + e] = tot;
  }
}

// coef[0][c] = mean(dz), coef[1][c] = mean(dz*xhat); dgamma/dbeta optional
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, int G, long P, int C, float* coef,
                                       float* dgamma, float* dbeta, int accumulate) {
  __shared__ double sm4[4];
  const int c = blockIdx.x;  // one 256-thread block per channel
  double s = 0.0, q = 0.0;
  for (int g = threadIdx.x; g < G; g += 256) {
    s += (double)partial[(long)g * 2 * C + c];
    q += (double)partial[(long)g * 2 * C + C + c];
  }
  s = block_sum_d(s, sm4);
  q = block_sum_d(q, sm4);
  if (threadIdx.x != 0) return;
  coef[c] = (float)(s / (double)P);
  coef[C + c] = (float)(q / (double)P);
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)q : (float)q;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
}

// backward pass 2: dx = gamma*invstd*(dz - c1 - xhat*c2) ; dres (=|+=) dz
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const T* __restrict__ y,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ coef, T* __restrict__ dx,
                                                           T* __restrict__ dres, long P, int C, int relu,
                                                           int acc_dx, int acc_dres) {
  const ColMap m = col_map(C);
  if (!m.active) return;
  f32x4 mu, is, gi, c1, c2;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = m.cv * 4 + t;
    mu[t] = mean[c];
    is[t] = invstd[c];
    gi[t] = gamma[c] * invstd[c];
    c1[t] = coef[c];
    c2[t] = coef[C + c];
  }
  for (long p = (long)blockIdx.x * m.rows + m.prow; p < P; p += (long)gridDim.x * m.rows) {
    const long o = p * C + m.cv * 4;
    f32x4 g = ld4(dy + o);
    if (relu) {
      const f32x4 yy = ld4(y + o);
#pragma unroll
      for (int t = 0; t < 4; ++t) g[t] = yy[t] > 0.f ? g[t] : 0.f;
    }
    const f32x4 xh = (ld4(x + o) - mu) * is;
    f32x4 d = gi * (g - c1 - xh * c2);
    if (acc_dx) d += ld4(dx + o);
    st4(dx + o, d);
    if (dres) {
      f32x4 r = g;
      if (acc_dres) r += ld4(dres + o);
      st4(dres + o, r);
    }
  }
}

// ---- two-launch forms (fami_bn_train_fwd2 / fami_bn_bwd2): the statistics pass adds its per-workgroup partial sums into
// NS slot rows of fp64 (global_atomic_add_f64; slot = workgroup % NS, so an address sees grid/NS adds), and the apply
// pass folds the NS rows itself in its prologue -- the one-workgroup-per-channel finalize launch in between (4-6.6 us of
// pure dependent latency per BatchNorm, ~600 launches per training step) is gone.  The slots must be zero on entry
// (the caller hands out slices of an arena it clears once per step).  fp64 adds in arrival order: the sums are
// reproducible to ~1e-16 relative, i.e. the fp32 mean / invstd almost always bit for bit, but not guaranteed -- the
// three-launch forms above stay for the deterministic mode.
// (BN_NS_MAX, bn_slots, bn_slots_bytes, bn_slots_pivot live in common.h: the convolution epilogues fill the same rows)

template <typename T>
__global__ __launch_bounds__(256) void bn_partial2_kernel(const T* __restrict__ x, double* __restrict__ slots, int NS,
                                                          long P, int C) {
  extern __shared__ float sm[];  // [rows][2][C]
  const ColMap m = col_map(C);
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
  if (m.active) {
    const f32x4 piv = ld4(x + m.cv * 4);
    const long step = (long)gridDim.x * m.rows;
    long p = (long)blockIdx.x * m.rows + m.prow;
    for (; p + (BN_U - 1) * step < P; p += BN_U * step) {
      f32x4 v[BN_U];
#pragma unroll
      for (int u = 0; u < BN_U; ++u) v[u] = ld4(x + (p + u * step) * C + m.cv * 4);
#pragma unroll
      for (int u = 0; u < BN_U; ++u) {
        const f32x4 d = v[u] - piv;
        s += d;
        q += d * d;
      }
    }
    for (; p < P; p += step) {
      const f32x4 v = ld4(x + p * C + m.cv * 4) - piv;
      s += v;
      q += v * v;
    }
    float* d = sm + (long)m.prow * 2 * C;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      d[m.cv * 4 + t] = s[t];
      d[C + m.cv * 4 + t] = q[t];
    }
  }
  __syncthreads();
  double* row = slots + (long)(blockIdx.x % NS) * 2 * C;
  for (int e = threadIdx.x; e < 2 * C; e += 256) {
    float tot = 0.f;
    for (int r = 0; r < m.rows; ++r) tot += sm[(long)r * 2 * C + e];
    unsafeAtomicAdd(row + e, (double)tot);
  }
}

// scale / shift of the normalisation exactly as the forward applies them; the backward recomputes the ReLU mask from them
__device__ __forceinline__ void bn_scale_shift(float mean, float invstd, float gamma, float beta, float& sc, float& sf) {
  sc = invstd * gamma;
  sf = __builtin_fmaf(-mean, sc, beta);
}

// y = [relu]( (x-mean)*invstd*gamma + beta [+ residual] ), mean / invstd folded from the slot rows by every workgroup
// (workgroup 0 also stores them and advances the running statistics)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply2_kernel(const T* __restrict__ x, const double* __restrict__ slots, int NS,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const T* __restrict__ residual, T* __restrict__ y,
                                                        float* __restrict__ mean, float* __restrict__ invstd,
                                                        float* running_mean, float* running_var, long P, int C,
                                                        int relu, float momentum, float eps, const float* pivot) {
  // pivot: null = the sums were shifted by the first pixel x[0][c] (bn_partial2_kernel); else the [C] pivots the
  // convolution epilogue that filled the slots wrote behind them (conv.hip, EpiBN mode 1)
  extern __shared__ float sm[];  // [2][C]: scale, shift
  for (int c = threadIdx.x; c < C; c += 256) {
    double s = 0.0, q = 0.0;
    for (int k = 0; k < NS; ++k) {
      s += slots[(long)k * 2 * C + c];
      q += slots[(long)k * 2 * C + C + c];
    }
    const double invP = 1.0 / (double)P;
    const double dm = s * invP;                       // mean of (x - pivot)
    double var = q * invP - dm * dm;
    if (var < 0.0) var = 0.0;
    const double mu = (double)(pivot ? pivot[c] : ld1(x + c)) + dm;
    const float muf = (float)mu;
    const float isf = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.x == 0) {
      mean[c] = muf;
      invstd[c] = isf;
      if (running_mean) {
        const double unb = P > 1 ? var * (double)P / (double)(P - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
      }
    }
    float sc, sf;
    bn_scale_shift(muf, isf, gamma[c], beta[c], sc, sf);
    sm[c] = sc;
    sm[C + c] = sf;
  }
  __syncthreads();
  const ColMap m = col_map(C);
  if (!m.active) return;
  const f32x4 sc = *reinterpret_cast<const f32x4*>(sm + m.cv * 4), sf = *reinterpret_cast<const f32x4*>(sm + C + m.cv * 4);
  for (long p = (long)blockIdx.x * m.rows + m.prow; p < P; p += (long)gridDim.x * m.rows) {
    const long o = p * C + m.cv * 4;
    const f32x4 xv = ld4(x + o);
    f32x4 v;
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = __builtin_fmaf(xv[t], sc[t], sf[t]);
    if (residual) v += ld4(residual + o);
    if (relu) {
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
    }
    st4(y + o, v);
  }
}

// ReLU mask of a BatchNorm output: relu 1 = from the stored output y, relu 2 = recomputed from x (only valid without a
// residual, and only against a forward that applied bn_scale_shift with explicit fused multiply-adds: bn_apply2_kernel)
__device__ __forceinline__ f32x4 bn_relu_mask(f32x4 g, int relu, f32x4 yy, f32x4 xv, f32x4 sc, f32x4 sf) {
  if (relu == 1) {
#pragma unroll
    for (int t = 0; t < 4; ++t) g[t] = yy[t] > 0.f ? g[t] : 0.f;
  } else if (relu == 2) {
#pragma unroll
    for (int t = 0; t < 4; ++t) g[t] = __builtin_fmaf(xv[t], sc[t], sf[t]) > 0.f ? g[t] : 0.f;
  }
  return g;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial2_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                              const T* __restrict__ y, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              double* __restrict__ slots, int NS, long P, int C, int relu) {
  extern __shared__ float sm[];
  const ColMap m = col_map(C);
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
  if (m.active) {
    f32x4 mu, is, sc, sf;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = m.cv * 4 + t;
      mu[t] = mean[c];
      is[t] = invstd[c];
      float a, b;
      bn_scale_shift(mu[t], is[t], gamma[c], beta[c], a, b);
      sc[t] = a;
      sf[t] = b;
    }
    const long step = (long)gridDim.x * m.rows;
    for (long p = (long)blockIdx.x * m.rows + m.prow; p < P; p += BN_U * step) {
      f32x4 g[BN_U], yy[BN_U], xx[BN_U];
      long o[BN_U];
#pragma unroll
      for (int u = 0; u < BN_U; ++u) {
        const long pu = p + u * step;
        o[u] = (pu < P ? pu : p) * C + m.cv * 4;
        g[u] = ld4(dy + o[u]);
      }
      if (relu == 1) {
#pragma unroll
        for (int u = 0; u < BN_U; ++u) yy[u] = ld4(y + o[u]);
      }
#pragma unroll
      for (int u = 0; u < BN_U; ++u) xx[u] = ld4(x + o[u]);
#pragma unroll
      for (int u = 0; u < BN_U; ++u) {
        g[u] = bn_relu_mask(g[u], relu, yy[u], xx[u], sc, sf);
        if (p + u * step >= P) g[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 xh = (xx[u] - mu) * is;
        s += g[u];
        q += g[u] * xh;
      }
    }
    float* d = sm + (long)m.prow * 2 * C;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      d[m.cv * 4 + t] = s[t];
      d[C + m.cv * 4 + t] = q[t];
    }
  }
  __syncthreads();
  double* row = slots + (long)(blockIdx.x % NS) * 2 * C;
  for (int e = threadIdx.x; e < 2 * C; e += 256) {
    float tot = 0.f;
    for (int r = 0; r < m.rows; ++r) tot += sm[(long)r * 2 * C + e];
    unsafeAtomicAdd(row + e, (double)tot);
  }
}

// dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)) ; dres (=|+=) dz ; the two means folded from the slot rows by
// every workgroup (workgroup 0 also stores dgamma / dbeta)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply2_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const T* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const double* __restrict__ slots, int NS, T* __restrict__ dx,
                                                            float* dgamma, float* dbeta, T* __restrict__ dres, long P,
                                                            int C, int relu, int acc_dx, int acc_param, int acc_dres) {
  extern __shared__ float sm[];  // [2][C]: mean(dz), mean(dz*xhat)
  for (int c = threadIdx.x; c < C; c += 256) {
    double s = 0.0, q = 0.0;
    for (int k = 0; k < NS; ++k) {
      s += slots[(long)k * 2 * C + c];
      q += slots[(long)k * 2 * C + C + c];
    }
    sm[c] = (float)(s / (double)P);
    sm[C + c] = (float)(q / (double)P);
    if (blockIdx.x == 0) {
      if (dgamma) dgamma[c] = acc_param ? dgamma[c] + (float)q : (float)q;
      if (dbeta) dbeta[c] = acc_param ? dbeta[c] + (float)s : (float)s;
    }
  }
  __syncthreads();
  const ColMap m = col_map(C);
  if (!m.active) return;
  f32x4 mu, is, gi, sc, sf;
  const f32x4 c1 = *reinterpret_cast<const f32x4*>(sm + m.cv * 4), c2 = *reinterpret_cast<const f32x4*>(sm + C + m.cv * 4);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = m.cv * 4 + t;
    mu[t] = mean[c];
    is[t] = invstd[c];
    gi[t] = gamma[c] * invstd[c];
    float a, b;
    bn_scale_shift(mu[t], is[t], gamma[c], beta[c], a, b);
    sc[t] = a;
    sf[t] = b;
  }
  for (long p = (long)blockIdx.x * m.rows + m.prow; p < P; p += (long)gridDim.x * m.rows) {
    const long o = p * C + m.cv * 4;
    f32x4 g = ld4(dy + o);
    const f32x4 xv = ld4(x + o);
    f32x4 yy = {0.f, 0.f, 0.f, 0.f};
    if (relu == 1) yy = ld4(y + o);
    g = bn_relu_mask(g, relu, yy, xv, sc, sf);
    const f32x4 xh = (xv - mu) * is;
    f32x4 d = gi * (g - c1 - xh * c2);
    if (acc_dx) d += ld4(dx + o);
    st4(dx + o, d);
    if (dres) {
      f32x4 r = g;
      if (acc_dres) r += ld4(dres + o);
      st4(dres + o, r);
    }
  }
}

// ---- small-tensor BatchNorm: statistics + apply (forward) / both reductions + apply (backward) in ONE launch.
// One workgroup per 4-channel column: the tensor of a low-resolution HRNet branch (P <= BN_SMALL_P pixels) is
// L2-resident, so the second pass re-reads it from cache and the three launches of the general path collapse
// into one (these layers are launch-latency bound, not bandwidth bound).  Same arithmetic: shifted moments,
// fp64 finalize, biased variance for normalisation / unbiased for running_var.
#define BN_SMALL_P 16384
// one workgroup per 4 channels is serial over pixels: measured slower than the three-launch path for everything but
// tiny tensors (the translation regressor's 16-channel maps below 24x18), where launch count is all that matters
// [fami_route_t] g_bn_small_elems (default 32768)  // fami_bn_tune_small: tensors up to this many elements take the one-launch kernels
static inline bool bn_small_ok(long P, int C) { return P * C <= g_bn_small_elems; }

__device__ __forceinline__ f32x4 block_sum4(f32x4 v, float* sm) {  // sm: 4 * 4 floats
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = wave_sum(v[t]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) sm[wave * 4 + t] = v[t];
  }
  __syncthreads();
  f32x4 r;
#pragma unroll
  for (int t = 0; t < 4; ++t) r[t] = (sm[t] + sm[4 + t]) + (sm[8 + t] + sm[12 + t]);
  return r;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const T* __restrict__ x, const T* __restrict__ residual,
                                                           T* __restrict__ y, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ mean,
                                                           float* __restrict__ invstd, float* running_mean,
                                                           float* running_var, long P, int C, int relu,
                                                           float momentum, float eps) {
  __shared__ float sm[16];
  const int c0 = blockIdx.x * 4;
  const f32x4 piv = ld4(x + c0);
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
  for (long p = threadIdx.x; p < P; p += 256) {
    const f32x4 v = ld4(x + p * C + c0) - piv;
    s += v;
    q += v * v;
  }
  s = block_sum4(s, sm);
  q = block_sum4(q, sm);
  f32x4 mu, is, sc, sf;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const double dm = (double)s[t] / (double)P;
    double var = (double)q[t] / (double)P - dm * dm;
    if (var < 0.0) var = 0.0;
    const double m = (double)piv[t] + dm;
    mu[t] = (float)m;
    is[t] = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
      mean[c0 + t] = mu[t];
      invstd[c0 + t] = is[t];
      if (running_mean) {
        const double unb = P > 1 ? var * (double)P / (double)(P - 1) : var;
        running_mean[c0 + t] = (float)((1.0 - momentum) * running_mean[c0 + t] + momentum * m);
        running_var[c0 + t] = (float)((1.0 - momentum) * running_var[c0 + t] + momentum * unb);
      }
    }
    sc[t] = is[t] * gamma[c0 + t];
    sf[t] = beta[c0 + t] - mu[t] * sc[t];
  }
  for (long p = threadIdx.x; p < P; p += 256) {
    const long o = p * C + c0;
    f32x4 v = ld4(x + o) * sc + sf;
    if (residual) v += ld4(residual + o);
    if (relu) {
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
    }
    st4(y + o, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const T* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, T* __restrict__ dx,
                                                           float* dgamma, float* dbeta, T* __restrict__ dres, long P,
                                                           int C, int relu, int acc_dx, int acc_param, int acc_dres) {
  __shared__ float sm[16];
  const int c0 = blockIdx.x * 4;
  f32x4 mu, is, gi;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    mu[t] = mean[c0 + t];
    is[t] = invstd[c0 + t];
    gi[t] = gamma[c0 + t] * is[t];
  }
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
  for (long p = threadIdx.x; p < P; p += 256) {
    const long o = p * C + c0;
    f32x4 g = ld4(dy + o);
    if (relu) {
      const f32x4 yy = ld4(y + o);
#pragma unroll
      for (int t = 0; t < 4; ++t) g[t] = yy[t] > 0.f ? g[t] : 0.f;
    }
    s += g;
    q += g * ((ld4(x + o) - mu) * is);
  }
  s = block_sum4(s, sm);
  q = block_sum4(q, sm);
  f32x4 c1, c2;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    c1[t] = (float)((double)s[t] / (double)P);
    c2[t] = (float)((double)q[t] / (double)P);
    if (threadIdx.x == 0) {
      if (dgamma) dgamma[c0 + t] = acc_param ? dgamma[c0 + t] + q[t] : q[t];
      if (dbeta) dbeta[c0 + t] = acc_param ? dbeta[c0 + t] + s[t] : s[t];
    }
  }
  for (long p = threadIdx.x; p < P; p += 256) {
    const long o = p * C + c0;
    f32x4 g = ld4(dy + o);
    if (relu) {
      const f32x4 yy = ld4(y + o);
#pragma unroll
      for (int t = 0; t < 4; ++t) g[t] = yy[t] > 0.f ? g[t] : 0.f;
    }
    const f32x4 xh = (ld4(x + o) - mu) * is;
    f32x4 d = gi * (g - c1 - xh * c2);
    if (acc_dx) d += ld4(dx + o);
    st4(dx + o, d);
    if (dres) {
      f32x4 r = g;
      if (acc_dres) r += ld4(dres + o);
      st4(dres + o, r);
    }
  }
}

// per-channel sum over pixels (bias gradients): partial then finalize
template <typename T>
__global__ __launch_bounds__(256) void chan_sum_partial_kernel(const T* __restrict__ x,
                                                               float* __restrict__ partial, long P, int C) {
  // scalar-channel version: works for any C (17, 2, 288, ...); C > 256 walks the channels in strides of 256
  extern __shared__ float sm[];  // [rows][C]
  if (C > 256) {
    for (int c = threadIdx.x; c < C; c += 256) {
      float s = 0.f;
      for (long p = blockIdx.x; p < P; p += gridDim.x) s += ld1(x + p * C + c);
      partial[(long)blockIdx.x * C + c] = s;
    }
    return;
  }
  const int rows = max(1, 256 / C);
  const int c = threadIdx.x % C, prow = threadIdx.x / C;
  const bool active = prow < rows;
  float s = 0.f;
  if (active) {
    for (long p = (long)blockIdx.x * rows + prow; p < P; p += (long)gridDim.x * rows) s += ld1(x + p * C + c);
    sm[prow * C + c] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < C; e += 256) {
    float tot = 0.f;
    for (int r = 0; r < rows; ++r) tot += sm[r * C + e];
    partial[(long)blockIdx.x * C + e] = tot;
  }
}
// four channels per thread, BN_U rows in flight (C % 4 == 0, C <= 1024): the bias gradients of the alignment head's 48-channel
// convolutions took 19 us per launch in the scalar form above (2-byte loads) against 3 us of HBM time
template <typename T>
__global__ __launch_bounds__(256) void chan_sum_partial4_kernel(const T* __restrict__ x, float* __restrict__ partial, long P,
                                                                int C) {
  extern __shared__ float sm[];  // [rows][C]
  const ColMap m = col_map(C);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (m.active) {
    const long step = (long)gridDim.x * m.rows;
    long p = (long)blockIdx.x * m.rows + m.prow;
    for (; p + (BN_U - 1) * step < P; p += BN_U * step) {
      f32x4 v[BN_U];
#pragma unroll
      for (int u = 0; u < BN_U; ++u) v[u] = ld4(x + (p + u * step) * C + m.cv * 4);
#pragma unroll
      for (int u = 0; u < BN_U; ++u) s += v[u];
    }
    for (; p < P; p += step) s += ld4(x + p * C + m.cv * 4);
    *reinterpret_cast<f32x4*>(sm + (long)m.prow * C + m.cv * 4) = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < C; e += 256) {
    float tot = 0.f;
    for (int r = 0; r < m.rows; ++r) tot += sm[r * C + e];
    partial[(long)blockIdx.x * C + e] = tot;
  }
}
__global__ void chan_sum_finalize_kernel(const float* __restrict__ partial, int G, int C, float* out,
                                         int accumulate) {
  __shared__ double sm4[4];
  const int c = blockIdx.x;  // one 256-thread block per channel
  double s = 0.0;
  for (int g = threadIdx.x; g < G; g += 256) s += (double)partial[(long)g * C + c];
  s = block_sum_d(s, sm4);
  if (threadIdx.x != 0) return;
  out[c] = accumulate ? out[c] + (float)s : (float)s;
}

static inline int bn_grid(long P, int C) {
  const int rows = 256 / (C >> 2);
  // at least 2*BN_U row steps per workgroup: on the low-resolution branches (12x9x384: 1080 row groups) a grid of 512
  // made the partial array as large as the tensor itself and the finalize pass, which walks it with a stride, the
  // longer half of the chain
  long g = (P + (long)rows * 2 * BN_U - 1) / ((long)rows * 2 * BN_U);
  if (g > BN_MAXG) g = BN_MAXG;
  if (g < 1) g = 1;
  return (int)g;
}
// the statistics passes of the two-launch forms write no partial array (fp64 slot atomics), so their grid is not tied to BN_MAXG:
// fami_bn_tune_small(-(1000 + n)) sets the cap (route field bn2_maxg)
// Measured inside the step (tools/ab_env.py, caps 512 / 1024 / 2048 / 4096): f32 storage 45.50 / 45.45 / 45.30 ms and 46.68 / - / 46.64 /
// 46.57 on a second box, config 4 (512x384, 8 frames) 116.8 / - / 116.5 / 116.2 ms; bf16 20.02 / 20.01 / 20.14 ms -- the f32 tensors of
// the stem stretch are twice the bytes and want more loads in flight: four times the cap in f32 storage.
static inline int bn_grid2(long P, int C, int esz) {
  const int rows = 256 / (C >> 2);
  long g = (P + (long)rows * 2 * BN_U - 1) / ((long)rows * 2 * BN_U);
  const long cap = (long)g_bn2_maxg * (esz == 4 ? 4 : 1);
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
static inline bool bn_shape_ok(long P, int C) { return P > 0 && C >= 4 && (C % 4) == 0 && C <= 1024; }

// ------------------------------------------------------------------ host side (templates over the storage type)
template <typename T>
static int bn_stats_impl(const T* x, long P, int C, float* mean, float* invstd, float* running_mean,
                         float* running_var, float momentum, float eps, float* ws, hipStream_t s, const char* nm) {
  FAMI_REQUIRE(x && mean && invstd && ws, nm, "null pointer");
  if (!bn_shape_ok(P, C)) {
    fami_set_error(nm, "C must be a multiple of 4, <= 1024");
    return FAMI_ESHAPE;
  }
  const int G = bn_grid(P, C);
  const int rows = 256 / (C >> 2);
  hipLaunchKernelGGL(bn_partial_kernel<T>, dim3(G), dim3(256), (size_t)rows * 2 * C * sizeof(float), s, x, ws, P, C);
  FAMI_CHECK_LAUNCH(nm);
  hipLaunchKernelGGL(bn_finalize_kernel<T>, dim3(C), dim3(256), 0, s, ws, x, G, P, C, mean, invstd, running_mean,
                     running_var, momentum, eps);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}

template <typename T>
static int bn_apply_impl(const T* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                         const T* residual, T* y, long P, int C, int relu, hipStream_t s, const char* nm) {
  FAMI_REQUIRE(x && mean && invstd && gamma && beta && y, nm, "null pointer");
  if (!bn_shape_ok(P, C)) {
    fami_set_error(nm, "C must be a multiple of 4, <= 1024");
    return FAMI_ESHAPE;
  }
  const int rows = 256 / (C >> 2);
  long g = (P + rows - 1) / rows;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(bn_apply_kernel<T>, dim3((int)g), dim3(256), 0, s, x, mean, invstd, gamma, beta, residual, y, P,
                     C, relu);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}

template <typename T>
static int bn_bwd_impl(const T* dy, const T* x, const T* y, const float* mean, const float* invstd,
                       const float* gamma, T* dx, float* dgamma, float* dbeta, T* dres, long P, int C, int relu,
                       int acc_dx, int acc_param, int acc_dres, float* ws, hipStream_t s, const char* nm) {
  FAMI_REQUIRE(dy && x && mean && invstd && gamma && dx && ws, nm, "null pointer");
  FAMI_REQUIRE(!relu || y, nm, "relu needs y");
  if (!bn_shape_ok(P, C)) {
    fami_set_error(nm, "C must be a multiple of 4, <= 1024");
    return FAMI_ESHAPE;
  }
  if (bn_small_ok(P, C)) {  // low-resolution branches: one launch (see bn_small_bwd_kernel)
    hipLaunchKernelGGL(bn_small_bwd_kernel<T>, dim3(C / 4), dim3(256), 0, s, dy, x, y, mean, invstd, gamma, dx, dgamma,
                       dbeta, dres, P, C, relu, acc_dx, acc_param, acc_dres);
    FAMI_CHECK_LAUNCH(nm);
    return FAMI_OK;
  }
  const int G = bn_grid(P, C);
  const int rows = 256 / (C >> 2);
  float* coef = ws + (long)BN_MAXG * 2 * C - 2 * C;  // tail of the workspace (G < BN_MAXG leaves it free)
  const int Gp = G < BN_MAXG ? G : BN_MAXG - 1;
  hipLaunchKernelGGL(bn_bwd_partial_kernel<T>, dim3(Gp), dim3(256), (size_t)rows * 2 * C * sizeof(float), s, dy, x, y,
                     mean, invstd, ws, P, C, relu);
  FAMI_CHECK_LAUNCH(nm);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, ws, Gp, P, C, coef, dgamma, dbeta, acc_param);
  FAMI_CHECK_LAUNCH(nm);
  long g = (P + rows - 1) / rows;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3((int)g), dim3(256), 0, s, dy, x, y, mean, invstd, gamma, coef, dx,
                     dres, P, C, relu, acc_dx, acc_dres);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}

// train-mode BatchNorm forward in one call: statistics (+ running-stat update) and apply (+ residual, + ReLU).
// Small tensors take the single-launch kernel, large ones the three-launch path.
template <typename T>
static int bn_train_fwd_impl(const T* x, const T* residual, T* y, const float* gamma, const float* beta, float* mean,
                             float* invstd, float* running_mean, float* running_var, long P, int C, int relu,
                             float momentum, float eps, float* ws, hipStream_t s, const char* nm) {
  FAMI_REQUIRE(x && y && gamma && beta && mean && invstd && ws, nm, "null pointer");
  if (!bn_shape_ok(P, C)) {
    fami_set_error(nm, "C must be a multiple of 4, <= 1024");
    return FAMI_ESHAPE;
  }
  if (bn_small_ok(P, C)) {
    hipLaunchKernelGGL(bn_small_fwd_kernel<T>, dim3(C / 4), dim3(256), 0, s, x, residual, y, gamma, beta, mean, invstd,
                       running_mean, running_var, P, C, relu, momentum, eps);
    FAMI_CHECK_LAUNCH(nm);
    return FAMI_OK;
  }
  int rc = bn_stats_impl<T>(x, P, C, mean, invstd, running_mean, running_var, momentum, eps, ws, s, nm);
  if (rc != FAMI_OK) return rc;
  return bn_apply_impl<T>(x, mean, invstd, gamma, beta, residual, y, P, C, relu, s, nm);
}

// two-launch forms: `slots` = fami_bn_slots_bytes(C) bytes, ZERO on entry
static inline int bn_apply_grid(long P, int C) {
  const int rows = 256 / (C >> 2);
  long g = (P + rows - 1) / rows;
  if (g > 1024) g = 1024;   // every workgroup folds the slot rows in its prologue: fewer, longer workgroups than bn_apply_impl
  return (int)g;
}

template <typename T>
static int bn_train_fwd2_impl(const T* x, const T* residual, T* y, const float* gamma, const float* beta, float* mean,
                              float* invstd, float* running_mean, float* running_var, long P, int C, int relu,
                              float momentum, float eps, void* slots, int pre, hipStream_t s, const char* nm) {
  FAMI_REQUIRE(x && y && gamma && beta && mean && invstd && slots, nm, "null pointer");
  if (!bn_shape_ok(P, C)) {
    fami_set_error(nm, "C must be a multiple of 4, <= 1024");
    return FAMI_ESHAPE;
  }
  if (bn_small_ok(P, C) && !pre) {
    hipLaunchKernelGGL(bn_small_fwd_kernel<T>, dim3(C / 4), dim3(256), 0, s, x, residual, y, gamma, beta, mean, invstd,
                       running_mean, running_var, P, C, relu, momentum, eps);
    FAMI_CHECK_LAUNCH(nm);
    return FAMI_OK;
  }
  const int G = bn_grid2(P, C, (int)sizeof(T)), NS = bn_slots(C);
  const int rows = 256 / (C >> 2);
  if (!pre) {   // pre: the producing convolution's epilogue has filled the slot rows and the pivots already
    hipLaunchKernelGGL(bn_partial2_kernel<T>, dim3(G), dim3(256), (size_t)rows * 2 * C * sizeof(float), s, x,
                       reinterpret_cast<double*>(slots), NS, P, C);
    FAMI_CHECK_LAUNCH(nm);
  }
  hipLaunchKernelGGL(bn_apply2_kernel<T>, dim3(bn_apply_grid(P, C)), dim3(256), (size_t)2 * C * sizeof(float), s, x,
                     reinterpret_cast<const double*>(slots), NS, gamma, beta, residual, y, mean, invstd, running_mean,
                     running_var, P, C, relu, momentum, eps, pre ? bn_slots_pivot(slots, C) : (const float*)nullptr);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}

template <typename T>
static int bn_bwd2_impl(const T* dy, const T* x, const T* y, const float* mean, const float* invstd, const float* gamma,
                        const float* beta, T* dx, float* dgamma, float* dbeta, T* dres, long P, int C, int relu,
                        int acc_dx, int acc_param, int acc_dres, void* slots, int pre, hipStream_t s, const char* nm) {
  FAMI_REQUIRE(dy && x && mean && invstd && gamma && beta && dx && slots, nm, "null pointer");
  FAMI_REQUIRE(relu != 1 || y, nm, "relu = 1 needs y");
  FAMI_REQUIRE(relu >= 0 && relu <= 2, nm, "relu must be 0, 1 (mask from y) or 2 (mask recomputed from x)");
  if (!bn_shape_ok(P, C)) {
    fami_set_error(nm, "C must be a multiple of 4, <= 1024");
    return FAMI_ESHAPE;
  }
  if (bn_small_ok(P, C) && !pre) {  // the one-launch kernel takes its mask from y
    FAMI_REQUIRE(relu != 2 || y, nm, "small tensors take the ReLU mask from y");
    hipLaunchKernelGGL(bn_small_bwd_kernel<T>, dim3(C / 4), dim3(256), 0, s, dy, x, y, mean, invstd, gamma, dx, dgamma,
                       dbeta, dres, P, C, relu ? 1 : 0, acc_dx, acc_param, acc_dres);
    FAMI_CHECK_LAUNCH(nm);
    return FAMI_OK;
  }
  const int G = bn_grid2(P, C, (int)sizeof(T)), NS = bn_slots(C);
  const int rows = 256 / (C >> 2);
  if (!pre) {   // pre: the input-gradient convolution that produced dy summed dz and dz*xhat in its epilogue
    hipLaunchKernelGGL(bn_bwd_partial2_kernel<T>, dim3(G), dim3(256), (size_t)rows * 2 * C * sizeof(float), s, dy, x, y, mean,
                       invstd, gamma, beta, reinterpret_cast<double*>(slots), NS, P, C, relu);
    FAMI_CHECK_LAUNCH(nm);
  }
  hipLaunchKernelGGL(bn_bwd_apply2_kernel<T>, dim3(bn_apply_grid(P, C)), dim3(256), (size_t)2 * C * sizeof(float), s, dy, x,
                     y, mean, invstd, gamma, beta, reinterpret_cast<const double*>(slots), NS, dx, dgamma, dbeta, dres, P,
                     C, relu, acc_dx, acc_param, acc_dres);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}

template <typename T>
static int channel_sum_impl(const T* x, long P, int C, float* out, int accumulate, float* ws, hipStream_t s,
                            const char* nm) {
  FAMI_REQUIRE(x && out && ws && P > 0 && C > 0, nm, "bad argument");
  FAMI_REQUIRE(C <= 4096, nm, "C > 4096 unsupported");
  int rows = C > 256 ? 1 : 256 / C;
  const bool vec4 = C % 4 == 0 && C <= 1024 && P >= 4096;
  if (vec4) rows = 256 / (C >> 2);
  long g = (P + rows - 1) / rows;
  if (vec4) g = (g + 2 * BN_U - 1) / (2 * BN_U);          // at least 2 BN_U row steps per workgroup
  if (g > BN_MAXG) g = BN_MAXG;
  if (g < 1) g = 1;
  if (vec4) hipLaunchKernelGGL(chan_sum_partial4_kernel<T>, dim3((int)g), dim3(256), (size_t)rows * C * sizeof(float), s, x, ws, P, C);
  else hipLaunchKernelGGL(chan_sum_partial_kernel<T>, dim3((int)g), dim3(256), C > 256 ? 0 : (size_t)rows * C * sizeof(float), s, x, ws, P, C);
  FAMI_CHECK_LAUNCH(nm);
  hipLaunchKernelGGL(chan_sum_finalize_kernel, dim3(C), dim3(256), 0, s, ws, (int)g, C, out, accumulate);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}

extern "C" {

long fami_bn_workspace(int C) { return (long)BN_MAXG * 2 * C * (long)sizeof(float); }
long fami_bn_slots_bytes(int C) { return bn_slots_bytes(C); }
int fami_bn_tune_small(long elems) {
  if (elems <= -1000) { g_bn2_maxg = (int)(-elems - 1000); return FAMI_OK; }      // benchmarks: grid cap of the two-launch statistics passes
  if (elems < 0) { g_bn_small_elems = 32768; g_bn2_maxg = 512; return FAMI_OK; }
  g_bn_small_elems = elems;
  return FAMI_OK;
}
int fami_bn_is_small(long P, int C) { return bn_small_ok(P, C) ? 1 : 0; }
long fami_channel_sum_workspace(int C) { return (long)BN_MAXG * C * (long)sizeof(float); }

int fami_bn_eval_stats_f32(const float* running_mean, const float* running_var, float* mean, float* invstd, int C,
                           float eps, hipStream_t s) {
  FAMI_REQUIRE(running_mean && running_var && mean && invstd && C > 0, "fami_bn_eval_stats_f32", "bad argument");
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(fami_cdiv(C, 64)), dim3(64), 0, s, running_mean, running_var, mean,
                     invstd, C, eps);
  FAMI_CHECK_LAUNCH("fami_bn_eval_stats_f32");
  return FAMI_OK;
}

int fami_bn_finalize_slots_f32(void* slots, long P, int C, float* mean, float* invstd, float* running_mean,
                               float* running_var, float momentum, float eps, hipStream_t s) {
  FAMI_REQUIRE(slots && mean && invstd && C > 0 && P > 0, "fami_bn_finalize_slots_f32", "bad argument");
  hipLaunchKernelGGL(bn_finalize_slots_kernel, dim3(fami_cdiv(C, 64)), dim3(64), 0, s, reinterpret_cast<const double*>(slots),
                     bn_slots(C), bn_slots_pivot(slots, C), P, C, mean, invstd, running_mean, running_var, momentum, eps);
  FAMI_CHECK_LAUNCH("fami_bn_finalize_slots_f32");
  return FAMI_OK;
}

int fami_bn_running_update_f32(float* running_mean, float* running_var, const float* mean, const float* invstd, int C,
                               long P, float momentum, float eps, hipStream_t s) {
  FAMI_REQUIRE(running_mean && running_var && mean && invstd && C > 0 && P > 0, "fami_bn_running_update_f32", "bad argument");
  hipLaunchKernelGGL(bn_running_update_kernel, dim3(fami_cdiv(C, 64)), dim3(64), 0, s, running_mean, running_var, mean,
                     invstd, C, P, momentum, eps);
  FAMI_CHECK_LAUNCH("fami_bn_running_update_f32");
  return FAMI_OK;
}


// n deferred running-statistics updates in call order: ptrs = host array of 4 n longs (running_mean, running_var, mean,
// invstd per entry), meta = host array of 4 n floats (C, P, momentum, eps per entry; C and P are exact in fp32 here).
int fami_bn_running_update_batch_f32(const long* ptrs, const float* meta, int n, hipStream_t s) {
  FAMI_REQUIRE(ptrs && meta && n > 0, "fami_bn_running_update_batch_f32", "bad argument");
  for (int i0 = 0; i0 < n; i0 += FAMI_BNRU_BATCH) {
    BnRunBatch b;
    b.n = n - i0 < FAMI_BNRU_BATCH ? n - i0 : FAMI_BNRU_BATCH;
    int maxc = 1;
    for (int i = 0; i < FAMI_BNRU_BATCH; ++i) {
      const int j = i < b.n ? i0 + i : i0;
      b.rm[i] = reinterpret_cast<float*>(ptrs[4 * j]);
      b.rv[i] = reinterpret_cast<float*>(ptrs[4 * j + 1]);
      b.mean[i] = reinterpret_cast<const float*>(ptrs[4 * j + 2]);
      b.invstd[i] = reinterpret_cast<const float*>(ptrs[4 * j + 3]);
      b.C[i] = (int)meta[4 * j];
      b.P[i] = meta[4 * j + 1];
      b.mom[i] = meta[4 * j + 2];
      b.eps[i] = meta[4 * j + 3];
      if (i < b.n && b.C[i] > maxc) maxc = b.C[i];
    }
    hipLaunchKernelGGL(bn_running_update_batch_kernel, dim3(fami_cdiv(maxc, 256)), dim3(256), 0, s, b);
    FAMI_CHECK_LAUNCH("fami_bn_running_update_batch_f32");
  }
  return FAMI_OK;
}

#define FAMI_BN_ABI(sfx, T)                                                                                            \
  /* train-mode statistics: mean/invstd out, running stats updated in place (may be null) */                          \
  int fami_bn_stats_##sfx(const T* x, long P, int C, float* mean, float* invstd, float* running_mean,                  \
                          float* running_var, float momentum, float eps, float* ws, hipStream_t s) {                   \
    return bn_stats_impl<T>(x, P, C, mean, invstd, running_mean, running_var, momentum, eps, ws, s, "fami_bn_stats_" #sfx); \
  }                                                                                                                    \
  int fami_bn_apply_##sfx(const T* x, const float* mean, const float* invstd, const float* gamma, const float* beta,   \
                          const T* residual, T* y, long P, int C, int relu, hipStream_t s) {                           \
    return bn_apply_impl<T>(x, mean, invstd, gamma, beta, residual, y, P, C, relu, s, "fami_bn_apply_" #sfx);          \
  }                                                                                                                    \
  /* dz = relu ? dy*(y>0) : dy ; dx (=|+=) BN-backward(dz) ; dgamma/dbeta (=|+=) ; dres (=|+=) dz */                   \
  int fami_bn_bwd_##sfx(const T* dy, const T* x, const T* y, const float* mean, const float* invstd,                   \
                        const float* gamma, T* dx, float* dgamma, float* dbeta, T* dres, long P, int C, int relu,      \
                        int acc_dx, int acc_param, int acc_dres, float* ws, hipStream_t s) {                           \
    return bn_bwd_impl<T>(dy, x, y, mean, invstd, gamma, dx, dgamma, dbeta, dres, P, C, relu, acc_dx, acc_param,       \
                          acc_dres, ws, s, "fami_bn_bwd_" #sfx);                                                       \
  }                                                                                                                    \
  /* train-mode forward, statistics + apply in one call (one launch for P <= 16384 pixels) */                         \
  int fami_bn_train_fwd_##sfx(const T* x, const T* residual, T* y, const float* gamma, const float* beta,              \
                              float* mean, float* invstd, float* running_mean, float* running_var, long P, int C,      \
                              int relu, float momentum, float eps, float* ws, hipStream_t s) {                         \
    return bn_train_fwd_impl<T>(x, residual, y, gamma, beta, mean, invstd, running_mean, running_var, P, C, relu,      \
                                momentum, eps, ws, s, "fami_bn_train_fwd_" #sfx);                                      \
  }                                                                                                                    \
  /* two-launch forms (statistics with fp64 slot atomics + apply that folds the slots itself): `slots` =           */ \
  /* fami_bn_slots_bytes(C) bytes, ZERO on entry.  relu (backward): 0 none, 1 mask from y, 2 mask recomputed from x   */ \
  /* (no residual; y may be null) -- valid against fami_bn_train_fwd2 only.                                           */ \
  int fami_bn_train_fwd2_##sfx(const T* x, const T* residual, T* y, const float* gamma, const float* beta,             \
                               float* mean, float* invstd, float* running_mean, float* running_var, long P, int C,     \
                               int relu, float momentum, float eps, void* slots, hipStream_t s) {                      \
    return bn_train_fwd2_impl<T>(x, residual, y, gamma, beta, mean, invstd, running_mean, running_var, P, C, relu,     \
                                 momentum, eps, slots, 0, s, "fami_bn_train_fwd2_" #sfx);                              \
  }                                                                                                                    \
  /* the same with the statistics pass already done: `slots` was filled by fami_conv2d_fwd_stats_* (sums + pivots) */  \
  int fami_bn_apply_slots_##sfx(const T* x, const T* residual, T* y, const float* gamma, const float* beta,            \
                                float* mean, float* invstd, float* running_mean, float* running_var, long P, int C,    \
                                int relu, float momentum, float eps, void* slots, hipStream_t s) {                     \
    return bn_train_fwd2_impl<T>(x, residual, y, gamma, beta, mean, invstd, running_mean, running_var, P, C, relu,     \
                                 momentum, eps, slots, 1, s, "fami_bn_apply_slots_" #sfx);                             \
  }                                                                                                                    \
  /* backward apply pass alone: `slots` was filled by fami_conv2d_dgrad_bnstats_* (sums of dz and dz*xhat); dy holds  */\
  /* dz (the ReLU mask is applied already), so no mask is taken here                                                  */\
  int fami_bn_bwd_apply_slots_##sfx(const T* dz, const T* x, const float* mean, const float* invstd,                   \
                                    const float* gamma, const float* beta, T* dx, float* dgamma, float* dbeta,         \
                                    T* dres, long P, int C, int acc_dx, int acc_param, int acc_dres, void* slots,      \
                                    hipStream_t s) {                                                                   \
    return bn_bwd2_impl<T>(dz, x, (const T*)nullptr, mean, invstd, gamma, beta, dx, dgamma, dbeta, dres, P, C, 0,      \
                           acc_dx, acc_param, acc_dres, slots, 1, s, "fami_bn_bwd_apply_slots_" #sfx);                 \
  }                                                                                                                    \
  int fami_bn_bwd2_##sfx(const T* dy, const T* x, const T* y, const float* mean, const float* invstd,                  \
                         const float* gamma, const float* beta, T* dx, float* dgamma, float* dbeta, T* dres, long P,   \
                         int C, int relu, int acc_dx, int acc_param, int acc_dres, void* slots, hipStream_t s) {       \
    return bn_bwd2_impl<T>(dy, x, y, mean, invstd, gamma, beta, dx, dgamma, dbeta, dres, P, C, relu, acc_dx,           \
                           acc_param, acc_dres, slots, 0, s, "fami_bn_bwd2_" #sfx);                                    \
  }                                                                                                                    \
  /* out[c] (=|+=) sum_p x[p][c] */                                                                      \
  int fami_channel_sum_##sfx(const T* x, long P, int C, float* out, int accumulate, float* ws, hipStream_t s) {        \
    return channel_sum_impl<T>(x, P, C, out, accumulate, ws, s, "fami_channel_sum_" #sfx);                             \
  }
FAMI_BN_ABI(f32, float)
FAMI_BN_ABI(bf16, bf16_t)
FAMI_BN_ABI(f16, f16_t)
#undef FAMI_BN_ABI

}  // extern "C"
