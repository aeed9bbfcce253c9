// Streaming (HBM-bound) helpers of the hot path: boundary layout converts
// (the drop-in boundary is NCHW, the kernels run NHWC), channel concat / slice,
// axpby, the HRNet cross-resolution fuse (BN-apply + nearest-upsample + sum +
// ReLU in one pass), ReLU-masked gradient pooling, and the fused Adam step.
// Every activation-touching kernel is a template over the storage type T (float | bf16); arithmetic is fp32.
//
// Replaces torch.cat / torch.chunk / `sup - kf` (Alignment_V15.py:117-125,132,139,143,160),
// HighResolutionModule.forward's fuse loop (hrnet.py:159-168) with
// Interpolate(nearest) (basic_model.py:116-125), and torch.optim.Adam
// (posetimation/optimizer/optimizer.py:66-68; lr 1e-3, betas (0.9,0.999), eps 1e-8).
#include "common.h"

// boundary: NCHW fp32 (images, heatmap gradients) -> NHWC T
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int N, int C, int H, int W) {
  const long total = (long)N * C * H * W;
  const long HW = (long)H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long p = i / C;  // n*HW + hw
    const long n = p / HW, hw = p - n * HW;
    st1(dst + i, src[(n * C + c) * HW + hw]);
  }
}
// NHWC T -> NCHW fp32 (heatmaps, MI operands, nn.Flatten order)
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int N, int C, int H, int W,
                                    int accumulate) {
  const long total = (long)N * C * H * W;
  const long HW = (long)H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long hw = i % HW;
    const long nc = i / HW;
    const long n = nc / C;
    const int c = (int)(nc - n * C);
    const float v = ld1(src + (n * HW + hw) * C + c);
    dst[i] = accumulate ? dst[i] + v : v;
  }
}

// frames[(f*B + b), y, x, c] : f = 0 key frame, f >= 1 supporting frame f-1 (channels 3(f-1)..3(f-1)+2 of sup)
template <typename T>
__global__ void pack_frames_kernel(const float* __restrict__ kf, const float* __restrict__ sup, T* __restrict__ out,
                                   int B, int S, int H, int W) {
  const long HW = (long)H * W;
  const long total = (long)(1 + S) * B * HW * 3;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % 3);
    const long p = i / 3;
    const long n = p / HW, hw = p - n * HW;
    const int f = (int)(n / B), b = (int)(n - (long)f * B);
    float v;
    if (f == 0)
      v = kf[((long)b * 3 + c) * HW + hw];
    else
      v = sup[((long)b * 3 * S + 3 * (f - 1) + c) * HW + hw];
    st1(out + i, v);
  }
}

// dst[p][dst_off + c] (=|+=) src[p][src_off + c], c < Cc ; 4 channels per thread when everything is 4-aligned
template <typename T, int V>
__global__ void copy_channels_kernel(const T* __restrict__ src, T* __restrict__ dst, long P, int Cs, int src_off,
                                     int Cd, int dst_off, int Cc, int accumulate) {
  const int CcV = Cc / V;
  const long total = P * CcV;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % CcV) * V;
    const long p = i / CcV;
    const T* sp = src + p * Cs + src_off + c;
    T* d = dst + p * Cd + dst_off + c;
    if (V == 4) {
      f32x4 v = ld4(sp);
      if (accumulate) v += ld4(d);
      st4(d, v);
    } else {
      const float v = ld1(sp);
      st1(d, accumulate ? ld1(d) + v : v);
    }
  }
}

// torch.cat on channels of up to four tensors in ONE launch (Alignment_V15.py:139,143,160: the head concatenates two to four
// 48-channel maps three times per step, forward and backward, on its one-lane chain -- a launch per source was 5 us each), and the
// reverse: slices of the concatenated gradient (=|+=) into the sources' gradients.  Channel counts are multiples of 4.
struct Cat4 { const void* src[4]; void* dst[4]; int c[4], acc[4]; int n; };
template <typename T, bool SPLIT>
__global__ void cat4_kernel(Cat4 a, const T* __restrict__ whole_src, T* __restrict__ whole_dst, long P, int Ct) {
  const int CtV = Ct >> 2;
  const long total = P * CtV;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % CtV) * 4;
    const long p = i / CtV;
    int k = 0, base = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (j + 1 < a.n && c >= base + a.c[k]) {
        base += a.c[k];
        ++k;
      }
    if (SPLIT) {
      T* d = reinterpret_cast<T*>(a.dst[k]);
      if (!d) continue;
      d += p * a.c[k] + (c - base);
      f32x4 v = ld4(whole_src + p * Ct + c);
      if (a.acc[k]) v += ld4(d);
      st4(d, v);
    } else {
      st4(whole_dst + p * Ct + c, ld4(reinterpret_cast<const T*>(a.src[k]) + p * a.c[k] + (c - base)));
    }
  }
}
template <typename T>
static int cat4_impl(bool split, const void* const* ptrs, const int* cs, const int* accs, int n, const T* whole_src, T* whole_dst, long P,
                     hipStream_t s, const char* nm) {
  FAMI_REQUIRE(ptrs && cs && n >= 1 && n <= 4 && P > 0 && (split ? (const void*)whole_src : (const void*)whole_dst), nm, "bad argument");
  Cat4 a;
  int Ct = 0;
  for (int k = 0; k < 4; ++k) {
    a.src[k] = nullptr; a.dst[k] = nullptr; a.c[k] = 0; a.acc[k] = 0;
    if (k < n) {
      FAMI_REQUIRE(cs[k] > 0 && (cs[k] & 3) == 0 && (split || ptrs[k]) && (reinterpret_cast<uintptr_t>(ptrs[k]) & 7) == 0, nm,
                   "channel counts must be multiples of 4, pointers 8-byte aligned");
      if (split) { a.dst[k] = const_cast<void*>(ptrs[k]); a.acc[k] = accs ? accs[k] : 0; }
      else a.src[k] = ptrs[k];
      a.c[k] = cs[k];
      Ct += cs[k];
    }
  }
  a.n = n;
  if (split) hipLaunchKernelGGL((cat4_kernel<T, true>), dim3(fami_ew_grid(P * Ct / 4)), dim3(256), 0, s, a, whole_src, (T*)nullptr, P, Ct);
  else hipLaunchKernelGGL((cat4_kernel<T, false>), dim3(fami_ew_grid(P * Ct / 4)), dim3(256), 0, s, a, (const T*)nullptr, whole_dst, P, Ct);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}

// out = alpha*a + beta*b (b may be null; out may alias a or b)
// out[i][0..1] (=|+=) in[i][0..1] * (sx, sy): the legacy kornia.warp_affine translation scaling (MODEL.WARP_ALIGN_CORNERS
// False: kornia <= 0.4 normalises the matrix for [0, W-1] but builds its grid with align_corners=False, so a translation
// of t pixels samples at x - t*W/(W-1); Alignment_V15.py:135) and its gradient.
__global__ void scale_pairs_kernel(const float* __restrict__ in, float* __restrict__ out, long n, float sx, float sy, int acc) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= 2 * n) return;
  const float v = in[i] * ((i & 1) ? sy : sx);
  out[i] = acc ? out[i] + v : v;
}

// g *= f over the flat gradient arena (1 / (world * loss_scale)); raises *flag when any element is inf / NaN
__global__ __launch_bounds__(256) void unscale_check_kernel(float* __restrict__ g, long n, float f, unsigned* __restrict__ flag) {
  bool bad = false;
  const long n4 = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = ld4(g + i * 4) * f;
#pragma unroll
    for (int t = 0; t < 4; ++t) bad |= !(fabsf(v[t]) <= 3.4028235e38f);
    st4(g + i * 4, v);
  }
  for (long i = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = g[i] * f;
    bad |= !(fabsf(v) <= 3.4028235e38f);
    g[i] = v;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

template <typename T>
__global__ void axpby_kernel(const T* a, const T* b, T* out, long n, float alpha, float beta, int vec) {
  const long n4 = vec ? n >> 2 : 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = ld4(a + i * 4) * alpha;
    if (b) v += ld4(b + i * 4) * beta;
    st4(out + i * 4, v);
  }
  for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = alpha * ld1(a + i);
    if (b) v += beta * ld1(b + i);
    st1(out + i, v);
  }
}

// dst (=|+=) src : fp32 accumulation buffers folded into T gradients (DCN input gradient)
template <typename T>
__global__ void cast_add_kernel(const float* __restrict__ src, T* dst, long n, int accumulate) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = src[i];
    st1(dst + i, accumulate ? ld1(dst + i) + v : v);
  }
}

template <typename T>
__global__ void fill_kernel(T* out, long n, float v) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) st1(out + i, v);
}

__global__ void incr_i64_kernel(long long* v, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) v[i] += 1;
}
__global__ void add_i64_kernel(long long* v, const long long* inc, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) v[i] += inc[i];
}

template <typename T>
struct FuseArgs {
  const T* x[4];
  const float* mean[4];
  const float* invstd[4];
  const float* gamma[4];
  const float* beta[4];
  int shift[4];
  int nterms;
};

// y[n,h,w,c] = relu( sum_k term_k ), term_k = bn_k(x_k[n, h>>s_k, w>>s_k, c]) (bn_k optional)
template <typename T>
__global__ __launch_bounds__(256) void fuse_sum_kernel(FuseArgs<T> a, T* __restrict__ y, int N, int H, int W, int C,
                                                       int relu) {
  const int CV = C >> 2;
  const long total = (long)N * H * W * CV;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const long n = p / H;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < a.nterms; ++k) {
      const int s = a.shift[k];
      const int Hk = H >> s, Wk = W >> s;
      const long o = ((n * Hk + (h >> s)) * Wk + (w >> s)) * C + cv * 4;
      f32x4 v = ld4(a.x[k] + o);
      if (a.mean[k]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int c = cv * 4 + t;
          const float sc = a.invstd[k][c] * a.gamma[k][c];
          v[t] = v[t] * sc + (a.beta[k][c] - a.mean[k][c] * sc);
        }
      }
      acc += v;
    }
    if (relu) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = fmaxf(acc[t], 0.f);
    }
    st4(y + i * 4, acc);
  }
}

// dx (=|+=) dy * (y > 0)
template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* dx, long n4, int accumulate) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 g = ld4(dy + i * 4);
    const f32x4 yy = ld4(y + i * 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) g[t] = yy[t] > 0.f ? g[t] : 0.f;
    if (accumulate) g += ld4(dx + i * 4);
    st4(dx + i * 4, g);
  }
}

// out[n,hl,wl,c] = sum_{window 2^s x 2^s} dy*(y>0) : gradient of nearest-upsample under the fuse ReLU
template <typename T>
__global__ void pool_relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ out, int N,
                                     int Hl, int Wl, int C, int s, int relu) {
  const int CV = C >> 2;
  const int f = 1 << s;
  const int H = Hl << s, W = Wl << s;
  const long total = (long)N * Hl * Wl * CV;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int wl = (int)(p % Wl);
    p /= Wl;
    const int hl = (int)(p % Hl);
    const long n = p / Hl;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int dh = 0; dh < f; ++dh)
      for (int dw = 0; dw < f; ++dw) {
        const long o = ((n * H + (hl * f + dh)) * W + (wl * f + dw)) * C + cv * 4;
        f32x4 g = ld4(dy + o);
        if (relu) {
          const f32x4 yy = ld4(y + o);
#pragma unroll
          for (int t = 0; t < 4; ++t) g[t] = yy[t] > 0.f ? g[t] : 0.f;
        }
        acc += g;
      }
    st4(out + i * 4, acc);
  }
}

// state = {step, lr, bc1, bc2}
// flag (optional): raised by fami_unscale_check_f32 when the gradient arena holds an inf / NaN (an fp16 overflow under the
// static loss scale).  The step is then SKIPPED: the step count does not advance, bc1 = 0 tells adam_kernel to return, and
// the flag is cleared for the next step -- one overflow no longer poisons m, v and the parameters for good.
__global__ void adam_prep_kernel(float* state, float beta1, float beta2, unsigned* flag) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (flag && *flag) {
      *flag = 0u;
      flag[1] += 1u;       // skipped-step counter (Trainer.skipped_steps): a persistent overflow must be visible, not silent
      state[2] = 0.f;
      return;
    }
    const float t = state[0] + 1.f;
    state[0] = t;
    state[2] = 1.f - powf(beta1, t);
    state[3] = 1.f - powf(beta2, t);
  }
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, const float* __restrict__ state, float beta1,
                            float beta2, float eps, float wd) {
  const float lr = state[1], bc1 = state[2], bc2s = sqrtf(state[3]);
  if (bc1 == 0.f) return;   // skipped step (non-finite gradients, see adam_prep_kernel); bc1 > 0 from step 1 on otherwise
  const float step_size = lr / bc1;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i];
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - step_size * (mi / (sqrtf(vi) / bc2s + eps));
  }
}


// ------------------------------------------------------------------ host side (templates over the storage type)
template <typename T>
static int nchw_to_nhwc_impl(const float* src, T* dst, int N, int C, int H, int W, hipStream_t s, const char* nm) {
  FAMI_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0, nm, "bad argument");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(fami_ew_grid((long)N * C * H * W)), dim3(256), 0, s, src, dst, N, C, H, W);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}
template <typename T>
static int nhwc_to_nchw_impl(const T* src, float* dst, int N, int C, int H, int W, int accumulate, hipStream_t s,
                             const char* nm) {
  FAMI_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0, nm, "bad argument");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(fami_ew_grid((long)N * C * H * W)), dim3(256), 0, s, src, dst, N, C, H, W, accumulate);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}
template <typename T>
static int pack_frames_impl(const float* kf, const float* sup, T* out, int B, int S, int H, int W, hipStream_t s,
                            const char* nm) {
  FAMI_REQUIRE(kf && out && B > 0 && S >= 0 && (S == 0 || sup), nm, "bad argument");
  hipLaunchKernelGGL(pack_frames_kernel<T>, dim3(fami_ew_grid((long)(1 + S) * B * H * W * 3)), dim3(256), 0, s, kf, sup, out, B, S, H, W);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}
template <typename T>
static int copy_channels_impl(const T* src, T* dst, long P, int Cs, int src_off, int Cd, int dst_off, int Cc,
                              int accumulate, hipStream_t s, const char* nm) {
  FAMI_REQUIRE(src && dst && P > 0 && Cc > 0 && src_off >= 0 && dst_off >= 0 && src_off + Cc <= Cs && dst_off + Cc <= Cd, nm, "bad argument");
  const bool v4 = ((Cs | src_off | Cd | dst_off | Cc) & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  if (v4)
    hipLaunchKernelGGL((copy_channels_kernel<T, 4>), dim3(fami_ew_grid(P * Cc / 4)), dim3(256), 0, s, src, dst, P, Cs, src_off, Cd, dst_off, Cc, accumulate);
  else
    hipLaunchKernelGGL((copy_channels_kernel<T, 1>), dim3(fami_ew_grid(P * Cc)), dim3(256), 0, s, src, dst, P, Cs, src_off, Cd, dst_off, Cc, accumulate);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}
template <typename T>
static int axpby_impl(const T* a, const T* b, T* out, long n, float alpha, float beta, hipStream_t s, const char* nm) {
  FAMI_REQUIRE(a && out && n > 0, nm, "bad argument");
  const int vec = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & (4 * sizeof(T) - 1)) == 0;
  hipLaunchKernelGGL(axpby_kernel<T>, dim3(fami_ew_grid(vec ? (n + 3) / 4 : n)), dim3(256), 0, s, a, b, out, n, alpha, beta, vec);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}
// dst[i] = (float)src[i]: the 16-bit gradient payload of the data-parallel exchange widened back into the fp32 arena
template <typename T>
__global__ void widen_kernel(const T* __restrict__ src, float* __restrict__ dst, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = ld1(src + i);
}
template <typename T>
static int widen_impl(const T* src, float* dst, long n, hipStream_t s, const char* nm) {
  FAMI_REQUIRE(src && dst && n > 0, nm, "bad argument");
  hipLaunchKernelGGL(widen_kernel<T>, dim3(fami_ew_grid(n)), dim3(256), 0, s, src, dst, n);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}

template <typename T>
static int cast_add_impl(const float* src, T* dst, long n, int accumulate, hipStream_t s, const char* nm) {
  FAMI_REQUIRE(src && dst && n > 0, nm, "bad argument");
  hipLaunchKernelGGL(cast_add_kernel<T>, dim3(fami_ew_grid(n)), dim3(256), 0, s, src, dst, n, accumulate);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}
template <typename T>
static int fill_impl(T* out, long n, float v, hipStream_t s, const char* nm) {
  FAMI_REQUIRE(out && n > 0, nm, "bad argument");
  hipLaunchKernelGGL(fill_kernel<T>, dim3(fami_ew_grid(n)), dim3(256), 0, s, out, n, v);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}
template <typename T>
static int fuse_sum_impl(int nterms, const T* const* x, const float* const* mean, const float* const* invstd,
                         const float* const* gamma, const float* const* beta, const int* shift, T* y, int N, int H,
                         int W, int C, int relu, hipStream_t s, const char* nm) {
  FAMI_REQUIRE(nterms >= 1 && nterms <= 4 && x && shift && y && (C % 4) == 0, nm, "bad argument");
  FuseArgs<T> a;
  a.nterms = nterms;
  for (int k = 0; k < 4; ++k) {
    const bool on = k < nterms;
    a.x[k] = on ? x[k] : nullptr;
    a.mean[k] = on && mean ? mean[k] : nullptr;
    a.invstd[k] = on && invstd ? invstd[k] : nullptr;
    a.gamma[k] = on && gamma ? gamma[k] : nullptr;
    a.beta[k] = on && beta ? beta[k] : nullptr;
    a.shift[k] = on ? shift[k] : 0;
    if (on) {
      FAMI_REQUIRE(a.x[k] && a.shift[k] >= 0 && (H % (1 << a.shift[k])) == 0 && (W % (1 << a.shift[k])) == 0, nm, "bad term");
    }
  }
  hipLaunchKernelGGL(fuse_sum_kernel<T>, dim3(fami_ew_grid((long)N * H * W * (C / 4))), dim3(256), 0, s, a, y, N, H, W, C, relu);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}
template <typename T>
static int relu_bwd_impl(const T* dy, const T* y, T* dx, long n, int accumulate, hipStream_t s, const char* nm) {
  FAMI_REQUIRE(dy && y && dx && n > 0 && (n % 4) == 0, nm, "bad argument");
  hipLaunchKernelGGL(relu_bwd_kernel<T>, dim3(fami_ew_grid(n / 4)), dim3(256), 0, s, dy, y, dx, n / 4, accumulate);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}
template <typename T>
static int pool_relu_bwd_impl(const T* dy, const T* y, T* out, int N, int Hl, int Wl, int C, int shift, int relu,
                              hipStream_t s, const char* nm) {
  FAMI_REQUIRE(dy && out && (!relu || y) && (C % 4) == 0 && shift >= 0 && shift <= 4, nm, "bad argument");
  hipLaunchKernelGGL(pool_relu_bwd_kernel<T>, dim3(fami_ew_grid((long)N * Hl * Wl * (C / 4))), dim3(256), 0, s, dy, y, out, N, Hl, Wl, C, shift, relu);
  FAMI_CHECK_LAUNCH(nm);
  return FAMI_OK;
}


// ------------------------------------------------------------------ batched small launches (the lane join of the engine)
// Folding the lane-private gradients of a module that ran on several stream lanes (the translation regressor: 37
// parameters x 3 lanes) was 111 axpby launches of 2-3 us back to back on the critical path of the head's backward pass
// (0.8 ms of the bf16 step in the kernel trace).  Up to 32 (a, b, out, n) entries per launch travel as kernel arguments
// (graph-safe, no device descriptor buffer): out = a + b.
#define FAMI_AXPBY_BATCH 32
struct AxpbyBatch { const float* a[FAMI_AXPBY_BATCH]; float* out[FAMI_AXPBY_BATCH]; int n[FAMI_AXPBY_BATCH]; };
__global__ __launch_bounds__(256) void add_batch_kernel(AxpbyBatch b) {
  const float* __restrict__ a = b.a[blockIdx.y];
  float* __restrict__ o = b.out[blockIdx.y];
  const int n = b.n[blockIdx.y];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) o[i] += a[i];
}

extern "C" {

#define FAMI_EW_ABI(sfx, T)                                                                                            \
  int fami_nchw_to_nhwc_##sfx(const float* src, T* dst, int N, int C, int H, int W, hipStream_t s) {                   \
    return nchw_to_nhwc_impl<T>(src, dst, N, C, H, W, s, "fami_nchw_to_nhwc_" #sfx);                                   \
  }                                                                                                                    \
  int fami_nhwc_to_nchw_##sfx(const T* src, float* dst, int N, int C, int H, int W, int accumulate, hipStream_t s) {   \
    return nhwc_to_nchw_impl<T>(src, dst, N, C, H, W, accumulate, s, "fami_nhwc_to_nchw_" #sfx);                       \
  }                                                                                                                    \
  /* kf [B,3,H,W], sup [B,3S,H,W] (NCHW fp32) -> frames [(1+S)*B, H, W, 3] (NHWC), frame-major */                      \
  int fami_pack_frames_##sfx(const float* kf, const float* sup, T* out, int B, int S, int H, int W, hipStream_t s) {   \
    return pack_frames_impl<T>(kf, sup, out, B, S, H, W, s, "fami_pack_frames_" #sfx);                                 \
  }                                                                                                                    \
  int fami_copy_channels_##sfx(const T* src, T* dst, long P, int Cs, int src_off, int Cd, int dst_off, int Cc,         \
                               int accumulate, hipStream_t s) {                                                        \
    return copy_channels_impl<T>(src, dst, P, Cs, src_off, Cd, dst_off, Cc, accumulate, s, "fami_copy_channels_" #sfx);\
  }                                                                                                                    \
  /* dst[P][c0 + .. + c(n-1)] = cat(src[0..n-1][P][c_k]); n <= 4, c_k % 4 == 0 */                                       \
  int fami_concat_channels_##sfx(const T* const* src, const int* c, int n, T* dst, long P, hipStream_t s) {            \
    return cat4_impl<T>(false, reinterpret_cast<const void* const*>(src), c, nullptr, n, (const T*)nullptr, dst, P, s, \
                        "fami_concat_channels_" #sfx);                                                                \
  }                                                                                                                    \
  /* dst[k][P][c_k] (=|+=, accumulate[k]) the k-th channel slice of src[P][sum c]; dst[k] may be null (slice skipped) */ \
  int fami_split_channels_##sfx(const T* src, T* const* dst, const int* c, const int* accumulate, int n, long P,       \
                                hipStream_t s) {                                                                       \
    return cat4_impl<T>(true, reinterpret_cast<const void* const*>(dst), c, accumulate, n, src, (T*)nullptr, P, s,     \
                        "fami_split_channels_" #sfx);                                                                 \
  }                                                                                                                    \
  int fami_axpby_##sfx(const T* a, const T* b, T* out, long n, float alpha, float beta, hipStream_t s) {               \
    return axpby_impl<T>(a, b, out, n, alpha, beta, s, "fami_axpby_" #sfx);                                            \
  }                                                                                                                    \
  int fami_cast_add_##sfx(const float* src, T* dst, long n, int accumulate, hipStream_t s) {                           \
    return cast_add_impl<T>(src, dst, n, accumulate, s, "fami_cast_add_" #sfx);                                        \
  }                                                                                                                    \
  int fami_widen_##sfx(const T* src, float* dst, long n, hipStream_t s) {                                              \
    return widen_impl<T>(src, dst, n, s, "fami_widen_" #sfx);                                                          \
  }                                                                                                                    \
  int fami_fill_##sfx(T* out, long n, float v, hipStream_t s) { return fill_impl<T>(out, n, v, s, "fami_fill_" #sfx); }\
  /* term k: x[k] [N, H>>shift[k], W>>shift[k], C]; mean[k]==null => identity term.  Arrays of length nterms (<=4). */ \
  int fami_fuse_sum_##sfx(int nterms, const T* const* x, const float* const* mean, const float* const* invstd,         \
                          const float* const* gamma, const float* const* beta, const int* shift, T* y, int N, int H,   \
                          int W, int C, int relu, hipStream_t s) {                                                     \
    return fuse_sum_impl<T>(nterms, x, mean, invstd, gamma, beta, shift, y, N, H, W, C, relu, s, "fami_fuse_sum_" #sfx);\
  }                                                                                                                    \
  int fami_relu_bwd_##sfx(const T* dy, const T* y, T* dx, long n, int accumulate, hipStream_t s) {                     \
    return relu_bwd_impl<T>(dy, y, dx, n, accumulate, s, "fami_relu_bwd_" #sfx);                                       \
  }                                                                                                                    \
  /* dy,y [N, Hl<<s, Wl<<s, C] -> out [N,Hl,Wl,C] */                                                                   \
  int fami_pool_relu_bwd_##sfx(const T* dy, const T* y, T* out, int N, int Hl, int Wl, int C, int shift, int relu,     \
                               hipStream_t s) {                                                                        \
    return pool_relu_bwd_impl<T>(dy, y, out, N, Hl, Wl, C, shift, relu, s, "fami_pool_relu_bwd_" #sfx);                \
  }
FAMI_EW_ABI(f32, float)
FAMI_EW_ABI(bf16, bf16_t)
FAMI_EW_ABI(f16, f16_t)
#undef FAMI_EW_ABI

int fami_scale_pairs_f32(const float* in, float* out, long n, float sx, float sy, int accumulate, hipStream_t s) {
  FAMI_REQUIRE(in && out && n > 0, "fami_scale_pairs_f32", "bad argument");
  hipLaunchKernelGGL(scale_pairs_kernel, dim3(fami_cdiv(2 * n, 256)), dim3(256), 0, s, in, out, n, sx, sy, accumulate);
  FAMI_CHECK_LAUNCH("fami_scale_pairs_f32");
  return FAMI_OK;
}

int fami_incr_i64(long long* v, long n, hipStream_t s) {
  FAMI_REQUIRE(v && n > 0, "fami_incr_i64", "bad argument");
  hipLaunchKernelGGL(incr_i64_kernel, dim3(fami_ew_grid(n)), dim3(256), 0, s, v, n);
  FAMI_CHECK_LAUNCH("fami_incr_i64");
  return FAMI_OK;
}

int fami_add_i64(long long* v, const long long* inc, long n, hipStream_t s) {
  FAMI_REQUIRE(v && inc && n > 0, "fami_add_i64", "bad argument");
  hipLaunchKernelGGL(add_i64_kernel, dim3(fami_ew_grid(n)), dim3(256), 0, s, v, inc, n);
  FAMI_CHECK_LAUNCH("fami_add_i64");
  return FAMI_OK;
}

// state (device float[4]) = {step, lr, 1-beta1^step, 1-beta2^step}; prep increments step
int fami_adam_prep_f32(float* state, float beta1, float beta2, hipStream_t s) {
  FAMI_REQUIRE(state, "fami_adam_prep_f32", "bad argument");
  hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, s, state, beta1, beta2, (unsigned*)nullptr);
  FAMI_CHECK_LAUNCH("fami_adam_prep_f32");
  return FAMI_OK;
}
// the same, skipping the step when *flag != 0 (and clearing the flag): see fami_unscale_check_f32
int fami_adam_prep_checked_f32(float* state, float beta1, float beta2, unsigned* flag, hipStream_t s) {
  FAMI_REQUIRE(state && flag, "fami_adam_prep_checked_f32", "bad argument");
  hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, s, state, beta1, beta2, flag);
  FAMI_CHECK_LAUNCH("fami_adam_prep_checked_f32");
  return FAMI_OK;
}
int fami_unscale_check_f32(float* g, long n, float f, unsigned* flag, hipStream_t s) {
  FAMI_REQUIRE(g && flag && n > 0, "fami_unscale_check_f32", "bad argument");
  hipLaunchKernelGGL(unscale_check_kernel, dim3(fami_ew_grid((n + 3) / 4)), dim3(256), 0, s, g, n, f, flag);
  FAMI_CHECK_LAUNCH("fami_unscale_check_f32");
  return FAMI_OK;
}
int fami_adam_f32(float* p, const float* g, float* m, float* v, long n, const float* state, float beta1, float beta2,
                  float eps, float weight_decay, hipStream_t s) {
  FAMI_REQUIRE(p && g && m && v && state && n > 0, "fami_adam_f32", "bad argument");
  hipLaunchKernelGGL(adam_kernel, dim3(fami_ew_grid(n)), dim3(256), 0, s, p, g, m, v, n, state, beta1, beta2, eps, weight_decay);
  FAMI_CHECK_LAUNCH("fami_adam_f32");
  return FAMI_OK;
}


// out[k] += a[k] for n pairs of fp32 tensors: ptrs = host array of 2 n longs (a_0, out_0, a_1, out_1, ...), counts = host
// array of n ints.  One launch per 32 pairs (the arguments travel in the kernarg segment).
int fami_add_batch_f32(const long* ptrs, const int* counts, int n, hipStream_t s) {
  FAMI_REQUIRE(ptrs && counts && n > 0, "fami_add_batch_f32", "bad argument");
  // The pairs of one launch run concurrently (one blockIdx.y each) with a plain read-modify-write, so a launch must not
  // hold the same `out` twice: a repeated output closes the launch and opens the next one (same stream => the adds into one
  // buffer happen in call order, and the result does not depend on how the pairs fall into launches).
  int i0 = 0;
  while (i0 < n) {
    AxpbyBatch b;
    int m = 0, maxn = 1;
    for (; m < FAMI_AXPBY_BATCH && i0 + m < n; ++m) {
      float* o = reinterpret_cast<float*>(ptrs[2 * (i0 + m) + 1]);
      bool repeated = false;
      for (int k = 0; k < m; ++k) repeated |= (b.out[k] == o);
      if (repeated) break;
      b.a[m] = reinterpret_cast<const float*>(ptrs[2 * (i0 + m)]);
      b.out[m] = o;
      b.n[m] = counts[i0 + m];
      if (b.n[m] > maxn) maxn = b.n[m];
    }
    for (int i = m; i < FAMI_AXPBY_BATCH; ++i) { b.a[i] = b.a[0]; b.out[i] = b.out[0]; b.n[i] = 0; }
    int gx = (maxn + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(add_batch_kernel, dim3(gx, m), dim3(256), 0, s, b);
    FAMI_CHECK_LAUNCH("fami_add_batch_f32");
    i0 += m;
  }
  return FAMI_OK;
}

}  // extern "C"
