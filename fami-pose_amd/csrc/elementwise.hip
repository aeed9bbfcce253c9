// Streaming (HBM-bound) helpers of the hot path: boundary layout converts
// (the drop-in boundary is NCHW, the kernels run NHWC), channel concat / slice,
// axpby, the HRNet cross-resolution fuse (BN-apply + nearest-upsample + sum +
// ReLU in one pass), ReLU-masked gradient pooling, and the fused Adam step.
//
// Replaces torch.cat / torch.chunk / `sup - kf` (Alignment_V15.py:117-125,132,139,143,160),
// HighResolutionModule.forward's fuse loop (hrnet.py:159-168) with
// Interpolate(nearest) (basic_model.py:116-125), and torch.optim.Adam
// (posetimation/optimizer/optimizer.py:66-68; lr 1e-3, betas (0.9,0.999), eps 1e-8).
#include "common.h"

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int H,
                                    int W) {
  const long total = (long)N * C * H * W;
  const long HW = (long)H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long p = i / C;  // n*HW + hw
    const long n = p / HW, hw = p - n * HW;
    dst[i] = src[(n * C + c) * HW + hw];
  }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int H,
                                    int W, int accumulate) {
  const long total = (long)N * C * H * W;
  const long HW = (long)H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long hw = i % HW;
    const long nc = i / HW;
    const long n = nc / C;
    const int c = (int)(nc - n * C);
    const float v = src[(n * HW + hw) * C + c];
    dst[i] = accumulate ? dst[i] + v : v;
  }
}

// frames[(f*B + b), y, x, c] : f = 0 key frame, f >= 1 supporting frame f-1 (channels 3(f-1)..3(f-1)+2 of sup)
__global__ void pack_frames_kernel(const float* __restrict__ kf, const float* __restrict__ sup,
                                   float* __restrict__ out, int B, int S, int H, int W) {
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
    out[i] = v;
  }
}

// dst[p][dst_off + c] (=|+=) src[p][src_off + c], c < Cc
__global__ void copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, long P, int Cs,
                                     int src_off, int Cd, int dst_off, int Cc, int accumulate) {
  const long total = P * Cc;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cc);
    const long p = i / Cc;
    const float v = src[p * Cs + src_off + c];
    float* d = dst + p * Cd + dst_off + c;
    *d = accumulate ? *d + v : v;
  }
}

// out = alpha*a + beta*b (b may be null; out may alias a or b)
__global__ void axpby_kernel(const float* a, const float* b, float* out, long n, float alpha, float beta) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = alpha * a[i];
    if (b) v += beta * b[i];
    out[i] = v;
  }
}

__global__ void fill_kernel(float* out, long n, float v) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = v;
}

__global__ void incr_i64_kernel(long long* v, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) v[i] += 1;
}
__global__ void add_i64_kernel(long long* v, const long long* inc, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) v[i] += inc[i];
}

struct FuseArgs {
  const float* x[4];
  const float* mean[4];
  const float* invstd[4];
  const float* gamma[4];
  const float* beta[4];
  int shift[4];
  int nterms;
};

// y[n,h,w,c] = relu( sum_k term_k ), term_k = bn_k(x_k[n, h>>s_k, w>>s_k, c]) (bn_k optional)
__global__ __launch_bounds__(256) void fuse_sum_kernel(FuseArgs a, float* __restrict__ y, int N, int H, int W,
                                                       int C, int relu) {
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
      f32x4 v = *reinterpret_cast<const f32x4*>(a.x[k] + o);
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
    *reinterpret_cast<f32x4*>(y + i * 4) = acc;
  }
}

// dx (=|+=) dy * (y > 0)
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* dx, long n4,
                                int accumulate) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
    const f32x4 yy = reinterpret_cast<const f32x4*>(y)[i];
#pragma unroll
    for (int t = 0; t < 4; ++t) g[t] = yy[t] > 0.f ? g[t] : 0.f;
    if (accumulate) g += reinterpret_cast<const f32x4*>(dx)[i];
    reinterpret_cast<f32x4*>(dx)[i] = g;
  }
}

// out[n,hl,wl,c] = sum_{window 2^s x 2^s} dy*(y>0) : gradient of nearest-upsample under the fuse ReLU
__global__ void pool_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                     float* __restrict__ out, int N, int Hl, int Wl, int C, int s, int relu) {
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
        f32x4 g = *reinterpret_cast<const f32x4*>(dy + o);
        if (relu) {
          const f32x4 yy = *reinterpret_cast<const f32x4*>(y + o);
#pragma unroll
          for (int t = 0; t < 4; ++t) g[t] = yy[t] > 0.f ? g[t] : 0.f;
        }
        acc += g;
      }
    *reinterpret_cast<f32x4*>(out + i * 4) = acc;
  }
}

// state = {step, lr, bc1, bc2}
__global__ void adam_prep_kernel(float* state, float beta1, float beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
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

extern "C" {

int fami_nchw_to_nhwc_f32(const float* src, float* dst, int N, int C, int H, int W, hipStream_t s) {
  FAMI_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "fami_nchw_to_nhwc_f32", "bad argument");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(fami_ew_grid((long)N * C * H * W)), dim3(256), 0, s, src, dst, N, C, H, W);
  FAMI_CHECK_LAUNCH("fami_nchw_to_nhwc_f32");
  return FAMI_OK;
}
int fami_nhwc_to_nchw_f32(const float* src, float* dst, int N, int C, int H, int W, int accumulate, hipStream_t s) {
  FAMI_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "fami_nhwc_to_nchw_f32", "bad argument");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(fami_ew_grid((long)N * C * H * W)), dim3(256), 0, s, src, dst, N, C, H, W, accumulate);
  FAMI_CHECK_LAUNCH("fami_nhwc_to_nchw_f32");
  return FAMI_OK;
}
// kf [B,3,H,W], sup [B,3S,H,W] (NCHW) -> frames [(1+S)*B, H, W, 3] (NHWC), frame-major
int fami_pack_frames_f32(const float* kf, const float* sup, float* out, int B, int S, int H, int W, hipStream_t s) {
  FAMI_REQUIRE(kf && out && B > 0 && S >= 0 && (S == 0 || sup), "fami_pack_frames_f32", "bad argument");
  hipLaunchKernelGGL(pack_frames_kernel, dim3(fami_ew_grid((long)(1 + S) * B * H * W * 3)), dim3(256), 0, s, kf, sup, out, B, S, H, W);
  FAMI_CHECK_LAUNCH("fami_pack_frames_f32");
  return FAMI_OK;
}
int fami_copy_channels_f32(const float* src, float* dst, long P, int Cs, int src_off, int Cd, int dst_off, int Cc,
                           int accumulate, hipStream_t s) {
  FAMI_REQUIRE(src && dst && P > 0 && Cc > 0 && src_off >= 0 && dst_off >= 0 && src_off + Cc <= Cs && dst_off + Cc <= Cd,
               "fami_copy_channels_f32", "bad argument");
  hipLaunchKernelGGL(copy_channels_kernel, dim3(fami_ew_grid(P * Cc)), dim3(256), 0, s, src, dst, P, Cs, src_off, Cd, dst_off, Cc, accumulate);
  FAMI_CHECK_LAUNCH("fami_copy_channels_f32");
  return FAMI_OK;
}
int fami_axpby_f32(const float* a, const float* b, float* out, long n, float alpha, float beta, hipStream_t s) {
  FAMI_REQUIRE(a && out && n > 0, "fami_axpby_f32", "bad argument");
  hipLaunchKernelGGL(axpby_kernel, dim3(fami_ew_grid(n)), dim3(256), 0, s, a, b, out, n, alpha, beta);
  FAMI_CHECK_LAUNCH("fami_axpby_f32");
  return FAMI_OK;
}
int fami_fill_f32(float* out, long n, float v, hipStream_t s) {
  FAMI_REQUIRE(out && n > 0, "fami_fill_f32", "bad argument");
  hipLaunchKernelGGL(fill_kernel, dim3(fami_ew_grid(n)), dim3(256), 0, s, out, n, v);
  FAMI_CHECK_LAUNCH("fami_fill_f32");
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

// term k: x[k] [N, H>>shift[k], W>>shift[k], C]; mean[k]==null => identity term.  Arrays of length nterms (<=4).
int fami_fuse_sum_f32(int nterms, const float* const* x, const float* const* mean, const float* const* invstd,
                      const float* const* gamma, const float* const* beta, const int* shift, float* y, int N, int H,
                      int W, int C, int relu, hipStream_t s) {
  FAMI_REQUIRE(nterms >= 1 && nterms <= 4 && x && shift && y && (C % 4) == 0, "fami_fuse_sum_f32", "bad argument");
  FuseArgs a;
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
      FAMI_REQUIRE(a.x[k] && a.shift[k] >= 0 && (H % (1 << a.shift[k])) == 0 && (W % (1 << a.shift[k])) == 0,
                   "fami_fuse_sum_f32", "bad term");
    }
  }
  hipLaunchKernelGGL(fuse_sum_kernel, dim3(fami_ew_grid((long)N * H * W * (C / 4))), dim3(256), 0, s, a, y, N, H, W, C, relu);
  FAMI_CHECK_LAUNCH("fami_fuse_sum_f32");
  return FAMI_OK;
}
int fami_relu_bwd_f32(const float* dy, const float* y, float* dx, long n, int accumulate, hipStream_t s) {
  FAMI_REQUIRE(dy && y && dx && n > 0 && (n % 4) == 0, "fami_relu_bwd_f32", "bad argument");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(fami_ew_grid(n / 4)), dim3(256), 0, s, dy, y, dx, n / 4, accumulate);
  FAMI_CHECK_LAUNCH("fami_relu_bwd_f32");
  return FAMI_OK;
}
// dy,y [N, Hl<<s, Wl<<s, C] -> out [N,Hl,Wl,C]
int fami_pool_relu_bwd_f32(const float* dy, const float* y, float* out, int N, int Hl, int Wl, int C, int shift,
                           int relu, hipStream_t s) {
  FAMI_REQUIRE(dy && out && (!relu || y) && (C % 4) == 0 && shift >= 0 && shift <= 4, "fami_pool_relu_bwd_f32", "bad argument");
  hipLaunchKernelGGL(pool_relu_bwd_kernel, dim3(fami_ew_grid((long)N * Hl * Wl * (C / 4))), dim3(256), 0, s, dy, y, out, N, Hl, Wl, C, shift, relu);
  FAMI_CHECK_LAUNCH("fami_pool_relu_bwd_f32");
  return FAMI_OK;
}
// state (device float[4]) = {step, lr, 1-beta1^step, 1-beta2^step}; prep increments step
int fami_adam_prep_f32(float* state, float beta1, float beta2, hipStream_t s) {
  FAMI_REQUIRE(state, "fami_adam_prep_f32", "bad argument");
  hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, s, state, beta1, beta2);
  FAMI_CHECK_LAUNCH("fami_adam_prep_f32");
  return FAMI_OK;
}
int fami_adam_f32(float* p, const float* g, float* m, float* v, long n, const float* state, float beta1, float beta2,
                  float eps, float weight_decay, hipStream_t s) {
  FAMI_REQUIRE(p && g && m && v && state && n > 0, "fami_adam_f32", "bad argument");
  hipLaunchKernelGGL(adam_kernel, dim3(fami_ew_grid(n)), dim3(256), 0, s, p, g, m, v, n, state, beta1, beta2, eps, weight_decay);
  FAMI_CHECK_LAUNCH("fami_adam_f32");
  return FAMI_OK;
}

}  // extern "C"
