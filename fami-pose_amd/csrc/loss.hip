// On-device targets, losses and decode for the heatmap head.  All tensors here
// are at the NCHW boundary: rows are (b, joint) or (b, channel) with H*W
// contiguous, which is what every reduction below runs over.
//
// Replaces
//   datasets/process/heatmaps_process.py:146-203  generate_heatmaps (Gaussian targets, sigma 3)
//   posetimation/loss/mse_loss.py:21-40           JointMSELoss.forward
//   posetimation/zoo/Alignment/Alignment_V15.py:250-277  the two MI estimators
//       kl_div(input=softmax(A/T) (probabilities!), target=softmax(Bt/T), 'mean'), T = 0.05
//   datasets/process/heatmaps_process.py:16-44    get_max_preds (flat argmax, first max on ties)
#include "common.h"

// joints [B,J,2] (x,y in input-image pixels), vis [B,J] -> target [B,J,Hh,Wh], weight [B,J]
__global__ void gauss_target_kernel(const float* __restrict__ joints, const float* __restrict__ vis,
                                    float* __restrict__ target, float* __restrict__ weight, int B, int J, int Hh,
                                    int Wh, double stride_x, double stride_y, int sigma) {
  const long HW = (long)Hh * Wh;
  const long total = (long)B * J * HW;
  const int r = sigma * 3;
  const float inv = (float)(2 * sigma * sigma);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long bj = i / HW;
    const int hw = (int)(i - bj * HW);
    const int y = hw / Wh, x = hw - y * Wh;
    // int(v + 0.5) of the reference truncates toward zero, in double
    const int mx = (int)((double)joints[bj * 2 + 0] / stride_x + 0.5);
    const int my = (int)((double)joints[bj * 2 + 1] / stride_y + 0.5);
    const int x0 = mx - r, y0 = my - r, x1 = mx + r + 1, y1 = my + r + 1;
    float w = vis[bj];
    const bool outside = x0 >= Wh || y0 >= Hh || x1 < 0 || y1 < 0;
    if (outside) w = 0.f;
    float v = 0.f;
    if (!outside && w > 0.5f && x >= x0 && x < x1 && y >= y0 && y < y1) {
      const int dx = x - mx, dy = y - my;
      v = expf(-((float)(dx * dx + dy * dy)) / inv);
    }
    target[i] = v;
    if (hw == 0) weight[bj] = w;
  }
}

__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ float block_max(float v, float* sm) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// rowsum[r] = sum_p (w_r * (pred - gt))^2
__global__ __launch_bounds__(256) void wmse_rows_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                        const float* __restrict__ w, float* __restrict__ rowsum,
                                                        int L) {
  __shared__ float sm[4];
  const long r = blockIdx.x;
  const float wr = w ? w[r] : 1.f;
  float s = 0.f;
  for (int l = threadIdx.x; l < L; l += 256) {
    const float d = pred[r * L + l] * wr - gt[r * L + l] * wr;
    s += d * d;
  }
  s = block_sum(s, sm);
  if (threadIdx.x == 0) rowsum[r] = s;
}
__global__ void scalar_sum_kernel(const float* __restrict__ rows, int R, double scale, float* out) {
  __shared__ double sm[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < R; i += 256) s += (double)rows[i];
  s = wave_sum_d(s);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sm[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)((sm[0] + sm[1] + sm[2] + sm[3]) * scale);
}
// dpred = g * 2 w^2 (pred - gt) * scale
__global__ void wmse_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                const float* __restrict__ w, float* __restrict__ dpred, long n, int L, float scale,
                                const float* gdev, int accumulate) {
  const float g = gdev ? gdev[0] * scale : scale;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float wr = w ? w[i / L] : 1.f;
    const float v = 2.f * g * wr * (pred[i] * wr - gt[i] * wr);
    dpred[i] = accumulate ? dpred[i] + v : v;
  }
}

// per row: stats[r] = {max_a, sum_a, max_b, sum_b, S = sum_l t_l*(log t_l + 1 - a_l)}, rowval[r] = sum_l t(log t - a)
// REG > 0: the row lives in registers (REG values per thread, L <= 256 * REG): all loads are requested up front and the
// three passes run out of registers.  The streaming form below walked the row three times with one dependent load per
// thread in flight -- 31 us per call for 68-192 rows of 6912 floats, six calls on the head's serial chain (the MI terms).
template <int REG>
__global__ __launch_bounds__(256) void softmax_kl_rows_kernel(const float* __restrict__ A,
                                                              const float* __restrict__ Bt,
                                                              float* __restrict__ stats, float* __restrict__ rowval,
                                                              int L, float temperature) {
  __shared__ float sm[4];
  const long r = blockIdx.x;
  const float* a = A + r * L;
  const float* b = Bt + r * L;
  constexpr int NR = REG > 0 ? REG : 1;
  float av[NR], bv[NR];
  if constexpr (REG > 0) {
#pragma unroll
    for (int j = 0; j < REG; ++j) {
      const int l = threadIdx.x + j * 256;
      av[j] = l < L ? a[l] / temperature : -INFINITY;
      bv[j] = l < L ? b[l] / temperature : -INFINITY;
    }
  }
  float ma = -INFINITY, mb = -INFINITY;
  if constexpr (REG > 0) {
#pragma unroll
    for (int j = 0; j < REG; ++j) {
      ma = fmaxf(ma, av[j]);
      mb = fmaxf(mb, bv[j]);
    }
  } else {
    for (int l = threadIdx.x; l < L; l += 256) {
      ma = fmaxf(ma, a[l] / temperature);
      mb = fmaxf(mb, b[l] / temperature);
    }
  }
  ma = block_max(ma, sm);
  mb = block_max(mb, sm);
  float sa = 0.f, sb = 0.f;
  if constexpr (REG > 0) {
#pragma unroll
    for (int j = 0; j < REG; ++j) {
      if (threadIdx.x + j * 256 < L) {
        sa += expf(av[j] - ma);
        sb += expf(bv[j] - mb);
      }
    }
  } else {
    for (int l = threadIdx.x; l < L; l += 256) {
      sa += expf(a[l] / temperature - ma);
      sb += expf(b[l] / temperature - mb);
    }
  }
  sa = block_sum(sa, sm);
  sb = block_sum(sb, sm);
  float val = 0.f, S = 0.f;
  auto term = [&](float al, float bl) {
    const float pa = expf(al - ma) / sa;
    const float t = expf(bl - mb) / sb;
    if (t > 0.f) {
      const float lt = logf(t);
      val += t * lt - t * pa;
      S += t * (lt + 1.f - pa);
    }
  };
  if constexpr (REG > 0) {
#pragma unroll
    for (int j = 0; j < REG; ++j)
      if (threadIdx.x + j * 256 < L) term(av[j], bv[j]);
  } else {
    for (int l = threadIdx.x; l < L; l += 256) term(a[l] / temperature, b[l] / temperature);
  }
  val = block_sum(val, sm);
  S = block_sum(S, sm);
  if (threadIdx.x == 0) {
    stats[r * 5 + 0] = ma;
    stats[r * 5 + 1] = sa;
    stats[r * 5 + 2] = mb;
    stats[r * 5 + 3] = sb;
    stats[r * 5 + 4] = S;
    rowval[r] = val;
  }
}
// dBt[r,l] (=|+=) g/(R*L*T) * t_l * (log t_l + 1 - a_l - S_r)   (0 where t_l underflows to 0)
template <int REG>
__global__ __launch_bounds__(256) void softmax_kl_bwd_kernel(const float* __restrict__ A,
                                                             const float* __restrict__ Bt,
                                                             const float* __restrict__ stats,
                                                             float* __restrict__ dBt, int L, float temperature,
                                                             float scale, const float* gdev, int accumulate) {
  const long r = blockIdx.x;
  const float g = (gdev ? gdev[0] * scale : scale) / temperature;
  const float ma = stats[r * 5 + 0], sa = stats[r * 5 + 1], mb = stats[r * 5 + 2], sb = stats[r * 5 + 3],
              S = stats[r * 5 + 4];
  auto grad = [&](float al, float bl) {
    const float pa = expf(al / temperature - ma) / sa;
    const float t = expf(bl / temperature - mb) / sb;
    return t > 0.f ? g * t * (logf(t) + 1.f - pa - S) : 0.f;
  };
  if constexpr (REG > 0) {
    float av[REG], bv[REG], dv[REG];
#pragma unroll
    for (int j = 0; j < REG; ++j) {          // every load of the row in flight at once (see softmax_kl_rows_kernel)
      const int l = threadIdx.x + j * 256;
      av[j] = l < L ? A[r * L + l] : 0.f;
      bv[j] = l < L ? Bt[r * L + l] : 0.f;
      dv[j] = (accumulate && l < L) ? dBt[r * L + l] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < REG; ++j) {
      const int l = threadIdx.x + j * 256;
      if (l < L) dBt[r * L + l] = dv[j] + grad(av[j], bv[j]);
    }
  } else {
    for (int l = threadIdx.x; l < L; l += 256) {
      const float v = grad(A[r * L + l], Bt[r * L + l]);
      float* d = dBt + r * L + l;
      *d = accumulate ? *d + v : v;
    }
  }
}

// flat argmax per row, first max on ties
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ hm, long long* __restrict__ idx,
                                                          float* __restrict__ maxval, int L) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const long r = blockIdx.x;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int l = threadIdx.x; l < L; l += 256) {
    const float v = hm[r * L + l];
    if (v > bv) {
      bv = v;
      bi = l;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sv[wave] = bv;
    si[wave] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k)
      if (sv[k] > bv || (sv[k] == bv && si[k] < bi)) {
        bv = sv[k];
        bi = si[k];
      }
    idx[r] = bi == 0x7fffffff ? 0 : bi;
    if (maxval) maxval[r] = bv;
  }
}

// get_final_preds (datasets/process/heatmaps_process.py:47-73): per (b, j) the argmax coordinate (zeroed when the
// maximum is <= 0), the quarter-pixel shift toward the higher neighbour, and transform_preds' inverse affine for
// rot = 0 (affine_transform.py:13-43: uniform scale s = 200*scale[0]/W about (W/2, H/2) -> image centre).
__global__ void final_preds_kernel(const float* __restrict__ hm, const long long* __restrict__ idx,
                                   const float* __restrict__ maxval, const float* __restrict__ center,
                                   const float* __restrict__ scale, float* __restrict__ preds, int B, int J, int H,
                                   int W) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B * J) return;
  const int b = r / J;
  const int i = (int)idx[r];
  float x = (float)(i % W), y = floorf((float)i / (float)W);
  if (!(maxval[r] > 0.f)) x = y = 0.f;
  const int px = (int)floorf(x + 0.5f), py = (int)floorf(y + 0.5f);
  if (1 < px && px < W - 1 && 1 < py && py < H - 1) {
    const float* m = hm + (long)r * H * W;
    const float dx = m[py * W + px + 1] - m[py * W + px - 1];
    const float dy = m[(py + 1) * W + px] - m[(py - 1) * W + px];
    x += (dx > 0.f ? 0.25f : (dx < 0.f ? -0.25f : 0.f));
    y += (dy > 0.f ? 0.25f : (dy < 0.f ? -0.25f : 0.f));
  }
  const float s = scale[b * 2] * 200.f / (float)W;
  preds[r * 2 + 0] = center[b * 2 + 0] + (x - 0.5f * (float)W) * s;
  preds[r * 2 + 1] = center[b * 2 + 1] + (y - 0.5f * (float)H) * s;
}

// PCK on heatmap argmax (engine/core/utils/evaluate.py:13-75 `calc_dists` / `dist_acc` / `accuracy`): one workgroup.
// pidx/tidx, pmax/tmax: argmax rows of the prediction / target stacks [B*J].  get_max_preds zeroes coordinates whose
// maximum is <= 0; a target joint counts only if both of its coordinates are > 1; x is normalised by H/10 and y by
// W/10 (the reference multiplies (x, y) by [h, w]/10 in that order); distances in fp64 as numpy computes them.
// out[0] = mean of the per-joint accuracies that exist (0 if none), out[1..J] = per-joint accuracy or -1,
// out[J+1] = avg_acc (== out[0] when cnt > 0), out[J+2] = cnt.
__global__ __launch_bounds__(256) void pck_kernel(const long long* __restrict__ pidx, const float* __restrict__ pmax,
                                                  const long long* __restrict__ tidx, const float* __restrict__ tmax,
                                                  float* __restrict__ out, int B, int J, int H, int W, double thr) {
  __shared__ int hit[64], val[64];
  for (int j = threadIdx.x; j < J; j += blockDim.x) {
    const double nx = (double)H / 10.0, ny = (double)W / 10.0;
    int h = 0, v = 0;
    for (int b = 0; b < B; ++b) {
      const int r = b * J + j;
      const long long pi = pidx[r], ti = tidx[r];
      const float pm = pmax[r] > 0.f ? 1.f : 0.f, tm = tmax[r] > 0.f ? 1.f : 0.f;
      const float px = (float)(pi % W) * pm, py = (float)(pi / W) * pm;
      const float tx = (float)(ti % W) * tm, ty = (float)(ti / W) * tm;
      if (tx > 1.f && ty > 1.f) {
        const double dx = (double)px / nx - (double)tx / nx, dy = (double)py / ny - (double)ty / ny;
        ++v;
        if (sqrt(dx * dx + dy * dy) < thr) ++h;
      }
    }
    hit[j] = h;
    val[j] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double avg = 0.0;
    int cnt = 0;
    for (int j = 0; j < J; ++j) {
      const double a = val[j] > 0 ? (double)hit[j] * 1.0 / (double)val[j] : -1.0;
      out[1 + j] = (float)a;
      if (a >= 0.0) {
        avg += a;
        ++cnt;
      }
    }
    avg = cnt != 0 ? avg / cnt : 0.0;
    out[0] = cnt != 0 ? (float)avg : 0.f;
    out[J + 1] = (float)avg;
    out[J + 2] = (float)cnt;
  }
}

extern "C" {

// joints [B,J,2] px, vis [B,J] -> target [B,J,Hh,Wh] (NCHW), weight [B,J]
int fami_gauss_target_f32(const float* joints, const float* vis, float* target, float* weight, int B, int J, int Hh,
                          int Wh, int img_h, int img_w, int sigma, hipStream_t s) {
  FAMI_REQUIRE(joints && vis && target && weight && B > 0 && J > 0 && Hh > 0 && Wh > 0 && sigma > 0,
               "fami_gauss_target_f32", "bad argument");
  hipLaunchKernelGGL(gauss_target_kernel, dim3(fami_ew_grid((long)B * J * Hh * Wh)), dim3(256), 0, s, joints, vis,
                     target, weight, B, J, Hh, Wh, (double)img_w / (double)Wh, (double)img_h / (double)Hh, sigma);
  FAMI_CHECK_LAUNCH("fami_gauss_target_f32");
  return FAMI_OK;
}

// loss[0] = scale * sum_{r,l} (w_r (pred-gt))^2 ; pred/gt [R,L], w [R] or null; ws >= R floats
int fami_wmse_fwd_f32(const float* pred, const float* gt, const float* w, float* loss, int R, int L, double scale,
                      float* ws, hipStream_t s) {
  FAMI_REQUIRE(pred && gt && loss && ws && R > 0 && L > 0, "fami_wmse_fwd_f32", "bad argument");
  hipLaunchKernelGGL(wmse_rows_kernel, dim3(R), dim3(256), 0, s, pred, gt, w, ws, L);
  FAMI_CHECK_LAUNCH("fami_wmse_fwd_f32/rows");
  hipLaunchKernelGGL(scalar_sum_kernel, dim3(1), dim3(256), 0, s, ws, R, scale, loss);
  FAMI_CHECK_LAUNCH("fami_wmse_fwd_f32/sum");
  return FAMI_OK;
}
// dpred (=|+=) (gdev? gdev[0] : 1) * scale * 2 w^2 (pred - gt)
int fami_wmse_bwd_f32(const float* pred, const float* gt, const float* w, float* dpred, int R, int L, float scale,
                      const float* gdev, int accumulate, hipStream_t s) {
  FAMI_REQUIRE(pred && gt && dpred && R > 0 && L > 0, "fami_wmse_bwd_f32", "bad argument");
  hipLaunchKernelGGL(wmse_bwd_kernel, dim3(fami_ew_grid((long)R * L)), dim3(256), 0, s, pred, gt, w, dpred, (long)R * L, L, scale, gdev, accumulate);
  FAMI_CHECK_LAUNCH("fami_wmse_bwd_f32");
  return FAMI_OK;
}

// value[0] = mean over R*L of t*(log t - a), a = softmax(A/T) rows, t = softmax(Bt/T) rows.
// stats [R,5] saved for the backward; ws >= R floats.
int fami_softmax_kl_fwd_f32(const float* A, const float* Bt, float* value, float* stats, int R, int L,
                            float temperature, float* ws, hipStream_t s) {
  FAMI_REQUIRE(A && Bt && value && stats && ws && R > 0 && L > 0 && temperature > 0.f, "fami_softmax_kl_fwd_f32", "bad argument");
  if (L <= 256 * 16) hipLaunchKernelGGL(softmax_kl_rows_kernel<16>, dim3(R), dim3(256), 0, s, A, Bt, stats, ws, L, temperature);
  else if (L <= 256 * 32) hipLaunchKernelGGL(softmax_kl_rows_kernel<32>, dim3(R), dim3(256), 0, s, A, Bt, stats, ws, L, temperature);
  else hipLaunchKernelGGL(softmax_kl_rows_kernel<0>, dim3(R), dim3(256), 0, s, A, Bt, stats, ws, L, temperature);
  FAMI_CHECK_LAUNCH("fami_softmax_kl_fwd_f32/rows");
  hipLaunchKernelGGL(scalar_sum_kernel, dim3(1), dim3(256), 0, s, ws, R, 1.0 / ((double)R * (double)L), value);
  FAMI_CHECK_LAUNCH("fami_softmax_kl_fwd_f32/sum");
  return FAMI_OK;
}
// dBt (=|+=) gscale * d value / d Bt (gradient flows through the target only, as in the reference)
int fami_softmax_kl_bwd_f32(const float* A, const float* Bt, const float* stats, float* dBt, int R, int L,
                            float temperature, float gscale, const float* gdev, int accumulate, hipStream_t s) {
  FAMI_REQUIRE(A && Bt && stats && dBt && R > 0 && L > 0, "fami_softmax_kl_bwd_f32", "bad argument");
  const float sc = (float)((double)gscale / ((double)R * (double)L));
  if (L <= 256 * 16) hipLaunchKernelGGL(softmax_kl_bwd_kernel<16>, dim3(R), dim3(256), 0, s, A, Bt, stats, dBt, L, temperature, sc, gdev, accumulate);
  else if (L <= 256 * 32) hipLaunchKernelGGL(softmax_kl_bwd_kernel<32>, dim3(R), dim3(256), 0, s, A, Bt, stats, dBt, L, temperature, sc, gdev, accumulate);
  else hipLaunchKernelGGL(softmax_kl_bwd_kernel<0>, dim3(R), dim3(256), 0, s, A, Bt, stats, dBt, L, temperature, sc, gdev, accumulate);
  FAMI_CHECK_LAUNCH("fami_softmax_kl_bwd_f32");
  return FAMI_OK;
}

// idx[r] = first flat argmax of row r, maxval[r] (optional); hm [R,L]
int fami_argmax2d_f32(const float* hm, long long* idx, float* maxval, int R, int L, hipStream_t s) {
  FAMI_REQUIRE(hm && idx && R > 0 && L > 0, "fami_argmax2d_f32", "bad argument");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(R), dim3(256), 0, s, hm, idx, maxval, L);
  FAMI_CHECK_LAUNCH("fami_argmax2d_f32");
  return FAMI_OK;
}

// hm [B,J,H,W] NCHW, center/scale [B,2] -> preds [B,J,2] image coordinates, maxvals [B,J]; idx_ws: B*J int64 scratch
int fami_final_preds_f32(const float* hm, const float* center, const float* scale, float* preds, float* maxvals,
                         long long* idx_ws, int B, int J, int H, int W, hipStream_t s) {
  FAMI_REQUIRE(hm && center && scale && preds && maxvals && idx_ws && B > 0 && J > 0 && H > 2 && W > 2,
               "fami_final_preds_f32", "bad argument");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(B * J), dim3(256), 0, s, hm, idx_ws, maxvals, H * W);
  FAMI_CHECK_LAUNCH("fami_final_preds_f32/argmax");
  hipLaunchKernelGGL(final_preds_kernel, dim3(fami_cdiv(B * J, 64)), dim3(64), 0, s, hm, idx_ws, maxvals, center, scale,
                     preds, B, J, H, W);
  FAMI_CHECK_LAUNCH("fami_final_preds_f32");
  return FAMI_OK;
}

// PCK accuracy of a heatmap stack against the target stack, entirely on the device (no host sync):
// out[J+3] as documented at pck_kernel; idx_ws: 2*B*J int64, max_ws: 2*B*J floats (scratch)
int fami_pck_accuracy_f32(const float* pred_hm, const float* target_hm, float* out, long long* idx_ws, float* max_ws,
                          int B, int J, int H, int W, float thr, hipStream_t s) {
  FAMI_REQUIRE(pred_hm && target_hm && out && idx_ws && max_ws && B > 0 && J > 0 && J <= 64 && H > 0 && W > 0,
               "fami_pck_accuracy_f32", "bad argument (J <= 64)");
  const int R = B * J;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(R), dim3(256), 0, s, pred_hm, idx_ws, max_ws, H * W);
  FAMI_CHECK_LAUNCH("fami_pck_accuracy_f32/argmax");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(R), dim3(256), 0, s, target_hm, idx_ws + R, max_ws + R, H * W);
  FAMI_CHECK_LAUNCH("fami_pck_accuracy_f32/argmax");
  hipLaunchKernelGGL(pck_kernel, dim3(1), dim3(256), 0, s, idx_ws, max_ws, idx_ws + R, max_ws + R, out, B, J, H, W, (double)thr);
  FAMI_CHECK_LAUNCH("fami_pck_accuracy_f32");
  return FAMI_OK;
}

}  // extern "C"
