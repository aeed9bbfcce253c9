// Library core: version / error reporting / device info, and the three tiny
// dense layers of the global-translation regressor (nn.Linear 16*h5*w5->64->64->2,
// Alignment_V15.py:69-71, no activations in between).  M = batch (<= a few
// dozen rows): a thread per output element is the right size; no MFMA.
#include "common.h"
#include <string.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

extern "C" void fami_set_error(const char* where, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where ? where : "?", what ? what : "?");
}

// ---- routing state (include/fami_route.h): one process default + the route each thread has bound
static fami_route_t make_default_route() {
  fami_route_t q;
  fami_route_set_defaults(&q);
  return q;
}
static fami_route_t g_process_route = make_default_route();
static thread_local fami_route_t* t_bound_route = nullptr;
extern "C" fami_route_t* fami_rt(void) { return t_bound_route ? t_bound_route : &g_process_route; }
extern "C" long fami_route_size(void) { return (long)sizeof(fami_route_t); }
// library defaults; the f32 arithmetic choice (split products on the bf16 matrix pipe or the exact-f32 MFMA) is the PROCESS
// default's -- what FAMI_F32_SPLIT / fami_tune_defaults stored there
extern "C" int fami_route_init(fami_route_t* r) {
  FAMI_REQUIRE(r, "fami_route_init", "null route");
  fami_route_set_defaults(r);
  r->s3_default = g_process_route.s3_default;
  r->use_t4_s3 = r->s3_default;
  r->wgs3_default = g_process_route.wgs3_default;
  r->wgs3 = r->wgs3_default;
  return FAMI_OK;
}
// entry points called by THIS thread route by *r from now on (r stays owned by the caller and must outlive the binding);
// NULL: back to the process default
extern "C" int fami_route_bind(fami_route_t* r) {
  if (r && r->size != (int)sizeof(fami_route_t)) {
    fami_set_error("fami_route_bind", "route not initialised by fami_route_init of this library (size mismatch)");
    return FAMI_EARG;
  }
  t_bound_route = r;
  return FAMI_OK;
}

// One WAVE per output element (lanes stride K, coalesced rows of x and w, wave reduction): a thread per output walked K = 144
// dependent global loads one at a time -- 52 us for the regressor's first layer (M = 4), on the head's serial chain.
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, int M, int K, int N) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= M * N) return;
  const int m = i / N, n = i - m * N;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += x[m * K + k] * w[n * K + k];
  s = wave_sum(s);
  if (lane == 0) y[i] = s + (b ? b[n] : 0.f);
}
// a thread per dx element (k fastest: rows of w coalesced); eight loads in flight per step instead of one
__global__ void linear_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* dx, int M,
                                    int K, int N, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * K) return;
  const int m = i / K, k = i - m * K;
  float s = 0.f;
  int n = 0;
  for (; n + 8 <= N; n += 8) {
    float wv[8], dv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      wv[j] = w[(n + j) * K + k];
      dv[j] = dy[m * N + n + j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += dv[j] * wv[j];
  }
  for (; n < N; ++n) s += dy[m * N + n] * w[n * K + k];
  dx[i] = accumulate ? dx[i] + s : s;
}
__global__ void linear_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* dw, float* db,
                                    int M, int K, int N, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * (K + 1)) return;
  const int n = i / (K + 1), k = i - n * (K + 1);
  float s = 0.f;
  if (k < K) {
    for (int m = 0; m < M; ++m) s += dy[m * N + n] * x[m * K + k];
    dw[n * K + k] = accumulate ? dw[n * K + k] + s : s;
  } else if (db) {
    for (int m = 0; m < M; ++m) s += dy[m * N + n];
    db[n] = accumulate ? db[n] + s : s;
  }
}

extern "C" {

const char* fami_version(void) { return "fami-pose_amd 0.1 (gfx950)"; }
const char* fami_last_error(void) { return g_err; }

// info[0]=CU count, [1]=wavefront size, [2]=LDS bytes per workgroup, [3]=clock kHz; name optional
int fami_device_info(int device, int* info, char* name, int name_len) {
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, device);
  if (e != hipSuccess) {
    fami_set_error("fami_device_info", hipGetErrorString(e));
    return FAMI_EHIP;
  }
  if (info) {
    info[0] = p.multiProcessorCount;
    info[1] = p.warpSize;
    info[2] = (int)p.sharedMemPerBlock;
    info[3] = p.clockRate;
  }
  if (name && name_len > 0) {
    strncpy(name, p.gcnArchName, name_len - 1);
    name[name_len - 1] = 0;
  }
  return FAMI_OK;
}

// y[M,N] = x[M,K] w[N,K]^T + b
int fami_linear_fwd_f32(const float* x, const float* w, const float* b, float* y, int M, int K, int N,
                        hipStream_t s) {
  FAMI_REQUIRE(x && w && y && M > 0 && K > 0 && N > 0, "fami_linear_fwd_f32", "bad argument");
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(fami_cdiv((long)M * N, 4)), dim3(256), 0, s, x, w, b, y, M, K, N);
  FAMI_CHECK_LAUNCH("fami_linear_fwd_f32");
  return FAMI_OK;
}
// dx (=|+=) dy w ; dw (=|+=) dy^T x ; db (=|+=) sum_m dy.  Any of dx / (dw,db) may be null.
int fami_linear_bwd_f32(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int M,
                        int K, int N, int acc_dx, int acc_param, hipStream_t s) {
  FAMI_REQUIRE(dy && x && w && M > 0 && K > 0 && N > 0, "fami_linear_bwd_f32", "bad argument");
  if (dx) {
    hipLaunchKernelGGL(linear_bwd_x_kernel, dim3(fami_cdiv((long)M * K, 64)), dim3(64), 0, s, dy, w, dx, M, K, N, acc_dx);
    FAMI_CHECK_LAUNCH("fami_linear_bwd_f32/dx");
  }
  if (dw) {
    hipLaunchKernelGGL(linear_bwd_w_kernel, dim3(fami_cdiv((long)N * (K + 1), 64)), dim3(64), 0, s, dy, x, dw, db, M, K, N, acc_param);
    FAMI_CHECK_LAUNCH("fami_linear_bwd_f32/dw");
  }
  return FAMI_OK;
}

}  // extern "C"
