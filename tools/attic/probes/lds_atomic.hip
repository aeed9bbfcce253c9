// Probe: throughput of LDS atomics on gfx950 -- ds_add_f32 vs ds_add_u32 vs ds_add_u64 vs plain (racy) LDS
// read-modify-write -- at the access pattern of the DCN backward's privatised input-gradient scatter
// (28 KB region per workgroup, 16 adds per work item at pseudo-random positions).  (GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void lds_add_f32_cas(float* a, float v) {
  unsigned* ua = reinterpret_cast<unsigned*>(a);
  unsigned old = *reinterpret_cast<volatile unsigned*>(ua), assumed;
  do {
    assumed = old;
    old = atomicCAS(ua, assumed, __float_as_uint(__uint_as_float(assumed) + v));
  } while (old != assumed);
}

template <int MODE>  // 5 = f32 add as a compare-and-swap loop; 0 f32 atomic, 1 u32 atomic, 2 u64 atomic, 3 plain f32 RMW (racy, rate reference), 4 f32 atomic with lane-rotated channel
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ unsigned long long reg64[7056];
  float* regf = reinterpret_cast<float*>(reg64);
  unsigned* regu = reinterpret_cast<unsigned*>(reg64);
  for (int i = threadIdx.x; i < 7056; i += 256) reg64[i] = 0;
  __syncthreads();
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  for (int it = 0; it < iters; ++it) {
    h = h * 1664525u + 1013904223u;
    const int pos = (h >> 8) % 420;              // (ry*RW + rx), leaving room for the +1 corners
    const int cl = ((h >> 4) & 3) * 4;
    const int base = pos * 16 + cl;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int cc = MODE == 4 ? ((c + threadIdx.x) & 3) : c;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int a = base + (k & 1) * 16 + (k >> 1) * 21 * 16 + cc;
        if (MODE == 0 || MODE == 4) atomicAdd(regf + a, 1.0f);
        else if (MODE == 5) lds_add_f32_cas(regf + a, 1.0f);
        else if (MODE == 1) atomicAdd(regu + a, 1u);
        else if (MODE == 2) atomicAdd(reg64 + a, 1ull);
        else regf[a] += 1.0f;
      }
    }
  }
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < 7056; i += 256) s += MODE == 2 ? (float)reg64[i] : MODE == 1 ? (float)regu[i] : regf[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int G = 1024, iters = 64;
  float* out;
  CK(hipMalloc(&out, G * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[6] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain f32 RMW (racy)", "ds_add_f32 lane-rotated channel", "f32 add by ds_cmpst loop"};
  for (int mode = 0; mode < 6; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(256), 0, 0, out, iters);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(G), dim3(256), 0, 0, out, iters);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(G), dim3(256), 0, 0, out, iters);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(G), dim3(256), 0, 0, out, iters);
      if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(G), dim3(256), 0, 0, out, iters);
      if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(G), dim3(256), 0, 0, out, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double ops = (double)G * 256 * iters * 16;
    float chk = 0.f; CK(hipMemcpy(&chk, out, 4, hipMemcpyDeviceToHost));
    printf("%-34s %8.1f us  %6.2f lane-ops / clk / CU (2.4 GHz, 256 CUs)\n", names[mode], best * 1e3,
           ops / (best * 1e-3 * 2.4e9 * 256));
  }
  return 0;
}
