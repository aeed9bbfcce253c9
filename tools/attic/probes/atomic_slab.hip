// Probe: cost of folding per-workgroup weight-gradient partials (20736 floats each) into S accumulation slabs with
// f32 atomics (agent scope / workgroup scope with XCC-local slabs) versus plain slab stores.  (GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>  // 0 plain store to own slab, 1 agent-scope atomics, 2 workgroup-scope atomics into the XCC's slabs
__global__ __launch_bounds__(256) void k(float* part, int n, int S, int* xcc_seen) {
  int slab;
  if (MODE == 0) slab = blockIdx.x;
  else if (MODE == 1) slab = blockIdx.x % S;
  else {
    const int xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15;  // HW_REG_XCC_ID[3:0]
    if (threadIdx.x == 0) xcc_seen[blockIdx.x] = xcc;
    slab = xcc * (S / 8) + (blockIdx.x / 8) % (S / 8);
  }
  float* dst = part + (long)slab * n;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = 1.0f + (float)(i & 3);
    if (MODE == 0) dst[i] = v;
    else if (MODE == 1) __hip_atomic_fetch_add(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

int main() {
  const int n = 9 * 48 * 48, G = 1024;
  float* part; int* seen;
  CK(hipMalloc(&part, (size_t)G * n * 4));
  CK(hipMalloc(&seen, G * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode)
    for (int S : {8, 64, 256, 1024}) {
      if (mode == 0 && S != 1024) continue;
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(part, 0, (size_t)G * n * 4));
        CK(hipDeviceSynchronize());
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(256), 0, 0, part, n, S, seen);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(G), dim3(256), 0, 0, part, n, S, seen);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(G), dim3(256), 0, 0, part, n, S, seen);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      // check: total over slabs of element 0 must be G * 1.0
      std::vector<float> h((size_t)G * n);
      CK(hipMemcpy(h.data(), part, (size_t)G * n * 4, hipMemcpyDeviceToHost));
      double s0 = 0, s3 = 0;
      for (int sIdx = 0; sIdx < G; ++sIdx) { s0 += h[(size_t)sIdx * n]; s3 += h[(size_t)sIdx * n + 3]; }
      printf("mode %d S=%4d: %.1f us  (sum elem0 = %.0f want %d, elem3 = %.0f want %d)\n", mode, S, best * 1e3, s0, G, s3, 4 * G);
    }
  std::vector<int> hs(G);
  CK(hipMemcpy(hs.data(), seen, G * 4, hipMemcpyDeviceToHost));
  printf("xcc of blocks 0..15:"); for (int i = 0; i < 16; ++i) printf(" %d", hs[i]); printf("\n");
  return 0;
}
