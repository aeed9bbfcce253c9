// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (run by hand): prints, per lane, the 4 LDS element indices it
// receives when lane l passes the address of elements [l*4 .. l*4+3].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out) {
  __shared__ unsigned short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + threadIdx.x * 4));
  for (int k = 0; k < 4; ++k) out[threadIdx.x * 4 + k] = (unsigned short)v[k];
}
int main() {
  unsigned short* d;
  hipMalloc(&d, 512);
  probe<<<1, 64>>>(d);
  unsigned short h[256];
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
