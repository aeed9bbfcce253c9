// Probe (GPU box): accuracy of an f32 dot product emulated on the bf16 matrix pipe by splitting each f32 operand into
// three bf16 terms (x = x0 + x1 + x2 exactly) and issuing 3 / 6 / 9 v_mfma_f32_16x16x32_bf16 products, against the exact-f32
// v_mfma_f32_16x16x4_f32 chain and an fp64 reference.   hipcc --offload-arch=gfx950 -O3 split_mfma.hip -o split_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x;
  const float r = x - (float)a;
  b = (__bf16)r;
  c = (__bf16)(r - (float)b);
}
// A [16][K] row-major, B [16][K] (column n of the product = row n), D[mode][16][16]
__global__ void probe(const float* A, const float* B, float* D, int K) {
  const int lane = threadIdx.x, rc = lane & 15, kq = lane >> 4;
  f32x4 d4 = {0, 0, 0, 0}, d3 = d4, d6 = d4, d9 = d4, d6s_hi = d4, d6s_lo = d4;
  for (int k0 = 0; k0 < K; k0 += 32) {
    bf16x8 a[3], b[3];
    for (int j = 0; j < 8; ++j) {
      __bf16 t0, t1, t2;
      split3(A[rc * K + k0 + kq * 8 + j], t0, t1, t2); a[0][j] = t0; a[1][j] = t1; a[2][j] = t2;
      split3(B[rc * K + k0 + kq * 8 + j], t0, t1, t2); b[0][j] = t0; b[1][j] = t1; b[2][j] = t2;
    }
    auto mm = [&](int i, int j, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], c, 0, 0, 0); };
    // 3 products (two-term split)
    d3 = mm(1, 0, d3); d3 = mm(0, 1, d3); d3 = mm(0, 0, d3);
    // 6 products, small terms first, one accumulator
    d6 = mm(2, 0, d6); d6 = mm(1, 1, d6); d6 = mm(0, 2, d6); d6 = mm(1, 0, d6); d6 = mm(0, 1, d6); d6 = mm(0, 0, d6);
    // 6 products, low-order terms in their own accumulator
    d6s_lo = mm(2, 0, d6s_lo); d6s_lo = mm(1, 1, d6s_lo); d6s_lo = mm(0, 2, d6s_lo); d6s_lo = mm(1, 0, d6s_lo); d6s_lo = mm(0, 1, d6s_lo);
    d6s_hi = mm(0, 0, d6s_hi);
    // 9 products
    d9 = mm(2, 2, d9); d9 = mm(2, 1, d9); d9 = mm(1, 2, d9); d9 = mm(2, 0, d9); d9 = mm(1, 1, d9); d9 = mm(0, 2, d9);
    d9 = mm(1, 0, d9); d9 = mm(0, 1, d9); d9 = mm(0, 0, d9);
    for (int s = 0; s < 8; ++s) {
      const float av = A[rc * K + k0 + s * 4 + kq], bv = B[rc * K + k0 + s * 4 + kq];
      d4 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, d4, 0, 0, 0);
    }
  }
  for (int r = 0; r < 4; ++r) {
    const int row = kq * 4 + r, col = rc;      // D[row][col] = sum_k A[row][k] * B[col][k]
    D[0 * 256 + row * 16 + col] = d4[r];
    D[1 * 256 + row * 16 + col] = d3[r];
    D[2 * 256 + row * 16 + col] = d6[r];
    D[3 * 256 + row * 16 + col] = d6s_hi[r] + d6s_lo[r];
    D[4 * 256 + row * 16 + col] = d9[r];
  }
}
int main() {
  const char* nm[5] = {"f32 mfma 16x16x4", "bf16 x3", "bf16 x6", "bf16 x6 (2 acc)", "bf16 x9"};
  for (int K : {448, 3456}) {
    for (int trial = 0; trial < 2; ++trial) {
      std::mt19937 g(K + trial);
      std::normal_distribution<float> na(trial ? 0.f : 0.5f, 1.f), nb(0.f, 0.05f);
      std::vector<float> A(16 * K), B(16 * K), D(5 * 256);
      for (auto& v : A) v = na(g);
      for (auto& v : B) v = nb(g);
      float *dA, *dB, *dD;
      hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
      hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
      double ref[256], refabs = 0, seq_err = 0;
      for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
        double s = 0; float fs = 0.f;
        for (int k = 0; k < K; ++k) { s += (double)A[r * K + k] * (double)B[c * K + k]; fs = fmaf(A[r * K + k], B[c * K + k], fs); }
        ref[r * 16 + c] = s; refabs = fmax(refabs, fabs(s)); seq_err = fmax(seq_err, fabs((double)fs - s));
      }
      printf("K=%d %s activations: max|ref| %.3f; sequential f32 fma chain: max err %.3e (%.2e of max)\n", K, trial ? "zero-mean" : "mean 0.5", refabs, seq_err, seq_err / refabs);
      for (int m = 0; m < 5; ++m) {
        double e = 0, q = 0;
        for (int i = 0; i < 256; ++i) { const double d = (double)D[m * 256 + i] - ref[i]; e = fmax(e, fabs(d)); q += d * d; }
        printf("   %-18s max err %.3e (%.2e of max)  rms %.3e\n", nm[m], e, e / refabs, sqrt(q / 256));
      }
    }
  }
  return 0;
}
