// Input pipeline on the device: the per-clip affine crop + ToTensor + Normalize that the
// reference runs on DataLoader workers with cv2 / torchvision:
//   cv2.warpAffine(frame, trans, (W, H), flags=cv2.INTER_LINEAR)   datasets/zoo/posetrack/PoseTrack_Alignment.py:421-427
//     (one transform shared by the key frame and every supporting frame of the clip; optional horizontal flip of the
//      source, :409-413; optional BGR->RGB, :299-300)
//   transforms.ToTensor + transforms.Normalize(mean, std)           datasets/transforms/build.py:12-23
// One thread owns one output pixel of all F frames: the source coordinate is computed once in the fixed-point
// arithmetic of OpenCV's generic warpAffine (10 fractional bits rounded to 1/32 pixel, integer bilinear weights summing
// to 2^15), so the 8-bit crop is bit-identical to the CPU restatement in oracle/ops.py::cv2_warp_affine_u8; the float
// conversion follows torch's operation order ((v / 255) - mean) / std with IEEE fp32 division.
#include "common.h"

struct WarpArgs {
  const unsigned char* src;  // [F][Hs][Ws][3]
  float* out;                // [F][3][Hd][Wd]
  double m[6];               // inverse map (dst -> src), row-major 2x3
  long src_stride;           // bytes between source frames
  long out_stride;           // floats between output frames
  int F, Hs, Ws, Hd, Wd, flip, swap_rb;
  float mean[3], stdv[3];
};

__global__ __launch_bounds__(256) void warp_normalize_kernel(WarpArgs p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.Hd * p.Wd) return;
  const int y = i / p.Wd, x = i - y * p.Wd;
  // saturate_cast<int>(M[0]*x*AB_SCALE) etc.: separate IEEE double operations (no contraction), round half to even
  const int adelta = __double2int_rn(__dmul_rn(__dmul_rn(p.m[0], (double)x), 1024.0));
  const int bdelta = __double2int_rn(__dmul_rn(__dmul_rn(p.m[3], (double)x), 1024.0));
  const int X0 = __double2int_rn(__dmul_rn(__dadd_rn(__dmul_rn(p.m[1], (double)y), p.m[2]), 1024.0)) + 16;
  const int Y0 = __double2int_rn(__dmul_rn(__dadd_rn(__dmul_rn(p.m[4], (double)y), p.m[5]), 1024.0)) + 16;
  const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
  const int sx = min(max(X >> 5, -32768), 32767), sy = min(max(Y >> 5, -32768), 32767);
  const int fx = X & 31, fy = Y & 31;
  const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
  const bool y0 = (unsigned)sy < (unsigned)p.Hs, y1 = (unsigned)(sy + 1) < (unsigned)p.Hs;
  const bool x0 = (unsigned)sx < (unsigned)p.Ws, x1 = (unsigned)(sx + 1) < (unsigned)p.Ws;
  // flip mirrors the source columns
  const int c0 = p.flip ? p.Ws - 1 - sx : sx, c1 = p.flip ? p.Ws - 2 - sx : sx + 1;
  for (int f = 0; f < p.F; ++f) {
    const unsigned char* s = p.src + f * p.src_stride;
    float* o = p.out + f * p.out_stride + (long)y * p.Wd + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int cs = p.swap_rb ? 2 - c : c;  // output channel c reads source channel cs
      int acc = 1 << 14;
      if (y0 && x0) acc += w00 * s[((long)sy * p.Ws + c0) * 3 + cs];
      if (y0 && x1) acc += w01 * s[((long)sy * p.Ws + c1) * 3 + cs];
      if (y1 && x0) acc += w10 * s[((long)(sy + 1) * p.Ws + c0) * 3 + cs];
      if (y1 && x1) acc += w11 * s[((long)(sy + 1) * p.Ws + c1) * 3 + cs];
      const float v = (float)(acc >> 15);
      o[(long)c * p.Hd * p.Wd] = __fdiv_rn(__fsub_rn(__fdiv_rn(v, 255.0f), p.mean[c]), p.stdv[c]);
    }
  }
}

extern "C" {

int fami_warp_normalize_u8(const unsigned char* src, float* out, int F, int Hs, int Ws, long src_stride, int Hd, int Wd,
                           long out_stride, double m00, double m01, double m02, double m10, double m11, double m12,
                           int flip, int swap_rb, float mean0, float mean1, float mean2, float std0, float std1,
                           float std2, hipStream_t s) {
  FAMI_REQUIRE(src && out && F > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "fami_warp_normalize_u8", "bad argument");
  FAMI_REQUIRE(Hs < 32768 && Ws < 32768 && (long)Hd * Wd < (1L << 31), "fami_warp_normalize_u8", "image too large");
  FAMI_REQUIRE(std0 != 0.f && std1 != 0.f && std2 != 0.f, "fami_warp_normalize_u8", "zero std");
  WarpArgs a;
  a.src = src; a.out = out;
  a.m[0] = m00; a.m[1] = m01; a.m[2] = m02; a.m[3] = m10; a.m[4] = m11; a.m[5] = m12;
  a.src_stride = src_stride; a.out_stride = out_stride;
  a.F = F; a.Hs = Hs; a.Ws = Ws; a.Hd = Hd; a.Wd = Wd; a.flip = flip; a.swap_rb = swap_rb;
  a.mean[0] = mean0; a.mean[1] = mean1; a.mean[2] = mean2;
  a.stdv[0] = std0; a.stdv[1] = std1; a.stdv[2] = std2;
  hipLaunchKernelGGL(warp_normalize_kernel, dim3(fami_cdiv((long)Hd * Wd, 256)), dim3(256), 0, s, a);
  FAMI_CHECK_LAUNCH("fami_warp_normalize_u8");
  return FAMI_OK;
}

}  // extern "C"
