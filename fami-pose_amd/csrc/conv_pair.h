// Horizontal fusion of a convolution's two backward launches (round 6).
// The input gradient (conv3x3_t6 / t7 kernel on dY) and the weight gradient (conv_wgrad6_kernel on X and dY) of one 3x3
// stride-1 convolution depend on the same tensor and on nothing of each other: the weight gradient is a LEAF of the backward
// graph, yet enqueued on its branch's stream lane it delays the chain (next BatchNorm backward) by its whole duration, and
// every attempt to move it to a stream of its own lost to the graph executor's scheduling (docs/HISTORY.md, round-5 log).
// fami_conv2d_bwd_pair_* runs both as ONE launch: the first workgroups of the grid execute the weight-gradient body, the rest the
// input-gradient body (conv_pair.hip).  The two bodies are the single kernels' code, so results are bitwise those of the
// two-launch form.
//
// Mechanism: while a capture is installed on the calling thread (fami_pair_capture() != null) the launch sites of those kernels
// record their argument struct, grid and template instance here instead of launching; the pair entry point then launches the
// combined instance, or -- when only one half was recorded or no combined instance exists -- re-runs the recorded half as the
// single kernel it would have been.
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>

struct PairHalf {
  int kind;        // 0: nothing recorded; 6: conv3x3_t6_kernel; 7: conv3x3_t7_kernel; 16: conv_wgrad6_kernel
  int half_kind;   // 0 bf16, 1 fp16
  int v[6];        // template arguments of the instance (t6 / t7: SG, NT, MT, EX, ACC, EM; wg6: KS, CIT, COT, XJ)
  unsigned gx, gy; // grid of the single kernel
  size_t lds;      // dynamic LDS bytes
  int slabs;       // wg6: partial slabs the launch writes (the caller's reduce needs it)
  alignas(16) unsigned char args[384];
};
struct PairCapture {
  PairHalf a, b;   // a: input gradient, b: weight gradient
};
PairCapture*& fami_pair_capture();      // thread-local (conv_pair.hip)

template <typename Args>
static inline void pair_record(PairHalf& h, int kind, int half_kind, const Args& a, dim3 grid, size_t lds, int v0, int v1, int v2, int v3,
                               int v4 = 0, int v5 = 0) {
  static_assert(sizeof(Args) <= sizeof(h.args), "argument struct does not fit the capture");
  h.kind = kind; h.half_kind = half_kind;
  h.v[0] = v0; h.v[1] = v1; h.v[2] = v2; h.v[3] = v3; h.v[4] = v4; h.v[5] = v5;
  h.gx = grid.x; h.gy = grid.y; h.lds = lds; h.slabs = 0;
  memcpy(h.args, &a, sizeof(Args));
}
// f32 storage (split-product kernels): only the persistent kernel's launches (conv_t5.hip: the 96x72 and 48x36 maps) are combined with
// their weight gradient -- f32 step 44.54 -> 44.07 ms (tools/ab_env.py, one box).  With the band kernel's launches of the 24x18 / 12x9 maps
// combined as well the gain was gone (44.35 -> 44.27 on another box): those launches are 160 long workgroups that run alone at the end
// of a module, and 160 + 192 weight-gradient workgroups are two rounds of the chip.
