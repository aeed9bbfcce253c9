/* fami.h -- C ABI of libfami_hip.so: the MI355X (gfx950) kernels behind the
 * FAMI-Pose temporal-alignment training hot path.
 *
 * The reference has NO native FFI on this path: its compute is
 * reached through torch.nn modules, torchvision's registered op
 * torchvision::deform_conv2d and kornia.geometry.warp_affine.  Each entry point
 * below therefore names the reference *call site / module* it replaces
 * (file:line relative to the reference tree).  A reference maintainer binds
 * them with ctypes; fami-pose_amd/_lib.py is that binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (torch tensors'
 *    data_ptr()); the library never allocates, frees or synchronises;
 *  - activations are fp32 NHWC ("[N,H,W,C]") unless stated; tensors at the
 *    drop-in boundary (network input, heatmaps, targets) are NCHW;
 *  - `stream` is a hipStream_t; every call only enqueues kernels on it and is
 *    therefore hipGraph-capturable;
 *  - `accumulate`/`acc_*` = 0: overwrite the output, 1: add into it;
 *  - returns 0, or FAMI_EARG (-1) bad argument, FAMI_ESHAPE (-2) unsupported
 *    shape, FAMI_EHIP (-3) HIP error; fami_last_error() has the text
 *    (thread-local);
 *  - workspaces: size from the matching fami_*_workspace() query (bytes).
 */
#ifndef FAMI_H
#define FAMI_H
#ifdef __cplusplus
extern "C" {
#endif

typedef void* fami_stream_t; /* hipStream_t */

/* ---- routing state: which kernel form an entry point dispatches to ---------------------------------------------------
 * "no global mutable state besides a kernel cache => re-entrant per stream".  Every such switch is a NAMED FIELD
 * of fami_route_t (fami_route.h documents each: default, meaning, the measurement behind it).  A caller that wants routes of its
 * own (the Python Engine does) keeps a fami_route_t, initialises it with fami_route_init -- library defaults, f32 arithmetic as
 * the process default has it -- edits fields, and binds it to its thread with fami_route_bind before enqueueing work; the
 * binding is thread-local, so engines with different routes interleave in one process and threads do not see each other's.
 * With nothing bound, entry points use the process default route.  The fami_*_tune* / fami_tune_reset functions further down
 * are SHIMS kept for tests and benchmarks: they write the corresponding fields of whichever route the calling thread has bound
 * (or of the process default); the code numbers in their comments are the shim's encoding, not the interface. */
#include "fami_route.h"
long fami_route_size(void);                 /* sizeof(fami_route_t) of this build */
int fami_route_init(fami_route_t* route);   /* library defaults */
int fami_route_bind(fami_route_t* route);   /* this thread's calls route by *route until rebound; NULL = process default */

const char* fami_version(void);
const char* fami_last_error(void);
/* info[0]=CUs, [1]=wavefront, [2]=LDS bytes/workgroup, [3]=clock kHz */
int fami_device_info(int device, int* info, char* name, int name_len);

/* ---- convolution family: nn.Conv2d fwd + autograd --------------------------------------
 * replaces every nn.Conv2d of posetimation/backbones/hrnet.py:569-629 (HRNet-W48: 293 convs),
 * posetimation/layers/basic_model.py:21-23,38-41,74-77 and basic_layer.py:18-19, and the 3x3
 * dilation-3 offset/mask predictors of posetimation/zoo/Alignment/Alignment_V15.py:79-100.
 * Weights are consumed in a fragment-packed image (mode 0 = forward, 1 = dgrad). */
/* benchmarks only: force the implicit-GEMM tile (MT x NT 16x16 tiles per wave, KS-way split-K); 0 = heuristic */
int fami_conv_tune(int mt, int nt, int ks);
int fami_conv_tune_lds(int on);          /* 1 / 0 = route eligible 3x3 stride-1 convs through the LDS-staged / direct kernels,
                                          * -1 = defaults: every storage type on the register-blocked LDS kernel (conv_t4.hip), f32 through
                                          * its split-product instance (operands split exactly into three bf16 terms, six products on the
                                          * bf16 matrix pipe, fp32 accumulation).  Benchmarks / tests: 10 / 11 register-blocked kernel off /
                                          * on; 20 / 21 its exact-f32-MFMA instance; 30 / 31 the split-product instance off / on (30 = exact
                                          * f32 MFMA kernels); 52 / 53 two / three pixel tiles per wave of it; 60 / 61 / 62 its producer-
                                          * consumer form off / one-band frames / always; 100 + bt tiles per band; 2000 + n only launches of
                                          * >= n workgroups; 102000 / 102001 CU-filling band rule off / on */
int fami_conv_tune_wgrad_lds(int on);    /* 0 = weight gradients on the scalar-operand kernels, 1 = LDS-staged kernels wherever eligible,
                                          * 1 + n = the same with n sub-chunks per workgroup of the 16-bit kernel (default 2),
                                          * -1 = defaults (16-bit storage: LDS-staged; f32: linear-address per-tap kernel on stride-1
                                          * same-size convs, general per-tap kernel elsewhere, LDS-staged kernel opt-in);
                                          * 50 / 51 = f32 per-tap kernel in its general / linear-address form (52 / 53 / 54: the latter in
                                          * 9- / 8- / 16-wave workgroups); 100 + mt, 1000 + n = benchmarks (tile cap, workgroup target);
                                          * 20000 + x: the pipelined 16-bit kernel (conv_wg16.hip): 0 / 1 off / on, 2 / 3 3x3-stride-1 only /
                                          * every centred geometry, 100 + bt, 1000 + target; 30000 + x: the f32 split-product kernel
                                          * (conv_wgs3.hip): 0 / 1 off / on (0 = exact-f32 MFMA kernels), 100 + bt, 1000 + target */
int fami_conv_tune_xcd(int mode);      /* benchmarks: bit 0 = XCD-contiguous workgroup->tile order in the implicit-GEMM kernels,
                                        * bit 1 = in the weight-gradient kernels; -1 = default (both on) */
int fami_conv_tune_stages(int stages); /* benchmarks only: register-pipeline depth 2..4, 0 = default;
                                        * 100 / 101 = linear-address form of the f32 implicit GEMM off / on (default on);
                                        * 110 / 111 = stride-2 input gradient: all taps / the pixel's parity class only (default) */
/* Every field of the calling thread's route (bound or process default) back to its default (tests: autouse fixture).  The
 * fields only select between kernels that compute the same function. */
int fami_tune_reset(void);
/* The library's default f32 arithmetic for 3x3 stride-1 convolutions (1 = split products on the bf16 matrix pipe,
 * 0 = exact-f32 MFMA, < 0 = keep) -- stored as the default that fami_tune_reset / fami_conv_tune_lds(-1) restore --
 * then fami_tune_reset.  _lib.py calls it once at load with FAMI_F32_SPLIT. */
int fami_tune_defaults(int f32_split);
/* would a 3x3 stride-1 pad-1 f32 convolution [N,H,W,Ci] -> Co take the persistent split-product kernel (conv_t5.hip)? (tests) */
int fami_conv_t5_eligible(int N, int H, int W, int Ci, int Co);
/* ... and the LDS-DMA-staged kernels of conv_t6.hip (round 4: 16-bit storage; 48 input channels: whole weight image LDS-resident,
 * patch and weights copied by LDS DMA -- fami_conv_tune_lds(8000 / 8001) off / on, 8100 + rows per band, 8201 / 8202 units of two /
 * four rows, 8400 + minimum jobs; 96 / 192 / 384 input channels in phases of 48 -- 8500 / 8501 off / on, 8600 + rows per band,
 * 8700 + workgroups per output-channel block)?  Returns 1 (the 48-channel kernel), 2 (the phased kernel) or 0.
 * fami_conv_tune_lds(9000 / 9001): the stem's dense-K 3 -> 64 stride-2 forward kernel (conv_stem.hip, every storage type) off / on.
 * Weight gradients: fami_conv_tune_wgrad_lds(23000 / 23001) the DMA-staged 3x3
 * kernel off / on (23002 / 23003 its 64-channel blocks, 23004 / 23005 stride 2, 23008 / 23009 the dilated 48 -> 216 / 108
 * predictors of the alignment head, 23100 + units per workgroup, 23400 + workgroup target), 24000 / 24001 the
 * DMA-staged wide 1x1 kernel off / on (24100 + workgroup target), 25000 / 25001 the stem's 3 -> 64 weight-gradient kernel. */
int fami_conv_t6_eligible(int N, int H, int W, int Ci, int Co);
long fami_packed_weight_elems(int Co, int Ci, int kh, int kw, int mode);
int fami_pack_conv_weight_f32(const float* w_oihw, float* wp, int Co, int Ci, int kh, int kw, int mode,
                              fami_stream_t stream);
/* all weight images of a step in one launch: desc = device array of n 32-byte records
 * {long src_elem_off (in params), long dst_elem_off (in packed), int Co, int Ci, int taps, int mode} */
int fami_pack_conv_weights_batch_f32(const float* params, float* packed, const void* desc, int n, fami_stream_t stream);
/* y[N,Ho,Wo,Co] (=|+=) relu?( conv(x[N,H,W,Ci]) + bias + addend ) ; bias/addend may be NULL */
int fami_conv2d_fwd_f32(const float* x, const float* wp, const float* bias, const float* addend, float* y, int N,
                        int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int relu,
                        int accumulate, fami_stream_t stream);
/* dx[N,H,W,Ci] (=|+=) conv_transpose(dy[N,Ho,Wo,Co]) + addend */
int fami_conv2d_dgrad_f32(const float* dy, const float* wp, const float* addend, float* dx, int N, int H, int W,
                          int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                          fami_stream_t stream);
/* Convolution with the statistics pass of the train-mode BatchNorm that follows it (basic_model.py:34-63: every conv of
 * a block feeds nn.BatchNorm2d) folded into the epilogue: per-channel sums of (y - pivot), (y - pivot)^2 go into the fp64
 * slot rows of fami_bn_slots_bytes(Co) bytes (ZERO on entry), pivot = pivot_src[c] (the running mean; NULL = 0) is
 * stored behind the rows.  Consumers: fami_bn_apply_slots_* / fami_bn_finalize_slots_f32.
 * fami_conv2d_dgrad_bnstats_*: the input-gradient convolution that makes the LAST contribution to dx = dL/d(output of a
 * train-mode BatchNorm [+ReLU] whose input was z): stores dz = relu-mask(dx) instead of dx and adds sum dz, sum dz*xhat
 * into `slots` (fami_bn_slots_bytes(Ci), ZERO on entry) for fami_bn_bwd_apply_slots_*.  relu: 0 none, 1 mask from the
 * BatchNorm output yrelu, 2 mask recomputed from z (forward run by fami_bn_train_fwd2 / fami_bn_apply_slots). */
int fami_conv2d_fwd_stats_f32(const float* x, const float* wp, const float* bias, float* y, int N, int H, int W, int Ci,
                              int Co, int kh, int kw, int stride, int pad, int dil, void* slots, const float* pivot_src,
                              fami_stream_t stream);
int fami_conv2d_dgrad_bnstats_f32(const float* dy, const float* wp, float* dx, int N, int H, int W, int Ci, int Co,
                                  int kh, int kw, int stride, int pad, int dil, int accumulate, const float* z,
                                  const float* yrelu, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, int relu, void* slots, fami_stream_t stream);
long fami_conv2d_wgrad_workspace(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil);
/* Deferred slab reduce (the weight-gradient kernels write per-workgroup partial slabs; the reference's autograd has no
 * counterpart): fami_conv2d_wgrad_defer_* launches the kernel and describes its reduce in desc_out -- a HOST buffer of
 * fami_wgrad_reduce_desc_longs() longs -- and fami_wgrad_reduce_batch launches up to 16 described reduces per kernel
 * launch (descs: n consecutive descriptors on the host).  Same summation order as the immediate reduce. */
int fami_wgrad_reduce_desc_longs(void);
int fami_wgrad_reduce_batch(const long* descs, int n, fami_stream_t stream);
int fami_conv2d_wgrad_defer_f32(const float* x, const float* dy, float* dw, float* workspace, long ws_bytes, int N,
                                int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil,
                                int accumulate, long* desc_out, fami_stream_t stream);
/* dw[Co,Ci,kh,kw] (OIHW, =|+=) */
int fami_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, float* workspace, long ws_bytes, int N, int H,
                          int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                          fami_stream_t stream);

/* ---- BatchNorm (+ReLU, +residual) : nn.BatchNorm2d(momentum 0.1, eps 1e-5) ---------------
 * replaces basic_model.py:34,42 (BasicBlock.bn1/bn2 + `out += residual` + ReLU :46-63),
 * :75-79 (Bottleneck), basic_layer.py:25-26, hrnet.py:107,127,138 (fuse BNs).  x is [P,C]. */
long fami_bn_workspace(int C);
int fami_bn_stats_f32(const float* x, long P, int C, float* mean, float* invstd, float* running_mean,
                      float* running_var, float momentum, float eps, float* ws, fami_stream_t stream);
/* train-mode forward in one call: batch statistics (+ running-stat update) and apply (+ residual, + ReLU);
 * tensors of <= 16384 pixels (the low-resolution branches) run as ONE launch */
int fami_bn_train_fwd_f32(const float* x, const float* residual, float* y, const float* gamma, const float* beta,
                          float* mean, float* invstd, float* running_mean, float* running_var, long P, int C,
                          int relu, float momentum, float eps, float* ws, fami_stream_t stream);
/* running_mean / running_var update from a finished (mean, invstd) of a train-mode call that was run with NULL running
 * pointers: shared BatchNorm modules can then run concurrently on several streams and be updated afterwards in call
 * order (var = 1/invstd^2 - eps, unbiased for running_var as nn.BatchNorm2d) */
int fami_bn_running_update_f32(float* running_mean, float* running_var, const float* mean, const float* invstd, int C,
                               long P, float momentum, float eps, fami_stream_t stream);
/* n deferred updates in call order, one launch per 32: ptrs = host array of 4 n longs (running_mean, running_var, mean,
 * invstd), meta = host array of 4 n floats (C, P, momentum, eps) */
int fami_bn_running_update_batch_f32(const long* ptrs, const float* meta, int n, fami_stream_t stream);
int fami_bn_eval_stats_f32(const float* running_mean, const float* running_var, float* mean, float* invstd, int C,
                           float eps, fami_stream_t stream);
int fami_bn_apply_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                      const float* residual, float* y, long P, int C, int relu, fami_stream_t stream);
int fami_bn_bwd_f32(const float* dy, const float* x, const float* y, const float* mean, const float* invstd,
                    const float* gamma, float* dx, float* dgamma, float* dbeta, float* dres, long P, int C, int relu,
                    int acc_dx, int acc_param, int acc_dres, float* ws, fami_stream_t stream);
/* Two-launch forms of the two calls above (fami_bn_{stats,finalize,apply,bwd}; the finalize launch is folded
 * into the apply pass): the statistics pass adds its partial sums into fp64 slot rows, every workgroup of the apply pass
 * folds them in its prologue.  `slots`: fami_bn_slots_bytes(C) bytes, ZERO on entry (the caller clears one arena per
 * step).  Backward relu: 0 none, 1 mask from y, 2 mask recomputed from x (no residual; y may be NULL) -- the latter only
 * against a forward run by fami_bn_train_fwd2 (same fused multiply-add of scale / shift).  Sums arrive in atomic order:
 * reproducible to fp64 rounding, not bit for bit -- the three-launch forms stay for the deterministic mode. */
long fami_bn_slots_bytes(int C);
int fami_bn_train_fwd2_f32(const float* x, const float* residual, float* y, const float* gamma, const float* beta,
                           float* mean, float* invstd, float* running_mean, float* running_var, long P, int C,
                           int relu, float momentum, float eps, void* slots, fami_stream_t stream);
int fami_bn_bwd2_f32(const float* dy, const float* x, const float* y, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta, float* dres, long P,
                     int C, int relu, int acc_dx, int acc_param, int acc_dres, void* slots, fami_stream_t stream);
/* The apply passes alone, for slot rows a convolution epilogue has filled (fami_conv2d_fwd_stats_* /
 * fami_conv2d_dgrad_bnstats_*): one launch per BatchNorm pass instead of two.  dz: the gradient with the ReLU mask
 * already applied.  fami_bn_is_small: tensors the one-launch small-tensor kernel takes (no point fusing those).
 * fami_bn_finalize_slots_f32: mean / invstd / running update only (HighResolutionModule fuse terms, hrnet.py:151-172). */
int fami_bn_is_small(long P, int C);
/* conv1 -> BatchNorm -> ReLU -> conv2 without the normalised tensor in HBM (16-bit modes; basic_model.py:34-63): conv2's
 * forward (fami_conv2d_fwd_xbn_*) and weight gradient (fami_conv2d_wgrad_defer_xbn_*) take the PRE-normalisation tensor z
 * and apply scale / shift / ReLU while they stage it into LDS.  fami_conv2d_xbn_ok: can a 3x3 stride-1 pad-1 convolution of
 * this shape do that (both kernels eligible)? */
int fami_conv2d_xbn_ok(int N, int H, int W, int Ci, int Co);
/* Round 6: the train-mode BatchNorm + ReLU in front of a 3x3 stride-1 convolution INSIDE that convolution's launch, with the
 * normalised tensor written out as well (a_out [N,H,W,Ci]: what the reference materialises between conv1 and conv2 of a BasicBlock,
 * basic_model.py:34-63; the backward pass reads it).  z = the BatchNorm's input, xslots = its statistics rows (filled by
 * fami_conv2d_fwd_stats_* of the producing convolution); mean / invstd / running statistics are published as by
 * fami_bn_apply_slots_*.  Results bit for bit those of fami_bn_apply_slots_* followed by fami_conv2d_fwd(_stats)_*. */
int fami_conv2d_fwd_bnin_ok(int N, int H, int W, int Ci, int Co);
/* Backward of a 3x3 stride-1 pad-1 convolution as ONE launch (16-bit modes; nn.Conv2d autograd of basic_model.py:44-63):
 * the input gradient (fami_conv2d_dgrad_* / fami_conv2d_dgrad_bnstats_* when slots != NULL) and the deferred weight gradient
 * (fami_conv2d_wgrad_defer_*) read the same dY and nothing of each other; fami_conv2d_bwd_pair_* runs the two kernels' bodies
 * side by side in one grid (weight-gradient workgroups first).  Bitwise the two-call form; where no combined instance exists
 * the two single launches run.  fami_conv2d_bwd_pair_ok: does a combined instance take this geometry (forward geometry)?
 * fami_conv2d_bwd_pair_key: the kernel instances of the two halves (tools / tests; out[9]). */
int fami_conv2d_bwd_pair_ok(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil);
int fami_conv2d_bwd_pair_key(int N, int H, int W, int Ci, int Co, int* out);
/* f32 storage: the split-product input gradient and the deferred split-product weight gradient of a 3x3 stride-1 convolution in one
 * launch (xmean != NULL: x is the input z of a BatchNorm + ReLU nobody materialised, as fami_conv2d_wgrad_defer_xbn_f32). */
int fami_conv2d_bwd_pair_ok_f32(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil);
int fami_conv2d_bwd_pair_key_f32(int N, int H, int W, int Ci, int Co, int* out);
int fami_conv2d_bwd_pair_f32(const float* x, const float* dy, const float* wpd, float* dx, float* dw, float* workspace,
                             long ws_bytes, int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil,
                             int acc_dx, int acc_dw, long* desc_out, const float* xmean, const float* xinvstd,
                             const float* xgamma, const float* xbeta, fami_stream_t stream);
/* the same in f32 storage (the split-product kernels; FAMI_XBN=1) */
int fami_conv2d_xbn_ok_f32(int N, int H, int W, int Ci, int Co);
int fami_conv2d_fwd_xbn_f32(const float* z, const float* wp, const float* bias, float* y, int N, int H, int W, int Ci,
                            int Co, void* slots, const float* pivot_src, const void* xslots, long xP,
                            const float* xgamma, const float* xbeta, float* xmean, float* xinvstd,
                            float* xrunning_mean, float* xrunning_var, float xmomentum, float xeps, fami_stream_t stream);
int fami_conv2d_wgrad_defer_xbn_f32(const float* z, const float* dy, float* dw, float* workspace, long ws_bytes, int N,
                                    int H, int W, int Ci, int Co, int accumulate, long* desc_out, const float* xmean,
                                    const float* xinvstd, const float* xgamma, const float* xbeta, fami_stream_t stream);
/* benchmarks: element count up to which a tensor takes the one-launch small-tensor kernels (< 0: default 32768) */
int fami_bn_tune_small(long elems);
int fami_bn_apply_slots_f32(const float* x, const float* residual, float* y, const float* gamma, const float* beta,
                            float* mean, float* invstd, float* running_mean, float* running_var, long P, int C,
                            int relu, float momentum, float eps, void* slots, fami_stream_t stream);
int fami_bn_bwd_apply_slots_f32(const float* dz, const float* x, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta,
                                float* dres, long P, int C, int acc_dx, int acc_param, int acc_dres, void* slots,
                                fami_stream_t stream);
int fami_bn_finalize_slots_f32(void* slots, long P, int C, float* mean, float* invstd, float* running_mean,
                               float* running_var, float momentum, float eps, fami_stream_t stream);
long fami_channel_sum_workspace(int C);
/* out[c] (=|+=) sum_p x[p][c] : bias gradients of the biased convs */
int fami_channel_sum_f32(const float* x, long P, int C, float* out, int accumulate, float* ws, fami_stream_t stream);

/* ---- streaming helpers ---------------------------------------------------------------------
 * boundary layout converts, torch.cat/chunk (Alignment_V15.py:117-125,139,143,160), `sup - kf`
 * (:132), HighResolutionModule fuse sum + Interpolate(nearest) + ReLU (hrnet.py:159-168,
 * basic_model.py:116-125), torch.optim.Adam (posetimation/optimizer/optimizer.py:66-68). */
int fami_nchw_to_nhwc_f32(const float* src, float* dst, int N, int C, int H, int W, fami_stream_t stream);
int fami_nhwc_to_nchw_f32(const float* src, float* dst, int N, int C, int H, int W, int accumulate,
                          fami_stream_t stream);
int fami_pack_frames_f32(const float* kf_nchw, const float* sup_nchw, float* frames_nhwc, int B, int S, int H, int W,
                         fami_stream_t stream);
int fami_copy_channels_f32(const float* src, float* dst, long P, int Cs, int src_off, int Cd, int dst_off, int Cc,
                           int accumulate, fami_stream_t stream);
/* torch.cat on channels of n <= 4 tensors in one launch (c[k] % 4 == 0) and its backward: the k-th channel slice of src (=|+=) into
 * dst[k] (null: skipped).  src / dst / c / accumulate are HOST arrays of length n. */
int fami_concat_channels_f32(const float* const* src, const int* c, int n, float* dst, long P, fami_stream_t stream);
int fami_split_channels_f32(const float* src, float* const* dst, const int* c, const int* accumulate, int n, long P,
                             fami_stream_t stream);
int fami_axpby_f32(const float* a, const float* b, float* out, long n, float alpha, float beta,
                   fami_stream_t stream);
int fami_fill_f32(float* out, long n, float v, fami_stream_t stream);
/* dst (=|+=) src : fp32 accumulation buffer folded into an activation-typed gradient */
int fami_cast_add_f32(const float* src, float* dst, long n, int accumulate, fami_stream_t stream);
/* dst[i] = (float)src[i] (fami_widen_bf16 / _f16: the 16-bit gradient payload of the data-parallel all-reduce widened
 * back into the fp32 gradient arena; the _f32 instance is a copy) */
int fami_widen_f32(const float* src, float* dst, long n, fami_stream_t stream);
/* out[i] (=|+=) in[i] * (sx, sy) over n (x, y) pairs: legacy kornia.warp_affine translation scaling and its gradient
 * (kornia <= 0.4 default align_corners=False at Alignment_V15.py:135: a shift of t pixels samples at x - t*W/(W-1)). */
int fami_scale_pairs_f32(const float* in, float* out, long n, float sx, float sy, int accumulate, fami_stream_t stream);
int fami_incr_i64(long long* v, long n, fami_stream_t stream);
/* v[i] += inc[i] : BatchNorm num_batches_tracked counters, all layers in one launch */
int fami_add_i64(long long* v, const long long* inc, long n, fami_stream_t stream);
int fami_fuse_sum_f32(int nterms, const float* const* x, const float* const* mean, const float* const* invstd,
                      const float* const* gamma, const float* const* beta, const int* shift, float* y, int N, int H,
                      int W, int C, int relu, fami_stream_t stream);
int fami_relu_bwd_f32(const float* dy, const float* y, float* dx, long n, int accumulate, fami_stream_t stream);
int fami_pool_relu_bwd_f32(const float* dy, const float* y, float* out, int N, int Hl, int Wl, int C, int shift,
                           int relu, fami_stream_t stream);
/* out_k += a_k for n pairs of fp32 tensors in one launch per 32 pairs (the engine's lane join: lane-private gradients of a
 * module that ran on several stream lanes).  ptrs: host array of 2 n longs (a_0, out_0, a_1, out_1, ...), counts: n ints.
 * The same `out` may appear several times: a repeated output starts a new launch, so its adds happen in call order. */
int fami_add_batch_f32(const long* ptrs, const int* counts, int n, fami_stream_t stream);
int fami_adam_prep_f32(float* state4, float beta1, float beta2, fami_stream_t stream);
int fami_adam_f32(float* p, const float* g, float* m, float* v, long n, const float* state4, float beta1,
                  float beta2, float eps, float weight_decay, fami_stream_t stream);
/* fp16 static loss scaling with an overflow guard (the reference trains in fp32 and has no counterpart; BASELINE config 5):
 * fami_unscale_check_f32 multiplies the gradient arena by f and raises *flag (device u32) on any inf / NaN;
 * fami_adam_prep_checked_f32 then skips the whole Adam step (no step count, no moment update), clears the flag and
 * counts the skip: flag is a device u32[2] = {raised, skipped steps so far}. */
int fami_unscale_check_f32(float* g, long n, float f, unsigned* flag, fami_stream_t stream);
int fami_adam_prep_checked_f32(float* state4, float beta1, float beta2, unsigned* flag, fami_stream_t stream);

/* ---- temporal alignment ----------------------------------------------------------------------
 * fami_shift_bilinear_*: kornia.geometry.warp_affine(src, [[1,0,tx],[0,1,ty]], dsize) at
 *   Alignment_V15.py:133-135 (differentiable wrt src and (tx,ty)); t is a device [B,2] = (tx,ty).
 * fami_dcn_*: torchvision.ops.DeformConv2d(C,C,3,padding=3,dilation=3).forward(x, offset, mask)
 *   at Alignment_V15.py:146,150,154,158 and its autograd; offset [B,Ho,Wo,2*G*K] ordered
 *   (group, tap, (dy,dx)), mask [B,Ho,Wo,G*K] raw (no sigmoid), G = offset groups (12). */
long fami_shift_workspace(int B);
int fami_shift_bilinear_fwd_f32(const float* src, const float* t, float* out, int B, int H, int W, int C,
                                fami_stream_t stream);
int fami_shift_bilinear_bwd_f32(const float* gout, const float* src, const float* t, float* gsrc, float* gt, int B,
                                int H, int W, int C, int acc_src, int acc_t, float* ws, fami_stream_t stream);
long fami_dcn_packed_weight_elems(int Co, int C, int kh, int kw, int G);
int fami_dcn_pack_weight_f32(const float* w_oihw, float* wp, int Co, int C, int kh, int kw, int G,
                             fami_stream_t stream);
/* the same plus the 16-bit image the bf16 / fp16 forward contracts with on the 16x16x32 matrix-core instruction (the gather's
 * vector arithmetic then overlaps the contraction); wp: fami_dcn_packed_weight_elems floats in every case */
int fami_dcn_pack_weight_bf16(const float* w_oihw, float* wp, int Co, int C, int kh, int kw, int G,
                              fami_stream_t stream);
int fami_dcn_pack_weight_f16(const float* w_oihw, float* wp, int Co, int C, int kh, int kw, int G,
                             fami_stream_t stream);
int fami_dcn_fwd_f32(const float* x, const float* off, const float* msk, const float* wp, const float* bias, float* y,
                     int B, int H, int W, int C, int Co, int G, int kh, int kw, int stride, int pad, int dil,
                     fami_stream_t stream);
int fami_dcn_tune(int mode);            /* benchmarks / tests: forward kernel 0 = dcn_fwd_kernel (LDS column tile), 1 = dcn_fwd_direct_kernel
                                           (samples fed to the MFMA from registers), -1 = default (direct below 4 GiB);
                                           16 + 2 / 16 + 0 = direct kernel built for 2 k groups in flight x 4 waves per SIMD / default build;
                                           2 = LDS-window forward kernel where eligible; 512 + m = backward input-gradient scatter: 0 f32
                                           compare-and-swap LDS region, 1 fixed-point region (64-bit for f32, 32-bit for 16-bit storage; default),
                                           2 64-bit region for every type; 1024 + bits = backward ablations (benchmarks) */
long fami_dcn_packed_weight_bwd_elems(int Co, int C, int kh, int kw, int G);
int fami_dcn_pack_weight_bwd_f32(const float* w_oihw, float* wpb, int Co, int C, int kh, int kw, int G,
                                 fami_stream_t stream);
/* Column count of the `col` matrix the fami_dcn_bwd_* / fami_dcn_bwd_om_* entry points write for this geometry and storage
 * size (elem_bytes 4 | 2): C*kh*kw in weight.view(Co, C*K) order on the general kernel (always in the deterministic form), or
 * more -- the register-fed kernel writes the modulated samples in its own (chunk, tap, group, channel) column order, padded to
 * whole 16-column blocks (fami_dcn_bwd_col_permuted says which; the widths can coincide).  In that case the caller's 1x1 weight
 * gradient over col is [Co][cols], and
 * fami_dcn_col_dw_unpermute_f32 (=|+=) brings it into the OIHW order of DeformConv2d.weight (Alignment_V15.py:83-101). */
long fami_dcn_bwd_col_width(int C, int Co, int G, int kh, int kw, int stride, int dil, int elem_bytes, int deterministic);
int fami_dcn_bwd_col_permuted(int C, int Co, int G, int kh, int kw, int stride, int dil, int elem_bytes, int deterministic);
int fami_dcn_col_dw_unpermute_f32(const float* dwp, float* dw, int Co, int C, int G, int kh, int kw, int stride, int dil,
                                  int elem_bytes, int accumulate, fami_stream_t stream);
/* autograd of DeformConv2d wrt input / offsets / masks (fused: column gradient dy x W stays in LDS).
 * col [P, C*K] (out, may be NULL) = modulated samples, column order (channel, tap) == weight.view(Co, C*K):
 * dW[co, kidx] = sum_p dy[p,co] * col[p,kidx], the caller runs that GEMM (fami_conv2d_wgrad_f32, 1x1).
 * gx [B,H,W,C] is ACCUMULATED with atomics (zero it first); goff/gmsk (=|+=) per acc_off; each may be NULL. */
int fami_dcn_bwd_f32(const float* x, const float* off, const float* msk, const float* dy, const float* wpb,
                     float* col, float* gx, float* goff, float* gmsk, int B, int H, int W, int C, int Co, int G,
                     int kh, int kw, int stride, int pad, int dil, int acc_off, fami_stream_t stream);

/* Run-to-run deterministic form of fami_dcn_bwd_f32 ("an fp32 deterministic path for the parity config"):
 * the scatter of the input gradient adds 64-bit fixed-point integers (associative, so independent of arrival order)
 * scaled by 2^(42 - ceil(log2 max|dy|)); gx [B,H,W,C] is then written (=|+= per acc_x) by a conversion pass.
 * Overflow needs one input element to collect more than 2^20 x max|dy|.  ws: fami_dcn_bwd_det_workspace bytes. */
long fami_dcn_bwd_det_workspace(int B, int H, int W, int C);
int fami_dcn_bwd_det_f32(const float* x, const float* off, const float* msk, const float* dy, const float* wpb,
                         float* col, float* gx, float* goff, float* gmsk, int B, int H, int W, int C, int Co, int G,
                         int kh, int kw, int stride, int pad, int dil, int acc_off, int acc_x, void* ws,
                         fami_stream_t stream);

/* The three DCN entry points above for offsets and masks held in ONE tensor om [B,Ho,Wo,3GK] -- per pixel the 2GK offsets
 * followed by the GK masks: what the offset and the mask predictor of a DCN layer (Alignment_V15.py:79-100, called at
 * :144-158) produce when they run as one 48 -> 324 convolution (weights concatenated on the output-channel axis; the
 * state_dict keeps the two modules).  gom: the gradient wrt om, same layout (=|+= per acc_om).  Same kernels, other strides. */
int fami_dcn_fwd_om_f32(const float* x, const float* om, const float* wp, const float* bias, float* y, int B, int H, int W,
                       int C, int Co, int G, int kh, int kw, int stride, int pad, int dil, fami_stream_t stream);
int fami_dcn_bwd_om_f32(const float* x, const float* om, const float* dy, const float* wpb, float* col, float* gx, float* gom,
                       int B, int H, int W, int C, int Co, int G, int kh, int kw, int stride, int pad, int dil, int acc_om,
                       fami_stream_t stream);
int fami_dcn_bwd_det_om_f32(const float* x, const float* om, const float* dy, const float* wpb, float* col, float* gx,
                           float* gom, int B, int H, int W, int C, int Co, int G, int kh, int kw, int stride, int pad,
                           int dil, int acc_om, int acc_x, void* ws, fami_stream_t stream);

/* ---- dense layers of the translation regressor: nn.Linear x3 (Alignment_V15.py:69-71) ------- */
int fami_linear_fwd_f32(const float* x, const float* w, const float* b, float* y, int M, int K, int N,
                        fami_stream_t stream);
int fami_linear_bwd_f32(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int M,
                        int K, int N, int acc_dx, int acc_param, fami_stream_t stream);

/* ---- targets / losses / decode (NCHW rows) ----------------------------------------------------
 * fami_gauss_target : datasets/process/heatmaps_process.py:146-203 (generate_heatmaps)
 * fami_wmse_*       : posetimation/loss/mse_loss.py:21-40 (JointMSELoss.forward)
 * fami_softmax_kl_* : Alignment_V15.py:250-277 (feat_label/feat_feat MI estimators, T = 0.05)
 * fami_argmax2d      : datasets/process/heatmaps_process.py:16-44 (get_max_preds) */
int fami_gauss_target_f32(const float* joints_xy, const float* vis, float* target, float* weight, int B, int J,
                          int Hh, int Wh, int img_h, int img_w, int sigma, fami_stream_t stream);
int fami_wmse_fwd_f32(const float* pred, const float* gt, const float* w, float* loss, int R, int L, double scale,
                      float* ws, fami_stream_t stream);
int fami_wmse_bwd_f32(const float* pred, const float* gt, const float* w, float* dpred, int R, int L, float scale,
                      const float* gdev, int accumulate, fami_stream_t stream);
int fami_softmax_kl_fwd_f32(const float* A, const float* Bt, float* value, float* stats, int R, int L,
                            float temperature, float* ws, fami_stream_t stream);
int fami_softmax_kl_bwd_f32(const float* A, const float* Bt, const float* stats, float* dBt, int R, int L,
                            float temperature, float gscale, const float* gdev, int accumulate,
                            fami_stream_t stream);
int fami_argmax2d_f32(const float* hm, long long* idx, float* maxval, int R, int L, fami_stream_t stream);
/* get_final_preds (datasets/process/heatmaps_process.py:47-73 + transform_preds, affine_transform.py:13-43, rot 0):
 * argmax, quarter-pixel shift, inverse affine to image coordinates.  hm [B,J,H,W]; center, scale [B,2];
 * preds [B,J,2]; maxvals [B,J]; idx_ws: B*J int64 scratch */
int fami_final_preds_f32(const float* hm, const float* center, const float* scale, float* preds, float* maxvals,
                         long long* idx_ws, int B, int J, int H, int W, fami_stream_t stream);

/* PCK accuracy on heatmap argmax, engine/core/utils/evaluate.py:39-75 (`accuracy`, called twice per training step at
 * engine/core/functions/alignment_mi_function_term6_1.py:159-163 after D2H copies of four heatmap stacks): here
 * three launches and no host sync.  pred_hm / target_hm [B,J,H,W]; out[J+3] = {acc[0] = mean of the defined
 * per-joint accuracies, acc[1..J] (-1 where no valid target), avg_acc, cnt}; idx_ws 2*B*J int64, max_ws 2*B*J floats */
int fami_pck_accuracy_f32(const float* pred_hm, const float* target_hm, float* out, long long* idx_ws, float* max_ws,
                          int B, int J, int H, int W, float thr, fami_stream_t stream);

/* ---- input pipeline on the device -------------------------------------------------------------
 * cv2.warpAffine(frame, trans, (Wd, Hd), flags=INTER_LINEAR) on the F 8-bit HWC frames of one clip (one transform
 * for key + supporting frames, datasets/zoo/posetrack/PoseTrack_Alignment.py:421-427; flip = source mirrored in x
 * first, :409-413; swap_rb = cv2.cvtColor(BGR2RGB), :299-300), then transforms.ToTensor + Normalize
 * (datasets/transforms/build.py:12-23): out[f][c][y][x] = ((crop / 255) - mean[c]) / std[c], NCHW fp32.
 * m00..m12: the INVERSE map (dst -> src) cv2 derives from `trans` (fami_pose_amd.data.invert_affine).  Coordinates and
 * interpolation follow OpenCV's generic fixed-point path bit for bit (see csrc/preproc.hip). */
int fami_warp_normalize_u8(const unsigned char* src, float* out, int F, int Hs, int Ws, long src_stride, int Hd, int Wd,
                           long out_stride, double m00, double m01, double m02, double m10, double m11, double m12,
                           int flip, int swap_rb, float mean0, float mean1, float mean2, float std0, float std1,
                           float std2, fami_stream_t stream);


/* ======================================================================================================
 * bf16 activation storage (BASELINE config 3: bf16 compute, fp32 master weights / accumulation / losses).
 * Every entry point above that touches an activation has a `_bf16` twin with identical semantics and argument
 * order; only the activation pointers change type.  Parameters, BatchNorm statistics, workspaces, the
 * translation (tx,ty), weight gradients, DCN input-gradient accumulators and everything at the NCHW boundary
 * stay fp32.  Convolutions run on v_mfma_f32_16x16x32_bf16 with a bf16 packed weight image (KC = Cin/32).
 * ====================================================================================================== */
typedef unsigned short fami_bf16_t; /* IEEE bfloat16 bit pattern */

long fami_packed_weight_elems_bf16(int Co, int Ci, int kh, int kw, int mode);
int fami_pack_conv_weight_bf16(const float* w_oihw, fami_bf16_t* wp, int Co, int Ci, int kh, int kw, int mode,
                               fami_stream_t stream);
int fami_pack_conv_weights_batch_bf16(const float* params, fami_bf16_t* packed, const void* desc, int n,
                                      fami_stream_t stream);
/* y is bf16, or fp32 when out_f32 (heatmap-producing layers) */
int fami_conv2d_fwd_bf16(const fami_bf16_t* x, const fami_bf16_t* wp, const float* bias, void* y, int N, int H, int W,
                         int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int relu, int accumulate,
                         int out_f32, fami_stream_t stream);
int fami_conv2d_dgrad_bf16(const fami_bf16_t* dy, const fami_bf16_t* wp, fami_bf16_t* dx, int N, int H, int W, int Ci,
                           int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                           fami_stream_t stream);
int fami_conv2d_fwd_xbn_bf16(const fami_bf16_t* z, const fami_bf16_t* wp, const float* bias, fami_bf16_t* y, int N, int H, int W,
                             int Ci, int Co, void* slots, const float* pivot_src, const void* xslots, long xP,
                             const float* xgamma, const float* xbeta, float* xmean, float* xinvstd,
                             float* xrunning_mean, float* xrunning_var, float xmomentum, float xeps, fami_stream_t stream);
int fami_conv2d_fwd_bnin_bf16(const fami_bf16_t* z, const fami_bf16_t* wp, const float* bias, fami_bf16_t* y, fami_bf16_t* a_out, int N, int H, int W,
                              int Ci, int Co, void* slots, const float* pivot_src, const void* xslots, long xP,
                              const float* xgamma, const float* xbeta, float* xmean, float* xinvstd,
                              float* xrunning_mean, float* xrunning_var, float xmomentum, float xeps, fami_stream_t stream);
int fami_conv2d_wgrad_defer_xbn_bf16(const fami_bf16_t* z, const fami_bf16_t* dy, float* dw, float* workspace, long ws_bytes,
                                     int N, int H, int W, int Ci, int Co, int accumulate, long* desc_out,
                                     const float* xmean, const float* xinvstd, const float* xgamma, const float* xbeta,
                                     fami_stream_t stream);
int fami_conv2d_fwd_stats_bf16(const fami_bf16_t* x, const fami_bf16_t* wp, const float* bias, fami_bf16_t* y, int N, int H, int W,
                               int Ci, int Co, int kh, int kw, int stride, int pad, int dil, void* slots,
                               const float* pivot_src, fami_stream_t stream);
int fami_conv2d_dgrad_bnstats_bf16(const fami_bf16_t* dy, const fami_bf16_t* wp, fami_bf16_t* dx, int N, int H, int W, int Ci,
                                   int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                                   const fami_bf16_t* z, const fami_bf16_t* yrelu, const float* mean, const float* invstd,
                                   const float* gamma, const float* beta, int relu, void* slots, fami_stream_t stream);
int fami_conv2d_bwd_pair_bf16(const fami_bf16_t* x, const fami_bf16_t* dy, const fami_bf16_t* wpd, fami_bf16_t* dx, float* dw, float* workspace,
                              long ws_bytes, int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad,
                              int dil, int acc_dx, int acc_dw, long* desc_out, const fami_bf16_t* z, const fami_bf16_t* yrelu,
                              const float* mean, const float* invstd, const float* gamma, const float* beta, int relu,
                              void* slots, fami_stream_t stream);
int fami_bn_apply_slots_bf16(const fami_bf16_t* x, const fami_bf16_t* residual, fami_bf16_t* y, const float* gamma,
                             const float* beta, float* mean, float* invstd, float* running_mean, float* running_var,
                             long P, int C, int relu, float momentum, float eps, void* slots, fami_stream_t stream);
int fami_bn_bwd_apply_slots_bf16(const fami_bf16_t* dz, const fami_bf16_t* x, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, fami_bf16_t* dx, float* dgamma, float* dbeta,
                                 fami_bf16_t* dres, long P, int C, int acc_dx, int acc_param, int acc_dres, void* slots,
                                 fami_stream_t stream);
int fami_conv2d_wgrad_defer_bf16(const fami_bf16_t* x, const fami_bf16_t* dy, float* dw, float* workspace, long ws_bytes,
                                 int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil,
                                 int accumulate, long* desc_out, fami_stream_t stream);
int fami_conv2d_wgrad_bf16(const fami_bf16_t* x, const fami_bf16_t* dy, float* dw, float* workspace, long ws_bytes,
                           int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil,
                           int accumulate, fami_stream_t stream);

int fami_bn_stats_bf16(const fami_bf16_t* x, long P, int C, float* mean, float* invstd, float* running_mean,
                       float* running_var, float momentum, float eps, float* ws, fami_stream_t stream);
int fami_bn_train_fwd_bf16(const fami_bf16_t* x, const fami_bf16_t* residual, fami_bf16_t* y, const float* gamma,
                           const float* beta, float* mean, float* invstd, float* running_mean, float* running_var,
                           long P, int C, int relu, float momentum, float eps, float* ws, fami_stream_t stream);
int fami_bn_apply_bf16(const fami_bf16_t* x, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, const fami_bf16_t* residual, fami_bf16_t* y, long P, int C, int relu,
                       fami_stream_t stream);
int fami_bn_train_fwd2_bf16(const fami_bf16_t* x, const fami_bf16_t* residual, fami_bf16_t* y, const float* gamma, const float* beta,
                           float* mean, float* invstd, float* running_mean, float* running_var, long P, int C,
                           int relu, float momentum, float eps, void* slots, fami_stream_t stream);
int fami_bn_bwd2_bf16(const fami_bf16_t* dy, const fami_bf16_t* x, const fami_bf16_t* y, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, fami_bf16_t* dx, float* dgamma, float* dbeta, fami_bf16_t* dres, long P,
                     int C, int relu, int acc_dx, int acc_param, int acc_dres, void* slots, fami_stream_t stream);
int fami_bn_bwd_bf16(const fami_bf16_t* dy, const fami_bf16_t* x, const fami_bf16_t* y, const float* mean,
                     const float* invstd, const float* gamma, fami_bf16_t* dx, float* dgamma, float* dbeta,
                     fami_bf16_t* dres, long P, int C, int relu, int acc_dx, int acc_param, int acc_dres, float* ws,
                     fami_stream_t stream);
int fami_channel_sum_bf16(const fami_bf16_t* x, long P, int C, float* out, int accumulate, float* ws,
                          fami_stream_t stream);

int fami_nchw_to_nhwc_bf16(const float* src, fami_bf16_t* dst, int N, int C, int H, int W, fami_stream_t stream);
int fami_nhwc_to_nchw_bf16(const fami_bf16_t* src, float* dst, int N, int C, int H, int W, int accumulate,
                           fami_stream_t stream);
int fami_pack_frames_bf16(const float* kf_nchw, const float* sup_nchw, fami_bf16_t* frames_nhwc, int B, int S, int H,
                          int W, fami_stream_t stream);
int fami_copy_channels_bf16(const fami_bf16_t* src, fami_bf16_t* dst, long P, int Cs, int src_off, int Cd,
                            int dst_off, int Cc, int accumulate, fami_stream_t stream);
/* torch.cat on channels of n <= 4 tensors in one launch (c[k] % 4 == 0) and its backward: the k-th channel slice of src (=|+=) into
 * dst[k] (null: skipped).  src / dst / c / accumulate are HOST arrays of length n. */
int fami_concat_channels_bf16(const fami_bf16_t* const* src, const int* c, int n, fami_bf16_t* dst, long P, fami_stream_t stream);
int fami_split_channels_bf16(const fami_bf16_t* src, fami_bf16_t* const* dst, const int* c, const int* accumulate, int n, long P,
                             fami_stream_t stream);
int fami_axpby_bf16(const fami_bf16_t* a, const fami_bf16_t* b, fami_bf16_t* out, long n, float alpha, float beta,
                    fami_stream_t stream);
int fami_widen_bf16(const fami_bf16_t* src, float* dst, long n, fami_stream_t stream);
int fami_cast_add_bf16(const float* src, fami_bf16_t* dst, long n, int accumulate, fami_stream_t stream);
int fami_fill_bf16(fami_bf16_t* out, long n, float v, fami_stream_t stream);
int fami_fuse_sum_bf16(int nterms, const fami_bf16_t* const* x, const float* const* mean, const float* const* invstd,
                       const float* const* gamma, const float* const* beta, const int* shift, fami_bf16_t* y, int N,
                       int H, int W, int C, int relu, fami_stream_t stream);
int fami_relu_bwd_bf16(const fami_bf16_t* dy, const fami_bf16_t* y, fami_bf16_t* dx, long n, int accumulate,
                       fami_stream_t stream);
int fami_pool_relu_bwd_bf16(const fami_bf16_t* dy, const fami_bf16_t* y, fami_bf16_t* out, int N, int Hl, int Wl,
                            int C, int shift, int relu, fami_stream_t stream);

int fami_shift_bilinear_fwd_bf16(const fami_bf16_t* src, const float* t, fami_bf16_t* out, int B, int H, int W, int C,
                                 fami_stream_t stream);
int fami_shift_bilinear_bwd_bf16(const fami_bf16_t* gout, const fami_bf16_t* src, const float* t, fami_bf16_t* gsrc,
                                 float* gt, int B, int H, int W, int C, int acc_src, int acc_t, float* ws,
                                 fami_stream_t stream);
int fami_dcn_fwd_bf16(const fami_bf16_t* x, const fami_bf16_t* off, const fami_bf16_t* msk, const float* wp,
                      const float* bias, fami_bf16_t* y, int B, int H, int W, int C, int Co, int G, int kh, int kw,
                      int stride, int pad, int dil, fami_stream_t stream);
int fami_dcn_bwd_bf16(const fami_bf16_t* x, const fami_bf16_t* off, const fami_bf16_t* msk, const fami_bf16_t* dy,
                      const float* wpb, fami_bf16_t* col, float* gx, fami_bf16_t* goff, fami_bf16_t* gmsk, int B,
                      int H, int W, int C, int Co, int G, int kh, int kw, int stride, int pad, int dil, int acc_off,
                      fami_stream_t stream);
int fami_dcn_bwd_det_bf16(const fami_bf16_t* x, const fami_bf16_t* off, const fami_bf16_t* msk, const fami_bf16_t* dy, const float* wpb, fami_bf16_t* col,
                         fami_bf16_t* gx, fami_bf16_t* goff, fami_bf16_t* gmsk, int B, int H, int W, int C, int Co, int G, int kh, int kw,
                         int stride, int pad, int dil, int acc_off, int acc_x, void* ws, fami_stream_t stream);
int fami_dcn_fwd_om_bf16(const fami_bf16_t* x, const fami_bf16_t* om, const float* wp, const float* bias, fami_bf16_t* y, int B, int H, int W,
                       int C, int Co, int G, int kh, int kw, int stride, int pad, int dil, fami_stream_t stream);
int fami_dcn_bwd_om_bf16(const fami_bf16_t* x, const fami_bf16_t* om, const fami_bf16_t* dy, const float* wpb, fami_bf16_t* col, float* gx, fami_bf16_t* gom,
                       int B, int H, int W, int C, int Co, int G, int kh, int kw, int stride, int pad, int dil, int acc_om,
                       fami_stream_t stream);
int fami_dcn_bwd_det_om_bf16(const fami_bf16_t* x, const fami_bf16_t* om, const fami_bf16_t* dy, const float* wpb, fami_bf16_t* col, fami_bf16_t* gx,
                           fami_bf16_t* gom, int B, int H, int W, int C, int Co, int G, int kh, int kw, int stride, int pad,
                           int dil, int acc_om, int acc_x, void* ws, fami_stream_t stream);


/* ======================================================================================================
 * fp16 activation storage (BASELINE config 5: "fp16 MFMA convs + fp32 loss accumulation").  Identical in every
 * respect to the bf16 family above -- same argument order, same fp32 accumulation / BatchNorm statistics / master
 * weights / losses -- with IEEE binary16 storage and v_mfma_f32_16x16x32_f16.  The packed weight image has the
 * bf16 geometry.  (posetimation/backbones/hrnet.py:569-629 driven by STAGE*.NUM_CHANNELS = 64/128/256/512.)
 * ====================================================================================================== */
typedef unsigned short fami_f16_t; /* IEEE binary16 bit pattern */

long fami_packed_weight_elems_f16(int Co, int Ci, int kh, int kw, int mode);
int fami_pack_conv_weight_f16(const float* w_oihw, fami_f16_t* wp, int Co, int Ci, int kh, int kw, int mode,
                               fami_stream_t stream);
int fami_pack_conv_weights_batch_f16(const float* params, fami_f16_t* packed, const void* desc, int n,
                                      fami_stream_t stream);
/* y is fp16, or fp32 when out_f32 (heatmap-producing layers) */
int fami_conv2d_fwd_f16(const fami_f16_t* x, const fami_f16_t* wp, const float* bias, void* y, int N, int H, int W,
                         int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int relu, int accumulate,
                         int out_f32, fami_stream_t stream);
int fami_conv2d_dgrad_f16(const fami_f16_t* dy, const fami_f16_t* wp, fami_f16_t* dx, int N, int H, int W, int Ci,
                           int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                           fami_stream_t stream);
int fami_conv2d_fwd_xbn_f16(const fami_f16_t* z, const fami_f16_t* wp, const float* bias, fami_f16_t* y, int N, int H, int W,
                             int Ci, int Co, void* slots, const float* pivot_src, const void* xslots, long xP,
                             const float* xgamma, const float* xbeta, float* xmean, float* xinvstd,
                             float* xrunning_mean, float* xrunning_var, float xmomentum, float xeps, fami_stream_t stream);
int fami_conv2d_fwd_bnin_f16(const fami_f16_t* z, const fami_f16_t* wp, const float* bias, fami_f16_t* y, fami_f16_t* a_out, int N, int H, int W,
                              int Ci, int Co, void* slots, const float* pivot_src, const void* xslots, long xP,
                              const float* xgamma, const float* xbeta, float* xmean, float* xinvstd,
                              float* xrunning_mean, float* xrunning_var, float xmomentum, float xeps, fami_stream_t stream);
int fami_conv2d_wgrad_defer_xbn_f16(const fami_f16_t* z, const fami_f16_t* dy, float* dw, float* workspace, long ws_bytes,
                                     int N, int H, int W, int Ci, int Co, int accumulate, long* desc_out,
                                     const float* xmean, const float* xinvstd, const float* xgamma, const float* xbeta,
                                     fami_stream_t stream);
int fami_conv2d_fwd_stats_f16(const fami_f16_t* x, const fami_f16_t* wp, const float* bias, fami_f16_t* y, int N, int H, int W,
                               int Ci, int Co, int kh, int kw, int stride, int pad, int dil, void* slots,
                               const float* pivot_src, fami_stream_t stream);
int fami_conv2d_dgrad_bnstats_f16(const fami_f16_t* dy, const fami_f16_t* wp, fami_f16_t* dx, int N, int H, int W, int Ci,
                                   int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                                   const fami_f16_t* z, const fami_f16_t* yrelu, const float* mean, const float* invstd,
                                   const float* gamma, const float* beta, int relu, void* slots, fami_stream_t stream);
int fami_conv2d_bwd_pair_f16(const fami_f16_t* x, const fami_f16_t* dy, const fami_f16_t* wpd, fami_f16_t* dx, float* dw, float* workspace,
                              long ws_bytes, int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad,
                              int dil, int acc_dx, int acc_dw, long* desc_out, const fami_f16_t* z, const fami_f16_t* yrelu,
                              const float* mean, const float* invstd, const float* gamma, const float* beta, int relu,
                              void* slots, fami_stream_t stream);
int fami_bn_apply_slots_f16(const fami_f16_t* x, const fami_f16_t* residual, fami_f16_t* y, const float* gamma,
                             const float* beta, float* mean, float* invstd, float* running_mean, float* running_var,
                             long P, int C, int relu, float momentum, float eps, void* slots, fami_stream_t stream);
int fami_bn_bwd_apply_slots_f16(const fami_f16_t* dz, const fami_f16_t* x, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, fami_f16_t* dx, float* dgamma, float* dbeta,
                                 fami_f16_t* dres, long P, int C, int acc_dx, int acc_param, int acc_dres, void* slots,
                                 fami_stream_t stream);
int fami_conv2d_wgrad_defer_f16(const fami_f16_t* x, const fami_f16_t* dy, float* dw, float* workspace, long ws_bytes,
                                 int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil,
                                 int accumulate, long* desc_out, fami_stream_t stream);
int fami_conv2d_wgrad_f16(const fami_f16_t* x, const fami_f16_t* dy, float* dw, float* workspace, long ws_bytes,
                           int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil,
                           int accumulate, fami_stream_t stream);

int fami_bn_stats_f16(const fami_f16_t* x, long P, int C, float* mean, float* invstd, float* running_mean,
                       float* running_var, float momentum, float eps, float* ws, fami_stream_t stream);
int fami_bn_train_fwd_f16(const fami_f16_t* x, const fami_f16_t* residual, fami_f16_t* y, const float* gamma,
                           const float* beta, float* mean, float* invstd, float* running_mean, float* running_var,
                           long P, int C, int relu, float momentum, float eps, float* ws, fami_stream_t stream);
int fami_bn_apply_f16(const fami_f16_t* x, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, const fami_f16_t* residual, fami_f16_t* y, long P, int C, int relu,
                       fami_stream_t stream);
int fami_bn_train_fwd2_f16(const fami_f16_t* x, const fami_f16_t* residual, fami_f16_t* y, const float* gamma, const float* beta,
                           float* mean, float* invstd, float* running_mean, float* running_var, long P, int C,
                           int relu, float momentum, float eps, void* slots, fami_stream_t stream);
int fami_bn_bwd2_f16(const fami_f16_t* dy, const fami_f16_t* x, const fami_f16_t* y, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, fami_f16_t* dx, float* dgamma, float* dbeta, fami_f16_t* dres, long P,
                     int C, int relu, int acc_dx, int acc_param, int acc_dres, void* slots, fami_stream_t stream);
int fami_bn_bwd_f16(const fami_f16_t* dy, const fami_f16_t* x, const fami_f16_t* y, const float* mean,
                     const float* invstd, const float* gamma, fami_f16_t* dx, float* dgamma, float* dbeta,
                     fami_f16_t* dres, long P, int C, int relu, int acc_dx, int acc_param, int acc_dres, float* ws,
                     fami_stream_t stream);
int fami_channel_sum_f16(const fami_f16_t* x, long P, int C, float* out, int accumulate, float* ws,
                          fami_stream_t stream);

int fami_nchw_to_nhwc_f16(const float* src, fami_f16_t* dst, int N, int C, int H, int W, fami_stream_t stream);
int fami_nhwc_to_nchw_f16(const fami_f16_t* src, float* dst, int N, int C, int H, int W, int accumulate,
                           fami_stream_t stream);
int fami_pack_frames_f16(const float* kf_nchw, const float* sup_nchw, fami_f16_t* frames_nhwc, int B, int S, int H,
                          int W, fami_stream_t stream);
int fami_copy_channels_f16(const fami_f16_t* src, fami_f16_t* dst, long P, int Cs, int src_off, int Cd,
                            int dst_off, int Cc, int accumulate, fami_stream_t stream);
/* torch.cat on channels of n <= 4 tensors in one launch (c[k] % 4 == 0) and its backward: the k-th channel slice of src (=|+=) into
 * dst[k] (null: skipped).  src / dst / c / accumulate are HOST arrays of length n. */
int fami_concat_channels_f16(const fami_f16_t* const* src, const int* c, int n, fami_f16_t* dst, long P, fami_stream_t stream);
int fami_split_channels_f16(const fami_f16_t* src, fami_f16_t* const* dst, const int* c, const int* accumulate, int n, long P,
                             fami_stream_t stream);
int fami_axpby_f16(const fami_f16_t* a, const fami_f16_t* b, fami_f16_t* out, long n, float alpha, float beta,
                    fami_stream_t stream);
int fami_widen_f16(const fami_f16_t* src, float* dst, long n, fami_stream_t stream);
int fami_cast_add_f16(const float* src, fami_f16_t* dst, long n, int accumulate, fami_stream_t stream);
int fami_fill_f16(fami_f16_t* out, long n, float v, fami_stream_t stream);
int fami_fuse_sum_f16(int nterms, const fami_f16_t* const* x, const float* const* mean, const float* const* invstd,
                       const float* const* gamma, const float* const* beta, const int* shift, fami_f16_t* y, int N,
                       int H, int W, int C, int relu, fami_stream_t stream);
int fami_relu_bwd_f16(const fami_f16_t* dy, const fami_f16_t* y, fami_f16_t* dx, long n, int accumulate,
                       fami_stream_t stream);
int fami_pool_relu_bwd_f16(const fami_f16_t* dy, const fami_f16_t* y, fami_f16_t* out, int N, int Hl, int Wl,
                            int C, int shift, int relu, fami_stream_t stream);

int fami_shift_bilinear_fwd_f16(const fami_f16_t* src, const float* t, fami_f16_t* out, int B, int H, int W, int C,
                                 fami_stream_t stream);
int fami_shift_bilinear_bwd_f16(const fami_f16_t* gout, const fami_f16_t* src, const float* t, fami_f16_t* gsrc,
                                 float* gt, int B, int H, int W, int C, int acc_src, int acc_t, float* ws,
                                 fami_stream_t stream);
int fami_dcn_fwd_f16(const fami_f16_t* x, const fami_f16_t* off, const fami_f16_t* msk, const float* wp,
                      const float* bias, fami_f16_t* y, int B, int H, int W, int C, int Co, int G, int kh, int kw,
                      int stride, int pad, int dil, fami_stream_t stream);
int fami_dcn_bwd_f16(const fami_f16_t* x, const fami_f16_t* off, const fami_f16_t* msk, const fami_f16_t* dy,
                      const float* wpb, fami_f16_t* col, float* gx, fami_f16_t* goff, fami_f16_t* gmsk, int B,
                      int H, int W, int C, int Co, int G, int kh, int kw, int stride, int pad, int dil, int acc_off,
                      fami_stream_t stream);
int fami_dcn_bwd_det_f16(const fami_f16_t* x, const fami_f16_t* off, const fami_f16_t* msk, const fami_f16_t* dy, const float* wpb, fami_f16_t* col,
                         fami_f16_t* gx, fami_f16_t* goff, fami_f16_t* gmsk, int B, int H, int W, int C, int Co, int G, int kh, int kw,
                         int stride, int pad, int dil, int acc_off, int acc_x, void* ws, fami_stream_t stream);
int fami_dcn_fwd_om_f16(const fami_f16_t* x, const fami_f16_t* om, const float* wp, const float* bias, fami_f16_t* y, int B, int H, int W,
                       int C, int Co, int G, int kh, int kw, int stride, int pad, int dil, fami_stream_t stream);
int fami_dcn_bwd_om_f16(const fami_f16_t* x, const fami_f16_t* om, const fami_f16_t* dy, const float* wpb, fami_f16_t* col, float* gx, fami_f16_t* gom,
                       int B, int H, int W, int C, int Co, int G, int kh, int kw, int stride, int pad, int dil, int acc_om,
                       fami_stream_t stream);
int fami_dcn_bwd_det_om_f16(const fami_f16_t* x, const fami_f16_t* om, const fami_f16_t* dy, const float* wpb, fami_f16_t* col, fami_f16_t* gx,
                           fami_f16_t* gom, int B, int H, int W, int C, int Co, int G, int kh, int kw, int stride, int pad,
                           int dil, int acc_om, int acc_x, void* ws, fami_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FAMI_H */
