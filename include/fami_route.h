/* Kernel-routing state of libfami_hip.so ("no global mutable state besides a kernel cache").
 * Every switch that selects between kernel forms of one entry point -- what the fami_*_tune* code numbers used to poke into
 * process-wide variables -- is a field of fami_route_t.  An Engine owns one (fami_route_init) and binds it to the calling
 * thread (fami_route_bind) before it enqueues work, so two engines with different routes coexist in one process; entry points
 * called with nothing bound use the process default route.  The fami_*_tune* functions remain as shims for tests and
 * benchmarks: they write the fields of the route bound to the calling thread (or of the process default).
 * Plain C: ints and longs only, so a ctypes.Structure mirrors it (fami-pose_amd/_lib.py builds it from this file). */
#ifndef FAMI_ROUTE_H
#define FAMI_ROUTE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct fami_route_t {
  int size;                       /* sizeof(fami_route_t), written by fami_route_init (layout check) */
  /* ---- align.hip */
  int  dcn_gather;            /* default -1.  fami_dcn_tune: 0 = dcn_fwd_kernel (LDS column tile), 1 = dcn_fwd_direct_kernel, 2 = dcn_fwd_win_kernel where eligible, -1 default (= 1) */
  int  dcn_abl;               /* default 0 */
  int  dcn_win_r;             /* default 0.  fami_dcn_tune(32 + r): force the window's offset reach (benchmarks); 0 = largest that fits, up to 4 */
  int  dcn_ksplit;            /* default 1.  fami_dcn_tune(256 + k): 2 = the 16-wave K-split build (measured slower: 35.5 vs 29.8 us) */
  int  dcn_pf;                /* default 0.  fami_dcn_tune(16 + 2): the 2-k-groups-in-flight x 4-waves-per-SIMD build of the direct kernel (benchmarks) */
  int  dcn_bwd2;              /* default 1.  fami_dcn_tune(2048 / 2049): off / on */
  int  dcn_bwd2_cap;          /* default 36.  fami_dcn_tune(4096 + KB): LDS budget of the fixed-point region (decides the groups per workgroup; benchmarks -- set BEFORE the weight pack) */
  int  dcn_bwd2_stage;        /* default 1.  fami_dcn_tune(8192 / 8193): offset / mask gradients staged in LDS and written with consecutive lanes on consecutive elements -- PMC WRITE_SIZE per B = 4 launch 218 -> 123 MB (f32), 181 -> 78 MB (bf16); time 84.4 -> 88.6 us (f32), 72.6 -> 71.7 us (bf16) */
  int  dcn_bwd_abl;           /* default 0.  fami_dcn_tune(1024 + bits): ablations of the general backward kernel (benchmarks) */
  int  dcn_bwd_scatter;       /* default -1.  fami_dcn_tune(512 + m): 0 = f32 compare-and-swap LDS adds, 1 / default = fixed-point LDS adds (64-bit for f32, 32-bit for 16-bit storage), 2 = 64-bit for every type */
  /* ---- conv.hip */
  int  prio;                  /* default 0.  fami_conv_tune_stages(120 / 121): s_setprio in the f32 MFMA kernels off / on */
  int  lin_conv;              /* default 1.  fami_conv_tune_stages(100 / 101): linear-address form of the f32 implicit GEMM off / on */
  int  par;                   /* default 1.  fami_conv_tune_stages(110 / 111): parity-class stride-2 input gradient off / on */
  int  force_mt;              /* default 0.  tuning overrides (fami_conv_tune) */
  int  force_nt;              /* default 0.  tuning overrides (fami_conv_tune) */
  int  force_ks;              /* default 0.  tuning overrides (fami_conv_tune) */
  int  stages;                /* default 0.  pipeline depth override (fami_conv_tune_stages) */
  int  use32;                 /* default 1.  fami_conv_tune(-1, ...) disables the 32x32-tile f32 kernel (benchmarks / tests) */
  int  xcd_w;                 /* default 1.  same switch for the weight-gradient kernels (fami_conv_tune_xcd bit 1) */
  int  xcd;                   /* default -1.  fami_conv_tune_xcd: 0 natural tile order, 1 XCD-contiguous, -1 default (= 1: PMC FETCH_SIZE of the */
  int  use_lds;               /* default -1 */
  int  lds_sim;               /* default 0.  fami_conv_tune_lds(2): LDS kernel in its split-operand cost-simulation form (benchmarks) */
  int  wgrad_nsub;            /* default 2.  sub-chunks per workgroup of the 16-bit LDS wgrad (fewer, larger partial slabs): 2 = half the slab */
  int  wgrad_ps;              /* default 0.  fami_conv_tune_wgrad_lds(1000 + n): pixel-split target of the per-tap f32 wgrad (benchmarks) */
  int  wgrad_mt;              /* default 0.  fami_conv_tune_wgrad_lds(100 + mt): cap on input-channel tiles per f32 wgrad workgroup */
  int  wgrad_lds;             /* default 1.  fami_conv_tune_wgrad_lds(0): bf16 weight gradients on the scalar-operand kernels */
  int  wgrad_lds_f32;         /* default 2.  f32 LDS weight gradient: 0 never, 1 whenever eligible, 2 only where it measured faster */
  int  wgrad_lin;             /* default 1.  fami_conv_tune_wgrad_lds(50 / 51): linear-address per-tap f32 kernel off / on */
  /* ---- conv_stem.hip */
  int  stem1;                 /* default 1.  fami_conv_tune_lds(9000 / 9001): off / on */
  /* ---- conv_t4.hip */
  int  use_t4;                /* default 1 */
  int  t4_dil;                /* default 1.  fami_conv_tune_lds(40 / 41): dilated 3x3 convolutions on the band kernels off / on */
  int  use_t4_f32;            /* default 0.  fami_conv_tune_lds(20 / 21): the f32 instance off / on.  Off by default: per launch it wins where */
  int  t4_bt;                 /* default 0.  fami_conv_tune_lds(100 + bt): force the tiles per band (benchmarks) */
  int  t4_s3_narrow;          /* default 0.  fami_conv_tune_lds(102030 / 102031): two channel tiles per workgroup on launches of < 200 workgroups off / on. */
  int  t4_s3_fill;            /* default 2.  fami_conv_tune_lds(102000 / 102001 / 102002): more, smaller bands on launches that leave CUs empty: off / all / tiny ones. */
  int  t4_s3_pc;              /* default 0.  fami_conv_tune_lds(60 / 61): producer / consumer form of the split-product instance off / on */
  int  t4_s3_mt_minft;        /* default 64.  ... only for frames of at least this many tiles (24x18 maps: 27 tiles = one band of 24 + one of 3) */
  int  t4_s3_mt;              /* default 3.  fami_conv_tune_lds(52 / 53): pixel tiles per wave of the split-product instance.  3 (bands of <= 24 tiles, 15 LDS */
  int  t4_s3_minwg;           /* default 0.  fami_conv_tune_lds(2000 + n): the split-product instance only for launches of >= n workgroups (benchmarks) */
  int  s3_default;            /* default 1.  fami_tune_defaults: what fami_conv_tune_lds(-1) restores (FAMI_F32_SPLIT=0 -> 0) */
  int  use_t4_s3;             /* default 1.  fami_conv_tune_lds(30 / 31): f32 storage on the bf16 matrix pipe (split products, see the kernel) off / on */
  /* ---- conv_t5.hip */
  int  use_t5;                /* default 1.  fami_conv_tune_lds(7000 / 7001): off / on */
  int  t5_rows;               /* default 0.  fami_conv_tune_lds(7100 + R): force the rows per band (benchmarks) */
  int  t5_maxwg;              /* default 256.  fami_conv_tune_lds(7500 + n): at most 8 n workgroups in the persistent grid (benchmarks; 7599: one job per workgroup) */
  int  t5_h16;                /* default 0.  fami_conv_tune_lds(7010 / 7011): the 16-bit instances off / on */
  int  t5_abl;                /* default 0.  fami_conv_tune_lds(7700 + n): ablation, see ConvT5Args.abl_chunks */
  int  t5_min_jobs;           /* default 200.  fami_conv_tune_lds(7600 + n): only launches of >= n jobs */
  int  t5_min_tiles;          /* default 0.  fami_conv_tune_lds(7400 + n): only frames of >= 8 n tiles (7401: >= 1)  // fami_conv_tune_lds(7400 + n): only frames of >= n tiles (benchmarks / routing experiments) */
  /* ---- conv_t6.hip */
  int  use_t6;                /* default 1.  fami_conv_tune_lds(8000 / 8001): off / on */
  int  t6_rows;               /* default 0.  fami_conv_tune_lds(8100 + RB): force the rows per band (benchmarks) */
  int  t6_min_jobs;           /* default 96.  fami_conv_tune_lds(8400 + n): only launches of >= n jobs */
  int  t6_mt;                 /* default 0.  fami_conv_tune_lds(8201 / 8202): units of two / four rows (0: four where the band allows) */
  int  t7_target;             /* default 240.  fami_conv_tune_lds(8700 + n): workgroups of a launch of the phased kernel (jobs are dealt consecutively) */
  int  use_t7;                /* default 1.  fami_conv_tune_lds(8500 / 8501): off / on */
  int  t7_rows;               /* default 0.  fami_conv_tune_lds(8600 + RB): force the rows per band (benchmarks) */
  int  t7_c64;                /* default 1.  fami_conv_tune_lds(8502 / 8503): the 32-channel-phase instances of the phased kernel (layers of 64-multiple channels: HRNet-W64, stage 1's 64 -> 64) off / on */
  int  bn_in;                 /* default 1.  fami_conv_tune_lds(8996 / 8997): the BatchNorm + ReLU in front of a 48-channel 3x3 convolution inside its launch (conv3x3_t6_kernel XB instances, fami_conv2d_fwd_bnin_*) off / on */
  /* ---- conv_pair.hip */
  int  bwd_pair;              /* default 1.  fami_conv_tune_lds(8998 / 8999): input gradient + weight gradient of a 3x3 stride-1 convolution (16-bit storage) as ONE launch off / on (conv_pair.h) */
  int  pair_wg6_target;       /* default 64.  fami_conv_tune_wgrad_lds(27000 + n): workgroup target of the weight-gradient half of a combined launch (0: wg6_target / wg6_target_c4).  bf16 step (tools/ab_env.py, one box): 48 / 80 / 120 / 160 / 240 -> 18.36 / 18.40 / 18.65 / 19.08 / 19.68 ms (two launches: 18.92); second box 24 / 32 / 40 / 48 / 64 -> 19.36 / 18.55 / 18.56 / 18.50 / 18.36 */
  /* ---- conv_wg16.hip */
  int  wg6_s2;                /* default 1.  fami_conv_tune_wgrad_lds(23004 / 23005): stride-2 launches off / on */
  int  wg6_dil;               /* default 1.  fami_conv_tune_wgrad_lds(23008 / 23009): the dilated (48 -> 216 / 108, dilation 3) launches off / on */
  int  wg6_c42;               /* default 1 */
  int  wg6_c4;                /* default 1.  fami_conv_tune_wgrad_lds(23002 / 23003): the 64-channel blocks off / on */
  int  wg6;                   /* default 1.  fami_conv_tune_wgrad_lds(23000 / 23001): off / on; 23100 + n: units per workgroup; 23400 + n: workgroup target */
  int  wg6_nu;                /* default 0.  fami_conv_tune_wgrad_lds(23000 / 23001): off / on; 23100 + n: units per workgroup; 23400 + n: workgroup target */
  int  wg6_target;            /* default 80.  fami_conv_tune_wgrad_lds(23000 / 23001): off / on; 23100 + n: units per workgroup; 23400 + n: workgroup target */
  int  wg6_target_c4;         /* default 160.  fami_conv_tune_wgrad_lds(26000 + n): workgroup target of the launches with 64-channel input blocks (0: wg6_target) */
  int  wg1;                   /* default 1.  inside the bf16 step (tools/ab_env.py): 48 / 96 / 192 workgroups 22.63 / 22.54 / 22.45 ms against 22.96 without the kernel      // fami_conv_tune_wgrad_lds(24000 / 24001): off / on; 24100 + n: workgroup target */
  int  wg1_target;            /* default 192.  inside the bf16 step (tools/ab_env.py): 48 / 96 / 192 workgroups 22.63 / 22.54 / 22.45 ms against 22.96 without the kernel      // fami_conv_tune_wgrad_lds(24000 / 24001): off / on; 24100 + n: workgroup target */
  int  wgs;                   /* default 1.  fami_conv_tune_wgrad_lds(25000 / 25001): off / on */
  int  wg16_abl;              /* default 0 */
  int  wg16;                  /* default 1.  18-tile aligned runs: per launch 27.6 -> 25.0 us (48 ch @96x72), inside the bf16 step 26.10 -> 26.24 / 25.99 -> 26.09 ms: off */
  int  wg16_bt;               /* default 0.  18-tile aligned runs: per launch 27.6 -> 25.0 us (48 ch @96x72), inside the bf16 step 26.10 -> 26.24 / 25.99 -> 26.09 ms: off */
  int  wg16_target;           /* default 0.  18-tile aligned runs: per launch 27.6 -> 25.0 us (48 ch @96x72), inside the bf16 step 26.10 -> 26.24 / 25.99 -> 26.09 ms: off */
  int  wg16_general;          /* default 1.  18-tile aligned runs: per launch 27.6 -> 25.0 us (48 ch @96x72), inside the bf16 step 26.10 -> 26.24 / 25.99 -> 26.09 ms: off */
  int  wg16_bt18;             /* default 0.  18-tile aligned runs: per launch 27.6 -> 25.0 us (48 ch @96x72), inside the bf16 step 26.10 -> 26.24 / 25.99 -> 26.09 ms: off */
  /* ---- conv_wgs3.hip */
  int  wgs3;                  /* default 1 */
  int  wgs3_bt;               /* default 0 */
  int  wgs3_target;           /* default 0 */
  int  wgs3_default;          /* default 1.  fami_tune_defaults (FAMI_F32_SPLIT) */
  /* ---- norm.hip */
  long bn_small_elems;        /* default 32768.  fami_bn_tune_small: tensors up to this many elements take the one-launch kernels */
  int  bn2_maxg;              /* default 512.  fami_bn_tune_small(-(1000 + n)): most workgroups of the statistics passes of the two-launch BatchNorm forms (bn_partial2 / bn_bwd_partial2) in 16-bit storage; four times as many in f32 storage */
} fami_route_t;

#ifdef __cplusplus
}
#endif
#endif
