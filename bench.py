#!/usr/bin/env python
"""Headline benchmark: train clips/sec of the FAMI-Pose temporal-alignment step
(5-frame 384x288 HRNet-W48 clips, MI loss on, backbone unfrozen, Adam) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One "step" = one optimisation step on a per-GPU batch of synthetic clips already resident
in HBM (targets are generated on device from synthetic joints inside the step).
Prints ONE JSON line on rank 0 (contract in the task statement): value = whole-job clips/s,
plus `roofline` (dominant kernel, live HIP-event timing) and, at N=1, `cpu_baseline`
(the oracle's CPU restatement timed on the host cores; baseline only).
"""
import argparse
import json
import gc
import os
import sys
import time
# ROCm runtime: kernel arguments in device memory -- 2-3 us less launch latency per kernel; with ~4000 dependent launches
# per training step that is -4 % (f32) / -5 % (bf16) step time on MI355X.  Must be set before the HIP runtime initialises.
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32-input MFMA dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 / fp16 MFMA dense peak (not the 2:1-sparsity figure)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak


def synth_batch(B, S, H, W, J, dev, seed):
    g = torch.Generator().manual_seed(seed)
    kf = torch.randn(B, 3, H, W, generator=g)
    sup = torch.randn(B, 3 * S, H, W, generator=g)
    joints = torch.rand(B, J, 2, generator=g) * torch.tensor([W, H], dtype=torch.float32)
    vis = (torch.rand(B, J, generator=g) < 0.8).float()
    return kf.to(dev), sup.to(dev), joints.to(dev), vis.to(dev)


def build(args, dev):
    import fami_pose_amd as fp
    from fami_pose_amd.init import realistic_init_     # weights at realistic scale
    cfg = fp.default_cfg(args.width, image_size=(args.img_w, args.img_h), num_sup=args.sup,
                         freeze_backbone=args.freeze_backbone)
    torch.manual_seed(19970808)
    model = fp.build_model(cfg, 'train')
    realistic_init_(model, seed=19970808)
    model.set_compute_dtype(args.dtype)
    if args.deterministic:
        model.set_deterministic(True)
    return model.to(dev)


def pmc_record(key):
    """The committed rocprofv3 --pmc record of the launch the roofline times (profiles/pmc_traffic.json, written by
    tools/pmc_sets.sh + tools/pmc_json.py on the MI355X): MFMA instruction count and matrix-pipe busy share."""
    try:
        return json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'))).get(key)
    except Exception:
        return None


def pmc_traffic(key):
    """HBM bytes per launch measured with rocprofv3 --pmc (profiles/pmc_traffic.json; separate FETCH_SIZE / WRITE_SIZE
    passes, gfx950 wide-read correction applied) for the SAME launch the roofline times; None if not on record."""
    try:
        rec = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))[key]
        return rec['read_bytes_corrected'] + rec['write_bytes']
    except Exception:
        return None


TORCH_DT = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}


def in_step(dtype):
    """What the committed rocprofv3 kernel trace of this round says about the step as a whole (profiles/in_step.json,
    written by tools/in_step_summary.py from the trace of `bench.py --dtype <dtype>` in graph mode): launches per step,
    the dominant kernel's average duration INSIDE the step, the share of the step's wall time with no kernel in flight."""
    try:
        return json.load(open(os.path.join(ROOT, 'profiles', 'in_step.json'))).get(dtype)
    except Exception:
        return None


def _time_launches(launch, s, reps):
    for _ in range(5):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps):
        launch()
    e1.record(s)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def conv_roofline(dev, N, dtype, reps=30, C=48, H=96, W=72):
    """Live HIP-event timing of the dominant kernel: the C->C 3x3 branch conv of the highest-resolution branch (26 % of
    the step's conv FLOPs at W48, 64 forward launches per step) on the stream it is launched on."""
    from fami_pose_amd._lib import lib
    L = lib()
    half = dtype != 'f32'
    tdt = TORCH_DT[dtype]
    x = torch.randn(N, H, W, C, device=dev).to(tdt)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    y = torch.empty(N, H, W, C, device=dev, dtype=tdt)
    s = torch.cuda.current_stream(dev)
    if half:
        wp = torch.empty(L.cdll.fami_packed_weight_elems_bf16(C, C, 3, 3, 0), device=dev, dtype=tdt)
        L.call('fami_pack_conv_weight_' + dtype, w.data_ptr(), wp.data_ptr(), C, C, 3, 3, 0, s.cuda_stream)

        def launch():
            L.call('fami_conv2d_fwd_' + dtype, x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1,
                   1, 0, 0, 0, s.cuda_stream)
    else:
        wp = torch.empty(L.cdll.fami_packed_weight_elems(C, C, 3, 3, 0), device=dev)
        L.call('fami_pack_conv_weight_f32', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, 0, s.cuda_stream)

        def launch():
            L.call('fami_conv2d_fwd_f32', x.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, 3, 3, 1,
                   1, 1, 0, 0, s.cuda_stream)
    ms = _time_launches(launch, s, reps)
    flops = 2.0 * N * H * W * C * 9 * C
    ach = flops / (ms * 1e-3) / 1e12
    # the library's default route for this shape is the register-blocked LDS kernel of conv_t4.hip in every storage type.
    # f32 storage: its split-product instance -- every f32 operand split exactly into three bf16 terms, six products per
    # f32 product on v_mfma_f32_16x16x32_bf16, fp32 accumulation -- so the peak that bounds it is a sixth of the dense
    # bf16 MFMA peak; the exact-f32 MFMA (v_mfma_f32_16x16x4_f32, peak 157.3) route is timed beside it.
    split = (not half) and os.environ.get('FAMI_F32_SPLIT', '1') != '0'
    t5 = split and os.environ.get('FAMI_T5', '1') != '0' and bool(L.cdll.fami_conv_t5_eligible(N, H, W, C, C))
    peak = PEAK_BF16_MFMA_TFLOPS if half else (round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1) if split else PEAK_F32_MFMA_TFLOPS)
    t6 = L.cdll.fami_conv_t6_eligible(N, H, W, C, C) if half else 0      # round 4: weight-resident, LDS-DMA-staged kernels (conv_t6.hip)
    kname = (({1: 'conv3x3_t6_kernel<%s, weight-resident, LDS DMA>', 2: 'conv3x3_t7_kernel<%s, 48-channel phases, LDS DMA>',
               3: 'conv3x3_t7_kernel<%s, 32-channel phases, LDS DMA>'}.get(t6, 'conv3x3_t4_kernel<%s>')) % dtype) if half else (
        ('conv3x3_t5_kernel<float, persistent split-product>' if t5 else 'conv3x3_t4_kernel<float, split-product>') if split else 'conv_igemm_f32')
    key = {'f32': ('conv3x3_t5_s3_f32' if t5 else 'conv3x3_t4_s3_f32') if split else 'conv_igemm_f32',
           'bf16': 'conv3x3_t6_bf16' if t6 else 'conv3x3_t4_bf16'}.get(dtype)
    abytes = int(2 * x.numel() * x.element_size())
    out = {"bound": "mfma", "kernel": "%s (%d->%d 3x3 @%dx%d, N=%d frames)" % (kname, C, C, H, W, N),
           "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
           "frac": round(ach / peak, 4), "traffic": pmc_traffic(key) if (N == 20 and C == 48 and H == 96 and key) else None,
           "algorithmic_bytes": abytes, "avg_launch_us": round(ms * 1e3, 2),
           # the SAME launch against the other roof: x read once + y written once over the live time vs the HBM peak.  At 48
           # channels the launch is below the ridge (216 FLOP/B against 312): its HBM floor is above its MFMA floor, so
           # hbm_frac is the tighter of the two bounds there
           "hbm_frac": round(abytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
    if not half:
        # f32 storage: `frac` prices the launch against the pipe it runs on (bf16 matrix pipe / 6 products);
        # frac_vs_guide_peak against the guide's f32-input MFMA peak (157.3 TFLOP/s), which an exact-f32 kernel is bound by
        out["frac_vs_guide_peak"] = round(ach / PEAK_F32_MFMA_TFLOPS, 4)
    rec_p = pmc_record(key) if (N == 20 and C == 48 and H == 96 and key) else None
    if rec_p and rec_p.get('mfma_busy') is not None:
        # north_star: "MFMA-busy reported against gfx950 peaks" -- SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles) of
        # the same launch, and the MFMA instruction count against 2 N H W C 9 C / 16384 (x 6 products in the split form).
        # RECORDED values (a rocprofv3 --pmc run committed under profiles/), not measured in this process: they describe
        # the kernel of `pmc_source`; `pmc_kernel_matches` says whether that is the kernel timed above.
        want = kname.split('<')[0]
        out["pmc_recorded"] = {"mfma_busy": rec_p['mfma_busy'], "mfma_insts": rec_p.get('sq_insts_mfma'),
                               "source": rec_p.get('source'), "kernel": rec_p.get('workload'),
                               "git_sha": rec_p.get('git_sha'),
                               "kernel_matches_live_route": want in (rec_p.get('workload') or '')}
        out["mfma_insts_expected"] = int(flops / 16384 * (6 if split else 1)) if (half or split) else int(flops / 2048)
    if split:
        out["peak_note"] = ("dense bf16 MFMA peak %.0f / 6 products per f32 product; achieved counts each f32 "
                            "multiply-add once" % PEAK_BF16_MFMA_TFLOPS)
        L.cdll.fami_conv_tune_lds(30)
        try:
            ms_x = _time_launches(launch, s, reps)
        finally:
            L.cdll.fami_conv_tune_lds(31)
        out["exact_f32_mfma"] = {"kernel": "conv_igemm_f32 (v_mfma_f32_16x16x4_f32)", "avg_launch_us": round(ms_x * 1e3, 2),
                                 "achieved": round(flops / (ms_x * 1e-3) / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                                 "frac": round(flops / (ms_x * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
    rec = in_step(dtype)
    if rec and N == 20 and C == 48 and H == 96 and rec.get('dominant_avg_us'):
        # the same launch inside the training step (committed trace; other stream lanes contend for the chip)
        out["in_step_avg_us"] = rec['dominant_avg_us']
        out["frac_in_step"] = round(flops / (rec['dominant_avg_us'] * 1e-6) / 1e12 / peak, 4)
        out["in_step_source"] = rec.get('source')
        if rec.get('profiled_step_ms'):
            out["in_step_profiled_step_ms"] = rec['profiled_step_ms']      # main() divides by the timed step: in_step_profiler_inflation
    return out


def dcn_roofline(dev, B, dtype, reps=30, C=48, G=12, H=96, W=72):
    """Secondary rooflines: the fused DCNv2 gather+contraction forward and its fused backward (HBM-bound; SURVEY.md 8d
    algorithmic bytes: fwd (Ci + 3GK + Co) * HW * s, bwd (2Ci + 6GK + Co) * HW * s per sample per layer)."""
    from fami_pose_amd._lib import lib
    L = lib()
    tdt = TORCH_DT[dtype]
    sz = 4.0 if dtype == 'f32' else 2.0
    x = torch.randn(B, H, W, C, device=dev).to(tdt)
    off = torch.randn(B, H, W, 18 * G, device=dev).to(tdt)
    msk = torch.randn(B, H, W, 9 * G, device=dev).to(tdt)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    bias = torch.zeros(C, device=dev)
    y = torch.empty(B, H, W, C, device=dev, dtype=tdt)
    wp = torch.empty(L.cdll.fami_dcn_packed_weight_elems(C, C, 3, 3, G), device=dev)
    s = torch.cuda.current_stream(dev)
    L.call('fami_dcn_pack_weight_' + dtype, w.data_ptr(), wp.data_ptr(), C, C, 3, 3, G, s.cuda_stream)

    def launch():
        L.call('fami_dcn_fwd_' + dtype, x.data_ptr(), off.data_ptr(), msk.data_ptr(), wp.data_ptr(), bias.data_ptr(),
               y.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, s.cuda_stream)
    ms = _time_launches(launch, s, reps)
    nbytes = (C + 3 * G * 9 + C) * H * W * sz * B
    ach = nbytes / (ms * 1e-3) / 1e9
    fwd = {"bound": "hbm", "kernel": "dcn_fwd_direct_kernel (%dch, %d groups, %dx%d, B=%d, %s)" % (C, G, H, W, B, dtype),
           "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
           "traffic": pmc_traffic('dcn_fwd_' + dtype) if (B == 4 and C == 48 and H == 96) else None,
           "algorithmic_bytes": int(nbytes), "avg_launch_us": round(ms * 1e3, 2)}
    # What a gather of this granularity can reach (review r5 item 3d): every sample point reads four corners, each a (group, pixel)
    # block of Ci / G channels = 16 bytes in f32 / 8 bytes in 16-bit storage at an offset-dependent address -- one vector-L1 line
    # lookup per corner and lane, and a CU's L1 serves one line per clock.  B * H * W * G * 9 * 4 corner loads over 256 CUs at
    # 2.4 GHz bound the launch from below whatever the HBM does; `ceiling` = algorithmic bytes / that time, as a fraction of the
    # HBM peak (the 0.60 target of the north star would need the 8- / 16-byte blocks to arrive at > 2x the L1's line rate).
    t_lines = B * H * W * G * 9 * 4 / (256 * 2.4e9)
    fwd["ceiling"] = {"frac": round(nbytes / t_lines / 1e9 / PEAK_HBM_GBS, 4), "min_launch_us": round(t_lines * 1e6, 2),
                      "bound": "vector-L1 line rate (one 128-byte line per clock and CU) on %d scattered %d-byte corner loads" % (
                          B * H * W * G * 9 * 4, int(C // G * sz))}
    fwd["frac_of_ceiling"] = round(fwd["frac"] / fwd["ceiling"]["frac"], 3)
    fwd["ceiling"]["note"] = ("a bound, not the binding constraint: a probe that cut the line lookups to a fourth (group-major copy of x) ran "
                              "only 10 % (f32) / 24 % (bf16) faster -- the launch's staging and gather phases add up "
                              "(profiles/r06/bench_dcn_group_major_probe.txt)")
    # the opt-in LDS-window forward kernel on the same launch (measured, not the default)
    L.cdll.fami_dcn_tune(2)
    try:
        fwd["lds_window_kernel_us"] = round(_time_launches(launch, s, reps) * 1e3, 2)
    finally:
        L.cdll.fami_dcn_tune(-1)
    # backward (the launch the training step issues: column gradient, offset / mask gradients, input-gradient scatter,
    # modulated samples for the weight gradient)
    wpb = torch.empty(L.cdll.fami_dcn_packed_weight_bwd_elems(C, C, 3, 3, G), device=dev)
    L.call('fami_dcn_pack_weight_bwd_f32', w.data_ptr(), wpb.data_ptr(), C, C, 3, 3, G, s.cuda_stream)
    dy = torch.randn(B, H, W, C, device=dev).to(tdt)
    colw = L.cdll.fami_dcn_bwd_col_width(C, C, G, 3, 3, 1, 3, int(sz), 0)      # (the register-fed kernel's own column order, padded)
    col = torch.empty(B * H * W, colw, device=dev, dtype=tdt)
    gx = torch.zeros(B, H, W, C, device=dev)
    goff, gmsk = torch.empty_like(off), torch.empty_like(msk)

    def launch_b():
        L.call('fami_dcn_bwd_' + dtype, x.data_ptr(), off.data_ptr(), msk.data_ptr(), dy.data_ptr(), wpb.data_ptr(),
               col.data_ptr(), gx.data_ptr(), goff.data_ptr(), gmsk.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, 0,
               s.cuda_stream)
    msb = _time_launches(launch_b, s, reps)
    nb = (2 * C + 6 * G * 9 + C) * H * W * sz * B
    achb = nb / (msb * 1e-3) / 1e9
    bwd = {"bound": "hbm", "kernel": "%s (%dch, %d groups, %dx%d, B=%d, %s; %d-bit fixed-point LDS scatter, f32 atomic flush)" % (
               "dcn_bwd2_kernel <register-fed>" if L.cdll.fami_dcn_bwd_col_permuted(C, C, G, 3, 3, 1, 3, int(sz), 0) else "dcn_bwd_kernel", C, G, H, W, B, dtype, 64 if dtype == 'f32' else 32),
           "achieved": round(achb, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(achb / PEAK_HBM_GBS, 4),
           "traffic": pmc_traffic('dcn_bwd_' + dtype) if (B == 4 and C == 48 and H == 96) else None,
           "algorithmic_bytes": int(nb), "avg_launch_us": round(msb * 1e3, 2)}
    # the backward gathers the same corners once more and adds as many read-modify-writes into its LDS regions: the forward's bound
    # holds for it with the backward's bytes
    bwd["ceiling"] = {"frac": round(nb / t_lines / 1e9 / PEAK_HBM_GBS, 4), "min_launch_us": round(t_lines * 1e6, 2),
                      "bound": "vector-L1 line rate on the corner gather alone (the LDS scatter and its atomic flush come on top)"}
    bwd["frac_of_ceiling"] = round(bwd["frac"] / bwd["ceiling"]["frac"], 3)
    # the deterministic (64-bit fixed-point) form of the same backward: zero + |dy| max + kernel + conversion pass
    gxd = torch.empty(B, H, W, C, device=dev, dtype=tdt)
    ws = torch.empty(L.cdll.fami_dcn_bwd_det_workspace(B, H, W, C) // 4 + 4, device=dev)
    col = torch.empty(B * H * W, C * 9, device=dev, dtype=tdt)          # (the deterministic form writes the OIHW column order)

    def launch_d():
        L.call('fami_dcn_bwd_det_' + dtype, x.data_ptr(), off.data_ptr(), msk.data_ptr(), dy.data_ptr(), wpb.data_ptr(),
               col.data_ptr(), gxd.data_ptr(), goff.data_ptr(), gmsk.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, 0, 0,
               ws.data_ptr(), s.cuda_stream)
    msd = _time_launches(launch_d, s, reps)
    bwd["deterministic_avg_us"] = round(msd * 1e3, 2)
    return fwd, bwd


def cpu_baseline_worker(args):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores: ONE clip,
    forward + loss + backward + Adam, fp32.  Baseline only.  Runs in its own process (see cpu_baseline)."""
    from oracle import model as om, ops as oops
    cores = min(os.cpu_count() or 1, args.cpu_threads if args.cpu_threads > 0 else (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = om.AlignmentOracle(om.make_cfg(args.width), True, args.sup, (args.img_h, args.img_w))
    om.realistic_init_(m, 1)
    oops.DCN_IMPL = 'gridsample'      # faster CPU formulation of the same op (cross-checked in tests/test_oracle.py)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    B = 1
    kf = torch.randn(B, 3, args.img_h, args.img_w)
    sup = torch.randn(B, 3 * args.sup, args.img_h, args.img_w)
    tgt = torch.rand(B, 17, args.img_h // 4, args.img_w // 4)
    w = torch.ones(B, 17, 1)
    def step():
        f, k, mi = m(kf, sup)
        loss = oops.total_loss(f, tgt, w, mi)
        opt.zero_grad()
        loss.backward()
        opt.step()
    step()                                  # warm-up (thread pool, allocator), untimed
    nstep = 3
    t0 = time.time()
    for _ in range(nstep):
        step()
    dt = (time.time() - t0) / nstep
    print(json.dumps({"value": round(B / dt, 4), "unit": "clips/s", "cores": cores, "kind": "port",
                      "sample": "1 clip (%d-frame %dx%d W%d) per step, 3 timed fwd+loss+bwd+Adam steps after 1 warm-up, "
                                "fp32, torch CPU (%d threads of %d host cores), %.1f s/step" %
                                (args.sup + 1, args.img_h, args.img_w, args.width, cores, os.cpu_count() or 1, dt)}),
          flush=True)


def cpu_baseline(args):
    """Run the CPU baseline in a child process with a hard time limit, so a slow or oversubscribed host can
    never stall the benchmark line (the child never touches the GPU)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--sup', str(args.sup), '--width',
           str(args.width), '--img-h', str(args.img_h), '--img-w', str(args.img_w), '--cpu-threads',
           str(args.cpu_threads)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.cpu_timeout)
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith('{'):
                return json.loads(line)
        return {"value": None, "unit": "clips/s", "cores": args.cpu_threads, "kind": "port",
                "sample": "failed: " + out.stderr.strip()[-200:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "clips/s", "cores": args.cpu_threads, "kind": "port",
                "sample": "1 clip fwd+bwd+Adam did not finish within %d s" % args.cpu_timeout}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)      # >= 20 warm-up + >= 50 timed steps
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=4, help='clips per GPU')
    ap.add_argument('--sup', type=int, default=4, help='supporting frames (4 => 5-frame clips)')
    ap.add_argument('--width', type=int, default=48)
    ap.add_argument('--img-h', type=int, default=384)
    ap.add_argument('--img-w', type=int, default=288)
    ap.add_argument('--freeze-backbone', action='store_true')
    ap.add_argument('--no-frozen', action='store_true', help='skip the extra frozen-backbone measurement')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--bucket-mb', type=int, default=32)
    ap.add_argument('--also', choices=['f32', 'bf16', 'f16', 'none'], default='bf16',
                    help='additionally time this dtype (fewer steps) and report it as an extra "also_<dtype>" object')
    ap.add_argument('--dtype', choices=['f32', 'bf16', 'f16'], default='f32',
                    help='activation storage / conv MFMA type (accumulation, master weights, losses are fp32 either way)')
    ap.add_argument('--cpu-baseline-only', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-threads', type=int, default=16,
                    help='host threads for the CPU baseline (0 = every core).  Default 16: the thread sweep on the 256-thread '
                         'GPU host (profiles/r02_cpu_threads.txt: 8 / 16 / 32 / 64 / 128 threads = 0.68 / 0.81 / 0.48 / 0.19 / '
                         '0.05 clips/s) peaks there -- beyond it torch\'s CPU backward of this many small convolutions '
                         'loses more to thread hand-offs across the sockets than it gains')
    ap.add_argument('--deterministic', action='store_true', help='fixed-point DCN backward (bitwise reproducible steps)')
    ap.add_argument('--cpu-timeout', type=int, default=240)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        return cpu_baseline_worker(args)
    if args.also == 'none':
        args.also = None

    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    # FAMI_DIST_BACKEND=gloo: validation mode for boxes with fewer GPUs than ranks (ranks share devices round-robin and
    # exchange gradients through gloo); the measured configuration is one rank per GPU over RCCL ('nccl')
    backend = os.environ.get('FAMI_DIST_BACKEND', 'nccl')
    if backend != 'nccl':
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from fami_pose_amd.train import Trainer

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # N>1: hipGraph(forward + backward) -> bucketed RCCL all-reduce of the gradient arena -> hipGraph(scale + Adam);
    # FAMI_DDP_GRAPH=0 (or --no-graph) selects the eager sequence with the all-reduce overlapped bucket by bucket.
    use_graph = (not args.no_graph) and (world == 1 or os.environ.get('FAMI_DDP_GRAPH', '1') != '0')
    kf, sup, joints, vis = synth_batch(args.batch, args.sup, args.img_h, args.img_w, 17, dev, 19970808 + rank)

    def make_trainer():
        """-> (trainer, model, plan name, fallback notes).  N>1: the launch plans in order of preference -- segmented graphs
        with overlapped all-reduce, serial graphs, eager hooks; a plan whose capture / first step fails on ANY rank is
        dropped by all of them (agreed through a MIN all-reduce), so a scaling run reports a number and says which plan
        produced it instead of dying in the capture."""
        if world == 1:
            model = build(args, dev)
            return Trainer(model, lr=1e-3, use_mi=True, use_graph=use_graph, targets_from_joints=True,
                           bucket_mb=args.bucket_mb), model, None, []
        first = os.environ.get('FAMI_DDP_PLAN', 'overlap')
        plans = ([(first, True)] + [(q, True) for q in ('overlap', 'serial') if q != first and first == 'overlap']
                 if use_graph else []) + [('eager-hooks', False)]
        notes = []
        for name, ug in plans:
            if ug:
                os.environ['FAMI_DDP_PLAN'] = name
            ok, trainer, model = 1, None, None
            try:
                model = build(args, dev)
                trainer = Trainer(model, lr=1e-3, use_mi=True, use_graph=ug, targets_from_joints=True, bucket_mb=args.bucket_mb)
                trainer.step(kf, sup, joints, vis)
                torch.cuda.synchronize(dev)
            except Exception as e:                       # noqa: BLE001 -- reported in the JSON line
                ok = 0
                notes.append('%s failed on rank %d: %s' % (name, rank, repr(e)[:200]))
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                return trainer, model, name, notes
            if ok:
                notes.append('%s dropped: failed on another rank' % name)
            del trainer, model
            torch.cuda.empty_cache()
        raise SystemExit('no data-parallel launch plan ran: ' + '; '.join(notes))

    def timed_run(dtype, steps, warmup):
        args.dtype = dtype
        trainer, model, plan_name, plan_notes = make_trainer()
        for _ in range(max(warmup, 1)):
            trainer.step(kf, sup, joints, vis)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            trainer.step(kf, sup, joints, vis)
        barrier()
        dt = time.perf_counter() - t0
        rank_ms = None
        if world > 1:
            # every rank's own wall time of the timed region: a straggler shows as max >> min (value uses the max)
            ts = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(ts, torch.tensor([dt], device=dev, dtype=torch.float64))
            per = [t.item() for t in ts]
            dt = max(per)
            rank_ms = [round(v / steps * 1e3, 3) for v in per]
        loss = trainer.loss_value()
        if not (loss == loss and abs(loss) < 1e30):
            # a throughput measured on NaN weights is not a measurement (round 4: a wrong-dtype gradient seed went unnoticed
            # in the bf16 line because NaN arithmetic runs at full speed)
            raise RuntimeError('bench: the training loss is not finite after %d steps (%r)' % (steps, loss))
        info = {"pck_final": round(trainer.accuracy()[0][1], 4), "pck_kf_backbone": round(trainer.accuracy()[1][1], 4),
                "conv_flops_per_step": int(trainer.conv_flops)}
        if rank_ms is not None:
            info.update({"rank_ms_per_step_min": min(rank_ms), "rank_ms_per_step_max": max(rank_ms), "rank_ms_per_step": rank_ms})
        if world > 1:
            info.update({"dist_ranks": dist.get_world_size(), "dist_backend": dist.get_backend(),
                         "ddp_plan": plan_name, "ddp_plan_fallbacks": plan_notes, "bucket_mb": args.bucket_mb,
                         "buckets": len(trainer.reducer.ranges()),
                         "gradient_bytes": int(trainer.grad.numel() * 4),
                         "gradient_payload": trainer.payload_name, "allreduce_algo": trainer.reducer.algo,
                         **trainer.plan_summary(),
                         "allreduce_ms_standalone": round(trainer.measure_allreduce_ms(), 3)})
            # what the exchange should cost over xGMI (point-to-point, 7 links x ~153 GB/s per GPU, MI355X_MICROARCH.md):
            # a ring is bound by one link, 2 (N-1)/N B / link; a direct reduce-scatter + all-gather over the full mesh
            # moves B/N per link and phase
            wire = trainer.grad.numel() * (4 if trainer.payload_name == 'f32' else 2)
            info.update({"allreduce_ms_expected_ring": round(2 * (world - 1) / world * wire / 153e9 * 1e3, 3),
                         "allreduce_ms_expected_mesh": round(2 * wire / world / 153e9 * 1e3, 3),
                         "allreduce_wire_bytes": int(wire)})
        del trainer, model
        gc.collect()                 # (the Trainer is a reference cycle: its hipGraph goes now, not inside the next capture)
        torch.cuda.empty_cache()
        return dt, loss, info

    primary = args.dtype
    dt, loss, info = timed_run(primary, args.steps, args.warmup)
    other = None
    if args.also and args.also != primary and world == 1:      # the extra dtype line is a single-GPU report
        o_steps = max(3, min(args.steps, 20))
        o_dt, o_loss, o_info = timed_run(args.also, o_steps, max(2, min(args.warmup, 5)))
        other = (args.also, o_dt, o_loss, o_steps, o_info)
    args.dtype = primary
    det = None
    if world == 1 and not args.deterministic and not args.no_frozen:
        # the price of run-to-run reproducibility (review r5 item 6c): the same step with the fixed-point DCN input gradient and the
        # three-launch BatchNorm forms (every kernel then has a fixed summation order; tests/test_train_gpu.py pins it bit for bit)
        args.deterministic = True
        d_steps = max(3, min(args.steps, 20))
        d_dt, d_loss, _ = timed_run(primary, d_steps, max(2, min(args.warmup, 5)))
        args.deterministic = False
        det = (d_dt, d_loss, d_steps)
    frozen = None
    if world == 1 and not args.freeze_backbone and not args.no_frozen:
        # the reference default freezes the backbone (Base_PoseTrack17.yaml:28); reported beside the
        # unfrozen headline (backbone forward only + the head's forward/backward/Adam)
        args.freeze_backbone = True
        f_steps = max(3, min(args.steps, 20))
        f_dt, f_loss, _ = timed_run(primary, f_steps, max(2, min(args.warmup, 5)))
        args.freeze_backbone = False
        frozen = (f_dt, f_loss, f_steps)

    if rank == 0:
        clips = args.batch * world * args.steps
        out = {
            "metric": "train clips/sec (%d-frame %dx%d HRNet-W%d)" % (args.sup + 1, args.img_h, args.img_w, args.width), "value": round(clips / dt, 3), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "HRNet-W%d %dx%d, %d-frame clips + MI loss, batch %d/GPU, backbone %s, Adam, "
                                   "on-device Gaussian targets" % (args.width, args.img_h, args.img_w, args.sup + 1,
                                                                   args.batch, "frozen" if args.freeze_backbone else "unfrozen"),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "hipgraph": use_graph, **({} if backend == 'nccl' or world == 1 else {"dist_backend": backend})},
            "loss": round(loss, 6), **info,
        }
        if det is not None:
            d_dt, d_loss, d_steps = det
            out["value_deterministic"] = {"value": round(args.batch * world * d_steps / d_dt, 3), "unit": "clips/s", "steps": d_steps,
                                          "ms_per_step": round(d_dt / d_steps * 1e3, 3), "loss": round(d_loss, 6),
                                          "note": "same workload and dtype with --deterministic (bitwise reproducible steps); `value` is "
                                                  "the default, non-deterministic step (float-atomic DCN input gradient, fp64 slot atomics)"}
        if args.dtype == 'f32':
            out["conv_arithmetic"] = (
                "f32 storage and accumulation; 3x3 stride-1 convolutions (forward, input and weight gradient): each f32 "
                "operand split exactly into three bf16 terms, six products on v_mfma_f32_16x16x32_bf16 (error vs fp64 1-2.5x "
                "the exact-f32 MFMA path's, tests/test_kernels_gpu.py); all other convolutions v_mfma_f32_16x16x4_f32"
                if os.environ.get('FAMI_F32_SPLIT', '1') != '0' else "f32 storage, v_mfma_f32_16x16x4_f32 everywhere")
        C, Hf, Wf = args.width, args.img_h // 4, args.img_w // 4
        G = 12 if C % 48 == 0 else C // 4
        def with_inflation(r, step_ms):
            # the committed in-step averages come from a rocprofv3 kernel trace, which slows the step down: the ratio of the
            # profiled step to the step timed here says by how much frac_in_step understates the kernel
            if r.get("in_step_profiled_step_ms"):
                r["in_step_profiler_inflation"] = round(r["in_step_profiled_step_ms"] / step_ms, 3)
            return r
        out["roofline"] = with_inflation(conv_roofline(dev, args.batch * (args.sup + 1), args.dtype, C=C, H=Hf, W=Wf),
                                         dt / args.steps * 1e3)

        def step_roofline(dtype, flops, ms):
            # every nn.Conv2d FLOP of the step (forward + input gradient + weight gradient, counted by the engine as the
            # launches are enqueued) over the measured step time, against the dtype's dense MFMA peak
            # (f32: the 3x3 stride-1 convolutions -- 90 % of those FLOPs -- run as six bf16 products per f32 product on the
            # bf16 matrix pipe, hence a sixth of the dense bf16 peak; the fraction of the exact-f32 MFMA peak beside it)
            split = dtype == 'f32' and os.environ.get('FAMI_F32_SPLIT', '1') != '0'
            peak = (round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1) if split else PEAK_F32_MFMA_TFLOPS) if dtype == 'f32' else PEAK_BF16_MFMA_TFLOPS
            r = {"bound": "mfma", "conv_flops_per_step": int(flops), "achieved": round(flops / (ms * 1e-3) / 1e12, 2),
                 "peak": peak, "unit": "TFLOP/s", "frac": round(flops / (ms * 1e-3) / 1e12 / peak, 4)}
            if split:
                r["frac_of_exact_f32_mfma_peak"] = round(flops / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
            rec = in_step(dtype)
            if rec:
                r.update({k: rec[k] for k in ('launches_per_step', 'idle_share', 'kernel_time_ms_per_step', 'source') if k in rec})
            return r
        out["roofline_step"] = step_roofline(args.dtype, info["conv_flops_per_step"], dt / args.steps * 1e3)   # per rank
        out["roofline_dcn"], out["roofline_dcn_bwd"] = dcn_roofline(dev, args.batch, args.dtype, C=C, G=G, H=Hf, W=Wf)
        if other is not None:
            o_dtype, o_dt, o_loss, o_steps, o_info = other
            out["also_" + o_dtype] = {
                "note": "same workload with %s activation storage / conv MFMA (fp32 accumulation, master weights, "
                        "losses); not the parity-gated configuration" % o_dtype if o_dtype != 'f32' else
                        "same workload in the fp32 parity configuration",
                "value": round(args.batch * world * o_steps / o_dt, 3), "unit": "clips/s", "steps": o_steps,
                "ms_per_step": round(o_dt / o_steps * 1e3, 3), "loss": round(o_loss, 6),
                **({"parity": "keypoint level only: PCK equal to the fp32 path's and peaks within 2 heatmap pixels for >= 95 % of joints "
                              "on a fitted model (tests/test_model_gpu.py::test_half_mode_keypoints_on_a_fitted_model); argmax indices are "
                              "bit-exact against the reference only in f32 storage (the `value` line)"} if o_dtype != 'f32' else {}),
                "roofline": with_inflation(conv_roofline(dev, args.batch * (args.sup + 1), o_dtype, C=C, H=Hf, W=Wf),
                                           o_dt / o_steps * 1e3),
                "roofline_step": step_roofline(o_dtype, o_info["conv_flops_per_step"], o_dt / o_steps * 1e3),
                "roofline_dcn": dcn_roofline(dev, args.batch, o_dtype, C=C, G=G, H=Hf, W=Wf)[0]}
        if frozen is not None:
            f_dt, f_loss, f_steps = frozen
            out["also_frozen_backbone"] = {
                "note": "same workload and dtype with MODEL.FREEZE_HRNET_WEIGHTS (the reference's default config): no "
                        "backbone backward / weight gradients, head trained",
                "value": round(args.batch * world * f_steps / f_dt, 3), "unit": "clips/s", "steps": f_steps,
                "ms_per_step": round(f_dt / f_steps * 1e3, 3), "loss": round(f_loss, 6)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
