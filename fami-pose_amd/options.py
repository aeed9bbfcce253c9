"""Every FAMI_* environment switch of the host layer, in ONE place: name, default, meaning.

The package reads the environment only through `get()` / `flag()` / `number()` below; a name that is not registered here
raises, so this table is the complete list (INTEGRATION.md points at it).  Kernel-routing state is NOT here: that is the
`fami_route_t` struct of include/fami_route.h, owned per Engine.  The switches below choose how the HOST arranges launches
(stream lanes, which passes are fused, launch plans of the data-parallel step) and a few measurement ablations.
Values are read when an Engine / Trainer is built (not cached), so tools can flip them between two Trainers of one process.
"""
import os

# name: (default, meaning).  Default None = "depends on the compute mode", resolved by the caller (documented there).
SWITCHES = {
    # ---- stream lanes (engine.py)
    'FAMI_LANES': ('1', 'independent sub-graphs (the parallel HRNet branches) on side streams between fork / join'),
    'FAMI_FUSE_LANES': ('1', 'fuse terms f_ij(x_j) of a HighResolutionModule on the lane of their source branch'),
    'FAMI_REGRESSOR_LANES': ('1', 'the S shared-weight translation regressors on separate lanes'),
    'FAMI_MI_LANES': ('1', 'the MI terms of the loss on three lanes'),
    'FAMI_PERSIST_LANES': ('1', 'lanes stay forked across the modules of an HRNet stage'),
    'FAMI_MERGE_FORK': ('1', 'branches and the fuse terms that read them in one forked region (modules.HighResolutionModule.run_both)'),
    'FAMI_WGRAD_LANE': ('0', 'weight gradients of the stage convolutions off their lane: 1 on the weight-gradient streams in turn, 2 on one stream per lane (both measured slower)'),
    'FAMI_HEAD_WGRAD_LANE': ('1', 'weight gradients of the serial aggregation / DCN stack on their own stream'),
    'FAMI_STEM_WGRAD_LANE': ('1', 'the same for stem / layer1 / transitions (round 5: on in f32 too, -0.3 ... -0.5 %)'),
    'FAMI_HEAD_WGRAD_LANES': ('4', 'number of weight-gradient streams of the head (taken in turn)'),
    'FAMI_STEM_WGRAD_LANES': ('4', 'number of weight-gradient streams of that stretch (taken in turn)'),
    'FAMI_DEBUG_STREAMS': ('', 'print the stream handles of every lane set-up'),
    # ---- fused passes (engine.py)
    'FAMI_BN2': ('1', 'two-launch BatchNorm (fp64 slot atomics, finalize folded into the apply pass)'),
    'FAMI_FUSE_BN': ('auto', "BatchNorm statistics in the neighbouring convolutions' epilogues: auto (f32: forward everywhere; 16-bit: forward in the DMA-staged 3x3 kernels only + backward where the input gradient runs on one) | autoall (16-bit: forward everywhere) | 0 | fwd | fwd3 | bwd | bwdauto | 1"),
    'FAMI_FUSE_BN_T7': ('2', 'which launches of the phased 16-bit kernel carry the backward statistics under auto: 0 none | 1 the non-accumulating ones with a recomputed mask | 2 all'),
    'FAMI_FUSE_BN_SKIP': ('', "probe: convolution classes whose epilogue does not take the forward statistics ('1x1', 's2'; comma or + separated)"),
    'FAMI_FUSE_BN_C64': ('2', 'statistics in the epilogues of the 32-channel-phase kernel (layers of 64-multiple channels): bit 0 forward, bit 1 backward'),
    'FAMI_FUSE_TERM_BN2': ('1', 'fuse-term BatchNorm backward on the two-launch form'),
    'FAMI_BN_IN': ('1', 'conv1 -> bn1 -> ReLU -> conv2 (16-bit storage): bn1\'s apply pass inside conv2\'s launch where the weight-resident 48-channel kernel takes conv2 (fami_conv2d_fwd_bnin_*)'),
    'FAMI_XBN': (None, 'BatchNorm + ReLU applied by the consumer convolution; default on in f32 storage, off in the 16-bit modes'),
    'FAMI_CONCAT_ONE': ('1', 'torch.cat of up to four maps, and its backward, in one launch each'),
    'FAMI_MERGE_PREDICTORS': ('1', 'offset + mask predictor of a DCN layer as one 48 -> 324 convolution (needs the Trainer arena)'),
    'FAMI_BWD_PAIR': ('1', 'input gradient + weight gradient of a 3x3 stride-1 convolution (16-bit storage) as ONE launch: 0 off | 1 outside the weight-gradient-stream scopes | 2 everywhere'),
    'FAMI_DEFER_REDUCE': ('1', 'weight-gradient slab reduces batched 16 per launch'),
    'FAMI_DETERMINISTIC': ('0', 'run-to-run reproducible kernels (fixed-point DCN input gradient, three-launch BatchNorm)'),
    # ---- train step (train.py)
    'FAMI_PACK_SPLIT': ('1', 'input-gradient weight images packed on a side lane beside the forward pass'),
    'FAMI_DGRAD_FIRST': (None, "a convolution's input gradient enqueued before its weight gradient (0 | 1 | 2 | 3); default 1 in f32 storage, 0 in the 16-bit modes"),
    'FAMI_PACK_EARLY': ('1', 'forward weight images of stem .. stage 2 first, the rest beside the stem'),
    'FAMI_EARLY_ADAM': ('1', 'Adam in two parts: everything but the stem stretch beside the end of the backward pass'),
    'FAMI_DDP_PLAN': ('overlap', 'data-parallel launch plan: overlap (graph segments + all-reduce between them) | serial'),
    'FAMI_DDP_GRAPH': ('1', '0: eager launches with the all-reduce fired from the backward hooks'),
    'FAMI_DDP_ALGO': ('ring', 'ring (one all_reduce per slice) | mesh (reduce_scatter_tensor -> all_gather_into_tensor)'),
    'FAMI_DDP_PAYLOAD': ('f32', 'gradient bytes on the wire: f32 | bf16 | f16 (the arena, the 1/world scale and Adam stay fp32)'),
    # ---- library load (_lib.py): written into the PROCESS DEFAULT route once
    'FAMI_F32_SPLIT': ('1', 'f32 3x3 convolutions as exact three-term bf16 splits on the matrix pipe; 0 = exact-f32 MFMA'),
    'FAMI_T5': ('1', 'A/B: 0 = the band kernel instead of the persistent split-product kernel'),
    'FAMI_T5_WG': ('', 'A/B: workgroups of the persistent grid / 8'),
    'FAMI_WG16_TARGET': ('', 'A/B: workgroup target of the pipelined 16-bit weight-gradient kernel'),
    'FAMI_WGS3_TARGET': ('', 'A/B: workgroup target of the split-product weight-gradient kernel'),
    # ---- measurement ablations: WRONG results on purpose, refused unless FAMI_ALLOW_WRONG=1 (see _lib.py)
    'FAMI_ALLOW_WRONG': ('', '1: allow the ablation switches below'),
    'FAMI_T5_ABL': ('', 'upper bound: n channel chunks per split-product convolution'),
    'FAMI_ABL_WGRAD': ('0', 'upper bound: no weight-gradient kernels at all'),
    'FAMI_ABL_REGTAIL': ('0', 'upper bound: the translation regressor stops after its first n - 1 stride-2 stages (n = 2: what a fused tail kernel could save)'),
    'FAMI_ABL_PACK': ('0', 'upper bound: the 16-bit / fragment weight images are packed in the first step only (what packing inside the optimizer could save)'),
    'FAMI_ABL_LANES': ('0', 'critical-path probe: bit i = every launch enqueued on stream lane i is skipped (what the step costs without that branch)'),
    'FAMI_ABL_BN1': ('0', 'upper bound: bit 1 no apply pass / bit 2 no backward of every conv1 -> bn1 -> ReLU -> conv2 BatchNorm'),
}
WRONG = ('FAMI_T5_ABL', 'FAMI_ABL_WGRAD', 'FAMI_ABL_LANES', 'FAMI_ABL_BN1', 'FAMI_ABL_REGTAIL', 'FAMI_ABL_PACK')      # produce wrong results by design


def get(name, default=None):
    """The switch's value (string): the environment's, else `default`, else the registered default."""
    if name not in SWITCHES:
        raise KeyError('%s is not a registered switch (fami_pose_amd/options.py)' % name)
    v = os.environ.get(name)
    if v is None or v == '':
        reg = SWITCHES[name][0]
        return default if (default is not None or reg is None) else reg
    if name in WRONG and v != '0' and os.environ.get('FAMI_ALLOW_WRONG') != '1':
        # checked at every read (the values are not cached: a tool may set one between two Trainers of a process)
        raise RuntimeError('%s set: this switch skips work and produces WRONG results (measurement ablation only); '
                           'set FAMI_ALLOW_WRONG=1 to run it on purpose' % name)
    return v


def flag(name, default=None):
    """Boolean reading: anything but '0' / '' is on."""
    v = get(name, default)
    return v not in (None, '', '0')


def number(name, default=None):
    v = get(name, default)
    return int(v) if v not in (None, '') else None


def describe():
    """-> text table of every switch (name, default, meaning)"""
    return '\n'.join('%-24s %-8s %s' % (k, 'mode' if d is None else (d or '-'), m) for k, (d, m) in SWITCHES.items())
