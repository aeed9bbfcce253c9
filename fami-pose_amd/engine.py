"""Host-side execution engine of the hot path: a tape of HIP kernel launches.

Every op below enqueues kernels of libfami_hip.so (through the C ABI) on the
current HIP stream and records the matching backward launch sequence; walking
the tape in reverse is the backward pass.  There is no tracing compiler and no
torch compute op on this path: torch provides device memory (tensors), the
stream and (optionally) hipGraph capture of the whole launch sequence.

Activations are NHWC (`T.data` of shape [N,H,W,C]) in the engine's activation dtype -- fp32 (parity
configuration), bf16 (BASELINE config 3: bf16 storage + bf16 MFMA convolutions, fp32 accumulation) or fp16
(BASELINE config 5: the same with IEEE half storage + fp16 MFMA) -- and every activation-touching C-ABI entry point
exists as a `_f32` / `_bf16` / `_f16` triple.  Heatmap-producing convolutions
write fp32 in either mode; parameters, BatchNorm statistics, weight gradients, the translation regressor's
dense layers ([M,K] tensors) and everything at the NCHW boundary are always fp32.
"""
import ctypes

import torch

from . import options
from ._lib import lib

BN_EPS = 1e-5


def _p(t):
    return None if t is None else t.data_ptr()


_ABL_WGRAD = options.flag('FAMI_ABL_WGRAD', '0')
# Upper-bound experiments for "BatchNorm apply inside the consumer" (WRONG results, refused by _lib unless
# FAMI_ALLOW_WRONG=1 when present at load; read per Engine): FAMI_ABL_BN1 bit 1 = the apply pass of every conv1 -> bn1 -> ReLU -> conv2 edge is skipped (conv2 reads z),
# bit 2 = the backward apply pass of the same BatchNorms is skipped (the gradient passes through unchanged).  What the step would
# gain if the consumer-side transform were free.
_SFX = {torch.float32: '_f32', torch.bfloat16: '_bf16', torch.float16: '_f16'}


def _sfx(t):
    return _SFX[t.dtype]


class T:
    """Engine tensor: NHWC (or 2-D) storage + lazily allocated gradient.  `f32grad`: the gradient is fp32
    regardless of the activation dtype (dense [M,K] tensors of the translation regressor)."""
    __slots__ = ('data', 'grad', 'requires_grad', 'parent', 'n0', 'n1', 'f32grad', 'uses', 'lanes', 'bnrec', 'nofuse', 'xbn', 'pending')

    def __init__(self, data, requires_grad=False, parent=None, n0=0, n1=0, f32grad=False):
        self.data = data
        self.grad = None
        self.requires_grad = requires_grad
        self.parent, self.n0, self.n1 = parent, n0, n1
        self.f32grad = f32grad
        self.uses = 0          # recorded consumers whose backward has not run yet (Engine.record_bwd / backward)
        self.lanes = None      # stream lanes of those consumers
        self.bnrec = None      # set on the output of a train-mode BatchNorm: what a fused backward statistics pass needs
        self.nofuse = parent is not None     # batch-slice views and their parents receive gradients through slices
        self.xbn = None        # set on a NOT materialised BatchNorm+ReLU output (Engine.conv_bn_relu_into): data is the
                               # pre-normalisation tensor, the one consumer applies scale / shift / ReLU while staging it
        self.pending = None    # set on a BatchNorm+ReLU output whose apply pass has not been launched (Engine.bn(defer=True)): the
                               # consumer convolution runs it inside its own launch, writing `data`, or launches it first

    @property
    def shape(self):
        return tuple(self.data.shape)


class CatParam:
    """Several parameters that are ADJACENT in memory, seen as one tensor concatenated on axis 0 -- no copy: the view spans
    their storage.  The offset and the mask predictor of a DCN layer (Alignment_V15.py:79-100) run as ONE 48 -> 324 convolution
    this way (one forward, one input gradient, one weight gradient instead of two each) while the module tree and the
    state_dict keep the two nn.Conv2d.  train.flatten_parameters lays the flat parameter (and so the gradient) arena out so that
    the parts are adjacent; anywhere else (parameters allocated one by one) `adjacent()` is False and callers take the
    two-convolution path."""

    def __init__(self, parts):
        self.parts = list(parts)
        self.shape = (sum(p.shape[0] for p in self.parts),) + tuple(self.parts[0].shape[1:])

    @property
    def requires_grad(self):
        return all(p.requires_grad for p in self.parts)

    def adjacent(self):
        ps = self.parts
        if any(not p.data.is_contiguous() or p.dtype != torch.float32 or p.shape[1:] != ps[0].shape[1:] for p in ps):
            return False
        if len({p.requires_grad for p in ps}) != 1:
            return False
        return all(b.data_ptr() == a.data_ptr() + a.numel() * 4 for a, b in zip(ps, ps[1:]))

    @property
    def data(self):
        a = self.parts[0].data
        return torch.as_strided(a, self.shape, a.stride())

    def numel(self):
        return sum(p.numel() for p in self.parts)


def _real_params(ps):
    """parameters of a tape entry with CatParams replaced by their parts (bucket hooks and counters work on real parameters)"""
    out = []
    for p in ps:
        if isinstance(p, CatParam):
            out.extend(p.parts)
        else:
            out.append(p)
    return out


class _RoutedQueries:
    """The library's size / eligibility queries (workspace bytes, slot rows, which kernel would take a shape) under an engine's route:
    they depend on the routing state, so the route is bound before the raw call, as it is before every launch."""

    def __init__(self, L, eng):
        self._L, self._eng = L, eng

    def __getattr__(self, name):
        fn, L, eng = getattr(self._L.cdll, name), self._L, self._eng

        def query(*args):
            L.bind(eng.route)
            return fn(*args)
        return query


class Engine:
    def __init__(self, device, grad_views=None, record=True, dtype=torch.float32, deterministic=None, route=None):
        """route: the kernel-routing state (include/fami_route.h, `lib().new_route()`) this engine's launches dispatch by; bound to
        the calling thread before every entry point, so engines with different routes interleave in one process.  None: the
        process default route (what the fami_*_tune shims of tests and benchmarks write)."""
        assert dtype in _SFX
        self.route = route
        self.dt = dtype                # activation storage dtype
        self.sfx = _SFX[dtype]
        self.half = dtype != torch.float32     # 16-bit storage (bf16 | fp16): 16x16x32 MFMA convolutions
        # deterministic: every kernel of the step is run-to-run reproducible.  The one order-dependent reduction of the
        # path is the DCN input-gradient scatter (float atomics); this flag routes it through the 64-bit fixed-point
        # form (fami_dcn_bwd_det_*).  Default from FAMI_DETERMINISTIC (0).
        self.deterministic = (options.flag('FAMI_DETERMINISTIC', '0')) if deterministic is None else bool(deterministic)
        self.L = lib()
        self.Q = _RoutedQueries(self.L, self)
        self.record = record           # False: forward only (no tape, no gradient flags)
        self.dev = device
        self.tape = []
        self.param_grads = {}          # id(param) -> grad tensor written this step
        self.grad_views = grad_views   # optional {id(param): preallocated view (flat gradient arena)}
        self.stream = None
        self.aux = {}                  # handles a model body leaves for the trainer (final T, MI seeds, ...)
        self.bn_trained = []           # BatchNorm modules that consumed a training batch this forward
        self.prepacked = None          # optional {(id(param), mode): packed weight image} filled by the Trainer
        # stream lanes: independent sub-graphs (the parallel HRNet branches) run on side streams between fork()/join()
        self.use_lanes = options.flag('FAMI_LANES', '1')
        self.lane = 0
        self._main = None              # torch stream object of lane 0
        self._side = []                # torch side streams (lanes 1..)
        self._forked = 0               # lanes currently forked (backward bookkeeping for deferred bucket hooks)
        self._epoch = 0                # id of the current forked region (see _lane_guard)
        self._lane_owner = {}
        self._lane_priv = {}           # (id(param), lane) -> private gradient buffer inside the current forked region
        self._merge = []               # [(id(param), private buffer)] folded into the owner's buffer at the join
        self.defer_bn = None           # list while BatchNorm running-stat updates are deferred (shared modules on lanes)
        self._side_events = []         # side_launch events lane 0 has not waited for yet (join_side)
        # weight-gradient lane (opt-in, FAMI_WGRAD_LANE=1): conv wgrad kernels are leaves of the backward graph, so they
        # can run on their own stream beside the dgrad -> BN chain.  Measured on MI355X: no gain (f32 75.5 vs 75.5 ms,
        # bf16 45.9 vs 44.6 ms per step) -- every kernel already fills the chip -- hence off by default.
        self.regressor_lanes = options.flag('FAMI_REGRESSOR_LANES', '1')   # shared-weight regressors on lanes
        self.mi_lanes = self.use_lanes and options.flag('FAMI_MI_LANES', '1')   # the six MI terms of the loss on three lanes
        self.fuse_lanes = options.flag('FAMI_FUSE_LANES', '1')   # fuse terms on the lane of their source branch
        # lanes stay forked across the modules of an HRNet stage (modules.HRNetBody.run): a module's fuse sum i runs on lane i behind
        # events of the other lanes instead of on lane 0 behind a join.  The Trainer switches it off when gradient buckets are
        # all-reduced during backward (their hooks fire between forked regions).
        self.persist_lanes = options.flag('FAMI_PERSIST_LANES', '1')
        self.use_wlane = options.number('FAMI_WGRAD_LANE', '0') if self.use_lanes else 0      # 1: every weight gradient on the weight-gradient streams in turn, 2: the stage convolutions' on one stream per lane
        self.wlane_scope_now = False
        self._wstreams = []        # weight-gradient streams of this step (the first is shared with the head's scope)
        self._wowner = {}          # parameter id -> index of the weight-gradient stream its gradients go to
        self._wflip = 0
        self.wlane_pair = False
        # round 4: two streams measured neutral (bf16 23.00 vs 23.01 ms, f32 47.47 vs 47.38); end of round 5 (tools/ab_env.py, two boxes):
        # f32 45.26 -> 44.96 and 44.72 -> 44.46 ms, bf16 20.15 -> 20.08 and 20.05 -> 19.96 with two; then stem / head streams 2 / 1, 2 / 2,
        # 4 / 2, 4 / 4, 8 / 8: bf16 19.28 / 19.18 / 19.12 / 19.05 / 19.07 (second run), f32 44.87 / 44.41 / 44.40 / 44.19 / 44.38 -- launches
        # on one capture stream are ordered among themselves in the graph although they are independent leaves: four each
        self.stem_wlanes = options.number('FAMI_STEM_WGRAD_LANES', '4')
        self._wdirty = False
        # ... but for the HEAD it pays (FAMI_HEAD_WGRAD_LANE, default 1): between the first DCN forward and the last DCN backward
        # the step is one serial chain of kernels (rocprof trace: 3.5 ms with exactly one kernel in flight), and the weight
        # gradients in it are leaves.  A model body brackets that part with wlane_scope = True.
        self.head_wlane = self.use_lanes and options.flag('FAMI_HEAD_WGRAD_LANE', '1')
        self.head_wlanes = options.number('FAMI_HEAD_WGRAD_LANES', '4')       # number of weight-gradient streams of the head's scope
        self.wlane_scope = False
        # the same for the backbone's serial head and tail: stem, layer1 and the transitions run before the branches fork
        # (their backward after the branches have joined), FAMI_STEM_WGRAD_LANE
        # round 3: measured neutral (f32 49.1 -> 49.3, bf16 26.6 -> 26.5 ms).  Round 4, interleaved graph replays on one box
        # (tools/ab_env.py): f32 equal on two boxes (46.39 / 46.15, 47.96 / 48.04); bf16 with the 128-workgroup weight gradients
        # 25.50 -> 25.10 ms: on in the 16-bit modes
        # round 5 (tools/ab_env.py, two runs of 6 / 10 rounds on two boxes): f32 45.68 -> 45.43 and 45.31 -> 45.17 ms with it: on in every mode
        self.stem_wlane = options.flag('FAMI_STEM_WGRAD_LANE')
        self._sliced = []              # parents of batch_slice views: their gradient buffers are created (zero-filled) on
                                       # lane 0 before backward starts, so no lane ever races a slice write against the fill
        self._keep = []                # every buffer handed out this step stays alive until the step has been enqueued:
                                       # the caching allocator must not recycle a block across lanes within a step
        # two-launch BatchNorm (fami_bn_train_fwd2 / fami_bn_bwd2: fp64 slot atomics, finalize folded into the apply pass).
        # Sums arrive in atomic order, so the deterministic mode keeps the three-launch forms.
        self.bn2 = (not self.deterministic) and options.flag('FAMI_BN2', '1')
        # BatchNorm statistics folded into the neighbouring convolutions' epilogues (conv.hip EpiBN): the forward
        # statistics into the producing convolution, the backward statistics into the input-gradient convolution that
        # makes the last contribution to the BatchNorm output's gradient.  FAMI_FUSE_BN = 0 | fwd | bwd | 1 (both).
        # Default fwd.  Measured on MI355X (profiles/r03_fused_bn.txt): per launch the epilogue costs about what the removed
        # pass cost -- +2..5 us on the forward convolution against a 4..6 us statistics kernel, +4..10 us on the input
        # gradient (it has to read the BatchNorm input tile) against 4..9 us -- and inside the step the stand-alone
        # statistics kernels were mostly hidden behind the other stream lanes.  Whole step, interleaved A/B with the
        # round-3 kernels: forward fusion f32 62.7 -> 62.3 ms, bf16 29.1 -> 29.0 ms (neutral, ~250 launches and as many
        # tensor reads fewer per step: on); backward fusion bf16 29.1 -> 29.3 ms, both 28.7 -> 29.1 ms (off, kept as a
        # tested option).
        # Round 4, 'auto' (the default): forward everywhere, backward only where the weight-resident 48-channel kernel
        # (conv_t6.hip) takes the input gradient -- it requests the BatchNorm input a unit ahead and reads the channel
        # constants from LDS, +1.7 us on a 14.5 us launch against 5.7 us saved on the BatchNorm backward
        # (tools/bench_epi2.py); on the other kernels the epilogue still costs more than the pass it removes.
        fz = options.get('FAMI_FUSE_BN', 'auto')
        # auto (default): f32 storage -- forward statistics in every convolution's epilogue (the split-product kernels carry them for
        # free and XBN needs them), no backward fusion; 16-bit storage -- forward statistics ONLY in the epilogues of the DMA-staged 3x3
        # kernels (conv3x3_t6 / t7 with 48-channel phases: statistics in the implicit-GEMM epilogues of the 1x1 / stride-2 convolutions
        # measured neutral at W48 and a loss at W64), backward statistics where the input-gradient kernel is one of those (see below).
        # tools/ab_env.py, one box per pair: bf16 W48 20.08 -> 19.98 ms, W64 fp16 28.42 -> 27.90 against 'autoall' (the rule up to round 5).
        # fwd3 / bwdauto: the two halves of the 16-bit rule alone (probes).
        self.fuse_bn_fwd = self.bn2 and fz in ('1', 'fwd', 'auto', 'autoall', 'fwd3')
        self.fuse_bn_fwd3 = fz == 'fwd3' or (fz == 'auto' and self.half)
        self.fuse_bn_bwd = self.bn2 and fz in ('1', 'bwd')
        self.fuse_bn_bwd_auto = self.bn2 and self.half and fz in ('auto', 'autoall', 'bwdauto')
        # ... and where the phased kernel (conv3x3_t7_kernel) takes it: 0 never, 1 (default) its non-accumulating launches with a
        # recomputed mask (-0.04 ms), 2 all of them (+0.12 ms: the accumulating variant spills)
        # round 6 (the input gradient now shares its launch with the weight gradient, csrc/conv_pair.h): 2 -- bf16 step 18.36 -> 18.24 ms
        # against 1 (tools/ab_env.py, one box), ~100 statistics launches fewer
        self.fuse_bn_bwd_t7 = options.number('FAMI_FUSE_BN_T7', '2')
        # order of a convolution's two backward launches: input gradient (the chain's next link) before the weight gradient (a leaf)?
        # tools/ab_env.py, two runs on two boxes: f32 storage 45.80 -> 45.48 and 45.95 -> 45.63 ms with it (half of it each from the
        # lanes with / without a weight-gradient stream); bf16 20.21 -> 20.40 and 20.13 -> 20.19 / 20.25: f32 only.
        # 1: everywhere, 2: only where the weight gradient has its own lane, 3: only elsewhere
        self.dgrad_first = options.number('FAMI_DGRAD_FIRST', '0' if self.half else '1')
        # ... and the forward statistics stay a pass of their own behind the 32-channel-phase instances of that kernel (layers of
        # 64-multiple channels: HRNet-W64's branches, stage 1's 64 -> 64).  W64 fp16 step, tools/ab_env.py on one box: epilogues
        # everywhere 29.40 ms, forward only 29.05, none 28.59 (while its EpiBN instances spilled); without spills: forward 28.47 vs 28.45
        # without, backward 28.25: bit 1 on
        self.fuse_bn_c64 = options.number('FAMI_FUSE_BN_C64', '2')      # bit 0: forward statistics, bit 1: backward statistics (under the rule of FAMI_FUSE_BN_T7)
        self.fuse_bn_skip = set(filter(None, options.get('FAMI_FUSE_BN_SKIP', '').replace('+', ',').split(',')))      # probes: '1x1', 's2'
        self.concat_one = options.flag('FAMI_CONCAT_ONE', '1')      # Engine.concat: one launch for up to four sources
        # the two predictor convolutions of a DCN layer as one (CatParam; needs the Trainer's arena layout): FAMI_MERGE_PREDICTORS
        self.merge_predictors = options.flag('FAMI_MERGE_PREDICTORS', '1')
        self.fuse_term_bn2 = options.flag('FAMI_FUSE_TERM_BN2', '1')
        # BatchNorm + ReLU applied by the CONSUMER convolution while it stages its input (Engine.conv_bn_relu_into, conv_epi.h
        # XBN): 104 launches and one tensor write + read per BasicBlock less.  FAMI_XBN = 0 | 1; default: on in f32 storage,
        # off in the 16-bit modes.  Measured on MI355X (interleaved A/B of the step, both orders, 10 rounds): f32 50.95 ->
        # 50.86 and 50.96 -> 50.78 ms (the split-product kernels already spend a dozen VALU instructions per staged element
        # on the split; two more are free); bf16 27.70 -> 27.91 ms (the transform moves into the staging path of kernels
        # that are bound by exactly that path).
        # Late round 4: off again in the 16-bit modes -- the DMA-staged kernels (conv_t6.hip, conv_wgrad6_kernel) copy their operands
        # global -> LDS without passing registers, so they take a materialised input; with every weight gradient on them the
        # bf16 step is 23.23 -> 22.64 ms (tools/ab_env.py, one box) against the consumer-side transform on the band kernels.
        self.use_xbn = options.flag('FAMI_XBN', '0' if self.half else '1')
        # Tried in round 4: backward fusion only on the SERIAL stretches of the step (stem, layer1: one lane, nothing beside it --
        # 3.6 ms of the f32 backward pass), `serial_scope` set by HRNetBody.run.  Interleaved bench runs on one box: f32 48.16
        # vs 48.26 ms, bf16 24.94 vs 25.15 ms -- the epilogue costs what the removed pass did there too.  `serial_fuse = True`
        # keeps it available.
        self.serial_scope = False
        self.serial_fuse = False       # (the FAMI_SERIAL_FUSE switch of rounds 4-5 is gone: measured neutral twice; tests set the attribute)
        self.nfused = {'fwd': 0, 'bwd': 0, 'xbn': 0}     # statistics passes that ran in a convolution epilogue (tests / reporting)
        self.conv_flops = 0                    # 2*MACs of every nn.Conv2d forward / input-gradient / weight-gradient launch enqueued (reporting)
        # deferred slab reduces of the weight-gradient kernels: described on the host as they are enqueued and launched
        # 16 at a time per stream (fami_wgrad_reduce_batch) at the joins / bucket boundaries / the end of backward --
        # 303 tiny launches per step otherwise sit between every weight gradient and the next kernel of its lane
        self.defer_reduce = options.flag('FAMI_DEFER_REDUCE', '1')
        # a 3x3 stride-1 convolution's input gradient and weight gradient as one launch (16-bit storage; csrc/conv_pair.h):
        # 0 off | 1 where the weight gradient would run on the convolution's own lane | 2 also inside the weight-gradient-stream scopes
        self.bwd_pair = options.number('FAMI_BWD_PAIR', '1')
        self.npair = 0                 # combined launches enqueued this step (tests / reporting)
        self._red = {}                 # raw stream -> [ctypes descriptor buffers]
        self._red_dw = {}              # raw stream -> {dw pointers with a pending reduce}
        self._red_longs = self.Q.fami_wgrad_reduce_desc_longs()
        self.abl_lanes = options.number('FAMI_ABL_LANES', '0')   # upper-bound experiment (WRONG results): bit i = no launches on stream lane i
        self.nbnin = 0                 # BatchNorm apply passes that ran inside their consumer convolution's launch (tests / reporting)
        self._pending = []             # BatchNorm apply passes deferred to their consumer convolution (Engine.bn(defer=True))
        self.bn_in = self.half and options.flag('FAMI_BN_IN', '1')      # conv1 -> bn1 -> ReLU -> conv2: bn1's apply pass inside conv2's launch (16-bit storage)
        self.abl_bn1 = options.number('FAMI_ABL_BN1', '0')      # upper-bound experiment, see the comment at the top of the file
        self.sync_stream()
        self._zero_begin()

    # ------------------------------------------------------------------ plumbing
    def rq(self, p):
        return self.record and p is not None and p.requires_grad

    def sync_stream(self):
        self._main = torch.cuda.current_stream(self.dev)
        self.stream = self._main.cuda_stream
        self.lane = 0

    # ------------------------------------------------------------------ stream lanes
    _side_pool = {}

    def _lanes(self, n):
        """-> n - 1 side streams, every one distinct from this step's main stream and from each other.  torch hands out
        streams round-robin from a fixed pool of 32 per device, so after enough torch.cuda.Stream() calls in a process a
        'new' stream IS an old one: a capture stream that coincided with a cached lane stream made fork / join wait on
        the very stream they were recorded on, and the hipGraph captured that way crashed in hipGraphLaunch
        (hip::Graph::UpdateStreams) -- the cache is therefore filtered by handle."""
        pool = Engine._side_pool.setdefault(self.dev, [])
        main = self._main.cuda_stream
        side = [s for s in pool if s.cuda_stream != main]
        tries = 0
        while len(side) < n - 1 and tries < 64:
            tries += 1
            s = torch.cuda.Stream(self.dev)
            if s.cuda_stream != main and all(s.cuda_stream != t.cuda_stream for t in pool):
                pool.append(s)
                side.append(s)
        if len(side) < n - 1:
            raise RuntimeError('could not obtain %d distinct side streams' % (n - 1))
        self._side = side
        if options.flag('FAMI_DEBUG_STREAMS'):
            print('[fami] lanes: main %#x side %s capturing %s' % (main, [hex(t.cuda_stream) for t in side[:n - 1]],
                                                                  torch.cuda.is_current_stream_capturing()), flush=True)
        return side

    def set_lane(self, i):
        self.lane = i
        self.stream = self._main.cuda_stream if i == 0 else self._side[i - 1].cuda_stream

    _wgrad_pool = {}

    def _lane_stream(self):
        return self._main if self.lane == 0 else self._side[self.lane - 1]

    def _new_wstream(self, extra=()):
        # same aliasing hazard as _lanes(): torch hands streams out of a 32-entry round-robin pool, so a cached stream can BE this
        # step's main / capture stream or one of its side lanes -- re-pick until distinct
        taken = {self._main.cuda_stream} | {t.cuda_stream for t in Engine._side_pool.get(self.dev, [])} | set(extra)
        ws, tries = None, 0
        while (ws is None or ws.cuda_stream in taken) and tries < 64:
            ws = torch.cuda.Stream(self.dev)
            tries += 1
        if ws.cuda_stream in taken:
            raise RuntimeError('could not obtain a distinct weight-gradient stream')
        return ws

    def _enter_wlane(self, pair=False, key=None):
        """Route the following calls to a weight-gradient stream, ordered after the current lane's work so far.  Inside a
        wlane_pair scope (HRNetBody: stem, layer1, transitions -- a serial chain whose weight gradients read tensors of up to
        70 MB and run at a fifth of the HBM rate each) `stem_wlanes` such streams take the launches in turn, elsewhere (the head)
        `head_wlanes`: in the kernel trace of the bf16 step the single stream was the critical path of the last 0.4 ms of the
        backward pass (launches on one stream are ordered among themselves although they are independent leaves)."""
        n = max(1, self.stem_wlanes if pair else self.head_wlanes)
        while len(self._wstreams) < n:
            k = len(self._wstreams)
            taken = {self._main.cuda_stream} | {t.cuda_stream for t in Engine._side_pool.get(self.dev, [])} | {t.cuda_stream for t in self._wstreams}
            ws = Engine._wgrad_pool.get((self.dev, k))
            if ws is None or ws.cuda_stream in taken:
                ws = self._new_wstream(tuple(t.cuda_stream for t in self._wstreams))
            Engine._wgrad_pool[(self.dev, k)] = ws
            self._wstreams.append(ws)
        # key: what the launches accumulate into (a parameter).  A module applied twice inside such a scope keeps its weight gradients
        # on ONE stream: they accumulate into the same buffer and their slab reduces are ordered per stream
        own = self._wowner.get(key) if key is not None else None
        if own is not None and own < n:
            target = self._wstreams[own]
        elif self.use_wlane == 2 and not pair and not self.wlane_scope_now:      # (probe: stage convolutions, one weight-gradient stream per lane)
            target = self._wstreams[self.lane % n]
        else:
            self._wflip = (self._wflip + 1) % n
            target = self._wstreams[self._wflip]
            if key is not None:
                self._wowner[key] = self._wflip
        ev = torch.cuda.Event()
        ev.record(self._lane_stream())
        target.wait_event(ev)
        saved = self.stream
        self.stream = target.cuda_stream
        self._wdirty = True
        return saved

    def sync_wgrad_lane(self):
        """Lane 0 continues after every weight-gradient kernel enqueued so far (before an all-reduce / the optimizer)."""
        if self._wdirty:
            for ws in self._wstreams:
                self.flush_reduces(ws.cuda_stream)
                ev = torch.cuda.Event()
                ev.record(ws)
                self._main.wait_event(ev)
            self._wdirty = False

    def side_launch(self, fn):
        """fn(raw_stream) on the first side lane, ordered after everything enqueued on lane 0 so far -> event to hand to
        wait_main().  For work nothing on the forward path reads (the dgrad weight images): it runs beside lane 0."""
        side = self._lanes(2)[0]
        ev = torch.cuda.Event()
        ev.record(self._main)
        side.wait_event(ev)
        fn(side.cuda_stream)
        done = torch.cuda.Event()
        done.record(side)
        return done

    late_weights_ready = None

    def join_late_weights(self):
        """Lane 0 waits for the side-lane pack of the later stages' weight images (train.Trainer sets the event)."""
        if self.late_weights_ready is not None:
            self._main.wait_event(self.late_weights_ready)
            self.late_weights_ready = None

    def join_side(self):
        """Lane 0 waits for the side-lane work launched so far whose events nobody has waited for."""
        for ev in self._side_events:
            self._main.wait_event(ev)
        self._side_events = []

    def wait_main(self, ev):
        self._main.wait_event(ev)

    def _do_fork(self, n):
        side = self._lanes(n)
        self.flush_reduces()
        ev = torch.cuda.Event()
        ev.record(self._main)
        for i in range(n - 1):
            side[i].wait_event(ev)
        self._forked = n
        self._epoch += 1

    def _do_join(self, n):
        side = self._lanes(n)
        self.flush_reduces()               # every lane's pending reduces go out on their own streams before the join events
        for i in range(n - 1):
            ev = torch.cuda.Event()
            ev.record(side[i])
            self._main.wait_event(ev)
        self.set_lane(0)
        self._forked = 0
        # parameters used on several lanes of the region: fold the lane-private gradients into the owner's buffer
        # ... in ONE launch per 32 buffers (fami_add_batch_f32): the translation regressors alone are 37 parameters x 3 lanes,
        # 111 launches of 2-3 us back to back on the critical path of the head's backward pass (0.8 ms in the kernel trace)
        if self._merge:
            n = len(self._merge)
            ptrs, counts = (ctypes.c_long * (2 * n))(), (ctypes.c_int * n)()
            for i, (key, buf) in enumerate(self._merge):
                g = self.param_grads[key]
                ptrs[2 * i], ptrs[2 * i + 1], counts[i] = buf.data_ptr(), g.data_ptr(), g.numel()
            self.call('fami_add_batch_f32', ptrs, counts, n)
        self._merge = []
        self._lane_priv = {}

    def _all_to_all(self, n):
        """every one of the n lanes continues after what the others have enqueued so far.  Through lane 0 as a hub (it waits for the
        side lanes' events, then they wait for its event): two side streams waiting on EACH OTHER's events segfault
        hipStreamEndCapture on this ROCm (tools/probes/capture_cross_wait.py: patterns 1, 2), a join immediately followed by a
        fork does not (pattern 4)."""
        side = [self._side[j - 1] for j in range(1, n)]
        for st in side:
            ev = torch.cuda.Event()
            ev.record(st)
            self._main.wait_event(ev)
        ev = torch.cuda.Event()
        ev.record(self._main)
        for st in side:
            st.wait_event(ev)

    def lanes_sync(self, n):
        """Inside a forked region of n lanes: an all-to-all dependency instead of a join + fork through lane 0 (a module's fuse
        sums read every lane's terms).  Its backward is the same all-to-all (the terms' backward reads every lane's gradients)."""
        self.flush_reduces()
        self._all_to_all(n)
        if self.record:
            def bwd():
                self.flush_reduces()
                self._all_to_all(n)
            self.tape.append((bwd, (), self.lane, ()))

    on_mark = None

    def mark(self, name):
        """A named point of the forward pass; the backward pass calls on_mark(name) when it gets back there (every operation
        recorded after the mark has had its backward enqueued by then)."""
        if self.record:
            self.tape.append((lambda: self.on_mark(name) if self.on_mark is not None else None, (), self.lane, ()))

    def fork(self, n):
        """Lanes 1..n-1 start after everything enqueued on lane 0 so far; the backward of a fork is a join."""
        if not self.use_lanes or n < 2:
            return False
        self._do_fork(n)
        if self.record:
            self.tape.append((lambda: self._do_join(n), (), 0, ()))
        return True

    def join(self, n):
        """Lane 0 continues after lanes 1..n-1; the backward of a join is a fork."""
        self._do_join(n)
        if self.record:
            self.tape.append((lambda: self._do_fork(n), (), 0, ()))

    def record_bwd(self, fn, params, inputs=()):
        """inputs: the engine tensors this op's backward contributes gradient to.  Counting them lets a backward closure
        know whether it makes the LAST contribution to a tensor (T.uses == 0 while it runs)."""
        for t in inputs:
            if t is None:
                continue
            t.uses += 1
            if t.lanes is None:
                t.lanes = {self.lane}
            else:
                t.lanes.add(self.lane)
        self.tape.append((fn, _real_params(params), self.lane, inputs))

    def cat_usable(self, *cats):
        """May these CatParams stand in for their parts in this step?  Adjacent in memory, and -- when gradients are recorded --
        with a gradient buffer of their own that spans the parts' buffers (train.Trainer registers those views)."""
        if not self.merge_predictors or not all(c.adjacent() for c in cats):
            return False
        if self.record and any(c.requires_grad for c in cats):
            return self.grad_views is not None and all(id(c) in self.grad_views for c in cats)
        return True

    def call(self, name, *args):
        if self.abl_lanes and (self.abl_lanes >> min(self.lane, 7)) & 1:
            return
        self.L.call_routed(self.route, name, *args, self.stream)

    # ------------------------------------------------------------------ weight gradients with deferred slab reduces
    def wgrad(self, x_data, dy, g, geo, acc, xbn=None):
        """dW (=|+=) of one convolution: the partial-slab kernel now, its reduce batched with the stream's others.
        xbn: x_data is the input of a not materialised BatchNorm+ReLU (mean, invstd, gamma, beta)."""
        if _ABL_WGRAD:          # upper-bound experiment (FAMI_ABL_WGRAD=1, WRONG gradients): no weight-gradient kernels at all
            return
        nb = self.Q.fami_conv2d_wgrad_workspace(*geo)
        ws = self.ws(nb)
        assert xbn is None or self.defer_reduce
        if not self.defer_reduce:
            self.acall('fami_conv2d_wgrad', _p(x_data), _p(dy), _p(g), _p(ws), ws.numel() * 4, *geo, acc)
            return
        st = self.stream
        dws = self._red_dw.setdefault(st, set())
        if g.data_ptr() in dws:            # a second reduce into the same gradient must follow the first
            self.flush_reduces(st)
            dws = self._red_dw.setdefault(st, set())
        desc = (ctypes.c_long * self._red_longs)()
        if xbn is not None:
            self.acall('fami_conv2d_wgrad_defer_xbn', _p(x_data), _p(dy), _p(g), _p(ws), ws.numel() * 4, *geo[:5], acc, desc,
                       *[_p(t) for t in xbn])
        else:
            self.acall('fami_conv2d_wgrad_defer', _p(x_data), _p(dy), _p(g), _p(ws), ws.numel() * 4, *geo, acc, desc)
        self._red.setdefault(st, []).append(desc)
        dws.add(g.data_ptr())
        if len(self._red[st]) >= 16:
            self.flush_reduces(st)

    def wgrad_pair(self, x_data, dy, wpd, gx, accx, g, accw, geo, stats, xbn=None):
        """Input gradient (stats: the arguments of the backward-statistics epilogue, or None) and weight gradient of a 3x3 stride-1
        convolution in one launch; the weight gradient's slab reduce is deferred as in wgrad().  xbn (f32 storage): x_data is the
        input of a not materialised BatchNorm + ReLU (mean, invstd, gamma, beta)."""
        ws = self.ws(self.Q.fami_conv2d_wgrad_workspace(*geo))
        st = self.stream
        dws = self._red_dw.setdefault(st, set())
        if g.data_ptr() in dws:
            self.flush_reduces(st)
            dws = self._red_dw.setdefault(st, set())
        desc = (ctypes.c_long * self._red_longs)()
        if not self.half:
            assert stats is None
            self.call('fami_conv2d_bwd_pair_f32', _p(x_data), _p(dy), _p(wpd), _p(gx), _p(g), _p(ws), ws.numel() * 4, *geo, accx, accw, desc,
                      *([None] * 4 if xbn is None else [_p(t) for t in xbn]))
        else:
            assert xbn is None
            if stats is None:
                stats = (None, None, None, None, None, None, 0, None)
            self.acall('fami_conv2d_bwd_pair', _p(x_data), _p(dy), _p(wpd), _p(gx), _p(g), _p(ws), ws.numel() * 4, *geo, accx, accw, desc, *stats)
        self.npair += 1
        self._red.setdefault(st, []).append(desc)
        dws.add(g.data_ptr())
        if len(self._red[st]) >= 16:
            self.flush_reduces(st)

    def flush_reduces(self, stream=None):
        """Launch the pending reduces of one raw stream (default: every stream), each batch on its own stream."""
        for st in ([stream] if stream is not None else list(self._red)):
            items = self._red.get(st)
            if not items:
                continue
            n = len(items)
            flat = (ctypes.c_long * (self._red_longs * n))()
            for i, d in enumerate(items):
                flat[i * self._red_longs:(i + 1) * self._red_longs] = d[:]
            self.L.call_routed(self.route, 'fami_wgrad_reduce_batch', flat, n, st)
            self._red[st] = []
            self._red_dw[st] = set()

    def empty(self, *shape, dtype=torch.float32):
        t = torch.empty(shape, dtype=dtype, device=self.dev)
        self._keep.append(t)
        return t

    def act(self, *shape):
        """Uninitialised activation-typed buffer."""
        t = torch.empty(shape, dtype=self.dt, device=self.dev)
        self._keep.append(t)
        return t

    def acall(self, name, *args):
        """Call the activation-dtype instance of an entry point."""
        if self.abl_lanes and (self.abl_lanes >> min(self.lane, 7)) & 1:
            return
        self.L.call_routed(self.route, name + self.sfx, *args, self.stream)

    def new_grad(self, t):
        g = torch.empty(t.data.shape, dtype=torch.float32 if t.f32grad else self.dt, device=self.dev)
        self._keep.append(g)
        return g

    def like(self, t):
        r = torch.empty_like(t)
        self._keep.append(r)
        return r

    def ws(self, nbytes):
        t = torch.empty((max(int(nbytes), 4) + 3) // 4, dtype=torch.float32, device=self.dev)
        self._keep.append(t)
        return t

    def fill(self, t, v=0.0):
        self.call('fami_fill' + _sfx(t), _p(t), t.numel(), float(v))
        return t

    # zero-initialised scratch for the step (the slot rows of the two-launch BatchNorm): slices of ONE arena per device
    # that is cleared by one launch on the step's main stream when its Engine is created (every lane forks after that),
    # sized from the steps before -- a graph-mode Trainer's eager warm-up steps size it before the capture.  A captured
    # graph keeps pointing at the arena it was captured with, so a buffer that is outgrown is retired, never freed.
    _zero_arenas = {}

    def _zero_begin(self):
        # The arena belongs to the Engine created LAST on the device; an older Engine that is still alive (two forwards
        # before one backward through runtime._EngineFn) takes fresh, separately cleared buffers from then on -- both
        # backwards handing fami_bn_bwd2 the same dirty slices was a silent wrong-gradient bug.  An Engine
        # on another stream than the previous owner's first waits for that stream's work so far (the previous owner's
        # kernels may still be reading their slots); inside a capture the Trainer has synchronised the device already.
        import weakref
        st = Engine._zero_arenas.setdefault(self.dev.index, {'buf': None, 'high': 0, 'retired': [], 'owner': None,
                                                             'stream': None})
        capturing = torch.cuda.is_current_stream_capturing()
        prev = st['stream']
        if prev is not None and prev.cuda_stream != self.stream and not capturing:
            ev = torch.cuda.Event()
            ev.record(prev)
            self._main.wait_event(ev)
        st['owner'] = weakref.ref(self)
        st['stream'] = None if capturing else self._main
        self._zst, self._zoff, self._zfilled = st, 0, 0
        cap = 0 if st['buf'] is None else st['buf'].numel() * 4
        if st['high'] > cap and not capturing:
            if st['buf'] is not None:
                st['retired'].append(st['buf'])
            st['buf'] = torch.empty((st['high'] * 5 // 4 + 1023) // 4, dtype=torch.float32, device=self.dev)
            cap = st['buf'].numel() * 4
        if st['buf'] is not None and st['high'] > 0:
            self._zfilled = min(cap, (st['high'] + 255) & ~255)
            self.fill(st['buf'][:self._zfilled // 4])

    def zeros_bytes(self, nbytes):
        """-> zero-filled fp32 buffer of >= nbytes (256-byte aligned slice of the step's arena; before the arena has been
        sized -- the first step -- or past its end, a fresh buffer cleared on the current lane)."""
        nbytes = (int(nbytes) + 255) & ~255
        st, off = self._zst, self._zoff
        self._zoff += nbytes
        st['high'] = max(st['high'], self._zoff)
        if off + nbytes <= self._zfilled and st['owner']() is self:
            return st['buf'][off // 4:(off + nbytes) // 4]
        t = torch.empty(nbytes // 4, dtype=torch.float32, device=self.dev)
        self._keep.append(t)
        return self.fill(t)

    def gbuf(self, t):
        """-> (gradient buffer of t, accumulate flag)."""
        if t.grad is None:
            if t.parent is not None:
                par = t.parent
                if par.grad is None:
                    par.grad = self.fill(self.new_grad(par))
                t.grad = par.grad[t.n0:t.n1]
                return t.grad, 1
            t.grad = self.new_grad(t)
            return t.grad, 0
        return t.grad, 1

    def pgrad(self, p):
        """-> (gradient buffer of parameter p, accumulate flag) for this step."""
        key = id(p)
        if self._forked:
            # a module applied on several lanes of one forked region (the translation regressor: one set of weights for
            # every supporting frame): the first lane owns the gradient buffer, the others accumulate privately and
            # are folded in at the join, in lane order
            owner = self._lane_owner.get(('grad', key))
            if owner is not None and owner[0] == self._epoch and owner[1] != self.lane:
                buf = self._lane_priv.get((key, self.lane))
                if buf is not None:
                    return buf, 1
                buf = self.like(p.data)
                self._lane_priv[(key, self.lane)] = buf
                self._merge.append((key, buf))
                return buf, 0
            self._lane_owner[('grad', key)] = (self._epoch, self.lane)
        g = self.param_grads.get(id(p))
        if g is not None:
            return g, 1
        if self.grad_views is not None:
            g = self.grad_views[id(p)]
        else:
            g = self.like(p.data)
        self.param_grads[id(p)] = g
        return g, 0

    def apply_deferred_bn(self):
        """Running-statistics updates of the BatchNorm calls made while `defer_bn` was a list, in call order (lane 0)."""
        items, self.defer_bn = self.defer_bn, None
        items = list(items or ())
        if items:      # one launch per 32 updates, applied in call order inside the kernel (was: one launch each)
            n = len(items)
            ptrs, meta = (ctypes.c_long * (4 * n))(), (ctypes.c_float * (4 * n))()
            for i, (bn, mean, invstd, P, mom) in enumerate(items):
                ptrs[4 * i:4 * i + 4] = [bn.running_mean.data_ptr(), bn.running_var.data_ptr(), mean.data_ptr(), invstd.data_ptr()]
                meta[4 * i:4 * i + 4] = [float(mean.numel()), float(P), float(mom), float(bn.eps)]
            # nothing in the step reads the running statistics: on a side lane, joined by join_side() (the caller)
            if self.use_lanes:
                self._side_events.append(self.side_launch(
                    lambda st: self.L.call_routed(self.route, 'fami_bn_running_update_batch_f32', ptrs, meta, n, st)))
            else:
                self.call('fami_bn_running_update_batch_f32', ptrs, meta, n)

    def _lane_guard(self, key):
        """Shared mutable state (a parameter's gradient accumulator, a BatchNorm's running statistics) may be touched by
        ONE lane per forked region: a module applied on two concurrent lanes would race.  Fails loudly instead."""
        if not self._forked:
            return
        seen = self._lane_owner.get(key)
        if seen is not None and seen[0] == self._epoch and seen[1] != self.lane:
            raise RuntimeError('engine: %s used on stream lanes %d and %d inside one forked region -- modules that share '
                               'parameters must run on one lane' % (key[0], seen[1], self.lane))
        self._lane_owner[key] = (self._epoch, self.lane)

    # packed weights: frozen nn.Parameters are packed once (cache lives ON the parameter object, keyed by mode /
    # dtype / version counter, so it can never outlive or be confused with another model's weights); trainable
    # parameters and ad-hoc tensors are packed every forward
    def packed(self, w, mode):
        if w.requires_grad and getattr(w, '_fami_packed', None):
            w._fami_packed = {}            # being trained again: an image cached while it was frozen must not survive
        if self.prepacked is not None:
            hit = self.prepacked.get((id(w), mode))
            if hit is not None:
                return hit
        Co, Ci, kh, kw = w.shape
        cacheable = isinstance(w, torch.nn.Parameter) and not w.requires_grad
        key = (mode, self.dt)
        if cacheable:
            hit = getattr(w, '_fami_packed', {}).get(key)
            if hit is not None and hit[0] == w._version and hit[2] == w.data_ptr():
                return hit[1]
        if self.half:
            n = self.Q.fami_packed_weight_elems_bf16(Co, Ci, kh, kw, mode)
            wp = self.act(n)
            self.acall('fami_pack_conv_weight', _p(w.data), _p(wp), Co, Ci, kh, kw, mode)
        else:
            n = self.Q.fami_packed_weight_elems(Co, Ci, kh, kw, mode)
            wp = self.empty(n)
            self.call('fami_pack_conv_weight_f32', _p(w.data), _p(wp), Co, Ci, kh, kw, mode)
        if cacheable:
            if not hasattr(w, '_fami_packed'):
                w._fami_packed = {}
            w._fami_packed[key] = (w._version, wp, w.data_ptr())
        return wp

    # ------------------------------------------------------------------ inputs / boundary
    def frames(self, kf_x, sup_x):
        """Alignment_V15.py:115-119: key + S supporting frames stacked on the batch axis (frame-major)."""
        B, _, H, W = kf_x.shape
        S = 0 if sup_x is None else sup_x.shape[1] // 3
        out = self.act((1 + S) * B, H, W, 3)
        self.acall('fami_pack_frames', _p(kf_x.contiguous()), _p(None if sup_x is None else sup_x.contiguous()),
                  _p(out), B, S, H, W)
        return T(out)

    def from_nchw(self, x, requires_grad=False):
        N, C, H, W = x.shape
        out = self.act(N, H, W, C)
        self.acall('fami_nchw_to_nhwc', _p(x.contiguous()), _p(out), N, C, H, W)
        return T(out, requires_grad)

    def to_nchw(self, x):
        """-> torch tensor [N,C,H,W]; gradient comes back through seed()."""
        N, H, W, C = x.shape
        out = self.empty(N, C, H, W)
        self.call('fami_nhwc_to_nchw' + _sfx(x.data), _p(x.data), _p(out), N, C, H, W, 0)
        return out

    def seed_nchw_split(self, x, g_nchw):
        """seed_nchw in two halves: the NCHW -> NHWC transpose now (on the current lane), the accumulation into x's gradient
        buffer by the returned closure (call it on lane 0 after the join: several lanes may seed the same tensor)."""
        if g_nchw is None or not x.requires_grad:
            return lambda: None
        N, H, W, C = x.shape
        # in the GRADIENT's storage type (new_grad: fp32 for f32grad tensors such as the heatmaps, the compute type otherwise)
        tmp = torch.empty(x.data.shape, dtype=x.grad.dtype if x.grad is not None else (torch.float32 if x.f32grad else self.dt),
                          device=self.dev)
        self._keep.append(tmp)
        self.call('fami_nchw_to_nhwc' + _sfx(tmp), _p(g_nchw.contiguous()), _p(tmp), N, C, H, W)

        def finish():
            g, acc = self.gbuf(x)
            self.call('fami_axpby' + _sfx(g), _p(tmp), _p(g) if acc else None, _p(g), g.numel(), 1.0, 1.0 if acc else 0.0)
        return finish

    def seed_nchw(self, x, g_nchw):
        """Add an NCHW gradient to NHWC tensor x."""
        if g_nchw is None or not x.requires_grad:
            return
        N, H, W, C = x.shape
        g, acc = self.gbuf(x)
        if acc:
            tmp = self.like(g)
            self.call('fami_nchw_to_nhwc' + _sfx(g), _p(g_nchw.contiguous()), _p(tmp), N, C, H, W)
            self.call('fami_axpby' + _sfx(g), _p(tmp), _p(g), _p(g), g.numel(), 1.0, 1.0)
        else:
            self.call('fami_nchw_to_nhwc' + _sfx(g), _p(g_nchw.contiguous()), _p(g), N, C, H, W)

    def flush_pending(self, pend):
        """Launch the apply pass a BatchNorm deferred to its consumer (Engine.bn(defer=True)), if nothing has run it yet."""
        if pend is None or pend['done']:
            return
        b = pend['bn']
        self.acall('fami_bn_apply_slots', _p(pend['z']), None, _p(pend['y']), _p(b.weight.data), _p(b.bias.data), _p(pend['mean']),
                   _p(pend['invstd']), _p(pend['rm']), _p(pend['rv']), pend['P'], pend['C'], 1, pend['mom'], pend['eps'], _p(pend['slots']))
        pend['done'] = True

    # ------------------------------------------------------------------ conv / bn
    def conv(self, x, weight, bias=None, stride=1, pad=0, dil=1, relu=False, out_f32=False, stats=None):
        """nn.Conv2d.  out_f32: write fp32 even in bf16 mode (heatmap-producing layers).  stats = (slots, pivot_src):
        the statistics pass of the BatchNorm that follows runs in the epilogue (fami_conv2d_fwd_stats_*)."""
        N, H, W, Ci = x.shape
        Co, Ci2, kh, kw = weight.shape
        assert Ci2 == Ci, (x.shape, weight.shape)
        Ho = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
        wp = self.packed(weight, 0)
        flops = 2 * N * Ho * Wo * Ci * Co * kh * kw
        self.conv_flops += flops
        xb = x.xbn
        pend = getattr(x, 'pending', None)
        if pend is not None and not pend['done']:
            # x is the output of a BatchNorm + ReLU whose apply pass nobody has launched: run it inside this launch where the
            # weight-resident kernel takes the convolution (it also writes x, which the backward pass reads), else launch it now
            if (self.half and xb is None and not relu and not out_f32 and (kh, kw, stride, pad, dil) == (3, 3, 1, 1, 1)
                    and self.Q.fami_conv2d_fwd_bnin_ok(N, H, W, Ci, Co)):
                y = self.act(N, Ho, Wo, Co)
                b = pend['bn']
                self.acall('fami_conv2d_fwd_bnin', _p(pend['z']), _p(wp), _p(None if bias is None else bias.data), _p(y), _p(x.data),
                           N, H, W, Ci, Co, _p(None if stats is None else stats[0]), _p(None if stats is None else stats[1]),
                           _p(pend['slots']), pend['P'], _p(b.weight.data), _p(b.bias.data), _p(pend['mean']), _p(pend['invstd']),
                           _p(pend['rm']), _p(pend['rv']), pend['mom'], pend['eps'])
                pend['done'] = True
                self.nbnin += 1
                if stats is not None:
                    self.nfused['fwd'] += 1
                return self._conv_out(x, y, weight, bias, stride, pad, dil, relu, flops, xb, N, H, W, Ci, Co, kh, kw, Ho, Wo)
            self.flush_pending(pend)
        if xb is not None:
            # x.data is z of a BatchNorm+ReLU nobody materialised: this convolution is its one consumer
            assert not relu and not out_f32 and (kh, kw, stride, pad, dil) == (3, 3, 1, 1, 1) and not xb['used']
            xb['used'] = True
            bnm = xb['bn']
            y = self.act(N, Ho, Wo, Co)
            self.acall('fami_conv2d_fwd_xbn', _p(x.data), _p(wp), _p(None if bias is None else bias.data), _p(y),
                       N, H, W, Ci, Co, _p(None if stats is None else stats[0]), _p(None if stats is None else stats[1]),
                       _p(xb['slots']), xb['P'], _p(bnm.weight.data), _p(bnm.bias.data), _p(xb['mean']), _p(xb['invstd']),
                       _p(bnm.running_mean), _p(bnm.running_var), float(xb['mom']), float(bnm.eps))
            self.nfused['xbn'] += 1
            if stats is not None:
                self.nfused['fwd'] += 1
        elif stats is not None:
            assert not relu and not out_f32
            y = self.act(N, Ho, Wo, Co)
            self.acall('fami_conv2d_fwd_stats', _p(x.data), _p(wp), _p(None if bias is None else bias.data), _p(y),
                       N, H, W, Ci, Co, kh, kw, stride, pad, dil, _p(stats[0]), _p(stats[1]))
            self.nfused['fwd'] += 1
        elif self.half:
            y = self.empty(N, Ho, Wo, Co) if out_f32 else self.act(N, Ho, Wo, Co)
            self.acall('fami_conv2d_fwd', _p(x.data), _p(wp), _p(None if bias is None else bias.data), _p(y),
                      N, H, W, Ci, Co, kh, kw, stride, pad, dil, int(relu), 0, int(out_f32))
        else:
            y = self.empty(N, Ho, Wo, Co)
            self.call('fami_conv2d_fwd_f32', _p(x.data), _p(wp), _p(None if bias is None else bias.data), None,
                      _p(y), N, H, W, Ci, Co, kh, kw, stride, pad, dil, int(relu), 0)
        return self._conv_out(x, y, weight, bias, stride, pad, dil, relu, flops, xb, N, H, W, Ci, Co, kh, kw, Ho, Wo)

    def _conv_out(self, x, y, weight, bias, stride, pad, dil, relu, flops, xb, N, H, W, Ci, Co, kh, kw, Ho, Wo):
        """The output tensor of Engine.conv and its backward closure (the forward launch has been enqueued)."""
        need_w = self.rq(weight) or self.rq(bias)
        out = T(y, x.requires_grad or need_w)
        wl = self.wlane_scope and self.head_wlane
        wpair = self.wlane_pair            # (captured now: the backward closure runs long after the scope has closed)
        fuse_here = self.bn2 and self.serial_scope and self.serial_fuse     # see serial_scope in __init__
        if out.requires_grad:
            assert not relu, "fused relu epilogue is forward-only"
            geo = (N, H, W, Ci, Co, kh, kw, stride, pad, dil)

            def bwd():
                if out.grad is None:
                    return
                dy = out.grad
                on_wl = (self.use_wlane or wl) and need_w
                self.wlane_scope_now = bool(wl)
                # input gradient + weight gradient as ONE launch (fami_conv2d_bwd_pair_*, csrc/conv_pair.h): the weight gradient is a
                # leaf that otherwise sits between the input gradient and the next link of the chain on this lane
                if (self.bwd_pair and x.requires_grad and self.rq(weight) and (xb is None or not self.half) and self.defer_reduce
                        and not _ABL_WGRAD and (kh, kw, stride, pad, dil) == (3, 3, 1, 1, 1)
                        and (self.bwd_pair == 2 or not on_wl)
                        and (self.Q.fami_conv2d_bwd_pair_ok(*geo) if self.half else self.Q.fami_conv2d_bwd_pair_ok_f32(*geo))):
                    do_x(pair=True)
                    if self.rq(bias):
                        do_bias()
                    return
                xfirst = self.dgrad_first == 1 or (self.dgrad_first == 2 and on_wl) or (self.dgrad_first == 3 and not on_wl)
                if xfirst:
                    do_x()
                saved = self._enter_wlane(wpair, id(weight)) if on_wl else None
                if self.rq(weight):
                    g, acc = self.pgrad(weight)
                    self.wgrad(x.data, dy, g, geo, acc,
                               None if xb is None else (xb['mean'], xb['invstd'], xb['bn'].weight.data, xb['bn'].bias.data))
                    self.conv_flops += flops
                if self.rq(bias):
                    do_bias()
                if saved is not None:
                    self.stream = saved
                if not xfirst:
                    do_x()

            def do_bias():
                g, acc = self.pgrad(bias)
                ws = self.ws(self.Q.fami_channel_sum_workspace(Co))
                self.acall('fami_channel_sum', _p(out.grad), N * Ho * Wo, Co, _p(g), acc, _p(ws))

            def do_x(pair=False):
                dy = out.grad
                if x.requires_grad:
                    self.conv_flops += flops
                    gx, acc = self.gbuf(x)
                    wpd = self.packed(weight, 1)
                    rec = x.bnrec
                    pays = False
                    if self.fuse_bn_bwd_auto and rec is not None and (kh, stride, pad, dil) == (3, 1, 1, 1):
                        kind = self.Q.fami_conv_t6_eligible(N, Ho, Wo, Co, Ci)
                        t7 = self.fuse_bn_bwd_t7
                        pays = kind == 1 or ((kind == 2 or (kind == 3 and (self.fuse_bn_c64 & 2))) and (t7 == 2 or (t7 == 1 and not acc and rec['rmode'] != 1)))
                    stats = None
                    if (rec is not None and (self.fuse_bn_bwd or fuse_here or pays) and x.uses == 0 and not x.nofuse and x.lanes is not None
                            and len(x.lanes) == 1 and not x.f32grad):
                        # x is the output of a train-mode BatchNorm and this is the last contribution to its gradient:
                        # the epilogue applies the ReLU mask and takes the two sums of the BatchNorm backward
                        slots = self.zeros_bytes(self.Q.fami_bn_slots_bytes(Ci))
                        stats = (_p(rec['z']), _p(rec['y'] if rec['rmode'] == 1 else None), _p(rec['mean']), _p(rec['invstd']),
                                 _p(rec['gamma']), _p(rec['beta']), rec['rmode'], _p(slots))
                        rec['pre_bwd'] = slots
                        self.nfused['bwd'] += 1
                    xbn_args = None if xb is None else (xb['mean'], xb['invstd'], xb['bn'].weight.data, xb['bn'].bias.data)
                    if pair and (self.half or stats is None):
                        gw, accw = self.pgrad(weight)
                        self.conv_flops += flops
                        self.wgrad_pair(x.data, dy, wpd, gx, acc, gw, accw, geo, stats, xbn_args)
                        return
                    if pair:          # (f32 storage with a statistics epilogue on the input gradient: two launches)
                        gw, accw = self.pgrad(weight)
                        self.wgrad(x.data, dy, gw, geo, accw, xbn_args)
                        self.conv_flops += flops
                    if stats is not None:
                        self.acall('fami_conv2d_dgrad_bnstats', _p(dy), _p(wpd), _p(gx), *geo, acc, *stats)
                    elif self.half:
                        self.acall('fami_conv2d_dgrad', _p(dy), _p(wpd), _p(gx), *geo, acc)
                    else:
                        self.call('fami_conv2d_dgrad_f32', _p(dy), _p(wpd), None, _p(gx), *geo, acc)
            self.record_bwd(bwd, [weight, bias], (x,))
        return out

    def bn_fusable(self, P, C):
        """Can the statistics passes of a train-mode BatchNorm over [P, C] run in a convolution epilogue?"""
        return C % 4 == 0 and 4 <= C <= 1024 and not self.Q.fami_bn_is_small(P, C)

    def conv_fuses_stats(self, N, Ho, Wo, Ci, Co, kh, st, pd, dl):
        """Does the forward statistics pass of the BatchNorm behind this convolution go into its epilogue (given bn_fusable)?"""
        if ('1x1' in self.fuse_bn_skip and kh == 1) or ('s2' in self.fuse_bn_skip and st == 2):
            return False
        if self.fuse_bn_fwd3:
            kinds = (1, 2, 3) if (self.fuse_bn_c64 & 1) else (1, 2)
            return self.half and (kh, st, pd, dl) == (3, 1, 1, 1) and self.Q.fami_conv_t6_eligible(N, Ho, Wo, Ci, Co) in kinds
        if self.half and not (self.fuse_bn_c64 & 1) and (kh, st, pd, dl) == (3, 1, 1, 1):
            return self.Q.fami_conv_t6_eligible(N, Ho, Wo, Ci, Co) != 3
        return True

    def conv_bn(self, x, conv, bn, relu=False, residual=None):
        """nn.Conv2d -> nn.BatchNorm2d (-> + residual) (-> ReLU): basic_model.py:34-63, basic_layer.py:25-26.  With a
        train-mode BatchNorm over a large enough tensor the statistics pass is the convolution's epilogue."""
        N, H, W, _ = x.shape
        Co, _, kh, kw = conv.weight.shape
        st, pd, dl = conv.stride[0], conv.padding[0], conv.dilation[0]
        Ho = (H + 2 * pd - dl * (kh - 1) - 1) // st + 1
        Wo = (W + 2 * pd - dl * (kw - 1) - 1) // st + 1
        if (bn.training and self.fuse_bn_fwd and self.bn_fusable(N * Ho * Wo, Co)
                and self.conv_fuses_stats(N, Ho, Wo, conv.weight.shape[1], Co, kh, st, pd, dl)):
            slots = self.zeros_bytes(self.Q.fami_bn_slots_bytes(Co))
            z = self.conv(x, conv.weight, conv.bias, st, pd, dl, stats=(slots, bn.running_mean))
            return self.bn(z, bn, relu=relu, residual=residual, pre=slots)
        return self.bn(self.conv(x, conv.weight, conv.bias, st, pd, dl), bn, relu=relu, residual=residual)

    def conv_bn_relu_into(self, x, conv, bn, nxt):
        """conv -> BatchNorm -> ReLU whose ONLY consumer is the 3x3 stride-1 convolution `nxt` (conv1 -> bn1 -> relu ->
        conv2 of a BasicBlock, basic_model.py:34-63).  Train-mode statistics, 16-bit storage or f32 on the split-product kernels: the normalised tensor is never
        written -- conv's epilogue takes the statistics, nxt's forward and weight-gradient kernels apply scale / shift /
        ReLU while they stage z (conv_epi.h XBN), and the BatchNorm backward recomputes the ReLU mask from z.  Anything
        else falls back to conv_bn(relu=True).  FAMI_XBN (see __init__)."""
        N, H, W, _ = x.shape
        Co, _, kh, kw = conv.weight.shape
        st, pd, dl = conv.stride[0], conv.padding[0], conv.dilation[0]
        Ho = (H + 2 * pd - dl * (kh - 1) - 1) // st + 1
        Wo = (W + 2 * pd - dl * (kw - 1) - 1) // st + 1
        P = N * Ho * Wo
        okq = self.Q.fami_conv2d_xbn_ok if self.half else self.Q.fami_conv2d_xbn_ok_f32
        ok = (self.use_xbn and self.bn2 and self.defer_reduce and bn.training and self.fuse_bn_fwd
              and self.defer_bn is None and bn.running_mean is not None and self.bn_fusable(P, Co)
              and tuple(nxt.weight.shape[1:]) == (Co, 3, 3) and nxt.stride[0] == 1 and nxt.padding[0] == 1
              and nxt.dilation[0] == 1 and okq(N, Ho, Wo, Co, nxt.weight.shape[0]))
        if not ok:
            if self.abl_bn1 and bn.training and self.fuse_bn_fwd and self.bn_fusable(P, Co):
                return self._abl_bn1(x, conv, bn, st, pd, dl, Co)
            # 16-bit storage (round 6): the normalised tensor IS written, but by `nxt`'s own launch (Engine.conv takes the pending
            # apply pass: fami_conv2d_fwd_bnin_*) -- where conv's epilogue takes the statistics and the weight-resident kernel takes nxt
            if (self.bn_in and bn.training and self.fuse_bn_fwd and self.bn_fusable(P, Co)
                    and self.conv_fuses_stats(N, Ho, Wo, conv.weight.shape[1], Co, kh, st, pd, dl)
                    and tuple(nxt.weight.shape[1:]) == (Co, 3, 3) and nxt.stride[0] == 1 and nxt.padding[0] == 1
                    and nxt.dilation[0] == 1 and self.Q.fami_conv2d_fwd_bnin_ok(N, Ho, Wo, Co, nxt.weight.shape[0])):
                slots = self.zeros_bytes(self.Q.fami_bn_slots_bytes(Co))
                z = self.conv(x, conv.weight, conv.bias, st, pd, dl, stats=(slots, bn.running_mean))
                return self.bn(z, bn, relu=True, pre=slots, defer=True)
            return self.conv_bn(x, conv, bn, relu=True)
        slots = self.zeros_bytes(self.Q.fami_bn_slots_bytes(Co))
        z = self.conv(x, conv.weight, conv.bias, st, pd, dl, stats=(slots, bn.running_mean))
        self._lane_guard(('running statistics', id(bn)))
        mean, invstd = self.empty(Co), self.empty(Co)
        self.bn_trained.append(bn)
        need_p = self.rq(bn.weight)
        out = T(z.data, z.requires_grad or need_p)
        out.xbn = {'bn': bn, 'slots': slots, 'mean': mean, 'invstd': invstd, 'P': P,
                   'mom': 0.1 if bn.momentum is None else bn.momentum, 'used': False}
        if out.requires_grad:
            def bwd():
                if out.grad is None:
                    return
                gx, accx = self.gbuf(z) if z.requires_grad else (self.act(N, Ho, Wo, Co), 0)
                gg = gb = None
                accp = 0
                if need_p:
                    gg, accp = self.pgrad(bn.weight)
                    gb, _ = self.pgrad(bn.bias)
                self.acall('fami_bn_bwd2', _p(out.grad), _p(z.data), None, _p(mean), _p(invstd), _p(bn.weight.data),
                           _p(bn.bias.data), _p(gx), _p(gg), _p(gb), None, P, Co, 2, accx, accp, 0,
                           _p(self.zeros_bytes(self.Q.fami_bn_slots_bytes(Co))))
            self.record_bwd(bwd, [bn.weight, bn.bias], (z,))
        return out

    _abl_skip = 0

    def _abl_bn1(self, x, conv, bn, st, pd, dl, Co):
        """FAMI_ABL_BN1 (WRONG results, timing only): conv -> statistics in its epilogue -> BatchNorm whose apply pass (bit 1:
        a one-workgroup finalize runs instead and the output aliases z) and / or backward apply pass (bit 2: the gradient
        passes through) are skipped."""
        slots = self.zeros_bytes(self.Q.fami_bn_slots_bytes(Co))
        z = self.conv(x, conv.weight, conv.bias, st, pd, dl, stats=(slots, bn.running_mean))
        self._abl_skip = self.abl_bn1
        try:
            y = self.bn(z, bn, relu=True, pre=slots)
        finally:
            self._abl_skip = 0
        return y

    def bn(self, x, bn, relu=False, residual=None, pre=None, defer=False):
        """nn.BatchNorm2d (+ residual add) (+ ReLU).  Train mode: batch statistics and running-stat update.
        pre: slot rows the producing convolution's epilogue has filled (Engine.conv_bn): apply pass only.
        defer (with pre, no residual): the apply pass is NOT launched -- the output carries `pending`, and the convolution that
        consumes it runs the pass inside its own launch (Engine.conv -> fami_conv2d_fwd_bnin_*) or launches it first."""
        shp = x.shape
        C = shp[-1]
        P = x.data.numel() // C
        mean, invstd = self.empty(C), self.empty(C)
        y = self.like(x.data)
        pending = None
        if bn.training:
            deferred = self.defer_bn is not None and bn.running_mean is not None
            if not deferred:
                self._lane_guard(('running statistics', id(bn)))
            mom = 0.1 if bn.momentum is None else bn.momentum
            bn2 = self.bn2
            if pre is not None and (self._abl_skip & 1):
                y = x.data
                self.call('fami_bn_finalize_slots_f32', _p(pre), P, C, _p(mean), _p(invstd), _p(None if deferred else bn.running_mean),
                          _p(None if deferred else bn.running_var), float(mom), float(bn.eps))
            elif pre is not None and defer and residual is None and relu:
                pending = {'z': x.data, 'y': y, 'slots': pre, 'P': P, 'C': C, 'bn': bn, 'mean': mean, 'invstd': invstd,
                           'rm': None if deferred else bn.running_mean, 'rv': None if deferred else bn.running_var,
                           'mom': float(mom), 'eps': float(bn.eps), 'done': False}
                self._pending.append(pending)
            elif pre is not None:
                self.acall('fami_bn_apply_slots', _p(x.data), _p(None if residual is None else residual.data), _p(y),
                           _p(bn.weight.data), _p(bn.bias.data), _p(mean), _p(invstd),
                           _p(None if deferred else bn.running_mean), _p(None if deferred else bn.running_var), P, C,
                           int(relu), float(mom), float(bn.eps), _p(pre))
            else:
                ws = self.zeros_bytes(self.Q.fami_bn_slots_bytes(C)) if bn2 else self.ws(self.Q.fami_bn_workspace(C))
                self.acall('fami_bn_train_fwd2' if bn2 else 'fami_bn_train_fwd', _p(x.data),
                           _p(None if residual is None else residual.data), _p(y),
                           _p(bn.weight.data), _p(bn.bias.data), _p(mean), _p(invstd),
                           _p(None if deferred else bn.running_mean), _p(None if deferred else bn.running_var), P, C,
                           int(relu), float(mom), float(bn.eps), _p(ws))
            if deferred:
                self.defer_bn.append((bn, mean, invstd, P, float(mom)))
            self.bn_trained.append(bn)
        else:
            assert pre is None
            self.call('fami_bn_eval_stats_f32', _p(bn.running_mean), _p(bn.running_var), _p(mean), _p(invstd), C,
                      float(bn.eps))
            self.acall('fami_bn_apply', _p(x.data), _p(mean), _p(invstd), _p(bn.weight.data), _p(bn.bias.data),
                       _p(None if residual is None else residual.data), _p(y), P, C, int(relu))
        need_p = self.rq(bn.weight)
        rg = x.requires_grad or need_p or (residual is not None and residual.requires_grad)
        out = T(y, rg)
        out.pending = pending
        if rg:
            training = bn.training
            # ReLU mask in backward: from y when a residual was added, else recomputed from x (one tensor read less)
            rmode = (1 if residual is not None else 2) if relu else 0
            rec = None
            if training and self.bn2 and self.bn_fusable(P, C):
                rec = out.bnrec = {'z': x.data, 'y': y, 'mean': mean, 'invstd': invstd, 'gamma': bn.weight.data,
                                   'beta': bn.bias.data, 'rmode': rmode, 'pre_bwd': None}

            abl_bwd = bool(self._abl_skip & 2)
            if abl_bwd:
                rec = out.bnrec = None

            def bwd():
                if out.grad is None:
                    return
                if abl_bwd:                  # FAMI_ABL_BN1 bit 2 (WRONG, timing only): no backward passes at all
                    if x.requires_grad:
                        x.grad = out.grad
                    return
                if not training:
                    raise NotImplementedError("backward through eval-mode BatchNorm is outside the training hot path")
                gx, accx = self.gbuf(x) if x.requires_grad else (self.act(*shp), 0)
                gg = gb = None
                accp = 0
                if need_p:
                    gg, accp = self.pgrad(bn.weight)
                    gb, _ = self.pgrad(bn.bias)
                gr, accr = (None, 0)
                if residual is not None and residual.requires_grad:
                    gr, accr = self.gbuf(residual)
                if rec is not None and rec['pre_bwd'] is not None:
                    # out.grad holds dz (mask applied) and the slot rows the two sums: fami_conv2d_dgrad_bnstats_*
                    self.acall('fami_bn_bwd_apply_slots', _p(out.grad), _p(x.data), _p(mean), _p(invstd),
                               _p(bn.weight.data), _p(bn.bias.data), _p(gx), _p(gg), _p(gb), _p(gr), P, C, accx, accp,
                               accr, _p(rec['pre_bwd']))
                elif self.bn2:
                    self.acall('fami_bn_bwd2', _p(out.grad), _p(x.data), _p(y), _p(mean), _p(invstd),
                               _p(bn.weight.data), _p(bn.bias.data), _p(gx), _p(gg), _p(gb), _p(gr), P, C, rmode, accx,
                               accp, accr, _p(self.zeros_bytes(self.Q.fami_bn_slots_bytes(C))))
                else:
                    ws = self.ws(self.Q.fami_bn_workspace(C))
                    self.acall('fami_bn_bwd', _p(out.grad), _p(x.data), _p(y), _p(mean), _p(invstd),
                               _p(bn.weight.data), _p(gx), _p(gg), _p(gb), _p(gr), P, C, int(relu), accx, accp, accr,
                               _p(ws))
            self.record_bwd(bwd, [bn.weight, bn.bias], (x, residual))
        return out

    # ------------------------------------------------------------------ fuse (hrnet.py:159-168)
    def conv_fuse_term(self, x, conv, bn, shift):
        """conv -> fuse term (hrnet.py:99-143: every cross-resolution path ends in conv + BatchNorm): the term's statistics
        come out of the convolution's epilogue when the tensor is large enough."""
        N, H, W, _ = x.shape
        Co, _, kh, kw = conv.weight.shape
        st, pd, dl = conv.stride[0], conv.padding[0], conv.dilation[0]
        Ho = (H + 2 * pd - dl * (kh - 1) - 1) // st + 1
        Wo = (W + 2 * pd - dl * (kw - 1) - 1) // st + 1
        if (bn.training and self.fuse_bn_fwd and self.bn_fusable(N * Ho * Wo, Co)
                and self.conv_fuses_stats(N, Ho, Wo, conv.weight.shape[1], Co, kh, st, pd, dl)):
            slots = self.zeros_bytes(self.Q.fami_bn_slots_bytes(Co))
            z = self.conv(x, conv.weight, conv.bias, st, pd, dl, stats=(slots, bn.running_mean))
            return self.fuse_term(z, bn, shift, pre=slots)
        return self.fuse_term(self.conv(x, conv.weight, conv.bias, st, pd, dl), bn, shift)

    def fuse_term(self, x, bn, shift, pre=None):
        """One term of a HighResolutionModule fuse sum (hrnet.py:151-172): its BatchNorm statistics now, its backward as
        its own tape entry on the CURRENT lane -- all terms fed by branch j run on lane j, so the gradient of x_j is
        accumulated by one stream.  -> handle for Engine.fuse.  pre: slot rows filled by the producing convolution."""
        C = x.shape[-1]
        h = {'x': x, 'bn': bn, 'shift': shift, 'stats': None, 'out': None}
        if bn is not None:
            Pk = x.data.numel() // C
            mean, invstd = self.empty(C), self.empty(C)
            if bn.training:
                self._lane_guard(('running statistics', id(bn)))
                mom = 0.1 if bn.momentum is None else bn.momentum
                if pre is not None:
                    self.call('fami_bn_finalize_slots_f32', _p(pre), Pk, C, _p(mean), _p(invstd), _p(bn.running_mean),
                              _p(bn.running_var), float(mom), float(bn.eps))
                else:
                    ws = self.ws(self.Q.fami_bn_workspace(C))
                    self.acall('fami_bn_stats', _p(x.data), Pk, C, _p(mean), _p(invstd), _p(bn.running_mean),
                               _p(bn.running_var), float(mom), float(bn.eps), _p(ws))
                self.bn_trained.append(bn)
            else:
                self.call('fami_bn_eval_stats_f32', _p(bn.running_mean), _p(bn.running_var), _p(mean), _p(invstd),
                          C, float(bn.eps))
            h['stats'] = (mean, invstd)
        need_p = bn is not None and self.rq(bn.weight)
        if self.record and (x.requires_grad or need_p):
            def bwd():
                out = h['out']
                if out is None or out.grad is None:
                    return
                dy, y = out.grad, out.data
                N, H, W, _ = out.shape
                if bn is None:
                    if x.requires_grad:
                        g, acc = self.gbuf(x)
                        self.acall('fami_relu_bwd', _p(dy), _p(y), _p(g), y.numel(), acc)
                    return
                if not bn.training:
                    raise NotImplementedError("backward through eval-mode BatchNorm")
                Pk = x.data.numel() // C
                gx, accx = self.gbuf(x) if x.requires_grad else (self.like(x.data), 0)
                gg = gb = None
                accp = 0
                if need_p:
                    gg, accp = self.pgrad(bn.weight)
                    gb, _ = self.pgrad(bn.bias)
                st = h['stats']
                # two-launch form (statistics into fp64 slot rows, finalize folded into the apply pass) where the BatchNorm is
                # large enough for it -- the three-launch form's one-workgroup finalize sat between the two passes of all 53
                # fuse-term BatchNorms of a step (5 us each); FAMI_FUSE_TERM_BN2=0 restores it
                two = self.bn2 and self.fuse_term_bn2 and self.bn_fusable(Pk, C)
                ws = self.zeros_bytes(self.Q.fami_bn_slots_bytes(C)) if two else self.ws(self.Q.fami_bn_workspace(C))
                if shift == 0:
                    if two:
                        self.acall('fami_bn_bwd2', _p(dy), _p(x.data), _p(y), _p(st[0]), _p(st[1]), _p(bn.weight.data),
                                   _p(bn.bias.data), _p(gx), _p(gg), _p(gb), None, Pk, C, 1, accx, accp, 0, _p(ws))
                    else:
                        self.acall('fami_bn_bwd', _p(dy), _p(x.data), _p(y), _p(st[0]), _p(st[1]),
                                   _p(bn.weight.data), _p(gx), _p(gg), _p(gb), None, Pk, C, 1, accx, accp, 0, _p(ws))
                else:
                    low = self.like(x.data)
                    self.acall('fami_pool_relu_bwd', _p(dy), _p(y), _p(low), N, H >> shift, W >> shift, C, shift, 1)
                    if two:
                        self.acall('fami_bn_bwd2', _p(low), _p(x.data), None, _p(st[0]), _p(st[1]), _p(bn.weight.data),
                                   _p(bn.bias.data), _p(gx), _p(gg), _p(gb), None, Pk, C, 0, accx, accp, 0, _p(ws))
                    else:
                        self.acall('fami_bn_bwd', _p(low), _p(x.data), None, _p(st[0]), _p(st[1]),
                                   _p(bn.weight.data), _p(gx), _p(gg), _p(gb), None, Pk, C, 0, accx, accp, 0, _p(ws))
            self.record_bwd(bwd, [bn.weight, bn.bias] if bn is not None else [], (x,))
        return h

    def fuse(self, handles):
        """handles from fuse_term (k <= 4): returns relu(sum_k up_{2^shift}(bn_k(x_k))).  The terms own the backward."""
        k = len(handles)
        assert 1 <= k <= 4
        big = [h for h in handles if h['shift'] == 0][0]['x']
        N, H, W, C = big.shape
        PA = ctypes.c_void_p * k
        xs = PA(*[_p(h['x'].data) for h in handles])
        mean_a = PA(*[None if h['stats'] is None else _p(h['stats'][0]) for h in handles])
        inv_a = PA(*[None if h['stats'] is None else _p(h['stats'][1]) for h in handles])
        gam_a = PA(*[None if h['bn'] is None else _p(h['bn'].weight.data) for h in handles])
        bet_a = PA(*[None if h['bn'] is None else _p(h['bn'].bias.data) for h in handles])
        sh_a = (ctypes.c_int * k)(*[h['shift'] for h in handles])
        y = self.act(N, H, W, C)
        self.acall('fami_fuse_sum', k, xs, mean_a, inv_a, gam_a, bet_a, sh_a, _p(y), N, H, W, C, 1)
        rg = self.record and any(h['x'].requires_grad or (h['bn'] is not None and self.rq(h['bn'].weight)) for h in handles)
        out = T(y, rg)
        for h in handles:
            h['out'] = out
        return out

    # ------------------------------------------------------------------ glue ops
    def batch_slice(self, x, n0, n1, terminal=False):
        """torch.chunk on the batch axis (Alignment_V15.py:121-125): a view; gradients land in the parent slice.
        terminal: a model output that nothing inside the graph consumes -- its gradient can only arrive through a seed
        (gbuf creates the parent's buffer then), so the parent is not zero-filled up front: without a seed the
        producer's backward is skipped, as autograd skips a branch whose .grad is None (kf_hm made the
        final layer's weight gradient and a dgrad of zeros run every step)."""
        x.nofuse = True         # gradients reach x through its slices: no "last contribution" bookkeeping on it
        if x.requires_grad and not terminal and all(x is not p for p in self._sliced):
            self._sliced.append(x)
        return T(x.data[n0:n1], x.requires_grad, parent=x, n0=n0, n1=n1)

    def sub(self, a, b):
        y = self.like(a.data)
        self.acall('fami_axpby', _p(a.data), _p(b.data), _p(y), y.numel(), 1.0, -1.0)
        out = T(y, a.requires_grad or b.requires_grad)
        if out.requires_grad:
            def bwd():
                if out.grad is None:
                    return
                for t, sgn in ((a, 1.0), (b, -1.0)):
                    if t.requires_grad:
                        g, acc = self.gbuf(t)
                        self.acall('fami_axpby', _p(out.grad), _p(g) if acc else None, _p(g), g.numel(), sgn, 1.0)
            self.record_bwd(bwd, (), (a, b))
        return out

    def concat(self, xs):
        """torch.cat on channels (Alignment_V15.py:139,143,160)."""
        N, H, W, _ = xs[0].shape
        Ct = sum(x.shape[3] for x in xs)
        P = N * H * W
        y = self.act(N, H, W, Ct)
        n = len(xs)
        # one launch for the lot (the head's three concatenations); distinct tensors only: the backward writes every slice at once
        one = self.concat_one and 2 <= n <= 4 and all(x.shape[3] % 4 == 0 for x in xs) and len({id(x) for x in xs}) == n
        cs = (ctypes.c_int * 4)(*[x.shape[3] for x in xs]) if one else None
        if one:
            self.acall('fami_concat_channels', (ctypes.c_void_p * 4)(*[x.data.data_ptr() for x in xs]), cs, n, _p(y), P)
        else:
            off = 0
            for x in xs:
                c = x.shape[3]
                self.acall('fami_copy_channels', _p(x.data), _p(y), P, c, 0, Ct, off, c, 0)
                off += c
        out = T(y, any(x.requires_grad for x in xs))
        if out.requires_grad:
            def bwd():
                if out.grad is None:
                    return
                if one:
                    dst, accs = (ctypes.c_void_p * 4)(), (ctypes.c_int * 4)()
                    for k, x in enumerate(xs):
                        if x.requires_grad:
                            g, acc = self.gbuf(x)
                            dst[k], accs[k] = g.data_ptr(), acc
                    self.acall('fami_split_channels', _p(out.grad), dst, cs, accs, n, P)
                    return
                o = 0
                for x in xs:
                    c = x.shape[3]
                    if x.requires_grad:
                        g, acc = self.gbuf(x)
                        self.acall('fami_copy_channels', _p(out.grad), _p(g), P, Ct, o, c, 0, c, acc)
                    o += c
            self.record_bwd(bwd, (), tuple(xs))
        return out

    def flatten_chw(self, x):
        """nn.Flatten on an NCHW tensor: [N,H,W,C] -> [N, C*H*W] in (c,h,w) order (Alignment_V15.py:68)."""
        N, H, W, C = x.shape
        y = self.empty(N, C * H * W)
        self.acall('fami_nhwc_to_nchw', _p(x.data), _p(y), N, C, H, W, 0)
        out = T(y, x.requires_grad, f32grad=True)
        if out.requires_grad:
            def bwd():
                if out.grad is None:
                    return
                g, acc = self.gbuf(x)
                if acc:
                    tmp = self.like(g)
                    self.acall('fami_nchw_to_nhwc', _p(out.grad), _p(tmp), N, C, H, W)
                    self.acall('fami_axpby', _p(tmp), _p(g), _p(g), g.numel(), 1.0, 1.0)
                else:
                    self.acall('fami_nchw_to_nhwc', _p(out.grad), _p(g), N, C, H, W)
            self.record_bwd(bwd, (), (x,))
        return out

    def linear(self, x, lin):
        M, K = x.shape
        Nn = lin.out_features
        y = self.empty(M, Nn)
        self.call('fami_linear_fwd_f32', _p(x.data), _p(lin.weight.data), _p(None if lin.bias is None else lin.bias.data),
                  _p(y), M, K, Nn)
        need_p = self.rq(lin.weight)
        out = T(y, x.requires_grad or need_p, f32grad=True)
        if out.requires_grad:
            def bwd():
                if out.grad is None:
                    return
                gx = gw = gb = None
                accx = accp = 0
                if x.requires_grad:
                    gx, accx = self.gbuf(x)
                if need_p:
                    gw, accp = self.pgrad(lin.weight)
                    if lin.bias is not None:
                        gb, _ = self.pgrad(lin.bias)
                self.call('fami_linear_bwd_f32', _p(out.grad), _p(x.data), _p(lin.weight.data), _p(gx), _p(gw),
                          _p(gb), M, K, Nn, accx, accp)
            self.record_bwd(bwd, [lin.weight, lin.bias], (x,))
        return out

    # ------------------------------------------------------------------ alignment ops
    def scale_pairs(self, t, sx, sy):
        """t [B,2] f32 -> t * (sx, sy) (gradient scaled the same way)."""
        B = t.shape[0]
        y = self.empty(B, 2)
        self.call('fami_scale_pairs_f32', _p(t.data), _p(y), B, float(sx), float(sy), 0)
        out = T(y, t.requires_grad, f32grad=True)
        if out.requires_grad:
            def bwd():
                if out.grad is None:
                    return
                g, acc = self.gbuf(t)
                self.call('fami_scale_pairs_f32', _p(out.grad), _p(g), B, float(sx), float(sy), acc)
            self.record_bwd(bwd, (), (t,))
        return out

    def shift(self, x, t, align_corners=True):
        """kornia warp_affine with a pure translation t=[B,2]=(tx,ty) (Alignment_V15.py:133-135).  align_corners=False:
        the kornia <= 0.4 default -- the matrix is normalised for [0, W-1] but the sampling grid is built and read with
        align_corners=False, which for a pure translation is exactly a shift by (tx*W/(W-1), ty*H/(H-1))."""
        B, H, W, C = x.shape
        if not align_corners:
            t = self.scale_pairs(t, W / max(W - 1, 1), H / max(H - 1, 1))
        y = self.like(x.data)
        self.acall('fami_shift_bilinear_fwd', _p(x.data), _p(t.data), _p(y), B, H, W, C)
        out = T(y, x.requires_grad or t.requires_grad)
        if out.requires_grad:
            def bwd():
                if out.grad is None:
                    return
                gs = gt = None
                accs = acct = 0
                if x.requires_grad:
                    gs, accs = self.gbuf(x)
                if t.requires_grad:
                    gt, acct = self.gbuf(t)
                ws = self.ws(self.Q.fami_shift_workspace(B))
                self.acall('fami_shift_bilinear_bwd', _p(out.grad), _p(x.data), _p(t.data), _p(gs), _p(gt), B, H,
                           W, C, accs, acct, _p(ws))
            self.record_bwd(bwd, (), (x, t))
        return out

    def dcn(self, x, off, msk, weight, bias, G, pad=3, dil=3):
        """torchvision DeformConv2d(C,Co,3,padding=3,dilation=3)(x, off, msk) (Alignment_V15.py:146-158).
        msk None: `off` is the merged predictor's output [B,H,W,3GK] = per pixel (2GK offsets | GK masks)."""
        om = msk is None
        B, H, W, C = x.shape
        Co, _, kh, kw = weight.shape
        K = kh * kw
        wp = (getattr(self, 'prepacked_dcn_fwd', None) or {}).get(id(weight))      # packed at the start of the step (Trainer)
        if wp is not None:
            ev = getattr(self, 'dcn_fwd_ready', None)
            if ev is not None:
                self.wait_main(ev)
                self.dcn_fwd_ready = None
        else:
            n = self.Q.fami_dcn_packed_weight_elems(Co, C, kh, kw, G)
            wp = self.empty(n)
            self.acall('fami_dcn_pack_weight', _p(weight.data), _p(wp), Co, C, kh, kw, G)   # 16-bit modes: + the 16-bit image
        y = self.act(B, H, W, Co)
        if om:
            assert off.shape[3] == 3 * G * K
            self.acall('fami_dcn_fwd_om', _p(x.data), _p(off.data), _p(wp), _p(bias.data), _p(y), B, H, W, C, Co, G, kh, kw, 1, pad, dil)
        else:
            self.acall('fami_dcn_fwd', _p(x.data), _p(off.data), _p(msk.data), _p(wp), _p(bias.data), _p(y), B, H, W,
                       C, Co, G, kh, kw, 1, pad, dil)
        rg = x.requires_grad or off.requires_grad or (not om and msk.requires_grad) or self.rq(weight)
        out = T(y, rg)
        wl = self.wlane_scope and self.head_wlane
        if rg:
            def bwd():
                if out.grad is None:
                    return
                dy = out.grad
                P = B * H * W
                CK = C * K
                wpb = (getattr(self, 'prepacked_dcn_bwd', None) or {}).get(id(weight))     # packed at the start of the step (Trainer)
                if wpb is None:
                    nwp = self.Q.fami_dcn_packed_weight_bwd_elems(Co, C, kh, kw, G)
                    wpb = self.empty(nwp)
                    self.call('fami_dcn_pack_weight_bwd_f32', _p(weight.data), _p(wpb), Co, C, kh, kw, G)
                # columns of the modulated-sample matrix: C*K (OIHW order) or the register-fed kernel's own order, padded
                esz = 2 if self.half else 4
                colw = self.Q.fami_dcn_bwd_col_width(C, Co, G, kh, kw, 1, dil, esz, int(self.deterministic))
                permuted = bool(self.Q.fami_dcn_bwd_col_permuted(C, Co, G, kh, kw, 1, dil, esz, int(self.deterministic)))
                col = self.act(P, colw) if self.rq(weight) else None
                gx = gx32 = goff = gmsk = None
                acco = accx = 0
                if off.requires_grad:
                    goff, acco = self.gbuf(off)
                    if not om:
                        gmsk, accm = self.gbuf(msk)
                        assert acco == accm
                if self.deterministic:
                    if x.requires_grad:
                        gx, accx = self.gbuf(x)
                    ws = self.ws(self.Q.fami_dcn_bwd_det_workspace(B, H, W, C))
                    if om:
                        self.acall('fami_dcn_bwd_det_om', _p(x.data), _p(off.data), _p(dy), _p(wpb), _p(col), _p(gx), _p(goff),
                                   B, H, W, C, Co, G, kh, kw, 1, pad, dil, acco, accx, _p(ws))
                    else:
                        self.acall('fami_dcn_bwd_det', _p(x.data), _p(off.data), _p(msk.data), _p(dy), _p(wpb), _p(col),
                                   _p(gx), _p(goff), _p(gmsk), B, H, W, C, Co, G, kh, kw, 1, pad, dil, acco, accx, _p(ws))
                else:
                    if x.requires_grad:
                        gx, accx = self.gbuf(x)
                        if self.half:      # the scatter accumulates in an fp32 buffer (float atomics)
                            gx32 = self.fill(self.empty(*gx.shape))
                        else:
                            gx32 = gx
                            if not accx:
                                self.fill(gx)
                    if om:
                        self.acall('fami_dcn_bwd_om', _p(x.data), _p(off.data), _p(dy), _p(wpb), _p(col), _p(gx32), _p(goff),
                                   B, H, W, C, Co, G, kh, kw, 1, pad, dil, acco)
                    else:
                        self.acall('fami_dcn_bwd', _p(x.data), _p(off.data), _p(msk.data), _p(dy), _p(wpb), _p(col),
                                   _p(gx32), _p(goff), _p(gmsk), B, H, W, C, Co, G, kh, kw, 1, pad, dil, acco)
                    if gx is not None and gx32 is not gx:
                        self.acall('fami_cast_add', _p(gx32), _p(gx), gx.numel(), accx)
                if self.rq(weight):
                    saved = self._enter_wlane(False, id(weight)) if wl else None      # leaves of the backward graph (see __init__)
                    g, acc = self.pgrad(weight)
                    if not permuted:
                        self.wgrad(col, dy, g, (1, 1, P, CK, Co, 1, 1, 1, 0, 1), acc)
                    else:
                        dwp = self.empty(Co, colw)
                        self.wgrad(col, dy, dwp, (1, 1, P, colw, Co, 1, 1, 1, 0, 1), 0)
                        self.flush_reduces(self.stream)      # (the deferred slab reduce writes dwp)
                        self.call('fami_dcn_col_dw_unpermute_f32', _p(dwp), _p(g), Co, C, G, kh, kw, 1, dil, esz, acc)
                    gb, accb = self.pgrad(bias)
                    ws2 = self.ws(self.Q.fami_channel_sum_workspace(Co))
                    self.acall('fami_channel_sum', _p(dy), P, Co, _p(gb), accb, _p(ws2))
                    if saved is not None:
                        self.stream = saved
            self.record_bwd(bwd, [weight, bias], (x, off) if om else (x, off, msk))
        return out

    # ------------------------------------------------------------------ MI estimators (Alignment_V15.py:250-277)
    def softmax_kl(self, a_nchw, b, temperature=0.05):
        """value = mean_{rows,l} t*(log t - a), a = softmax(A/T) (detached), t = softmax(Bt/T); rows = (n,c), l = h*w.
        a_nchw: torch tensor [N,C,H,W]; b: T (NHWC).  Returns (scalar tensor [1], seed function(gscale))."""
        N, H, W, C = b.shape
        bt = self.to_nchw(b)
        R, Ln = N * C, H * W
        val = self.empty(1)
        stats = self.empty(R, 5)
        ws = self.ws(R * 4)
        self.call('fami_softmax_kl_fwd_f32', _p(a_nchw), _p(bt), _p(val), _p(stats), R, Ln, float(temperature), _p(ws))

        def seed(gscale, gdev=None, split=False):
            """split: -> closure that adds the gradient to b on the caller's lane (the six MI terms of the training step run
            their backward kernels on parallel lanes and accumulate after the join)."""
            if not b.requires_grad:
                return (lambda: None) if split else None
            d = self.empty(N, C, H, W)
            self.call('fami_softmax_kl_bwd_f32', _p(a_nchw), _p(bt), _p(stats), _p(d), R, Ln, float(temperature),
                      float(gscale), _p(gdev), 0)
            if split:
                return self.seed_nchw_split(b, d)
            self.seed_nchw(b, d)
        return val, seed

    # ------------------------------------------------------------------ backward driver
    def backward(self, on_params_done=None):
        """Walk the tape in reverse.  `on_params_done(list_of_params)` (optional) fires once a parameter's
        last gradient contribution has been enqueued -- the hook the data-parallel bucket all-reduce hangs on."""
        self.sync_stream()
        assert all(q['done'] for q in self._pending), 'a deferred BatchNorm apply pass was never launched'
        remaining = None
        if on_params_done is not None:
            remaining = {}
            for _, ps, _lane, _ins in self.tape:
                for p in ps:
                    if p is not None and p.requires_grad:
                        remaining[id(p)] = remaining.get(id(p), 0) + 1
        for par in self._sliced:
            if par.grad is None:
                par.grad = self.fill(self.new_grad(par))
        pending = []
        for fn, ps, lane, ins in reversed(self.tape):
            if lane != self.lane:
                self.set_lane(lane)
            for t in ins:
                if t is not None:
                    t.uses -= 1
            fn()
            if remaining is not None and ps:
                for p in ps:
                    if p is not None and p.requires_grad:
                        remaining[id(p)] -= 1
                        if remaining[id(p)] == 0:
                            pending.append(p)
            # bucket hooks only fire from lane 0 outside a forked region: by then every lane's gradient
            # kernels are ordered before whatever the hook enqueues on the main stream
            if pending and on_params_done is not None and self._forked == 0 and self.lane == 0:
                self.flush_reduces()       # the completed parameters' gradients must be final before the hook reads them
                on_params_done(pending)
                pending = []
        self.set_lane(0)
        self.flush_reduces()
        self.sync_wgrad_lane()
        if pending and on_params_done is not None:
            on_params_done(pending)
        self.tape = []
