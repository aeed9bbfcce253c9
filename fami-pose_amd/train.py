"""The training step of the hot path, MI355X-first.

Restates engine/core/functions/alignment_mi_function_term6_1.py:104-156
(model call, JointMSELoss * W_mse, MI combination with alpha 0.5 / beta 0.1,
zero_grad / backward / Adam step) -- the reference loop itself is not importable
 -- with these design choices:

* parameters, gradients and both Adam moments live in four flat fp32 arenas
  (one allocation each); nn.Parameter objects are views, so state_dict() is
  unchanged.  Adam is ONE kernel over the arena; the data-parallel all-reduce
  needs no flatten/unflatten copies: a bucket is a slice of the gradient arena.
* one process per GPU; gradients are averaged with RCCL all-reduce over xGMI,
  bucket by bucket, each bucket launched as soon as the backward tape has
  enqueued the last contribution to it (reverse registration order: the head
  and stage 4 first), overlapping with the rest of backward.  BatchNorm
  statistics stay per replica (what the reference's nn.DataParallel does).
* the whole launch sequence of a step is captured once into hipGraphs and
  replayed (torch.cuda.CUDAGraph is used purely as the capture/replay handle).
  With world_size > 1 the backward is cut into one graph per bucket so the
  collectives run between graph launches on RCCL's stream.
"""
import torch
import torch.distributed as dist

from . import options
from ._lib import lib
from .engine import Engine, _p


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


class FlatAdam:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0) on one flat fp32 arena
    (posetimation/optimizer/optimizer.py:66-68 builds the reference's).  The step count, the learning rate and the two
    bias corrections live in a device vector, so an LR schedule needs no re-capture of a hipGraph; betas / eps /
    weight decay are kernel arguments: changing them (set_hyper, or loading a checkpoint) bumps `hyper_version`,
    which makes a graph-mode Trainer re-capture."""

    def __init__(self, flat_param, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.p = flat_param
        self.grad = torch.zeros_like(flat_param)
        self.m = torch.zeros_like(flat_param)
        self.v = torch.zeros_like(flat_param)
        self.betas, self.eps, self.wd = tuple(betas), eps, weight_decay
        self.lr = float(lr)
        # the undecayed rate an LR schedule scales (torch stores it as param_groups[0]['initial_lr'] the first time a
        # scheduler is attached; checkpoints carry it so a resumed MultiStepLR does not decay an already decayed rate)
        self.initial_lr = float(lr)
        self.hyper_version = 0
        self.state = torch.tensor([0.0, lr, 1.0, 1.0], device=flat_param.device)   # step, lr, bc1, bc2

    def set_lr(self, lr):
        self.lr = float(lr)
        self.state[1] = self.lr

    def set_hyper(self, betas=None, eps=None, weight_decay=None):
        new = (tuple(betas) if betas is not None else self.betas, self.eps if eps is None else eps,
               self.wd if weight_decay is None else weight_decay)
        if new != (self.betas, self.eps, self.wd):
            self.betas, self.eps, self.wd = new
            self.hyper_version += 1

    def step(self, flag=None):
        """flag: device u32 raised by fami_unscale_check_f32 on non-finite gradients -> the step is skipped."""
        s = _stream(self.p.device)
        if flag is None:
            lib().call('fami_adam_prep_f32', _p(self.state), self.betas[0], self.betas[1], s)
        else:
            lib().call('fami_adam_prep_checked_f32', _p(self.state), self.betas[0], self.betas[1], _p(flag), s)
        lib().call('fami_adam_f32', _p(self.p), _p(self.grad), _p(self.m), _p(self.v), self.p.numel(), _p(self.state),
                   self.betas[0], self.betas[1], self.eps, self.wd, s)

    def prep(self, s=None):
        """the step's scalars alone (step count, bias corrections): a step split into ranges prepares once, up front"""
        lib().call('fami_adam_prep_f32', _p(self.state), self.betas[0], self.betas[1], _stream(self.p.device) if s is None else s)

    def apply(self, lo, hi, s=None):
        """the update of arena[lo:hi] with the prepared scalars"""
        if hi > lo:
            lib().call('fami_adam_f32', _p(self.p[lo:hi]), _p(self.grad[lo:hi]), _p(self.m[lo:hi]), _p(self.v[lo:hi]), hi - lo,
                       _p(self.state), self.betas[0], self.betas[1], self.eps, self.wd, _stream(self.p.device) if s is None else s)


class MultiStepLR:
    """torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones, gamma, last_epoch) for the flat Adam: what
    posetimation/optimizer/scheduler.py:14-35 builds from TRAIN.LR_STEP / TRAIN.LR_FACTOR ([8, 12, 16] x 0.1 in
    configs/Alignment/Base_PoseTrack17.yaml) and engine/defaults/trainer.py steps once per epoch.  `opt` is a FlatAdam
    or a Trainer; the new rate is written into the optimizer's device-resident state, so captured graphs keep replaying."""

    def __init__(self, opt, milestones, gamma=0.1, last_epoch=-1, base_lr=None):
        import bisect
        self._bisect = bisect.bisect_right
        self.opt = opt.opt if hasattr(opt, 'opt') else opt
        self.milestones = sorted(int(m) for m in milestones)
        self.gamma = float(gamma)
        # torch's rule (lr_scheduler.LRScheduler.__init__): a fresh schedule (last_epoch == -1) adopts the optimizer's
        # current rate as `initial_lr`; a resumed one REQUIRES the stored `initial_lr` -- never the current, already
        # decayed `lr` (resuming at epoch 10 gave 1e-5 instead of 1e-4)
        if base_lr is not None:
            self.base_lr = float(base_lr)
        elif int(last_epoch) == -1:
            self.base_lr = float(self.opt.lr)
        else:
            self.base_lr = float(getattr(self.opt, 'initial_lr', self.opt.lr))
        self.opt.initial_lr = self.base_lr
        self.last_epoch = int(last_epoch)
        self.step()          # like torch: construction performs the step to epoch last_epoch + 1

    def get_last_lr(self):
        return [self.base_lr * self.gamma ** self._bisect(self.milestones, self.last_epoch)]

    def step(self):
        self.last_epoch += 1
        self.opt.set_lr(self.get_last_lr()[0])

    def state_dict(self):
        return {'milestones': list(self.milestones), 'gamma': self.gamma, 'base_lr': self.base_lr,
                'last_epoch': self.last_epoch}

    def load_state_dict(self, sd):
        self.milestones, self.gamma = sorted(sd['milestones']), float(sd['gamma'])
        self.base_lr, self.last_epoch = float(sd['base_lr']), int(sd['last_epoch'])
        self.opt.set_lr(self.get_last_lr()[0])


def flatten_parameters(model):
    """Move every trainable parameter into one flat arena (views keep names/shapes). -> (flat, [(param, off, n)])"""
    ps = [p for p in model.parameters() if p.requires_grad]
    # groups the model wants adjacent (engine.CatParam: the two predictor convolutions of a DCN layer run as one): the later
    # members move directly behind the first one
    groups = [g for g in (model.adjacent_parameters() if hasattr(model, 'adjacent_parameters') else [])
              if all(q.requires_grad for q in g)]
    follow = {id(g[0]): g[1:] for g in groups}
    moved = {id(q) for g in groups for q in g[1:]}
    ordered = []
    for p in ps:
        if id(p) in moved:
            continue
        ordered.append(p)
        ordered.extend(follow.get(id(p), ()))
    total = sum(p.numel() for p in ps)
    flat = torch.empty(total, dtype=torch.float32, device=ps[0].device)
    where, off = {}, 0
    for p in ordered:                      # the arena's layout
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view(p.shape)
        if hasattr(p, '_fami_packed'):          # a packed image cached while the parameter was frozen is stale now
            del p._fami_packed
        where[id(p)] = off
        off += n
    # the table stays in model.parameters() order: torch.optim.Adam numbers its state that way (checkpoint.py)
    return flat, [(p, where[id(p)], p.numel()) for p in ps]


class WeightPacker:
    """MFMA-fragment images of every trainable nn.Conv2d weight (forward and dgrad orientation), rebuilt from the
    flat fp32 parameter arena in ONE launch per step (fami_pack_conv_weights_batch_*) instead of ~600."""

    def __init__(self, model, flat, table, dtype, cats=None, route=None):
        """route: the Trainer's kernel-routing state (None: the process default) -- the size queries and the pack launches run
        under the route the model's engines dispatch by (today no route field changes an image's geometry; the DCN backward
        image already did once)."""
        import numpy as np
        self.route = route
        lib().bind(route)
        L = lib().cdll
        from .engine import _SFX
        bf = dtype != torch.float32      # 16-bit images (bf16 | fp16) share one geometry
        elems = L.fami_packed_weight_elems_bf16 if bf else L.fami_packed_weight_elems
        convs = {id(m.weight) for m in model.modules() if isinstance(m, torch.nn.Conv2d)}
        recs, self.views, off = [], {}, 0
        spans = []
        # merged convolutions (engine.CatParam over adjacent weights): ONE image of the concatenated weight, keyed by the
        # CatParam; the parts get no image of their own
        cats = {id(c.parts[0]): c for c in (cats or ()) if len(c.shape) == 4}
        skip = {id(q) for c in cats.values() for q in c.parts[1:]}
        for prm, src, _ in table:
            if id(prm) not in convs or id(prm) in skip:
                continue
            if id(prm) in cats:
                prm = cats[id(prm)]
            Co, Ci, kh, kw = prm.shape
            for mode in (0, 1):
                n = elems(Co, Ci, kh, kw, mode)
                recs.append((src, off, Co, Ci, kh * kw, mode))
                spans.append((id(prm), mode, off, n))
                off += n
        # forward images first, input-gradient images behind them: the two halves can be packed by separate launches
        # (run(stream, 0) before the forward pass, run(side_stream, 1) beside it -- nothing reads a mode-1 image before
        # the backward pass)
        recs.sort(key=lambda r: r[5])
        self.n = len(recs)
        self.n_fwd = sum(1 for r in recs if r[5] == 0)
        # ... and the forward images in two: what the stem, layer1 and stage 2 read (packed on the main lane before the first
        # convolution) and the rest (stage 3 / 4 and the head: ~95 % of the bytes), packed on a side lane beside the stem --
        # the whole forward pack sat alone at the top of every step (0.36-0.42 ms + the f32 split images)
        names = {id(p): n for n, p in model.named_parameters()}
        early = ('hrnet.conv1.', 'hrnet.conv2.', 'hrnet.layer1.', 'hrnet.transition1.', 'hrnet.stage2.', 'conv1.', 'conv2.',
                 'layer1.', 'transition1.', 'stage2.')
        order = [prm for prm, _, _ in table if id(prm) in convs and id(prm) not in skip]
        self.n_early = 0
        for prm in order:
            if names.get(id(prm), '').startswith(early):
                self.n_early += 1
            else:
                break
        self.flat = flat
        self.arena = torch.empty(max(off, 1), dtype=dtype, device=flat.device)
        desc = np.array(recs, dtype=[('src', '<i8'), ('dst', '<i8'), ('Co', '<i4'), ('Ci', '<i4'), ('taps', '<i4'),
                                     ('mode', '<i4')])
        self.desc = torch.from_numpy(desc.view(np.uint8).copy()).to(flat.device)
        for pid, mode, o, n in spans:
            self.views[(pid, mode)] = self.arena[o:o + n]
        self.fn = 'fami_pack_conv_weights_batch' + _SFX[dtype]

    def run(self, stream, part=None):
        """part None: every image; 0: the forward images; 1: the input-gradient images; 'early' / 'late': the forward images of
        the stem .. stage 2 / of everything behind them."""
        lo, hi = {None: (0, self.n), 0: (0, self.n_fwd), 1: (self.n_fwd, self.n), 'early': (0, self.n_early),
                  'late': (self.n_early, self.n_fwd)}[part]
        if hi > lo:
            lib().call_routed(self.route, self.fn, _p(self.flat), _p(self.arena), self.desc.data_ptr() + 32 * lo, hi - lo, stream)


class BucketReducer:
    """Data-parallel gradient exchange over one flat gradient arena (pure host logic + torch.distributed;
    no HIP dependency, so the N>1 path is covered by gloo tests on CPU).

    The arena is cut into fixed-size slices from the TAIL (parameters register head-last, so backward
    completes the tail first).  `hook(done_params)` is what Engine.backward calls when a parameter's last
    gradient contribution has been enqueued; once every parameter overlapping a slice is complete the slice
    is handed to `on_bucket(lo, hi)` (an async all-reduce, or a graph cut during capture)."""

    def __init__(self, grad, table, bucket_elems, process_group=None, payload=None, cast=None, widen=None, algo=None):
        """payload: None (the fp32 slices themselves are reduced in place) or a 16-bit torch dtype: a slice is cast into a
        16-bit mirror of the arena, the mirror slice is all-reduced (half the bytes over xGMI: 129 MB instead of 258.6 MB
        per step for W48) and widened back into the fp32 arena in wait().  cast(src_f32, dst_16) / widen(src_16, dst_f32):
        the device kernels of the caller (Trainer: fami_cast_add_* / fami_widen_*); default = torch copies (CPU tests).
        algo (default FAMI_DDP_ALGO or 'ring'): 'ring' = one all_reduce per slice (the library picks its algorithm; a ring
        over xGMI is bound by ONE 153 GB/s link: ~3 ms for 258 MB); 'mesh' = the slice as reduce_scatter_tensor ->
        all_gather_into_tensor, i.e. every rank owns 1/world of the slice, receives the other ranks' copies of it over
        its 7 point-to-point links at once, sums, and sends the sum back the same way (~0.4 ms for 258 MB on
        the full xGMI mesh).  The remainder of a slice that is not a multiple of `world` (< world elements, at most one
        slice per step) goes through a plain all_reduce.  Replaces engine/defaults/trainer.py:57-58 (nn.DataParallel)."""
        self.payload = payload
        self._cast = cast or (lambda src, dst: dst.copy_(src))
        self._widen = widen or (lambda src, dst: dst.copy_(src))
        self._mirror = None
        self.grad = grad
        self.offset = {id(p): (o, n) for p, o, n in table}
        self.total = grad.numel()
        self.bucket_elems = max(1, int(bucket_elems))
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.algo = (algo or options.get('FAMI_DDP_ALGO')).lower()
        if self.algo not in ('ring', 'mesh'):
            raise ValueError("FAMI_DDP_ALGO must be 'ring' or 'mesh', not %r" % self.algo)
        self._shards = {}        # (lo, hi) -> this rank's 1/world of the slice (mesh plan; persistent: graph-plan replays reuse it)
        # gloo runs queued collectives on a worker pool: the all-gather of a slice must not start before its
        # reduce-scatter has finished, so the handle is waited for in between.  RCCL orders both on its own stream.
        self._ordered = dist.is_initialized() and dist.get_backend(process_group) == 'nccl'
        self.works = []

    def ranges(self):
        out, hi = [], self.total
        while hi > 0:
            lo = max(0, hi - self.bucket_elems)
            out.append((lo, hi))
            hi = lo
        return out

    def begin(self, on_bucket=None):
        """-> hook for Engine.backward(on_params_done=...)."""
        self._on_bucket = on_bucket or self.allreduce
        self._ranges = self.ranges()
        self._next, self._done_lo, self._pending = 0, self.total, {}
        self.works = []
        return self._hook

    def _hook(self, done_params):
        for p in done_params:
            o, n = self.offset[id(p)]
            self._pending[o] = n
        while self._done_lo in _ends(self._pending):        # advance the contiguous frontier from the tail
            o = _ends(self._pending)[self._done_lo]
            self._done_lo = o
            del self._pending[o]
        while self._next < len(self._ranges) and self._ranges[self._next][0] >= self._done_lo:
            self._on_bucket(*self._ranges[self._next])
            self._next += 1

    def flush(self):
        """Slices whose parameters never receive a gradient (hrnet.final_layer feeds only detached MI terms)."""
        while self._next < len(self._ranges):
            self._on_bucket(*self._ranges[self._next])
            self._next += 1

    def _exchange(self, buf, key):
        """Sum `buf` (a slice of the arena or of its 16-bit mirror) over the ranks, asynchronously -> last work handle."""
        if self.algo == 'ring' or self.world == 1:
            return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        n = buf.numel()
        main = n - n % self.world
        wk = None
        if main:
            shard = self._shards.get(key)
            if shard is None or shard.numel() != main // self.world or shard.dtype != buf.dtype:
                shard = self._shards[key] = torch.empty(main // self.world, dtype=buf.dtype, device=buf.device)
            wk = dist.reduce_scatter_tensor(shard, buf[:main], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            if self._ordered:
                wk = dist.all_gather_into_tensor(buf[:main], shard, group=self.pg, async_op=True)
            else:
                # (gloo: the gather must follow the scatter's completion -- chained in wait(), not here: the backward hook that
                #  issues the exchange must not block the host enqueue of the rest of the backward pass)
                wk = _Chained(wk, lambda b=buf[:main], sh=shard: dist.all_gather_into_tensor(b, sh, group=self.pg))
        if main < n:
            if wk is not None:
                self.works.append((wk, None))
            wk = dist.all_reduce(buf[main:], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        return wk

    def allreduce(self, lo, hi):
        if self.payload is None:
            wk = self._exchange(self.grad[lo:hi], (lo, hi))
            self.works.append((wk, None))
            return wk
        if self._mirror is None:
            self._mirror = torch.empty(self.total, dtype=self.payload, device=self.grad.device)
        buf = self._mirror[lo:hi]
        self._cast(self.grad[lo:hi], buf)
        wk = self._exchange(buf, (lo, hi))
        self.works.append((wk, (lo, hi)))
        return wk

    def wait(self):
        for wk, rng in self.works:
            wk.wait()
            if rng is not None:
                self._widen(self._mirror[rng[0]:rng[1]], self.grad[rng[0]:rng[1]])
        self.works = []


class _Chained:
    """A work handle followed by a blocking step that may only start once it has completed."""

    def __init__(self, work, then):
        self.work, self.then = work, then

    def wait(self):
        self.work.wait()
        self.then()


def _ends(pending):
    return {o + n: o for o, n in pending.items()}


class Trainer:
    """One optimisation step per call: forward, loss, backward, (all-reduce), Adam, PCK of both heatmap outputs -- all
    HIP kernels, no host synchronisation (core fn :104-174).

    loss_scale: gradients of the 16-bit activation modes are seeded with `loss_scale * dLoss` and the flat fp32
    gradient arena is divided by it before Adam (static scaling; fp16's smallest normal is 6e-5 while the MSE seed is
    ~2e-6 * error).  Default: 8192 for fp16, 1 otherwise (bf16 has fp32's exponent range).
    """

    def __init__(self, model, lr=1e-3, mse_weight=1.0, alpha=0.5, beta=0.1, use_mi=True, bucket_mb=32,
                 process_group=None, use_graph=True, targets_from_joints=False, sigma=3, force_ddp=False,
                 loss_scale=None, pck=True, data_parallel=None):
        self.model = model
        self.route = getattr(model, 'route', None)      # kernel-routing state (None: process default), as the model's engines
        self.targets_from_joints, self.sigma = targets_from_joints, sigma
        self.dev = next(model.parameters()).device
        if self.dev.type != 'cuda':
            raise RuntimeError('Trainer needs the model on the GPU (HIP path only)')
        self.mse_weight, self.alpha, self.beta, self.use_mi = mse_weight, alpha, beta, use_mi
        self.act_dtype = getattr(model, 'act_dtype', torch.float32)
        self.loss_scale = float(loss_scale if loss_scale is not None else
                                (8192.0 if self.act_dtype == torch.float16 else 1.0))
        self.flat, self.table = flatten_parameters(model)
        # overflow guard of the static loss scale: a device flag raised by the unscale pass, consumed by the optimizer
        self.overflow = torch.zeros(2, dtype=torch.int32, device=self.dev) if self.loss_scale != 1.0 else None   # {raised, skipped steps}
        self.opt = FlatAdam(self.flat, lr=lr)
        self.grad = self.opt.grad
        self._adam_split = self._find_adam_split(model)
        self._adam_done = False
        self.views = {id(p): self.grad[o:o + n].view(p.shape) for p, o, n in self.table}
        # merged parameters (engine.CatParam): adjacent in the arena by construction -- one gradient view spanning the parts
        self.cats = []
        if hasattr(model, 'merged_predictors'):
            offs = {id(p): o for p, o, n in self.table}
            for pair in model.merged_predictors().values():
                for c in pair:
                    if c.adjacent() and c.requires_grad and id(c.parts[0]) in offs:
                        o = offs[id(c.parts[0])]
                        self.views[id(c)] = self.grad[o:o + c.numel()].view(c.shape)
                        self.cats.append(c)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (process_group is not None or dist.is_initialized()) else 1
        if data_parallel is False:          # a single-process trainer inside an initialised process group (tests, tools)
            self.world = 1
        if self.world > 1 and process_group is None:
            self.pg = dist.group.WORLD
        # force_ddp: run the bucketed all-reduce path even with one rank (exercises the hooks / RCCL stream ordering
        # on a single GPU; an all-reduce over one rank is the identity)
        self.ddp = self.world > 1 or (force_ddp and dist.is_initialized())
        # FAMI_DDP_PAYLOAD = f32 (default) | bf16 | f16: gradient bytes on the wire (the sum over ranks is then taken in
        # that type; master gradients, the 1/world scale and Adam stay fp32).  The reference all-reduces fp32 gradients
        # (nn.DataParallel, trainer.py:57-58) and so does the default here in every compute mode: a 16-bit wire format
        # halves the bytes (129 MB instead of 258.6 MB per step for W48) but sums over ranks with 8 / 11 significand bits,
        # and no multi-GPU convergence run has validated that yet -- opt in with FAMI_DDP_PAYLOAD.
        self.payload_name = options.get('FAMI_DDP_PAYLOAD')
        pay = {'f32': None, 'bf16': torch.bfloat16, 'f16': torch.float16}[self.payload_name]
        if pay == torch.float16 and self.overflow is None:
            # without the loss-scale guard an fp16 sum that overflows on the wire would reach Adam as inf
            raise ValueError('FAMI_DDP_PAYLOAD=f16 needs the overflow guard of a loss-scaled (fp16) model; use bf16')
        sfx = {torch.bfloat16: 'bf16', torch.float16: 'f16'}.get(pay)
        self.reducer = BucketReducer(
            self.grad, self.table, bucket_mb * (1 << 20) // 4, self.pg, payload=pay,
            cast=lambda src, dst: self._lc('fami_cast_add_' + sfx, _p(src), _p(dst), src.numel(), 0, _stream(self.dev)),
            widen=lambda src, dst: self._lc('fami_widen_' + sfx, _p(src), _p(dst), src.numel(), _stream(self.dev)))
        self.reducer.world = self.world
        self.packer = WeightPacker(model, self.flat, self.table, self.act_dtype, cats=self.cats, route=self.route)
        # data-parallel launch plan: 'overlap' (default) = hipGraph segments cut at the bucket boundaries with each
        # bucket's all-reduce issued between two segment replays (graph replay AND overlap with the rest of backward);
        # 'serial' = one graph for forward + backward, then every all-reduce; FAMI_DDP_GRAPH=0 = no graphs at all
        # (eager launches, all-reduces fired from the Engine.backward hooks)
        self.ddp_plan = options.get('FAMI_DDP_PLAN')
        self.use_graph = use_graph and (not self.ddp or options.flag('FAMI_DDP_GRAPH'))
        self.loss_parts = torch.zeros(7, device=self.dev)     # mse, mi_1..6 (device scalars of the last step)
        self.pck = pck
        self.acc = None           # [2, J+3] device floats of the last step: PCK of final_hm / kf_bb_hm (see accuracy())
        self._cache = {}          # batch shapes -> (plan, static inputs, static outputs, optimizer hyper version)
        if hasattr(model, '_ensure_nbt'):
            model._ensure_nbt(self.dev)      # BatchNorm counters move into their arena BEFORE any snapshot is taken
        if self.ddp:
            self.broadcast_parameters()

    def _lc(self, name, *args):
        """a library entry point under this trainer's route (weight packs and plans must see what its engines see)"""
        return lib().call_routed(self.route, name, *args)

    # ------------------------------------------------------------------ data parallel
    def broadcast_parameters(self):
        dist.broadcast(self.flat, src=0, group=self.pg)
        for b in self.model.buffers():
            if b.dtype.is_floating_point:
                dist.broadcast(b, src=0, group=self.pg)

    def measure_allreduce_ms(self, reps=3):
        """Wall time of one bucketed all-reduce of the whole gradient arena with nothing else running (reporting only;
        the arena content is scaled back afterwards)."""
        if not self.ddp:
            return 0.0
        import time
        torch.cuda.synchronize(self.dev)
        best = None
        for _ in range(reps):
            dist.barrier(group=self.pg)
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            for lo, hi in self.reducer.ranges():
                self.reducer.allreduce(lo, hi)
            self.reducer.wait()
            torch.cuda.synchronize(self.dev)
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
            self.grad.mul_(1.0 / self.world)
        return best

    def _dcn_bwd_images(self):
        """[(weight, persistent image buffer, (Co, C, kh, kw, G))] of the model's DCN layers (built once)."""
        if getattr(self, '_dcn_imgs', None) is None:
            from .zoo.alignment_v15 import DeformConv2d
            G = getattr(self.model, 'G', None)
            imgs = []
            lib().bind(self.route)       # (the image size follows the route's plan of the register-fed backward kernel)
            if G is not None:
                for m in self.model.modules():
                    if isinstance(m, DeformConv2d):
                        Co, C, kh, kw = m.weight.shape
                        n = lib().cdll.fami_dcn_packed_weight_bwd_elems(Co, C, kh, kw, G)
                        nf = lib().cdll.fami_dcn_packed_weight_elems(Co, C, kh, kw, G)
                        imgs.append((m.weight, torch.empty(n, device=self.dev), (Co, C, kh, kw, G),
                                     torch.empty(nf, device=self.dev)))
            self._dcn_imgs = imgs
        return self._dcn_imgs

    # ------------------------------------------------------------------ the step (eager launch sequence)
    def _forward_backward(self, kf_x, sup_x, target, weight, on_bucket=None, fused_opt=False):
        """Enqueue forward, loss and backward of one batch.  fused_opt (set only by _eager_step / _capture_plan): the Adam update
        may be enqueued from INSIDE the backward pass in two parts (_find_adam_split) -- the call then mutates the parameters and
        sets `_adam_done` for the `_opt_step` that must follow.  Direct callers (tools, tests) get a pure gradient computation."""
        model = self.model
        eng = Engine(self.dev, grad_views=self.views, dtype=self.act_dtype,
                     deterministic=getattr(model, 'deterministic', None), route=self.route)
        self.last_nfused = eng.nfused       # (tools read the counters; not the engine itself: its callbacks point back here, and a
                                            #  Trainer kept alive by that cycle is collected at a random time -- e.g. inside the next capture)
        if on_bucket is not None:
            eng.persist_lanes = False       # bucket hooks fire between forked regions: keep a join per module
        early = {'ev': None}
        split = self._adam_split if (fused_opt and on_bucket is None and not self.ddp) else None
        if split is not None:
            # Adam in two parts (see _find_adam_split): the scalars now, everything but the stem stretch when backward reaches the
            # stage-2 boundary (side lane), the stem stretch behind the backward pass
            self.opt.prep(eng.stream)

            def on_mark(name):
                if name == 'stage2' and early['ev'] is None:
                    eng.flush_reduces()
                    eng.sync_wgrad_lane()
                    n = self.flat.numel()

                    def upd(st):
                        self.opt.apply(0, split[0], st)
                        self.opt.apply(split[1], n, st)
                    early['ev'] = eng.side_launch(upd)
            eng.on_mark = on_mark
        # conv weight images of this step: the forward orientation in one launch now, the input-gradient orientation in a
        # second launch on a side lane beside the forward pass (first read by the backward pass, which waits for it)
        # ... and the DCN layers' backward weight images + their fixed-point scale bound (a one-workgroup reduction): they
        # depend on the weights only, and packed inside the backward closures they sat on the head's serial chain
        dcn = self._dcn_bwd_images()

        from .engine import _SFX
        pack_fwd_fn = 'fami_dcn_pack_weight' + _SFX[self.act_dtype]

        def pack_dcn_fwd(st):
            # the DCN layers' FORWARD weight images (three small launches per layer) sat on the head's serial chain too:
            # packed on a side lane at the start of the step, joined before the head
            for w, _, geo, fbuf in dcn:
                self._lc(pack_fwd_fn, _p(w.data), _p(fbuf), *geo, st)

        def pack_bwd(st):
            self.packer.run(st, 1)
            for w, buf, geo, _ in dcn:
                self._lc('fami_dcn_pack_weight_bwd_f32', _p(w.data), _p(buf), *geo, st)
        if options.flag('FAMI_ABL_PACK') and getattr(self, '_abl_packed', False):
            packed_fwd = packed_bwd = None          # (upper-bound experiment, WRONG results: the images of the first step stay)
        elif eng.use_lanes and options.flag('FAMI_PACK_SPLIT'):
            # (round 5: starting the late / input-gradient packs at stage 2 / 3 on the weight-gradient stream instead of beside the stem
            #  stretch measured no gain -- bf16 20.26 vs 20.23 ms, f32 45.92 vs 45.72; all packing removed is worth 0.37 / 0.72 ms, FAMI_ABL_PACK)
            if self.packer.n_early > 0 and options.flag('FAMI_PACK_EARLY'):
                self.packer.run(eng.stream, 'early')
                eng.late_weights_ready = eng.side_launch(lambda st: self.packer.run(st, 'late'))   # HRNetBody.run waits before stage 3
            else:
                self.packer.run(eng.stream, 0)
            packed_fwd = eng.side_launch(pack_dcn_fwd) if dcn else None
            packed_bwd = eng.side_launch(pack_bwd)
        else:
            self.packer.run(eng.stream)
            pack_dcn_fwd(eng.stream)
            for w, buf, geo, _ in dcn:
                self._lc('fami_dcn_pack_weight_bwd_f32', _p(w.data), _p(buf), *geo, eng.stream)
            packed_fwd = packed_bwd = None
        self._abl_packed = True
        eng.prepacked = self.packer.views
        eng.prepacked_dcn_bwd = {id(w): buf for w, buf, _, _ in dcn}
        eng.prepacked_dcn_fwd = {id(w): fbuf for w, _, _, fbuf in dcn}
        eng.dcn_fwd_ready = packed_fwd          # event the first DCN forward waits for
        if self.targets_from_joints:
            # on-device Gaussian targets (generate_heatmaps): `target` carries joints [B,J,2], `weight` visibility [B,J]
            joints, vis = target, weight
            B, J = joints.shape[:2]
            Hh, Wh = kf_x.shape[2] // 4, kf_x.shape[3] // 4
            target = eng.empty(B, J, Hh, Wh)
            weight = eng.empty(B, J)
            eng.call('fami_gauss_target_f32', _p(joints), _p(vis), _p(target), _p(weight), B, J, Hh, Wh,
                     kf_x.shape[2], kf_x.shape[3], self.sigma)
        outs, _ = model._body(eng, kf_x, sup_x)
        model._advance_bn_counters(eng)
        aux = eng.aux
        final_nchw = outs[0]
        B, J, Hh, Wh = final_nchw.shape
        L = Hh * Wh
        scale = self.mse_weight / (B * L * J)
        w = weight.reshape(B * J)
        ws = eng.ws(B * J * 4)
        eng.call('fami_wmse_fwd_f32', _p(final_nchw), _p(target), _p(w), _p(self.loss_parts[0:1]), B * J, L,
                 float(scale), _p(ws))
        if self.pck:
            # accuracy(pred_heatmaps, target) and accuracy(kf_bb_heatmaps, target), core fn :159-163, without the four
            # device-to-host heatmap copies
            if self.acc is None or self.acc.shape[1] != J + 3:
                self.acc = torch.zeros(2, J + 3, device=self.dev)
            iws = eng.empty(2 * B * J, dtype=torch.int64)
            mws = eng.empty(2 * B * J)
            for k, hm in enumerate((final_nchw, outs[1])):
                eng.call('fami_pck_accuracy_f32', _p(hm), _p(target), _p(self.acc[k]), _p(iws), _p(mws), B, J, Hh, Wh,
                         0.5)
        ls = self.loss_scale
        dpred = torch.empty_like(final_nchw)
        eng.call('fami_wmse_bwd_f32', _p(final_nchw), _p(target), _p(w), _p(dpred), B * J, L, float(scale * ls), None, 0)
        eng.seed_nchw(aux['final'], dpred)
        if self.use_mi and aux['mis']:
            a, b = self.alpha, self.beta
            coef = [-b * a, b * a, a, -a, a, -a]        # core fn :119-148
            lanes = eng.use_lanes and eng.mi_lanes
            if lanes:
                eng._do_fork(3)        # (not taped: the seeds are enqueued here, ahead of the tape walk)
            pending = []
            same = aux.get('mi_same', {})      # repeated terms (mi_6 is mi_2): one gradient pass with the summed coefficient
            for k, j in same.items():
                coef[j] += coef[k]
            for k, ((val, seed), c) in enumerate(zip(aux['mis'], coef)):
                if lanes:
                    eng.set_lane(k % 3)
                eng.call('fami_axpby_f32', _p(val), None, _p(self.loss_parts[1 + k:2 + k]), 1, 1.0, 0.0)
                if k not in same:
                    pending.append(seed(c * ls, None, True))
            if lanes:
                eng._do_join(3)
            for fin in pending:        # the accumulations into the (shared) gradient buffers, in term order, on lane 0
                fin()
        hook = None
        if on_bucket is not None:
            def bucket_ready(lo, hi):
                eng.sync_wgrad_lane()       # the slice's weight gradients live on the engine's wgrad stream
                return on_bucket(lo, hi)
            hook = self.reducer.begin(bucket_ready)
        if packed_bwd is not None:
            eng.wait_main(packed_bwd)
        eng.backward(on_params_done=hook)
        if split is not None:
            if early['ev'] is not None:
                self.opt.apply(split[0], split[1], eng.stream)
                eng.wait_main(early['ev'])
            else:
                self.opt.apply(0, self.flat.numel(), eng.stream)
            self._adam_done = True
        self.conv_flops = eng.conv_flops        # nn.Conv2d FLOPs of one step (forward + both gradients; reporting)
        if on_bucket is not None:
            self._flushing = True
            try:
                self.reducer.flush()
            finally:
                self._flushing = False
        return outs

    _flushing = False
    conv_flops = 0

    def skipped_steps(self):
        """Optimizer steps skipped so far because the loss-scaled gradients held inf / NaN (fp16 mode; synchronises).
        The scale is static: a count that keeps growing means the scale is too large for this model."""
        return 0 if self.overflow is None else int(self.overflow[1].item())

    def _find_adam_split(self, model):
        """(lo, hi): the arena range of the stem / layer1 / first-transition parameters, whose gradients are the LAST the backward
        pass completes.  Everything outside it is final when backward reaches the boundary of stage 2 (modules.HRNetBody.run marks
        it), so its Adam update can run on a side lane beside the rest of the backward pass instead of behind it (the one-launch
        Adam is 0.33 ms of pure HBM traffic at the end of every step).  None: no such split (FAMI_EARLY_ADAM=0, fp16's checked step,
        data-parallel plans, a frozen backbone)."""
        if not options.flag('FAMI_EARLY_ADAM') or self.loss_scale != 1.0:
            return None
        hr = getattr(model, 'hrnet', None)
        if hr is None or not hasattr(hr, 'stage2') or not hasattr(hr, 'conv1'):
            return None
        off = {id(p): o for p, o, n in self.table}
        first = [p for p in hr.stage2.parameters() if id(p) in off]
        if id(hr.conv1.weight) not in off or not first:
            return None
        lo, hi = off[id(hr.conv1.weight)], min(off[id(p)] for p in first)
        return (lo, hi) if 0 <= lo < hi else None

    def _opt_step(self):
        if self._adam_done:
            self._adam_done = False
            return
        self._unscale()
        self.opt.step(self.overflow)

    def _unscale(self):
        """gradient arena *= 1 / (world * loss_scale): the data-parallel mean and the static loss scale in one pass."""
        f = 1.0 / ((self.world if self.ddp else 1) * self.loss_scale)
        if self.overflow is not None:
            # static loss scaling (fp16): the same pass also looks for inf / NaN; Adam then skips the step
            self._lc('fami_unscale_check_f32', _p(self.grad), self.grad.numel(), f, _p(self.overflow), _stream(self.dev))
        elif f != 1.0:
            self._lc('fami_axpby_f32', _p(self.grad), None, _p(self.grad), self.grad.numel(), f, 0.0, _stream(self.dev))

    def _eager_step(self, kf_x, sup_x, target, weight):
        if self.ddp:
            outs = self._forward_backward(kf_x, sup_x, target, weight, on_bucket=self.reducer.allreduce)
            self.reducer.wait()
        else:
            outs = self._forward_backward(kf_x, sup_x, target, weight, fused_opt=True)
        self._opt_step()
        return outs

    # ------------------------------------------------------------------ hipGraph capture / replay
    def _capture(self, kf_x, sup_x, target, weight):
        """Capture with the cyclic garbage collector off: a collection that starts inside a capture may destroy an older Trainer's
        graph (Engine / Trainer objects are reference cycles through their callbacks, so `del trainer` leaves them to the collector),
        and destroying a graph while a stream captures in the global mode aborts the process (hipErrorStreamCaptureUnsupported out
        of ~CUDAGraph).  torch.cuda.graph() no longer collects on entry in this PyTorch, so it is done here."""
        import gc
        was = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            return self._capture_plan(kf_x, sup_x, target, weight)
        finally:
            if was:
                gc.enable()

    def _capture_plan(self, kf_x, sup_x, target, weight):
        st = {'kf': kf_x.clone(), 'sup': sup_x.clone(), 'target': target.clone(), 'weight': weight.clone()}
        # warm-up on a side stream (allocator + pack caches).  The warm-up steps must not count as training steps: the
        # parameters, Adam state and module buffers (BN running statistics, batch counters) are restored afterwards, so
        # the first step() of a graph-mode Trainer is exactly one optimisation step, like the eager one.
        snap = [t.clone() for t in (self.flat, self.opt.m, self.opt.v, self.opt.state)]
        ovf = None if self.overflow is None else self.overflow.clone()     # {raised, skipped steps}: warm-up steps do not count
        bufs = [(b, b.clone()) for b in self.model.buffers()]
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._eager_step(st['kf'], st['sup'], st['target'], st['weight'])
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        for dst, src in zip((self.flat, self.opt.m, self.opt.v, self.opt.state), snap):
            dst.copy_(src)
        for b, src in bufs:
            b.copy_(src)
        if ovf is not None:
            self.overflow.copy_(ovf)
        torch.cuda.synchronize(self.dev)
        if not self.ddp:
            g = torch.cuda.CUDAGraph()
            # (round 6 probe: capturing on a high-priority stream -- lane 0 is the step's critical path -- changes nothing:
            #  bf16 18.75 vs 18.70 ms, f32 44.78 vs 44.78, profiles/r06/ab_lane0_high_priority_*.txt)
            with torch.cuda.graph(g):
                outs = self._forward_backward(st['kf'], st['sup'], st['target'], st['weight'], fused_opt=True)
                self._opt_step()
            plan = [('graph', g)]
        elif self.ddp_plan == 'serial':
            # graph 1 = forward + backward, then the bucketed all-reduce of the flat gradient arena (RCCL, outside any
            # graph), then graph 2 = 1/world scale + Adam
            pool = torch.cuda.graph_pool_handle()
            g1 = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may query events while this thread captures
            with torch.cuda.graph(g1, pool=pool, capture_error_mode='thread_local'):
                outs = self._forward_backward(st['kf'], st['sup'], st['target'], st['weight'])
            plan = [('graph', g1)]
            plan += [('allreduce', r) for r in self.reducer.ranges()]
            plan.append(('wait', None))
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=pool, capture_error_mode='thread_local'):
                self._unscale()
                self.opt.step(self.overflow)
            plan.append(('graph', g2))
        else:
            outs, plan = self._capture_overlap(st)
        entry = (plan, st, outs, self.opt.hyper_version)
        return entry

    def _capture_overlap(self, st):
        """Data-parallel plan with graph replay AND overlap: the launch sequence of forward + backward is captured into
        one hipGraph per gradient bucket -- a segment ends where the backward tape has enqueued the last contribution
        to a bucket (BucketReducer's frontier; only outside forked stream-lane regions, so no cross-segment event) --
        and at replay the bucket's RCCL all-reduce is issued between two segment launches: it runs on RCCL's stream
        beside the next segment's kernels.  The tail segment is followed by wait + (1/world scale, Adam)."""
        pool = torch.cuda.graph_pool_handle()
        plan = []
        cap = torch.cuda.Stream(self.dev)
        cap.wait_stream(torch.cuda.current_stream(self.dev))
        torch.cuda.synchronize(self.dev)
        late = []
        with torch.cuda.stream(cap):
            cur = [torch.cuda.CUDAGraph()]
            cur[0].capture_begin(pool=pool, capture_error_mode='thread_local')
            mark = [lib().ncalls]

            def cut(lo, hi):
                if self._flushing:              # buckets without a gradient: nothing was enqueued since the last cut
                    late.append((lo, hi))
                    return
                if lib().ncalls == mark[0]:     # two buckets completed by the same tape entry: no empty segment
                    plan.append(('allreduce', (lo, hi)))
                    return
                cur[0].capture_end()
                plan.append(('graph', cur[0]))
                plan.append(('allreduce', (lo, hi)))
                cur[0] = torch.cuda.CUDAGraph()
                cur[0].capture_begin(pool=pool, capture_error_mode='thread_local')
                mark[0] = lib().ncalls

            try:
                outs = self._forward_backward(st['kf'], st['sup'], st['target'], st['weight'], on_bucket=cut)
                # a (possibly tiny) tail always exists: the stem's gradients complete after the last bucket boundary
                self._lc('fami_axpby_f32', _p(self.loss_parts), None, _p(self.loss_parts), 7, 1.0, 0.0, _stream(self.dev))
            finally:
                cur[0].capture_end()
            plan.append(('graph', cur[0]))
            plan += [('allreduce', r) for r in late]
            plan.append(('wait', None))
            g2 = torch.cuda.CUDAGraph()
            g2.capture_begin(pool=pool, capture_error_mode='thread_local')
            try:
                self._unscale()
                self.opt.step(self.overflow)
            finally:
                g2.capture_end()
            plan.append(('graph', g2))
        torch.cuda.current_stream(self.dev).wait_stream(cap)
        torch.cuda.synchronize(self.dev)
        return outs, plan

    def step(self, kf_x, sup_x, target, weight):
        """-> (final_hm, kf_bb_hm, ...) of this step; self.loss_parts holds [mse*W, mi_1..mi_6] on device, self.acc
        the PCK rows.  Graph mode keeps one captured plan per batch shape: the DataLoader of the reference has no
        drop_last, so the last batch of an epoch is usually smaller (datasets/zoo/build.py:32-50)."""
        weight = weight.reshape(weight.shape[0], -1).float().contiguous()
        target = target.float().contiguous()
        if not (kf_x.shape[0] == sup_x.shape[0] == target.shape[0] == weight.shape[0]):
            raise ValueError('batch sizes differ: kf %s sup %s target %s weight %s' %
                             (tuple(kf_x.shape), tuple(sup_x.shape), tuple(target.shape), tuple(weight.shape)))
        if not self.use_graph:
            return self._eager_step(kf_x, sup_x, target, weight)
        key = (tuple(kf_x.shape), tuple(sup_x.shape), tuple(target.shape), tuple(weight.shape))
        entry = self._cache.get(key)
        if entry is None or entry[3] != self.opt.hyper_version:
            entry = self._cache[key] = self._capture(kf_x, sup_x, target, weight)
        plan, st, outs, _ = entry
        st['kf'].copy_(kf_x, non_blocking=True)
        st['sup'].copy_(sup_x, non_blocking=True)
        st['target'].copy_(target, non_blocking=True)
        st['weight'].copy_(weight, non_blocking=True)
        for kind, obj in plan:
            if kind == 'graph':
                obj.replay()
            elif kind == 'allreduce':
                self.reducer.allreduce(*obj)
            else:
                self.reducer.wait()
        return outs

    def plan_summary(self):
        """{'graphs': n, 'allreduces': n} of the most recently captured plan (reporting)."""
        if not self._cache:
            return {'graphs': 0, 'allreduces': len(self.reducer.ranges()) if self.ddp else 0}
        plan = list(self._cache.values())[-1][0]
        return {'graphs': sum(1 for k, _ in plan if k == 'graph'), 'allreduces': sum(1 for k, _ in plan if k == 'allreduce')}

    def loss_value(self):
        """Total loss of the last step (host float; forces a sync -- logging only)."""
        p = self.loss_parts.tolist()
        a, b = self.alpha, self.beta
        mi = a * (-b * p[1] + b * p[2] + p[3] - p[4] + p[5] - p[6]) if self.use_mi else 0.0
        return p[0] + mi

    def accuracy(self):
        """PCK of the last step as the reference's `accuracy` returns it, for (final_hm, kf_bb_hm):
        -> [(acc[J+1] ndarray, avg_acc, cnt), (...)]  (host values; forces a sync -- logging only)."""
        if self.acc is None:
            return None
        rows = self.acc.cpu().numpy().astype('float64')
        J = rows.shape[1] - 3
        return [(r[:J + 1].copy(), float(r[J + 1]), int(r[J + 2])) for r in rows]
