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
import os

import torch
import torch.distributed as dist

from ._lib import lib
from .engine import Engine, _p


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


class FlatAdam:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0) on one flat fp32 arena."""

    def __init__(self, flat_param, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.p = flat_param
        self.grad = torch.zeros_like(flat_param)
        self.m = torch.zeros_like(flat_param)
        self.v = torch.zeros_like(flat_param)
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        self.state = torch.tensor([0.0, lr, 1.0, 1.0], device=flat_param.device)   # step, lr, bc1, bc2

    def set_lr(self, lr):
        self.state[1] = lr

    def step(self):
        s = _stream(self.p.device)
        lib().call('fami_adam_prep_f32', _p(self.state), self.betas[0], self.betas[1], s)
        lib().call('fami_adam_f32', _p(self.p), _p(self.grad), _p(self.m), _p(self.v), self.p.numel(), _p(self.state),
                   self.betas[0], self.betas[1], self.eps, self.wd, s)


def flatten_parameters(model):
    """Move every trainable parameter into one flat arena (views keep names/shapes). -> (flat, [(param, off, n)])"""
    ps = [p for p in model.parameters() if p.requires_grad]
    total = sum(p.numel() for p in ps)
    flat = torch.empty(total, dtype=torch.float32, device=ps[0].device)
    table, off = [], 0
    for p in ps:
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view(p.shape)
        table.append((p, off, n))
        off += n
    return flat, table


class WeightPacker:
    """MFMA-fragment images of every trainable nn.Conv2d weight (forward and dgrad orientation), rebuilt from the
    flat fp32 parameter arena in ONE launch per step (fami_pack_conv_weights_batch_*) instead of ~600."""

    def __init__(self, model, flat, table, dtype):
        import numpy as np
        L = lib().cdll
        bf = dtype == torch.bfloat16
        elems = L.fami_packed_weight_elems_bf16 if bf else L.fami_packed_weight_elems
        convs = {id(m.weight) for m in model.modules() if isinstance(m, torch.nn.Conv2d)}
        recs, self.views, off = [], {}, 0
        spans = []
        for prm, src, _ in table:
            if id(prm) not in convs:
                continue
            Co, Ci, kh, kw = prm.shape
            for mode in (0, 1):
                n = elems(Co, Ci, kh, kw, mode)
                recs.append((src, off, Co, Ci, kh * kw, mode))
                spans.append((id(prm), mode, off, n))
                off += n
        self.n = len(recs)
        self.flat = flat
        self.arena = torch.empty(max(off, 1), dtype=dtype, device=flat.device)
        desc = np.array(recs, dtype=[('src', '<i8'), ('dst', '<i8'), ('Co', '<i4'), ('Ci', '<i4'), ('taps', '<i4'),
                                     ('mode', '<i4')])
        self.desc = torch.from_numpy(desc.view(np.uint8).copy()).to(flat.device)
        for pid, mode, o, n in spans:
            self.views[(pid, mode)] = self.arena[o:o + n]
        self.fn = 'fami_pack_conv_weights_batch_bf16' if bf else 'fami_pack_conv_weights_batch_f32'

    def run(self, stream):
        if self.n:
            lib().call(self.fn, _p(self.flat), _p(self.arena), _p(self.desc), self.n, stream)


class BucketReducer:
    """Data-parallel gradient exchange over one flat gradient arena (pure host logic + torch.distributed;
    no HIP dependency, so the N>1 path is covered by gloo tests on CPU).

    The arena is cut into fixed-size slices from the TAIL (parameters register head-last, so backward
    completes the tail first).  `hook(done_params)` is what Engine.backward calls when a parameter's last
    gradient contribution has been enqueued; once every parameter overlapping a slice is complete the slice
    is handed to `on_bucket(lo, hi)` (an async all-reduce, or a graph cut during capture)."""

    def __init__(self, grad, table, bucket_elems, process_group=None):
        self.grad = grad
        self.offset = {id(p): (o, n) for p, o, n in table}
        self.total = grad.numel()
        self.bucket_elems = max(1, int(bucket_elems))
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
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

    def allreduce(self, lo, hi):
        wk = dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self.works.append(wk)
        return wk

    def wait(self):
        for wk in self.works:
            wk.wait()
        self.works = []


def _ends(pending):
    return {o + n: o for o, n in pending.items()}


class Trainer:
    """One optimisation step per call: forward, loss, backward, (all-reduce), Adam -- all HIP kernels."""

    def __init__(self, model, lr=1e-3, mse_weight=1.0, alpha=0.5, beta=0.1, use_mi=True, bucket_mb=32,
                 process_group=None, use_graph=True, targets_from_joints=False, sigma=3, force_ddp=False):
        self.model = model
        self.targets_from_joints, self.sigma = targets_from_joints, sigma
        self.dev = next(model.parameters()).device
        if self.dev.type != 'cuda':
            raise RuntimeError('Trainer needs the model on the GPU (HIP path only)')
        self.mse_weight, self.alpha, self.beta, self.use_mi = mse_weight, alpha, beta, use_mi
        self.flat, self.table = flatten_parameters(model)
        self.opt = FlatAdam(self.flat, lr=lr)
        self.grad = self.opt.grad
        self.views = {id(p): self.grad[o:o + n].view(p.shape) for p, o, n in self.table}
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (process_group is not None or dist.is_initialized()) else 1
        if self.world > 1 and process_group is None:
            self.pg = dist.group.WORLD
        # force_ddp: run the bucketed all-reduce path even with one rank (exercises the hooks / RCCL stream ordering
        # on a single GPU; an all-reduce over one rank is the identity)
        self.ddp = self.world > 1 or (force_ddp and dist.is_initialized())
        self.reducer = BucketReducer(self.grad, self.table, bucket_mb * (1 << 20) // 4, self.pg)
        self.packer = WeightPacker(model, self.flat, self.table, getattr(model, 'act_dtype', torch.float32))
        # FAMI_DDP_GRAPH=0 forces the eager, hook-overlapped launch sequence on the data-parallel path
        self.use_graph = use_graph and (not self.ddp or os.environ.get('FAMI_DDP_GRAPH', '1') != '0')
        self.loss_parts = torch.zeros(7, device=self.dev)     # mse, mi_1..6 (device scalars of the last step)
        self._graphs = None
        self._static = None
        if self.ddp:
            self.broadcast_parameters()

    # ------------------------------------------------------------------ data parallel
    def broadcast_parameters(self):
        dist.broadcast(self.flat, src=0, group=self.pg)
        for b in self.model.buffers():
            if b.dtype.is_floating_point:
                dist.broadcast(b, src=0, group=self.pg)

    # ------------------------------------------------------------------ the step (eager launch sequence)
    def _forward_backward(self, kf_x, sup_x, target, weight, on_bucket=None):
        model = self.model
        eng = Engine(self.dev, grad_views=self.views, dtype=getattr(model, 'act_dtype', torch.float32))
        self.packer.run(eng.stream)             # every conv weight image of this step, one launch
        eng.prepacked = self.packer.views
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
        B, J = final_nchw.shape[:2]
        L = final_nchw[0, 0].numel()
        scale = self.mse_weight / (B * L * J)
        w = weight.reshape(B * J)
        ws = eng.ws(B * J * 4)
        eng.call('fami_wmse_fwd_f32', _p(final_nchw), _p(target), _p(w), _p(self.loss_parts[0:1]), B * J, L,
                 float(scale), _p(ws))
        dpred = torch.empty_like(final_nchw)
        eng.call('fami_wmse_bwd_f32', _p(final_nchw), _p(target), _p(w), _p(dpred), B * J, L, float(scale), None, 0)
        eng.seed_nchw(aux['final'], dpred)
        if self.use_mi and aux['mis']:
            a, b = self.alpha, self.beta
            coef = [-b * a, b * a, a, -a, a, -a]        # core fn :119-148
            for k, ((val, seed), c) in enumerate(zip(aux['mis'], coef)):
                eng.call('fami_axpby_f32', _p(val), None, _p(self.loss_parts[1 + k:2 + k]), 1, 1.0, 0.0)
                seed(c)
        hook = None
        if on_bucket is not None:
            def bucket_ready(lo, hi):
                eng.sync_wgrad_lane()       # the slice's weight gradients live on the engine's wgrad stream
                return on_bucket(lo, hi)
            hook = self.reducer.begin(bucket_ready)
        eng.backward(on_params_done=hook)
        if on_bucket is not None:
            self.reducer.flush()
        return outs

    def _eager_step(self, kf_x, sup_x, target, weight):
        if self.ddp:
            outs = self._forward_backward(kf_x, sup_x, target, weight, on_bucket=self.reducer.allreduce)
            self.reducer.wait()
            lib().call('fami_axpby_f32', _p(self.grad), None, _p(self.grad), self.grad.numel(), 1.0 / self.world, 0.0,
                       _stream(self.dev))
        else:
            outs = self._forward_backward(kf_x, sup_x, target, weight)
        self.opt.step()
        return outs

    # ------------------------------------------------------------------ hipGraph capture / replay
    def _capture(self, kf_x, sup_x, target, weight):
        st = {'kf': kf_x.clone(), 'sup': sup_x.clone(), 'target': target.clone(), 'weight': weight.clone()}
        # warm-up on a side stream (allocator + pack caches).  The warm-up steps must not count as training steps: the
        # parameters, Adam state and module buffers (BN running statistics) are restored afterwards, so the first
        # step() of a graph-mode Trainer is exactly one optimisation step, like the eager one.
        snap = [t.clone() for t in (self.flat, self.opt.m, self.opt.v, self.opt.state)]
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
        torch.cuda.synchronize(self.dev)
        if not self.ddp:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = self._forward_backward(st['kf'], st['sup'], st['target'], st['weight'])
                self.opt.step()
            self._graphs = [('graph', g)]
        else:
            # data parallel: graph 1 = forward + backward, then the bucketed all-reduce of the flat gradient arena
            # (RCCL, outside any graph), then graph 2 = 1/world scale + Adam.  The exchange is ~260 MB of fp32 per
            # step, ~2 ms on 8 xGMI-linked GPUs against a ~75 ms step, so it is not overlapped with backward here; the
            # eager path (use_graph=False) overlaps it bucket by bucket through the Engine.backward hooks.
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
                lib().call('fami_axpby_f32', _p(self.grad), None, _p(self.grad), self.grad.numel(), 1.0 / self.world,
                           0.0, _stream(self.dev))
                self.opt.step()
            plan.append(('graph', g2))
            self._graphs = plan
        self._static = st
        self._static_outs = outs
        return outs

    def step(self, kf_x, sup_x, target, weight):
        """-> (final_hm, kf_bb_hm, ...) of this step; self.loss_parts holds [mse*W, mi_1..mi_6] on device."""
        weight = weight.reshape(weight.shape[0], -1).float().contiguous()
        target = target.float().contiguous()
        if not self.use_graph:
            return self._eager_step(kf_x, sup_x, target, weight)
        if self._graphs is None:
            self._capture(kf_x, sup_x, target, weight)
        st = self._static
        st['kf'].copy_(kf_x, non_blocking=True)
        st['sup'].copy_(sup_x, non_blocking=True)
        st['target'].copy_(target, non_blocking=True)
        st['weight'].copy_(weight, non_blocking=True)
        for kind, obj in self._graphs:
            if kind == 'graph':
                obj.replay()
            elif kind == 'allreduce':
                self.reducer.allreduce(*obj)
            else:
                self.reducer.wait()
        return self._static_outs

    def loss_value(self):
        """Total loss of the last step (host float; forces a sync -- logging only)."""
        p = self.loss_parts.tolist()
        a, b = self.alpha, self.beta
        mi = a * (-b * p[1] + b * p[2] + p[3] - p[4] + p[5] - p[6]) if self.use_mi else 0.0
        return p[0] + mi
