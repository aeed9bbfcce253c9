"""Glue between the nn.Module boundary (torch tensors, torch autograd) and the
HIP engine: one autograd node per model call.  Its forward runs the whole
forward launch sequence, its backward walks the engine tape; torch's own
autograd only sees this single node plus whatever the caller does with the
outputs (callers do `loss.backward(); optimizer.step()`).
"""
import torch
import torch.nn as nn

from .engine import Engine, _p


class _EngineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, body, n_in, *tensors):
        inputs, params = tensors[:n_in], tensors[n_in:]
        eng = Engine(inputs[0].device, grad_views=getattr(owner, '_grad_views', None), record=True,
                     dtype=owner.act_dtype, deterministic=owner.deterministic, route=owner.route)
        outs, seeds = body(eng, *inputs)
        owner._advance_bn_counters(eng)
        ctx.eng, ctx.seeds, ctx.params, ctx.n_in = eng, seeds, params, n_in
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        eng = ctx.eng
        eng.sync_stream()
        for fn, g in zip(ctx.seeds, gouts):
            if fn is not None and g is not None:
                fn(g)
        eng.backward()
        grads = [eng.param_grads.get(id(p)) for p in ctx.params]
        ctx.eng = ctx.seeds = None
        return (None, None, None) + (None,) * ctx.n_in + tuple(grads)


class EngineModule(nn.Module):
    """Base class of the drop-in models: runs `body(eng, *inputs)` on the HIP engine."""

    _nbt_flat = None
    _nbt_index = None
    _nbt_inc = None
    act_dtype = torch.float32      # activation storage / conv MFMA dtype of the HIP engine (fp32 | bf16 | fp16)

    def set_compute_dtype(self, dtype):
        """'f32' (parity configuration: exact-f32 MFMA), 'bf16' (bf16 activations + bf16 MFMA, fp32 accumulation,
        fp32 master weights -- BASELINE config 3) or 'f16' (the same with IEEE half storage + fp16 MFMA -- BASELINE
        config 5).  Parameters, BatchNorm statistics, heatmaps and losses stay fp32 in every mode."""
        dtype = {'f32': torch.float32, 'fp32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16,
                 'fp16': torch.float16}.get(dtype, dtype)
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError('compute dtype must be f32, bf16 or f16')
        self.act_dtype = dtype
        return self

    deterministic = None           # None: FAMI_DETERMINISTIC decides; True / False: this model's engines
    route = None                   # kernel-routing state of this model's engines (None: the library's process default)

    def set_route(self, route):
        """Give this model's engines a routing state of their own (`lib().new_route()`, fields documented in
        include/fami_route.h): models with different routes run interleaved in one process, each launch dispatched by its own."""
        self.route = route
        return self


    def set_deterministic(self, on=True):
        """Run-to-run reproducible kernels for this model (the DCN input-gradient scatter takes its fixed-point form)."""
        self.deterministic = on
        return self

    def _trainable(self):
        return [p for p in self.parameters() if p.requires_grad]

    def _launch(self, body, *inputs):
        if not inputs[0].is_cuda:
            raise RuntimeError("fami_pose_amd models run on the MI355X HIP path only; move the model and inputs to "
                               "'cuda' (there is no CPU fallback)")
        inputs = tuple(t.float().contiguous() for t in inputs)
        params = self._trainable() if torch.is_grad_enabled() else []
        if not params:
            eng = Engine(inputs[0].device, record=False, dtype=self.act_dtype, deterministic=self.deterministic, route=self.route)
            outs, _ = body(eng, *inputs)
            self._advance_bn_counters(eng)
            return tuple(outs)
        return _EngineFn.apply(self, body, len(inputs), *inputs, *params)

    # nn.BatchNorm2d.num_batches_tracked: all counters live in one int64 arena and advance in ONE launch
    def _ensure_nbt(self, dev):
        """Move every BatchNorm's num_batches_tracked into one int64 arena on `dev` (idempotent).  A Trainer calls this
        before it snapshots the module buffers, so the capture warm-up's rollback restores the live counters."""
        if self._nbt_flat is None or self._nbt_flat.device != dev:
            bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None]
            flat = torch.stack([m.num_batches_tracked.to(dev) for m in bns]) if bns else None
            self._nbt_index = {}
            for i, m in enumerate(bns):
                m._buffers['num_batches_tracked'] = flat[i]
                self._nbt_index[id(m)] = i
            self._nbt_flat = flat
            self._nbt_inc = {}

    def _advance_bn_counters(self, eng):
        if not eng.bn_trained:
            return
        dev = eng.dev
        self._ensure_nbt(dev)
        counts = {}
        for m in eng.bn_trained:
            i = self._nbt_index[id(m)]
            counts[i] = counts.get(i, 0) + 1
        key = tuple(sorted(counts.items()))
        inc = self._nbt_inc.get(key)
        if inc is None:
            host = torch.zeros(self._nbt_flat.numel(), dtype=torch.int64)
            for i, c in counts.items():
                host[i] = c
            inc = host.to(dev)
            self._nbt_inc[key] = inc
        eng.call('fami_add_i64', _p(self._nbt_flat), _p(inc), self._nbt_flat.numel())
