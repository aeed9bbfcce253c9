"""Child process of tests/test_model_gpu.py::test_ddp_path_single_rank_rccl (see its docstring): the data-parallel step
with a one-rank RCCL process group on the MI355X against the plain single-GPU step."""
import os
import sys

os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np
import pytest
import torch

from test_model_gpu import _pair


def main(dev):
    import torch.distributed as dist
    from fami_pose_amd.train import Trainer
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29517')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        S, H, W, B = 2, 128, 96, 2
        gen = torch.Generator().manual_seed(9)
        kf, sup = torch.randn(B, 3, H, W, generator=gen).to(dev), torch.randn(B, 3 * S, H, W, generator=gen).to(dev)
        joints = (torch.rand(B, 17, 2, generator=gen) * torch.tensor([W, H], dtype=torch.float32)).to(dev)
        vis = (torch.rand(B, 17, generator=gen) < 0.8).float().to(dev)
        def run(force, graph, steps):
            m, _ = _pair(48, S, (H, W), 'train', 5)
            tr = Trainer(m.to(dev), lr=1e-3, use_graph=graph, targets_from_joints=True, force_ddp=force, bucket_mb=8)
            assert tr.ddp == force
            for _ in range(steps):
                tr.step(kf, sup, joints, vis)
            return tr.loss_value(), tr.grad.clone(), tr.flat.clone()

        # ONE step: identical parameters in, so the gradients must agree (DCN input gradients use float atomics whose
        # summation order varies run to run -> ~1e-5 of the gradient scale, not bitwise).  Adam turns a noise-level
        # gradient into a +-lr step, so parameters are compared only where the gradient is well above that noise.
        l0, g0, p0 = run(False, False, 1)
        l1, g1, p1 = run(True, False, 1)
        assert l1 == pytest.approx(l0, rel=1e-5)
        assert ((g1 - g0).abs().max() / g0.abs().max()).item() < 1e-3
        big = g0.abs() > 1e-2 * g0.abs().max()
        assert (p1 - p0)[big].abs().max().item() < 2e-4
        # graph-mode data parallel plan (hipGraph fwd+bwd -> bucketed all-reduce -> hipGraph scale+Adam): the first
        # step() is exactly one optimisation step (the capture warm-up is rolled back), same gradients as eager
        l2, g2, p2 = run(True, True, 1)
        assert l2 == pytest.approx(l0, rel=1e-5)
        assert ((g2 - g0).abs().max() / g0.abs().max()).item() < 1e-3
        assert (p2 - p0)[big].abs().max().item() < 2e-4
        # several steps keep training on both plans (four Adam steps at lr 1e-3 amplify the atomics' run-to-run noise to a
        # few percent of the loss: the one-step comparisons above are the strict ones)
        l4, _, _ = run(False, False, 4)
        ld, gd, _ = run(True, True, 4)
        assert np.isfinite(ld) and torch.isfinite(gd).all()
        assert ld == pytest.approx(l4, rel=0.15) and ld < l0
        os.environ['FAMI_DDP_GRAPH'] = '0'          # eager, hook-overlapped plan
        try:
            le, ge, _ = run(True, True, 4)
        finally:
            del os.environ['FAMI_DDP_GRAPH']
        assert np.isfinite(le) and le == pytest.approx(l4, rel=0.15)
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main(torch.device('cuda:0'))
    print('DDP_SINGLE_RANK_OK', flush=True)
