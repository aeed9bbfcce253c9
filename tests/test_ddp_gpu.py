"""Data-parallel equivalence on hardware without an 8-GPU box (SURVEY.md 4 item v; the reference's mechanism is
torch.nn.DataParallel at engine/defaults/trainer.py:57-58): two ranks share the one MI355X (gloo process group -- RCCL
refuses two ranks on one device), each takes its own shard of the minibatch, and
  * rank r's gradients BEFORE the exchange equal a single-process run on shard r (bitwise, deterministic mode),
  * the exchanged gradients equal the mean over ranks (bitwise: a two-term sum is order independent),
  * parameters after Adam are identical on both ranks,
  * the graph plans ('overlap': hipGraph segments cut at the bucket boundaries with the all-reduces issued between
    segment launches; 'serial') reproduce the eager, hook-driven plan bit for bit.
BatchNorm statistics stay per replica, as under DataParallel."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

S, H, W, B = 2, 96, 64, 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, backend='gloo'):
    """backend 'gloo': both ranks share cuda:0 (the one-GPU test box).  'nccl': one rank per device over RCCL -- the
    measured configuration (bench.py --gpus N); runs only where torch.cuda.device_count() >= world."""
    try:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.pop('FAMI_DDP_ALGO', None)
        import torch.distributed as dist
        import fami_pose_amd as fp
        from fami_pose_amd.train import Trainer
        from oracle import model as om
        if backend == 'nccl':
            dev = torch.device('cuda', rank)
            torch.cuda.set_device(dev)
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dev = torch.device('cuda:0')
            dist.init_process_group('gloo', rank=rank, world_size=world)

        def gather(t):
            """-> [rank 0's t, rank 1's t, ...] on the host (RCCL moves device tensors only)."""
            src = t.to(dev) if backend == 'nccl' else t.cpu()
            out = [torch.empty_like(src) for _ in range(world)]
            dist.all_gather(out, src.contiguous())
            return [o.cpu() for o in out]

        sd = om.realistic_init_(om.AlignmentOracle(om.make_cfg(48), True, S, (H, W)), 5).state_dict()

        def model():
            m = fp.build_model(fp.default_cfg(48, image_size=(W, H), num_sup=S), 'train')
            m.load_state_dict(sd)
            return m.to(dev).set_deterministic(True)

        gen = torch.Generator().manual_seed(100 + rank)            # this rank's shard of the global batch
        kf, sup = torch.randn(B, 3, H, W, generator=gen).to(dev), torch.randn(B, 3 * S, H, W, generator=gen).to(dev)
        joints = (torch.rand(B, 17, 2, generator=gen) * torch.tensor([W, H], dtype=torch.float32)).to(dev)
        vis = (torch.rand(B, 17, generator=gen) < 0.8).float().to(dev)

        # (1) single-process run on shard r
        tr_a = Trainer(model(), use_graph=False, targets_from_joints=True, data_parallel=False)
        assert not tr_a.ddp
        tr_a.step(kf, sup, joints, vis)
        g_a = tr_a.grad.clone()

        # (2) two ranks, eager plan: all-reduces fired from the backward hooks
        tr_b = Trainer(model(), use_graph=False, targets_from_joints=True, bucket_mb=8)
        assert tr_b.ddp and tr_b.world == world
        pre = torch.zeros_like(tr_b.grad)
        inner = tr_b.reducer.allreduce

        def spy(lo, hi):
            pre[lo:hi].copy_(tr_b.grad[lo:hi])
            return inner(lo, hi)
        tr_b.reducer.allreduce = spy
        tr_b.step(kf, sup, joints, vis)
        assert torch.equal(pre, g_a), 'rank %d: pre-exchange gradients differ from the single-process run on its shard' % rank
        both = gather(g_a)
        mean = (both[0] + both[1]) * 0.5
        assert torch.equal(tr_b.grad.cpu(), mean), 'exchanged gradients are not the mean over ranks'
        ps = gather(tr_b.flat)
        assert torch.equal(ps[0], ps[1]), 'parameters diverged across ranks after Adam'
        assert not torch.equal(tr_b.flat, tr_a.flat)
        # BatchNorm statistics are per replica
        rm = gather(tr_b.model.hrnet.bn1.running_mean)
        assert not torch.equal(rm[0], rm[1])

        # (2b) FAMI_DDP_ALGO=mesh (reduce_scatter_tensor -> all_gather_into_tensor per slice) is held to the all_reduce
        # plan below through the overlap graph plan (a two-rank sum is order independent, so bitwise), and in the eager form by
        # tests/test_ddp_gloo.py on CPU
        # (3) graph plans == eager plan (rank 0 also checks the serial plan; the single-rank RCCL test covers both)
        # (the serial plan -- one graph, every exchange, one graph -- runs its launch sequence over RCCL with one rank in
        # test_ddp_path_single_rank_rccl and with two ranks in the RCCL variant of this test wherever two devices exist: each
        # capture here costs ~25 s of two processes sharing one GPU)
        plans = (('overlap', 'ring'), ('overlap', 'mesh')) if backend == 'gloo' else (('overlap', 'ring'), ('serial', 'ring'), ('overlap', 'mesh'))
        for plan, algo in plans:
            os.environ['FAMI_DDP_PLAN'] = plan
            os.environ['FAMI_DDP_ALGO'] = algo
            tr_c = Trainer(model(), use_graph=True, targets_from_joints=True, bucket_mb=8)
            tr_c.step(kf, sup, joints, vis)
            assert torch.equal(tr_c.flat, tr_b.flat), (plan, algo)
            summ = tr_c.plan_summary()
            if plan == 'overlap':
                assert summ['graphs'] >= 4 and summ['allreduces'] == len(tr_c.reducer.ranges()), summ
            else:
                assert summ['graphs'] == 2, summ
            tr_c.step(kf, sup, joints, vis)                         # replay keeps working
            assert torch.isfinite(tr_c.loss_parts).all()
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, 'ok'))
    except Exception as e:       # noqa: BLE001 -- reported to the parent
        import traceback
        q.put((rank, 'FAILED: %s\n%s' % (e, traceback.format_exc())))


def _run(backend):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, msg = q.get(timeout=900)
        res[r] = msg
    for p in procs:
        p.join(60)
    assert res == {0: 'ok', 1: 'ok'}, res


def test_two_ranks_on_one_gpu_equal_their_shards(dev):
    _run('gloo')


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two MI355X: one rank per device over RCCL')
def test_two_ranks_two_gpus_rccl(dev):
    """The same assertions with the measured transport: one rank per GPU, RCCL over xGMI, the eager / overlap / serial plans
    and the mesh exchange.  Skipped on the one-GPU test box; a multi-GPU node (the driver's scaling box) runs it."""
    _run('nccl')
