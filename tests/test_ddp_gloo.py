"""The N>1 path's host logic on CPU: two processes over gloo (world_size 2) drive BucketReducer -- the
bucketed, backward-overlapped gradient all-reduce of fami_pose_amd/train.py -- exactly as Engine.backward
drives it on the GPU (a hook per completed parameter), and the result must equal the mean of the per-rank
gradients.  (shard clips across ranks, one exchange step per iteration.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from fami_pose_amd.train import BucketReducer
        sizes = [7, 120, 33, 64, 5, 250, 18, 90]                  # registration order: backbone ... head
        params = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
        table, off = [], 0
        for p, n in zip(params, sizes):
            table.append((p, off, n))
            off += n
        total = off
        g = torch.Generator().manual_seed(1234 + rank)
        local = torch.randn(total, generator=g)
        grad = local.clone()
        red = BucketReducer(grad, table, bucket_elems=128)
        assert red.world == world
        ranges = red.ranges()
        assert ranges[0][1] == total and ranges[-1][0] == 0 and all(hi - lo <= 128 for lo, hi in ranges)
        assert [r[0] for r in ranges[:-1]] == [r[1] for r in ranges[1:]]          # contiguous, tail first

        fired = []

        def on_bucket(lo, hi):
            fired.append((lo, hi))
            red.allreduce(lo, hi)

        hook = red.begin(on_bucket)
        # backward completes parameters head-first, with two out-of-order completions and param 0 never done
        order = [7, 5, 6, 4, 2, 3, 1]
        seen_lo = total
        for i in order:
            hook([params[i]])
            done = set(order[:order.index(i) + 1])
            frontier = total
            for j in range(len(sizes) - 1, -1, -1):                                # contiguous completed tail
                if j in done:
                    frontier = table[j][1]
                else:
                    break
            assert all(lo >= frontier for lo, hi in fired), 'bucket fired before its parameters were complete'
            seen_lo = min([lo for lo, _ in fired], default=total)
        assert seen_lo > 0                       # param 0 never completed: its slice is still pending
        red.flush()                              # parameters without gradient (e.g. hrnet.final_layer)
        assert fired == ranges
        red.wait()
        both = [torch.randn(total, generator=torch.Generator().manual_seed(1234 + r)) for r in range(world)]
        want = (both[0] + both[1])
        assert torch.allclose(grad, want, atol=1e-6)
        mean = grad / world
        assert torch.allclose(mean, torch.stack(both).mean(0), atol=1e-6)

        # 16-bit payload (FAMI_DDP_PAYLOAD=bf16): the wire carries bf16 slices, the sum is taken in bf16, the arena stays
        # fp32 -- exactly round(round(g0) + round(g1)) per element, i.e. within 2^-8 of the fp32 sum's magnitude
        grad16 = local.clone()
        red16 = BucketReducer(grad16, table, bucket_elems=128, payload=torch.bfloat16)
        hook16 = red16.begin()
        for i in range(len(sizes) - 1, -1, -1):
            hook16([params[i]])
        red16.wait()
        want16 = (both[0].to(torch.bfloat16) + both[1].to(torch.bfloat16)).float()
        assert torch.equal(grad16, want16)
        assert (grad16 - want).abs().max() <= 2.0 ** -7 * want.abs().max()

        # FAMI_DDP_ALGO=mesh: every slice as reduce_scatter_tensor -> all_gather_into_tensor (the direct exchange
        # over the full xGMI mesh) + a plain all_reduce for the < world elements a slice has beyond a multiple of world.
        # Sums of two terms are order independent: bitwise the ring result.  Odd slice lengths (bucket 127, total 587)
        # exercise the remainder; the second begin() replays the plan on the persistent shard buffers (graph plans do).
        for bucket in (128, 127, 1000):
            gm = local.clone()
            redm = BucketReducer(gm, table, bucket_elems=bucket, algo='mesh')
            assert redm.algo == 'mesh'
            for rep in range(2):
                gm.copy_(local)
                hookm = redm.begin()
                for i in range(len(sizes) - 1, -1, -1):
                    hookm([params[i]])
                redm.flush()
                redm.wait()
                assert torch.equal(gm, grad), (bucket, rep)
        gm16 = local.clone()
        redm16 = BucketReducer(gm16, table, bucket_elems=127, payload=torch.bfloat16, algo='mesh')
        hookm16 = redm16.begin()
        for i in range(len(sizes) - 1, -1, -1):
            hookm16([params[i]])
        redm16.wait()
        assert torch.equal(gm16, want16)
        try:
            BucketReducer(local.clone(), table, bucket_elems=128, algo='tree')
            raise AssertionError('unknown algo accepted')
        except ValueError:
            pass

        # parameter / buffer broadcast from rank 0 (what Trainer.broadcast_parameters does with the flat arena)
        flat = torch.full((total,), float(rank + 1))
        dist.broadcast(flat, src=0)
        assert torch.all(flat == 1.0)
        q.put((rank, 'ok'))
    except Exception as e:      # noqa: BLE001 -- report to the parent instead of hanging the peer
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res


def test_bucket_ranges_single_process():
    from fami_pose_amd.train import BucketReducer
    ps = [torch.nn.Parameter(torch.zeros(10)), torch.nn.Parameter(torch.zeros(3))]
    red = BucketReducer(torch.zeros(13), [(ps[0], 0, 10), (ps[1], 10, 3)], bucket_elems=4)
    assert red.world == 1
    assert red.ranges() == [(9, 13), (5, 9), (1, 5), (0, 1)]
    fired = []
    hook = red.begin(lambda lo, hi: fired.append((lo, hi)))
    hook([ps[1]])
    assert fired == []                       # slice (9,13) also covers the tail of ps[0]
    hook([ps[0]])
    assert fired == red.ranges()
