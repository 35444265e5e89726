"""world_size-2 gloo tests (CPU) of the multi-GPU layer: view-sharded fusion equals the
single-process result (all-reduce path for pool fusers, ordered all-gather path for the GRU),
hypothesis sharding + loss gather preserve order.  The per-view encoder is irrelevant here
(views are independent up to the fuser), so random per-view volumes stand in for it; the GRU
fuser's gate convolutions are evaluated by the oracle."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lf_oracle import nets


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleGRUFuser:
    """Same call signature as latentfusion_amd.recon.fusion.GRUFuser, evaluated on CPU."""

    def __init__(self, ck):
        self.ck = ck

    def __call__(self, z_obj, a, b, c):
        return nets.fuse(self.ck, z_obj), {}


def _worker(rank, size, port, case, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    try:
        from latentfusion_amd import parallel
        from latentfusion_amd.recon.fusion import PoolFuser
        g = torch.Generator().manual_seed(0)
        V, C, S = 5, 4, 6                       # 5 views over 2 ranks: ragged 3 + 2
        z = torch.randn(1, V, C, S, S, S, generator=g)
        b, e = parallel.shard_range(V, rank, size)
        if case in ('mean', 'max', 'median'):
            fuser = PoolFuser(case)
            want = fuser(z, None, None, None)[0]
        else:
            gen = torch.Generator().manual_seed(1)
            sd = {}
            for gate in ('update_gate', 'reset_gate', 'out_gate'):
                sd[f'gru.{gate}.module.weight'] = torch.randn(C, 2 * C + 3, 3, 3, 3, generator=gen)
                sd[f'gru.{gate}.bias'] = torch.randn(C, generator=gen) * 0.1
            fuser = OracleGRUFuser({'type': 'GRUFuser', 'state_dict': sd})
            want = fuser(z, None, None, None)[0]
        got = parallel.fuse_sharded(fuser, z[:, b:e].contiguous(), V)
        err = (got - want).abs().max().item()
        # hypotheses
        losses = torch.arange(7, dtype=torch.float32) * 1.5
        hb, he = parallel.shard_range(7, rank, size)
        gathered = parallel.gather_losses(losses[hb:he].clone(), 7)
        q.put((rank, err, gathered.tolist() == losses.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('case', ['mean', 'max', 'median', 'gru'])
def test_view_sharded_fusion_world2(case):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, order_ok in results:
        # mean: (a+b)+(c+d+e) vs sequential mean -> fp32 re-association only; others bit-identical
        assert err <= (1e-6 if case == 'mean' else 0.0), (case, rank, err)
        assert order_ok


def test_shard_range_partition():
    from latentfusion_amd.parallel import shard_range
    for n in (0, 1, 7, 16, 17):
        for size in (1, 2, 3, 8):
            spans = [shard_range(n, r, size) for r in range(size)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(size - 1))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


def _allreduce_worker(rank, size, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=size)
    from latentfusion_amd import parallel
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    parallel.allreduce_flat_(flat, bucket_bytes=1024)          # 256 floats per bucket -> 4 collectives
    q.put((rank, flat.clone()))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2():
    """Bucketed mean all-reduce of the flat gradient buffer (data-parallel training step)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29650 + (os.getpid() % 200)
    ps = [ctx.Process(target=_allreduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    want = torch.arange(1000, dtype=torch.float32) * 1.5
    for r in (0, 1):
        assert torch.equal(got[r], want)
