"""CPU, world_size 2, gloo: the data-parallel plumbing of the N>1 path (sharding, flat gradient bucket,
all-reduce mean == gradient of the global-batch mean loss, noise broadcast)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from vae_gp_ode_amd.parallel import FlatGrads, GradAllReduce, broadcast_noise, shard_batch, shard_bounds
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 1))
    X = torch.randn(10, 5)  # the GLOBAL batch, identical on every rank
    # reference: gradient of the global-batch mean loss
    ref = torch.autograd.grad(net(X).pow(2).mean(), list(net.parameters()))
    fg = FlatGrads(net.parameters())
    lo, hi = shard_bounds(10, rank, world)
    sync = GradAllReduce(fg, dist, weight=(hi - lo) / 10.0)
    fg.zero()
    net(shard_batch(X, rank, world)).pow(2).mean().backward()   # accumulates into the flat views in place
    assert all(p.grad.data_ptr() == fg.flat.data_ptr() + 4 * o for p, o in zip(fg.params, fg.offsets))
    sync.all_reduce_grads()
    ok = all(torch.allclose(p.grad, r, atol=1e-6) for p, r in zip(net.parameters(), ref))
    # uneven shards
    lo3, hi3 = shard_bounds(11, rank, world)
    ok = ok and (hi3 - lo3) in (5, 6)
    # every rank ends with rank 0's draw
    nz = {'eps_u': torch.full((3, 2), float(rank)), 'rff_u': torch.full((1, 4, 2), float(rank) + 5)}
    broadcast_noise(nz, dist, src=0)
    ok = ok and float(nz['eps_u'][0, 0]) == 0.0 and float(nz['rff_u'][0, 0, 0]) == 5.0
    # cross-rank BatchNorm plumbing: the gather is rank-ordered, weights are relative to the own share, counts need no communication
    from vae_gp_ode_amd.parallel import BatchNormSync
    bs = BatchNormSync(dist, shares=[3, 2])
    got = bs.gather(torch.tensor([float(rank), 10.0 + rank]))
    ok = ok and got.tolist() == [[0.0, 10.0], [1.0, 11.0]]
    ok = ok and torch.allclose(bs.weights('cpu'), torch.tensor([1.0, 2.0 / 3.0] if rank == 0 else [1.5, 1.0]))
    ok = ok and abs(bs.count_all(3 * 49 if rank == 0 else 2 * 49) - 5 * 49) < 1e-9
    # several independent statistics vectors in one collective == the separate gathers, bit for bit; and the per-step accounting
    from vae_gp_ode_amd import parallel
    a, b = torch.randn(17) + rank, torch.randn(2, 5) * (rank + 1)
    with parallel.stats.step():
        sep = [bs.gather(a), bs.gather(b)]
    ok = ok and parallel.stats.summary() == {'all_gather': {'count': 2, 'payload_bytes': 4 * (17 + 10)}}
    with parallel.stats.step():
        packed = bs.gather_many([a, b])
        sync.all_reduce_grads()
    ok = ok and parallel.stats.summary() == {'all_gather': {'count': 1, 'payload_bytes': 4 * 27},
                                             'all_reduce': {'count': 1, 'payload_bytes': 4 * fg.total}}
    ok = ok and all(torch.equal(x, y) for x, y in zip(sep, packed)) and tuple(packed[1].shape) == (world, 2, 5)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gradient_allreduce_equals_global_batch_gradient():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(30)
    assert res == [(0, True), (1, True)]


def test_shard_bounds_cover_everything():
    from vae_gp_ode_amd.parallel import shard_bounds
    for n in (1, 7, 8, 256, 2049):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
