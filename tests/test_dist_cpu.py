"""world_size-2 gloo tests (CPU) of the multi-GPU path: the data-parallel gradient step
(lion_amd/dist.py, replaces utils/utils.py:717-770) and the sharding / gathering of a sampling job
(lion_amd/sampling.py, trainers/base_trainer.py:447-487)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(fn, world=2):
    port = _free_port()
    mp.spawn(_entry, args=(world, port, fn), nprocs=world, join=True)


def _entry(rank, world, port, fn):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world)
    finally:
        dist.destroy_process_group()


def _model(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(13, 32), torch.nn.SiLU(), torch.nn.Linear(32, 32),
                               torch.nn.SiLU(), torch.nn.Linear(32, 5))


def _w_bucketed(rank, world):
    from lion_amd.dist import BucketedGradAverager, average_gradients, broadcast_params
    m = _model(100 + rank)                       # different init per rank ...
    broadcast_params(m.parameters())             # ... made identical by ONE flat broadcast
    ref = _model(100)
    for p, q in zip(m.parameters(), ref.parameters()):
        assert torch.equal(p, q)
    torch.manual_seed(7 + rank)
    x, y = torch.randn(16, 13), torch.randn(16, 5)
    # reference semantics: utils.average_gradients == mean of the per-rank gradients
    m2 = _model(100)
    (m2(x) - y).square().mean().backward()
    average_gradients(m2.parameters())
    # bucketed + overlapped (tiny buckets force several of them, launched from the grad hooks)
    avg = BucketedGradAverager(m.parameters(), bucket_bytes=256, overlap=True)
    assert len(avg.buckets) >= 3
    for step in range(2):                         # second step: buffers reused, hooks still armed
        avg.zero_grad()
        (m(x) - y).square().mean().backward()
        avg.finish()
        for p, q in zip(m.parameters(), m2.parameters()):
            torch.testing.assert_close(p.grad, q.grad, rtol=1e-6, atol=1e-7)
    # every rank ends with the same averaged gradient
    flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert torch.equal(other[0], other[1])
    # a parameter that gets no gradient must not dead-lock its bucket
    m3 = _model(100)
    avg3 = BucketedGradAverager(m3.parameters(), bucket_bytes=256, overlap=True)
    avg3.zero_grad()
    (m3[0](x)).sum().backward()                   # only the first layer receives gradients
    avg3.finish()
    # ... and, like in the reference (utils/utils.py:725-727), ends the step WITHOUT a gradient, so that
    # Adam / EMA skip it instead of stepping it with zeros (weight decay, momentum)
    assert all(p.grad is not None for p in m3[0].parameters())
    assert all(p.grad is None for p in list(m3[2].parameters()) + list(m3[4].parameters()))
    # a foreign zero_grad(set_to_none=True) (torch.optim's default) breaks the .grad <-> bucket aliasing:
    # the hook moves the fresh gradient back into the bucket, the averaged result is still right
    opt = torch.optim.SGD(m3.parameters(), lr=0.0)
    avg3.zero_grad()
    opt.zero_grad(set_to_none=True)
    (m3(x) - y).square().mean().backward()
    avg3.finish()
    for p, q in zip(m3.parameters(), m2.parameters()):
        assert p.grad.data_ptr() == avg3._view_of[p].data_ptr()
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-6, atol=1e-7)
    # two backward passes before finish() would reduce a bucket after the first one: refused loudly
    avg3.zero_grad()
    (m3(x) - y).square().mean().backward()
    with pytest.raises(RuntimeError, match="second gradient"):
        (m3(x) - y).square().mean().backward()
    avg3.finish()
    avg3.remove_hooks()
    # non-overlapped mode gives the same numbers
    m4 = _model(100)
    avg4 = BucketedGradAverager(m4.parameters(), bucket_bytes=4096, overlap=False)
    avg4.zero_grad()
    (m4(x) - y).square().mean().backward()
    avg4.finish()
    for p, q in zip(m4.parameters(), m2.parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-6, atol=1e-7)
    # hooks armed but told not to launch (GraphedTrainStep's split mode: the backward is a replayed graph, the
    # collectives run eagerly behind it): finish() reduces every bucket itself; reduce_all() is the replay-time form
    m5 = _model(100)
    avg5 = BucketedGradAverager(m5.parameters(), bucket_bytes=256, overlap=True)
    avg5.launch_in_hooks = False
    avg5.zero_grad()
    (m5(x) - y).square().mean().backward()
    assert not avg5._launched and not avg5._works
    avg5.finish()
    for p, q in zip(m5.parameters(), m2.parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-6, atol=1e-7)
    avg5.zero_grad()
    (m5(x) - y).square().mean().backward()
    avg5.reduce_all()
    for p, q in zip(m5.parameters(), m2.parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-6, atol=1e-7)
    avg5._reset()


def _w_sampling(rank, world):
    from lion_amd.sampling import gather_samples, rank_seed, shard_batch
    total = 7                                     # ragged split: 4 + 3
    mine = shard_batch(total, rank, world)
    assert mine == (4 if rank == 0 else 3)
    seeds = [rank_seed(1234, r) for r in range(world)]
    assert len(set(seeds)) == world              # the reference reseeds every rank identically (bug)
    torch.manual_seed(rank_seed(1234, rank))
    pts = torch.randn(mine, 2048, 3)
    allp = gather_samples(pts, world)
    assert tuple(allp.shape) == (total, 2048, 3)
    lo = 0 if rank == 0 else 4
    assert torch.equal(allp[lo:lo + mine], pts)
    # different ranks really drew different clouds
    assert not torch.equal(allp[0], allp[4])


def _w_pair_matrix(rank, world):
    """lion_amd.metrics.pairwise_distance shards the ROWS of the pair matrix over the ranks and all-gathers them
    (the reference evaluates everything on rank 0, base_trainer.py:491-495).  The distance kernels need a GPU; here the
    sharding / gathering logic runs with a CPU stand-in for the two distance functions (ragged split: 7 rows, 2 ranks)."""
    from lion_amd import metrics
    torch.Tensor.cuda = lambda self, *a, **k: self

    def cd(a, b):                                   # brute-force Chamfer, same return convention as the kernels
        P = ((a[:, :, None, :] - b[:, None, :, :]) ** 2).sum(-1)
        return P.min(2)[0], P.min(1)[0]

    metrics.distChamferCUDAnograd = cd
    metrics.emd_approx = lambda a, b, require_grad=False: (a.mean(1) - b.mean(1)).abs().sum(-1)
    g = torch.Generator().manual_seed(5)
    smp, ref = torch.rand(7, 20, 3, generator=g), torch.rand(5, 20, 3, generator=g)
    full = metrics.pairwise_distance("CD", smp, ref, pair_batch=6, rank=0, world=1)
    mine = metrics.pairwise_distance("CD", smp, ref, pair_batch=6, rank=rank, world=world)
    assert tuple(mine.shape) == (7, 5) and torch.equal(mine, full)
    emd = metrics.pairwise_distance("EMD", smp, ref, pair_batch=4, rank=rank, world=world)
    assert torch.allclose(emd, metrics.pairwise_distance("EMD", smp, ref, pair_batch=4))
    res = metrics.compute_all_metrics(smp, ref, batch_size=8, rank=rank, world=world)
    res1 = metrics.compute_all_metrics(smp, ref, batch_size=8)
    assert res == res1 and "1-NN-CD-acc" in res


def test_sharded_pair_matrix_two_ranks():
    _run(_w_pair_matrix)


def test_bucketed_gradient_averaging_two_ranks():
    _run(_w_bucketed)


def test_sampling_shard_and_gather_two_ranks():
    _run(_w_sampling)


def test_single_process_is_a_noop():
    """is_distributed=False / world 1: every helper degenerates (utils/utils.py:719,768)."""
    from lion_amd.dist import BucketedGradAverager, average_gradients, broadcast_params
    from lion_amd.sampling import gather_samples, shard_batch
    m = _model(0)
    broadcast_params(m.parameters(), is_distributed=False)
    (m(torch.randn(4, 13))).sum().backward()
    g = [p.grad.clone() for p in m.parameters()]
    average_gradients(m.parameters(), is_distributed=False)
    for a, b in zip(g, m.parameters()):
        assert torch.equal(a, b.grad)
    avg = BucketedGradAverager(m.parameters())
    avg.zero_grad()
    (m(torch.randn(4, 13))).sum().backward()
    avg.finish()
    assert shard_batch(32, 0, 1) == 32
    x = torch.randn(3, 8, 3)
    assert gather_samples(x, 1) is x
