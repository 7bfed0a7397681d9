"""N > 1 plumbing on CPU: two gloo ranks, frame sharding without a collective, barrier-fenced timing with max over ranks."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from egorear_amd.dist import allreduce_gradients_, grad_seed_scale, shard_range, timed_steps


def test_shard_range_partitions_frames():
    for n in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from egorear_amd import synth
        frames = synth.synth_images(6, 4, seed=3, size=8)            # the "dataset": 6 independent frames
        b, e = shard_range(frames.shape[0], rank, world)
        mine = frames[b:e]
        calls = []

        def run():
            calls.append(1)
            # stand-in for the per-rank forward: rank 1 is slower, so max-over-ranks must report its time
            if rank == 1:
                import time
                time.sleep(0.02)
            return float(mine.sum())

        elapsed = timed_steps(run, steps=3, warmup=2, sync=lambda: None)
        gathered = [None] * world
        dist.all_gather_object(gathered, (b, e, float(mine.double().sum()), elapsed, len(calls)))
        if rank == 0:
            out.put((gathered, float(frames.double().sum())))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_timing():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, total = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (b0, e0, s0, t0, c0), (b1, e1, s1, t1, c1) = gathered
    assert (b0, e0, b1, e1) == (0, 3, 3, 6)                   # disjoint, complete shards
    assert abs((s0 + s1) - total) < 1e-6                      # every frame processed exactly once, no exchange needed
    assert c0 == c1 == 5                                      # 2 warm-up + exactly 3 timed steps on each rank
    assert t0 == t1 and t0 >= 0.06                            # MAX over ranks (rank 1 sleeps 3 x 20 ms)


def _train_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # toy data-parallel step with the training path's exchange: per-rank mean loss over its shard, backward seeded with
        # 1/world, ONE sum all-reduce of the flat gradient
        g = torch.Generator().manual_seed(5)
        x, y = torch.randn(8, 16, generator=g, dtype=torch.float64), torch.randn(8, generator=g, dtype=torch.float64)
        w = torch.linspace(-1, 1, 16, dtype=torch.float64).requires_grad_(True)
        b, e = shard_range(8, rank, world)
        loss = ((x[b:e] @ w - y[b:e]) ** 2).mean()
        (loss * grad_seed_scale()).backward()
        flat = w.grad.clone()
        allreduce_gradients_(flat)
        if rank == 0:
            w2 = w.detach().clone().requires_grad_(True)
            (((x @ w2 - y) ** 2).mean()).backward()            # what one process with the global batch computes
            out.put((flat, w2.grad))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_average_is_one_sum_allreduce():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got, ref = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.allclose(got, ref, rtol=1e-12, atol=1e-14)
    assert grad_seed_scale() == 1.0 and allreduce_gradients_(got) is got   # single process: no-ops
