"""N > 1 plumbing on CPU: two gloo ranks, frame sharding without a collective, barrier-fenced timing with max over ranks."""
import os
import socket
import sys

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


# --------------------------------------------------------------------------- training side channels

def _full_net_on_meta():
    import copy
    from egorear_amd import configs
    from egorear_amd.estimator import EgoPoseFormerMVFEX
    with torch.device("meta"):
        return EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_rw")))


def test_gradient_stage_buckets_cover_the_flat_buffer_exactly_once():
    """The flat gradient buffer is exchanged as one all-reduce per stage of the reverse pass: the stage ranges must tile
    [0, total) without gap or overlap, every parameter must sit in exactly one of them, 16-byte aligned, no-decay before decay."""
    from egorear_amd.train import N_GRAD_STAGES, flat_layout, grad_stage, is_no_decay
    net = _full_net_on_meta()
    named = list(net.named_parameters())
    order, slots, stage_range, total = flat_layout(named)
    assert sorted(stage_range) == list(range(N_GRAD_STAGES))
    spans = [tuple(stage_range[s]) for s in sorted(stage_range)]
    assert spans[0][0] == 0 and spans[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))          # contiguous, disjoint
    assert sum(e - b for b, e in spans) == total
    seen = set()
    end = 0
    for (k, off, n, decay), (k2, p) in zip(slots, order):
        assert k == k2 and n == p.numel() and off % 4 == 0 and off >= end
        b, e = stage_range[grad_stage(k)]
        assert b <= off and off + n <= e                                   # inside its own stage's bucket
        assert decay == (not is_no_decay(k))
        seen.add(k)
        end = off + n
    assert seen == {k for k, _ in named} and len(slots) == len(named)
    assert total >= sum(p.numel() for _, p in named) and total - sum(p.numel() for _, p in named) < 4 * len(named)
    # bucket sizes in the order they are sent: the lifting head's 268 MB mlp_pred.0 goes first
    mb = [4 * (e - b) / 1e6 for b, e in spans]
    assert mb[0] > 250 and sum(mb) > 500


def _side_channel_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import copy
        from egorear_amd import configs
        from egorear_amd.dist import BufferSync, allreduce_mean_
        from egorear_amd.estimator import EgoPoseFormerHeatmap
        from egorear_amd.train import flat_layout
        torch.manual_seed(100 + rank)                                      # ranks start from DIFFERENT buffers and gradients
        net = EgoPoseFormerHeatmap(**copy.deepcopy(configs.heatmap_cfg()))
        for b in net.buffers():
            if b.dtype.is_floating_point:
                b.uniform_(0.5, 1.5)
            else:
                b.fill_(rank + 1)
        keys = list(net.state_dict().keys())
        sync = BufferSync(net)
        assert list(net.state_dict().keys()) == keys                       # re-homing keeps the state_dict contract
        mine = {k: v.clone().numpy() for k, v in net.named_buffers()}       # numpy: tensors do not survive the queue
        sync.broadcast(0)
        after = {k: v.clone().numpy() for k, v in net.named_buffers()}
        # the stage buckets of the real layout, exchanged one all-reduce per stage
        named = list(net.named_parameters())
        _, slots, stage_range, total = flat_layout(named)
        flat = torch.full((total,), float(rank + 1), dtype=torch.float64)
        for s in sorted(stage_range):
            b, e = stage_range[s]
            allreduce_gradients_(flat[b:e])
        terms = torch.tensor([1.0 + rank, 10.0 * (rank + 1)], dtype=torch.float64)
        mean = allreduce_mean_(terms.clone())
        gathered = [None] * world
        dist.all_gather_object(gathered, (mine, after, float(flat.min()), float(flat.max()), mean.tolist(), sync.numel()))
        if rank == 0:
            out.put(gathered)
    finally:
        dist.destroy_process_group()


def test_two_rank_buffer_broadcast_stage_allreduce_and_metric_mean():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_side_channel_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    g0, g1 = out.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mine0, after0, lo0, hi0, mean0, n0 = g0
    mine1, after1, lo1, hi1, mean1, n1 = g1
    import numpy as np
    assert any(not np.array_equal(mine0[k], mine1[k]) for k in mine0)      # they did differ before
    for k in mine0:                                                        # DDP broadcast_buffers: rank 0 wins everywhere
        assert np.array_equal(after0[k], mine0[k]) and np.array_equal(after1[k], mine0[k]), k
    assert n0 == n1 == sum(v.size for v in mine0.values())
    assert lo0 == hi0 == lo1 == hi1 == 3.0                                 # every element summed exactly once (1 + 2)
    assert mean0 == mean1 == [1.5, 15.0]                                   # sync_dist=True: mean over ranks


def test_one_distinct_gpu_per_rank_or_refuse(monkeypatch):
    """bench.py's N > 1 line must be verifiable: LOCAL_RANK beyond the visible GPUs is refused (it would make ranks share a device
    silently), duplicates in the gathered per-rank records are refused, and both have an explicit rehearsal override."""
    import pytest
    from egorear_amd import dist as D
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    assert [D.claim_device(r) for r in range(4)] == [0, 1, 2, 3]
    with pytest.raises(RuntimeError, match="share a device"):
        D.claim_device(4)
    assert D.claim_device(5, allow_shared=True) == 1
    recs = [{"host": "n0", "device_index": i, "pci_domain_id": 0, "pci_bus_id": 0x10 + i, "pci_device_id": 0} for i in range(4)]
    assert D.check_distinct_devices(recs) is True
    dup = recs[:3] + [dict(recs[0], rank=3)]
    with pytest.raises(RuntimeError, match="share a GPU"):
        D.check_distinct_devices(dup)
    assert D.check_distinct_devices(dup, allow_shared=True) is False
    assert D.check_distinct_devices([{"host": "n0", "device_index": 0}, {"host": "n0", "device_index": 1}]) is True
    assert D.gather_rank_records({"rank": 0}) == [{"rank": 0}]            # single process: no collective
    # a launcher that isolates one GPU per rank: every rank sees exactly one device and takes it; sharing is caught on the PCI addresses
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(RuntimeError, match="share a device"):
        D.claim_device(3)                                                  # one visible GPU and no isolation declared: refused
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "3")
    assert D.claim_device(3) == 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    with pytest.raises(RuntimeError, match="share a device"):
        D.claim_device(3)


def _skewed_worker(rank, world, port, out):
    """The bucket exchange of the segmented-graph training step (Trainer._capture cuts the graph at every stage marker and starts
    FusedAdamW.reduce_stage there), with the ranks deliberately out of step: rank 1 reaches every marker late by a different amount,
    rank 0 races ahead and parks several asynchronous all-reduces before rank 1 has issued its first."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        import types
        from egorear_amd import train
        from egorear_amd.train import FusedAdamW, N_GRAD_STAGES, Step, flat_layout
        named = list(_full_net_on_meta().named_parameters())
        _, slots, stage_range, total = flat_layout(named)
        opt = types.SimpleNamespace(flat_g=torch.full((total,), float(rank + 1), dtype=torch.float32), stage_range=stage_range, pg=None,
                                    force_collective=False, pending=[])
        reduce_stage = types.MethodType(FusedAdamW.reduce_stage, opt)          # the real method, on a CPU stand-in of the flat buffers
        # the order comes from the tape, not from timing: the markers are recorded by the forward (stages 1, 0, 2 in recording order),
        # the reverse pass runs them backwards, the encoders' stage is started by _finish_backward
        S = types.SimpleNamespace(tape=[], record=True, stage_hook=None, finish_param_grads=lambda: None)
        for st in (1, 0, 2):
            Step.mark_stage(S, st)
        order = []
        S.stage_hook = lambda st: (order.append(st), time.sleep((0.4 * ((st * 7 + 3) % 4)) if rank == 1 else 0.0), reduce_stage(st))
        t0 = time.perf_counter()
        Step.backward(S)
        S.stage_hook(N_GRAD_STAGES - 1)
        issued = time.perf_counter() - t0
        for h in opt.pending:
            h.wait()
        gathered = [None] * world
        dist.all_gather_object(gathered, (order, float(opt.flat_g.min()), float(opt.flat_g.max()), len(opt.pending), issued))
        if rank == 0:
            out.put(gathered)
    finally:
        dist.destroy_process_group()


def test_two_skewed_ranks_exchange_the_stage_buckets_in_tape_order_without_deadlock():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_skewed_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    g0, g1 = out.get(timeout=240)          # a mismatched collective order would hang here
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from egorear_amd.train import N_GRAD_STAGES
    assert g0[0] == g1[0] == [2, 0, 1, N_GRAD_STAGES - 1]        # the reverse of the recording order, then the encoders
    assert g0[1] == g0[2] == g1[1] == g1[2] == 3.0                # every element of every bucket summed exactly once (1 + 2)
    assert g0[3] == g1[3] == N_GRAD_STAGES                        # one asynchronous all-reduce per stage
    assert g1[4] > g0[4] + 0.5                                    # rank 1 really was late; rank 0 did not wait to ISSUE its buckets


# --------------------------------------------------------------------------- bench.py's gradient-exchange leg

def _exchange_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import types
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        # a stand-in for train.Trainer: the flat gradient buffer and its {stage: [begin, end)} buckets (train.flat_layout)
        flat = torch.full((1000,), float(rank + 1))
        tr = types.SimpleNamespace(opt=types.SimpleNamespace(stage_range={2: [600, 1000], 0: [0, 100], 1: [100, 600], 3: [1000, 1000]},
                                                             flat_g=flat, pg=None))
        rec = bench._exchange_leg(tr, world, "gloo", torch.device("cpu"))
        if rank == 0:
            out.put(rec)
    finally:
        dist.destroy_process_group()


def test_bench_exchange_leg_walks_the_stage_buckets():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    rec = out.get(timeout=60)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [st["bytes"] for st in rec["stages"]] == [400, 2000, 1600, 0]          # stage order 0, 1, 2, 3
    assert [st["stage"] for st in rec["stages"]] == ["lifting head", "refiners", "initial heat-map heads", "encoders"]
    assert rec["bytes_per_step"] == 4000 and rec["communicator_size"] == 2
    assert all(st["allreduce_ms"] > 0 for st in rec["stages"])


# --------------------------------------------------------------------------- eight-rank rehearsal of bench.py's N > 1 plumbing

def _eight_rank_worker(rank, world, port, out):
    """Everything bench.py does around the timed region when the driver launches it on an 8-GPU node, on stand-in work and host
    tensors: process-group bring-up with the RCCL attempt (fails on every rank here: no GPU) and the unanimous gloo fallback, the
    strong-scaling share of --global-batch 512, barrier-fenced timing with max over ranks, the per-rank device records and the
    distinct-device refusal, the stage-bucket gradient exchange (pl_wrappers/egoposeformer/pose_3d_mvf_ex.py:208, 253-256)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import time
    import types
    import bench
    from egorear_amd import dist as D
    res = {}
    try:
        backend, note, rccl_ok = bench.init_distributed(world, rank, "nccl", None)
        res["backend"], res["rccl_ok"], res["note_has_reason"] = backend, rccl_ok, note.startswith("gloo (RCCL failed:")
        B = bench.frames_per_rank(64, 512, world)
        res["B"] = B
        b, e = shard_range(512, rank, world)
        res["span"] = (b, e)
        own = {}
        elapsed = timed_steps(lambda: time.sleep(0.005 * (1 + (rank == 5) * 4)), 4, 1, lambda: None, None, detail=own)
        res["elapsed"], res["own_s"] = elapsed, own["own_s"]
        rec = {"rank": rank, "local_rank": rank, "host": "node0", "device_index": rank, "pci_domain_id": 0, "pci_bus_id": 0x10 + rank, "pci_device_id": 0,
               "frames_per_s": B * 4 / own["own_s"]}
        records = D.gather_rank_records(rec)
        res["n_records"], res["distinct"] = len(records), D.check_distinct_devices(records)
        try:
            D.check_distinct_devices(records[:7] + [dict(records[2], rank=7)])
            res["refused"] = False
        except RuntimeError:
            res["refused"] = True
        flat = torch.full((1000,), float(rank + 1))
        tr = types.SimpleNamespace(opt=types.SimpleNamespace(stage_range={2: [600, 1000], 0: [0, 100], 1: [100, 600], 3: [1000, 1000]}, flat_g=flat, pg=None))
        ex = bench._exchange_leg(tr, world, backend, torch.device("cpu"))
        res["exchange_bytes"], res["comm"] = ex["bytes_per_step"], ex["communicator_size"]
        # 7 warm-up/timed all-reduces per bucket: every element went through SUM 7 times -> value x world^7 is out of range; check one fresh reduce instead
        g = torch.full((16,), float(rank + 1))
        D.allreduce_gradients_(g, None)
        res["sum"] = float(g[0])
        gathered = [None] * world
        dist.all_gather_object(gathered, res)
        if rank == 0:
            out.put(gathered)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_eight_rank_rehearsal_of_the_bench_plumbing():
    import pytest
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 8
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_eight_rank_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = out.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(got) == world
    for r, g in enumerate(got):
        assert g["backend"] == "gloo" and g["rccl_ok"] is False and g["note_has_reason"]          # unanimous fallback, reason in the line
        assert g["B"] == 64 and g["span"] == (64 * r, 64 * (r + 1))                               # strong scaling: 512 / 8 frames each
        assert g["n_records"] == world and g["distinct"] is True and g["refused"] is True
        assert g["exchange_bytes"] == 4000 and g["comm"] == world
        assert g["sum"] == float(sum(range(1, world + 1)))
    assert len({g["elapsed"] for g in got}) == 1                                                   # max over ranks: the same figure everywhere
    assert got[0]["elapsed"] >= got[5]["own_s"] - 1e-3 and got[5]["own_s"] > 3 * got[0]["own_s"]   # ... and it is the slow rank's
    # a mixed RCCL outcome is refused, and an uneven strong-scaling share too
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    with pytest.raises(SystemExit, match="refusing to mix backends"):
        bench.check_unanimous([True] * 7 + [False])
    bench.check_unanimous([False] * 8)
    bench.check_unanimous([True] * 8)
    with pytest.raises(SystemExit, match="not divisible"):
        bench.frames_per_rank(64, 500, 8)
    assert bench.frames_per_rank(64, 0, 8) == 64


def test_bench_pmc_kernel_families_and_live_pass_fallback(monkeypatch):
    """bench.py's live PMC passes (round 6): the kernel-name -> family rule of the reduction (the same as tools/pmc_traffic.py) and the
    fallback when rocprofv3 does not exist (the committed summary then stays in charge) - no GPU."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    fam = bench._pmc_family
    assert fam("void (anonymous namespace)::conv_tapx_kernel<1, 4, 2, 2, false, 0>(egrc::ConvArgs)") == "f16x2"
    assert fam("void (anonymous namespace)::conv_pw_x6_kernel<8, 4, 0, 2>(egrc::ConvArgs)") == "f16x2"
    assert fam("void (anonymous namespace)::conv_pw_x6_kernel<8, 4, 0, 3>(egrc::ConvArgs)") == "bf16x3"
    assert fam("void (anonymous namespace)::conv_pw2_kernel<0, 8>((anonymous namespace)::ChainArgs)") == "f16x2"
    assert fam("void (anonymous namespace)::conv_igemm_kernel<64, 64, 2, 2>(egrc::ConvArgs)") == "f32"
    assert fam("void (anonymous namespace)::stem_x6_kernel<true, 2>(StemArgs)") == "other"
    assert bench._pmc_is_tapx3x3("conv_tapx_kernel<2, 2, 2, 1, true, 0>(egrc::ConvArgs)") and bench._pmc_is_tapx3x3("conv_tapx_kernel<1, 4, 2, 2, false, 0>(x)")
    assert not bench._pmc_is_tapx3x3("conv_tapx_kernel<1, 4, 1, 0, false, 0>(egrc::ConvArgs)") and not bench._pmc_is_tapx3x3("conv_pw_x6_kernel<8, 4, 0, 2>(x)")
    import shutil
    monkeypatch.setattr(shutil, "which", lambda name: None)
    monkeypatch.setattr(os.path, "exists", lambda p: False if "rocprofv3" in p else True)
    res, why = bench.live_pmc_traffic(64)
    assert res is None and "rocprofv3" in why
