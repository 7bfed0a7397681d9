"""The reference's OWN multi-GPU route with the drop-in module: Lightning wraps `self.network` in
torch.nn.parallel.DistributedDataParallel (pl_wrappers/egoposeformer/pose_3d_mvf_ex.py:117-153, 208, 253-256; run.py:7-22 with
`--trainer.devices N`, strategy ddp_find_unused_parameters_true) - nothing of egorear_amd.train.Trainer is involved.  Two gloo ranks
sharing cuda:0 (the real job: RCCL, one GPU per rank) and one rank on RCCL: train-mode forward through DDP, the wrapper's loss,
loss.backward() -> every .grad is the MEAN of the ranks' single-process gradients (DDP's reducer hooks fire on the one autograd
node that carries the HIP reverse pass), the never-used parameters stay None, BatchNorm running statistics follow rank 0 at the
next forward (DDP's buffer broadcast against the module's stacked normalisation buffers).  Fresh child processes only."""
import copy
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build():
    from egorear_amd import configs, synth
    from egorear_amd.estimator import EgoPoseFormerMVFEX
    net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_rw")))
    synth.load_synth(net, 42)
    return net.to(DEV)


def _data(seed):
    from egorear_amd import synth
    from egorear_amd.metrics import generate_target
    B = 1
    return (synth.synth_images(B, 4, seed=10 + seed).to(DEV), synth.synth_coord_trans_mat(B, seed=20 + seed).to(DEV),
            synth.synth_gt_pose(B, seed=30 + seed).to(DEV), generate_target(synth.synth_joint_px(B, seed=40 + seed).to(DEV)).contiguous())


def _sample(t, n=16):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    v = f[::step][:n].float().cpu().numpy()
    return np.pad(v, (0, n - len(v)))


def _loss(preds, hms, gt_pose, gt_hm):
    """training_step of the wrapper (pose_3d_mvf_ex.py:117-153): MPJPE terms x 0.1 + heat-map terms x 10 - restated in
    oracle/train_oracle.py (test infrastructure)."""
    from oracle import train_oracle as TO
    return sum(TO.training_loss(preds, hms, gt_pose, gt_hm).values())


def _buffers(net, suffix):
    return {k: b.detach().double().cpu().numpy() for k, b in net.named_buffers() if k.endswith(suffix)}


def _worker(rank, world, port, backend, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP
        net = _build()
        ddp = DDP(net, device_ids=[0], find_unused_parameters=True)
        ddp.train()
        img, ctm, gt_pose, gt_hm = _data(rank)
        preds, hms = ddp(img, ctm, None)
        _loss(preds, hms, gt_pose, gt_hm).backward()
        torch.cuda.synchronize()
        grads = {k: (None if p.grad is None else (float(p.grad.double().norm()), _sample(p.grad))) for k, p in net.named_parameters()}
        rm1 = _buffers(net, "running_mean")
        # second iteration on the SAME frames on every rank: DDP broadcasts rank 0's buffers in front of the forward, so the
        # running statistics agree afterwards if and only if that broadcast reached the buffers the HIP forward reads
        for p in net.parameters():
            p.grad = None
        img, ctm, gt_pose, gt_hm = _data(7)
        preds, hms = ddp(img, ctm, None)
        _loss(preds, hms, gt_pose, gt_hm).backward()
        torch.cuda.synchronize()
        rm2 = _buffers(net, "running_mean")
        nbt = {k: int(b) for k, b in net.named_buffers() if k.endswith("num_batches_tracked")}
        g2 = {k: (None if p.grad is None else float(p.grad.double().norm())) for k, p in net.named_parameters()}
        out.put((rank, grads, rm1, rm2, nbt, g2, None))
    except Exception as exc:  # noqa: BLE001 - reported to the parent
        import traceback
        out.put((rank, None, None, None, None, None, f"{type(exc).__name__}: {exc}\n{traceback.format_exc()[-1500:]}"))
    finally:
        dist.destroy_process_group()


def _spawn(world, backend):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, *rest = out.get(timeout=900)
        got[rank] = rest
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in got:
        assert got[r][-1] is None, got[r][-1]
    return got


def _reference_grads(seeds):
    from egorear_amd import train
    ref = []
    for sd in seeds:
        S, _ = train.forward_backward(_build(), *_data(sd))
        ref.append({k: v.detach().clone() for k, v in S.pgrads.items()})
    torch.cuda.synchronize()
    return ref


def _check_mean(grads, ref):
    unused = 0
    for k, g in grads.items():
        if k not in ref[0]:
            assert g is None, f"{k}: a parameter the forward never uses must keep .grad = None under find_unused_parameters"
            unused += 1
            continue
        assert g is not None, k
        avg = sum(r[k] for r in ref) / len(ref)
        n_ref = float(avg.double().norm())
        assert abs(g[0] - n_ref) <= 1e-4 * n_ref + 1e-7, (k, g[0], n_ref)
        np.testing.assert_allclose(g[1], _sample(avg), rtol=1e-3, atol=max(1e-5 * n_ref, 1e-8), err_msg=k)   # (k_proj.bias: zero gradient up to noise)
    return unused


def test_ddp_around_the_dropin_module_two_ranks():
    got = _spawn(2, "gloo")
    ref = _reference_grads([0, 1])
    for r in (0, 1):
        assert _check_mean(got[r][0], ref) == 28          # the reference's 28 never-used parameters (golden `grad_present`)
    # both ranks hold the same reduced gradient
    for k, g in got[0][0].items():
        if g is not None:
            np.testing.assert_array_equal(g[1], got[1][0][k][1], err_msg=k)
    # per-rank batch statistics after the first iteration (no SyncBN) ...
    rm1_0, rm1_1 = got[0][1], got[1][1]
    assert any(not np.array_equal(rm1_0[k], rm1_1[k]) for k in rm1_0)
    # ... rank 0's everywhere in front of the second forward: same frames + same weights -> the same statistics afterwards
    rm2_0, rm2_1 = got[0][2], got[1][2]
    for k in rm2_0:
        assert np.array_equal(rm2_0[k], rm2_1[k]), k
    assert got[0][3] == got[1][3] and len(set(got[0][3].values())) == 1       # every BatchNorm counted two batches on top of the loaded state
    # and the second iteration reduced again (the reducer was re-armed: no "expected to have finished reduction" error)
    for k, n in got[0][4].items():
        assert (n is None) == (got[0][0][k] is None), k
        if n is not None:
            assert abs(n - got[1][4][k]) <= 1e-6 * max(n, 1e-12), k


def test_ddp_around_the_dropin_module_one_rank_rccl():
    got = _spawn(1, "nccl")
    ref = _reference_grads([0])
    assert _check_mean(got[0][0], ref) == 28
    assert len(set(got[0][3].values())) == 1
