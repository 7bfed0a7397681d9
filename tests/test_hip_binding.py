"""Host-side bookkeeping of the ctypes binding that needs no GPU: the device of a launch is taken from its operands."""
import ctypes as C

import pytest
import torch


class _Dev:
    def __init__(self, index):
        self.index = index


class _FakeCudaTensor:
    """Quacks like a device tensor for hip._p (no GPU in the CPU test tier)."""
    is_cuda = True

    def __init__(self, index, dtype=torch.float32, ptr=0x1000):
        self.device, self.dtype, self._ptr = _Dev(index), dtype, ptr

    def data_ptr(self):
        return self._ptr


def test_pointer_helper_tracks_one_device_per_launch(monkeypatch):
    from egorear_amd import hip
    hip._DEV[0] = None
    a, b = _FakeCudaTensor(1), _FakeCudaTensor(1, ptr=0x2000)
    assert isinstance(hip._p(a), C.c_void_p) and hip._DEV[0] == 1
    hip._p(b)
    assert hip._DEV[0] == 1
    with pytest.raises(RuntimeError, match="different devices"):
        hip._p(_FakeCudaTensor(0))
    assert hip._DEV[0] is None          # a failed assembly leaves nothing behind
    with pytest.raises(RuntimeError, match="expected"):
        hip._p(_FakeCudaTensor(1, dtype=torch.float64))
    assert hip._DEV[0] is None
    with pytest.raises(RuntimeError, match="no CPU path"):
        hip._p(torch.zeros(2))


def test_launch_switches_to_the_operands_device(monkeypatch):
    from egorear_amd import hip
    seen = []

    class _Ctx:
        def __init__(self, idx):
            self.idx = idx

        def __enter__(self):
            seen.append(("enter", self.idx))

        def __exit__(self, *a):
            seen.append(("exit", self.idx))

    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "device", _Ctx)
    hip._DEV[0] = None
    hip._p(_FakeCudaTensor(3))
    hip._launch("fake", lambda *a: seen.append(("call", a)) or 0, 7)
    assert seen == [("enter", 3), ("call", (7,)), ("exit", 3)] and hip._DEV[0] is None
    seen.clear()
    hip._p(_FakeCudaTensor(0))
    hip._launch("fake", lambda *a: seen.append(("call", a)) or 0, 8)
    assert seen == [("call", (8,))]      # already on the right device: no switch
    with pytest.raises(RuntimeError, match="fake failed"):
        hip._launch("fake", lambda *a: -1)


def test_launch_error_carries_the_return_code(monkeypatch):
    """A non-zero return of a C-ABI entry point becomes hip.LaunchError (a RuntimeError) with the code, so that a caller with a second
    way to run the launch (train.Step.conv: statistics epilogue -> plain launch + pass) can tell EINVAL / EWORKSPACE from a fault."""
    from egorear_amd import hip
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    for rc, text in ((hip.EINVAL, "EGR_EINVAL"), (hip.ENULL, "EGR_ENULL"), (hip.EWORKSPACE, "EGR_EWORKSPACE"), (700, "hipError_t 700")):
        hip._DEV[0] = None
        with pytest.raises(hip.LaunchError) as ei:
            hip._launch("egr_fake", lambda *a, rc=rc: rc)
        assert isinstance(ei.value, RuntimeError) and ei.value.code == rc and ei.value.name == "egr_fake" and text in str(ei.value)
    hip._DEV[0] = None
    hip._launch("egr_fake", lambda *a: 0)            # success: nothing raised


def test_arena_exhaustion_is_counted(monkeypatch):
    """hip.AmaxArena hands out 64-slot records in launch order and COUNTS the requests it has to refuse (bench.py reports the
    process-wide counter: each refusal is a launch that silently left the fp16 scheme)."""
    from egorear_amd import hip

    class _Arena(hip.AmaxArena):
        def __init__(self, records):
            self.buf = torch.zeros(records * 64, dtype=torch.int32)
            self.records, self.k, self.exhausted = records, 0, 0

    before = hip.ARENA_EXHAUSTED
    a = _Arena(2)
    r0, r1 = a.new(), a.new()
    assert r0.numel() == 64 and r1.numel() == 64 and r0.data_ptr() != r1.data_ptr()
    assert a.new() is None and a.new() is None
    assert a.exhausted == 2 and hip.ARENA_EXHAUSTED == before + 2
    a.k = 0                                           # (begin() rewinds; the counters keep their history)
    assert a.new() is not None and a.exhausted == 2


def test_grad_free_rule_names_real_parameters():
    """train._mvfex_grad_free: the parameters the reference's graph never reaches (stereo estimators' own conv_heatmap, refiners'
    frame_feat_proj_layers) exist under those names in the drop-in module - the rule DDP(find_unused_parameters=True) relies on."""
    import copy
    from egorear_amd import configs, train
    from egorear_amd.estimator import EgoPoseFormerMVFEX
    net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_rw")))
    names = [k for k, _ in net.named_parameters()]
    free = [k for k in names if train._mvfex_grad_free(k)]
    heads = [k for k in free if ".conv_heatmap." in k]
    proj = [k for k in free if ".frame_feat_proj_layers." in k]
    assert heads and proj and len(heads) + len(proj) == len(free)
    assert all("heatmap_estimator_stereo_" in k for k in heads)
    assert len({k.split(".frame_feat_proj_layers.")[0] for k in proj}) == 4          # the four refiners
    assert len(names) - len(free) == 536                                             # the gradients the golden step holds
