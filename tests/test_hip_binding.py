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
