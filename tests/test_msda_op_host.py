"""CPU-side checks of the mmcv-boundary op (egorear_amd/msda.py): the oracle's two statements of the algorithm agree
(the gather form used as the checker and the grid_sample form the golden-generating shim gave the real reference), the
known-answer cases hold for the oracle, the shim registers under the mmcv module path, and the host layer rejects bad
operands and CPU tensors before anything is launched."""
import sys

import pytest
import torch


def _case(seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = torch.tensor([[6, 5], [3, 4], [2, 2]], dtype=torch.int64)
    starts = torch.tensor([0, 30, 42], dtype=torch.int64)
    value = torch.randn(2, 46, 3, 8, generator=g, requires_grad=True)
    loc = (torch.rand(2, 5, 3, 3, 4, 2, generator=g) * 1.4 - 0.2).requires_grad_()
    attn = torch.rand(2, 5, 3, 3, 4, generator=g).requires_grad_()
    return value, shapes, starts, loc, attn


def test_oracle_forms_agree():
    from oracle import egorear_oracle as O
    from oracle.ref_shims import _MSDAFunction
    value, shapes, starts, loc, attn = _case()
    a = O.msda_levels(value, shapes, starts, loc, attn)
    b = _MSDAFunction.apply(value, shapes, starts, loc, attn, 64)
    assert torch.allclose(a, b, atol=2e-6)
    go = torch.randn(a.shape, generator=torch.Generator().manual_seed(1))
    ga = torch.autograd.grad(a, [value, loc, attn], go)
    gb = torch.autograd.grad(b, [value, loc, attn], go)
    for x, y in zip(ga, gb):
        assert torch.allclose(x, y, atol=5e-5, rtol=1e-4)


def test_oracle_known_answers():
    from oracle import egorear_oracle as O
    h, w = 5, 7
    value = torch.randn(1, h * w, 2, 8, generator=torch.Generator().manual_seed(1))
    shapes, starts = torch.tensor([[h, w]]), torch.tensor([0])
    loc = torch.tensor([(3 + 0.5) / w, (2 + 0.5) / h]).view(1, 1, 1, 1, 1, 2).expand(1, 1, 2, 1, 1, 2)
    out = O.msda_levels(value, shapes, starts, loc, torch.ones(1, 1, 2, 1, 1))
    assert torch.allclose(out[0, 0], value[0, 2 * w + 3].reshape(-1), atol=1e-6)
    outside = torch.tensor([1.0 + 1.5 / w, 0.5]).view(1, 1, 1, 1, 1, 2).expand(1, 1, 2, 1, 1, 2)
    assert float(O.msda_levels(value, shapes, starts, outside, torch.ones(1, 1, 2, 1, 1)).abs().max()) == 0.0


def test_shim_registers_under_mmcv_path():
    from egorear_amd import msda
    saved = {k: sys.modules.get(k) for k in ("mmcv", "mmcv.ops", "mmcv.ops.multi_scale_deform_attn")}
    try:
        for k in saved:
            sys.modules.pop(k, None)
        msda.install_mmcv_shim()
        from mmcv.ops.multi_scale_deform_attn import MultiScaleDeformableAttnFunction as Fn
        assert Fn is msda.MultiScaleDeformableAttnFunction
        assert issubclass(Fn, torch.autograd.Function)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_host_layer_rejects_bad_operands():
    from egorear_amd import msda
    value, shapes, starts, loc, attn = [t.detach() for t in _case()]
    with pytest.raises(RuntimeError, match="no CPU path"):
        msda.msda_forward(value, shapes, starts, loc, attn)
    with pytest.raises(ValueError):
        msda.msda_forward(value[:, :, 0], shapes, starts, loc, attn)
    with pytest.raises(ValueError):
        msda.msda_forward(value, shapes, starts, loc[..., :1], attn)
    with pytest.raises(ValueError):
        msda.msda_forward(value, shapes, starts, loc, attn[:, :, :, :, :2])
    with pytest.raises(ValueError):
        msda.msda_forward(value, shapes.float(), starts, loc, attn)
    with pytest.raises(ValueError):
        msda.msda_backward(value, shapes, starts, loc, attn, torch.zeros(2, 5, 7))
