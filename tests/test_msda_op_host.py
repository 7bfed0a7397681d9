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
        assert callable(Fn.apply)
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


def test_op_is_a_torch_library_operator_with_a_meta_implementation():
    """the launcher is registered as a torch.library op with a Meta (fake) implementation, so the reference's own
    MSDeformAttn.forward statement sequence around it (deform_attn.py:121-162) traces under Dynamo with no graph break - checked here
    without a GPU on meta tensors: one graph, the operator in it, output shape / dtype from the fake kernel, and a backward graph
    through the registered autograd formula."""
    import torch.nn.functional as F
    from egorear_amd import msda
    assert hasattr(torch.ops.egorear_amd, "msda_fwd") and hasattr(torch.ops.egorear_amd, "msda_bwd")
    n, lq, heads, d, points, hw = 2, 15, 4, 64, 16, 64
    C = heads * d
    dev = "meta"
    shapes = torch.tensor([[hw, hw]], dtype=torch.int64, device=dev)
    starts = torch.tensor([0], dtype=torch.int64, device=dev)
    lin_v, lin_o, lin_a, lin_out = (torch.nn.Linear(C, C, device=dev), torch.nn.Linear(C, heads * points * 2, device=dev),
                                    torch.nn.Linear(C, heads * points, device=dev), torch.nn.Linear(C, C, device=dev))

    def module_sequence(query, ref_pts, tokens):
        value = lin_v(tokens).view(n, hw * hw, heads, d)
        off = lin_o(query).view(n, lq, heads, 1, points, 2)
        aw = F.softmax(lin_a(query).view(n, lq, heads, points), -1).view(n, lq, heads, 1, points)
        normalizer = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
        loc = ref_pts[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
        out = msda.MultiScaleDeformableAttnFunction.apply(value.to(dtype=torch.float32), shapes, starts, loc, aw, 64)
        return lin_out(out)

    query = torch.empty(n, lq, C, device=dev, requires_grad=True)
    ref = torch.empty(n, lq, 1, 2, device=dev)
    tokens = torch.empty(n, hw * hw, C, device=dev, requires_grad=True)
    out = module_sequence(query, ref, tokens)                       # eager on meta tensors: the fake kernel infers the shape
    assert out.shape == (n, lq, C) and out.dtype == torch.float32 and out.requires_grad
    gq, gt = torch.autograd.grad(out, [query, tokens], torch.empty_like(out))   # the registered autograd formula, on meta
    assert gq.shape == query.shape and gt.shape == tokens.shape
    import torch._dynamo as dynamo
    dynamo.reset()
    ex = dynamo.explain(module_sequence)(query.detach(), ref, tokens.detach())
    assert ex.graph_break_count == 0 and ex.graph_count == 1, ex
    targets = [str(nd.target) for g in ex.graphs for nd in g.graph.nodes if nd.op == "call_function"]
    assert any("egorear_amd.msda_fwd" in t for t in targets), targets
    with pytest.raises((NotImplementedError, RuntimeError)):       # no CPU kernel, no fallback
        msda.MultiScaleDeformableAttnFunction.apply(torch.zeros(1, 4, 1, 4), torch.tensor([[2, 2]]), torch.tensor([0]),
                                                     torch.zeros(1, 1, 1, 1, 1, 2), torch.ones(1, 1, 1, 1, 1), 64)
