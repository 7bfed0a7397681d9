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


def test_chain_entry_refuses_unsupported_shapes_before_any_launch():
    """egr_conv1x1_chain_f32 (two 1x1 convs in one launch): every refusal of the C entry point comes back as a return code BEFORE a
    kernel is launched - checked here without a GPU through the raw C ABI (fake, aligned pointers: none is dereferenced on these paths)."""
    from egorear_amd import hip
    lib = hip.lib

    def desc(**kw):
        d = hip.ConvDesc()
        d.n, d.h, d.w, d.cin, d.cout = 2, 8, 8, 64, 128
        d.kh = d.kw = d.stride = 1
        d.pad, d.ho, d.wo = 0, 8, 8
        d.ldx, d.ldy, d.ldr = 64, 128, 128
        d.xmap = hip.NMap(2, 8 * 8 * 64, 0)
        d.ymap = hip.NMap(2, 8 * 8 * 128, 0)
        d.rmap = hip.NMap(1, 0, 0)
        d.act, d.res_mode, d.split_k, d.groups, d.w_format = 1, 0, 1, 1, 4
        for k, v in kw.items():
            setattr(d, k, v)
        return d

    P = 0x10000          # 16-byte aligned fake device pointer
    aux = hip.ConvAux(P, P, None, None, None, 0)
    ch = hip.ChainAux(P, P, None, 128, 1, 0, 0)

    def call(d, x=P, w1=P, res=None, aux_=aux, ch_=ch):
        return lib.egr_conv1x1_chain_f32(C.byref(d), x, w1, None, res, P, C.byref(aux_) if aux_ is not None else None,
                                          C.byref(ch_) if ch_ is not None else None, None)

    assert call(desc(), x=None) == hip.ENULL
    assert call(desc(), aux_=None) == hip.ENULL
    assert call(desc(), ch_=None) == hip.ENULL
    assert call(desc(), aux_=hip.ConvAux(P, None, None, None, None, 0)) == hip.ENULL        # the input's abs-max record is required
    assert call(desc(cin=96)) == hip.EINVAL                                                  # cin 64 / 128 only
    assert call(desc(cout=132)) == hip.EINVAL and call(desc(cout=126)) == hip.EINVAL
    assert call(desc(kh=3, kw=3, pad=1)) == hip.EINVAL and call(desc(stride=2, ho=4, wo=4)) == hip.EINVAL
    assert call(desc(w_format=1)) == hip.EINVAL                                              # fp16 scheme only
    assert call(desc(act=2)) == hip.EINVAL                                                   # GELU is not one of the chain's activations
    assert call(desc(), ch_=hip.ChainAux(P, P, None, 64, 1, 0, 0)) == hip.EINVAL             # the intermediate has 128 channels
    assert call(desc(res_mode=1), res=None) == hip.ENULL
    assert call(desc(res_mode=3, ho=7, wo=7, h=7, w=7), res=P) == hip.EINVAL                 # up-sampled residual: even output size
    assert call(desc(ldx=66)) == hip.EINVAL and call(desc(), x=P + 4) == hip.EINVAL          # 16-byte rows
    # the streamed form: 256 -> 256 -> <= 128 only, without a residual
    big = hip.ChainAux(P, P, None, 256, 1, 0, 0)
    assert call(desc(cin=256, ldx=256), ch_=hip.ChainAux(P, P, None, 128, 1, 0, 0)) == hip.EINVAL      # 256 -> 128 -> ..: no such kernel
    assert call(desc(cin=128, ldx=128), ch_=big) == hip.EINVAL
    assert call(desc(cin=256, ldx=256, res_mode=1), res=P, ch_=big) == hip.EINVAL
    assert call(desc(cin=256, ldx=256, cout=132), ch_=big) == hip.EINVAL


def test_chain_eligibility_rule():
    from egorear_amd import hip

    class W(hip.W6):
        def __init__(self, npad, K, groups=1, h2=True):
            super().__init__(None, npad, K, groups, 0)
            self.h2 = object() if h2 else None

    class X:
        def __init__(self, n, h, w, c, amax=True, ld=None, ptr=0x10000):
            self.n, self.h, self.w, self.c, self.amax = n, h, w, c, (object() if amax else None)
            self.ld = c if ld is None else ld
            self.nstride = h * w * self.ld
            self.t = type("T", (), {"data_ptr": staticmethod(lambda: ptr)})()

    big = X(64, 64, 64, 64)
    assert hip.chain_eligible(big, W(128, 64), W(128, 128), 128, 128, 1)
    assert not hip.chain_eligible(X(64, 64, 64, 64, amax=False), W(128, 64), W(128, 128), 128, 128, 1)     # no record: no fp16 pre-scale
    assert not hip.chain_eligible(big, W(128, 64, h2=False), W(128, 128), 128, 128, 1)
    assert not hip.chain_eligible(big, W(128, 64), W(128, 128), 128, 128, 1, scale1=object())                # BatchNorm-scaled conv
    assert not hip.chain_eligible(X(64, 64, 64, 256), W(128, 256), W(128, 128), 128, 128, 1)                 # cin 256 -> 128: W1 does not fit LDS
    assert hip.chain_eligible(X(64, 64, 64, 256), W(256, 256), W(128, 256), 256, 128, 1)                     # 256 -> 256 -> 128: streamed
    assert not hip.chain_eligible(X(64, 64, 64, 256), W(256, 256), W(128, 256), 256, 128, 1, res_mode=1)     # ... without a residual only
    assert not hip.chain_eligible(X(8, 32, 32, 256), W(256, 256), W(128, 256), 256, 128, 1)                  # ... from 65536 pixels
    assert not hip.chain_eligible(big, W(256, 64), W(128, 256), 256, 128, 1)                                 # 256-channel intermediate
    assert not hip.chain_eligible(X(2, 8, 8, 64), W(128, 64), W(128, 128), 128, 128, 1)                      # too few pixels to stream
    assert not hip.chain_eligible(X(63, 64, 64, 64), W(128, 64, groups=2), W(128, 128, groups=2), 128, 128, 2)   # images not divisible by groups
    # the entry's layout checks, mirrored: a channel slice at an odd offset / with odd strides stays on the single launches
    assert not hip.chain_eligible(X(64, 64, 64, 64, ptr=0x10004), W(128, 64), W(128, 128), 128, 128, 1)
    assert not hip.chain_eligible(X(64, 64, 64, 64, ld=66), W(128, 64), W(128, 128), 128, 128, 1)
    assert hip.chain_eligible(X(64, 64, 64, 64, ld=192), W(128, 64), W(128, 128), 128, 128, 1)               # an aligned slice of a wider buffer
    # a policy of its own (round 6): the thread's active LaunchPolicy decides, the process default is untouched
    with hip.use_policy(hip.POLICY.replace(chain=False)):
        assert not hip.chain_eligible(big, W(128, 64), W(128, 128), 128, 128, 1) and hip.CHAIN is False
    assert hip.CHAIN is True and hip.chain_eligible(big, W(128, 64), W(128, 128), 128, 128, 1)
    saved = hip.CHAIN
    try:
        hip.CHAIN = False
        assert not hip.chain_eligible(big, W(128, 64), W(128, 128), 128, 128, 1)
    finally:
        hip.CHAIN = saved


def test_token_chain_entries_refuse_before_any_launch():
    """egr_jqa_query_f32 / egr_pose_query_f32 / the head-offset tail of egr_joint_layer_f32 (round 5): every refusal is a return code
    BEFORE a launch - raw C ABI, fake aligned pointers, no GPU."""
    from egorear_amd import hip
    lib = hip.lib
    P = 0x10000
    ptrs_q = ("t", "s32", "w_hp2", "b_hp2", "w_bfb", "b_bfb", "embed", "w_q", "b_q", "w_ol", "b_ol", "x_out", "ol_out")

    def q(**kw):
        d = hip.JqaQueryDesc()
        d.B, d.J, d.C, d.groups, d.kb, d.pool_hw, d.ol_n, d.w_packed = 2, 15, 256, 4, 512, 64, 192, 2
        for n in ptrs_q:
            setattr(d, n, P)
        for k, v in kw.items():
            setattr(d, k, v)
        return lib.egr_jqa_query_f32(C.byref(d), None)

    assert lib.egr_jqa_query_f32(None, None) == hip.ENULL
    for n in ptrs_q:
        assert q(**{n: None}) == hip.ENULL, n
    assert q(C=128) == hip.EINVAL and q(kb=256) == hip.EINVAL                # the shipped refiner: 256 channels, 512-wide stride-32 features
    assert q(J=17) == hip.EINVAL and q(J=0) == hip.EINVAL and q(B=0) == hip.EINVAL and q(groups=0) == hip.EINVAL
    assert q(ol_n=190) == hip.EINVAL and q(w_packed=3) == hip.EINVAL and q(pool_hw=0) == hip.EINVAL
    assert q(w_q=P + 4) == hip.EINVAL                                        # 16-byte weight loads

    ptrs_p = ("h1", "w_m2", "b_m2", "cams", "w_qg0", "b_qg0", "w_qg2", "b_qg2", "w_qg4", "b_qg4", "w_ol", "b_ol", "pred_out", "anchors3d_out",
              "anchors2d_out", "valid_out", "x_out", "ol_out")

    def p(**kw):
        d = hip.PoseQueryDesc()
        d.B, d.J, d.C, d.ol_n, d.w_packed = 2, 16, 128, 192, 2
        for n in ptrs_p:
            setattr(d, n, P)
        for k, v in kw.items():
            setattr(d, k, v)
        return lib.egr_pose_query_f32(C.byref(d), None)

    assert lib.egr_pose_query_f32(None, None) == hip.ENULL
    for n in ptrs_p:
        assert p(**{n: None}) == hip.ENULL, n
    assert p(J=15) == hip.EINVAL and p(C=256) == hip.EINVAL and p(B=0) == hip.EINVAL and p(ol_n=8) == hip.EINVAL and p(w_packed=-1) == hip.EINVAL
    assert p(w_m2=P + 8) == hip.EINVAL

    # the layer launch with the head-offset tail: needs post_norm, 256 channels, 64 head channels, no regression head
    need = ("x", "g", "sigma", "rowmask", "w_fold", "c_fold", "w_out", "b_out", "w_fuse", "b_fuse", "ln1_g", "ln1_b", "w_qkv", "b_qkv", "w_mo", "b_mo",
            "ln2_g", "ln2_b", "w_f0", "b_f0", "w_f1", "b_f1", "ln3_g", "ln3_b", "x_out")

    def layer(**kw):
        d = hip.LayerDesc()
        d.B, d.J, d.V, d.C, d.heads, d.cf, d.groups, d.ffn_dim = 2, 15, 4, 256, 4, 128, 4, 512
        d.w_packed = 2
        for n in need + ("lnp_g", "lnp_b", "w_h0", "b_h0", "h0_out"):
            setattr(d, n, P)
        d.h0_n = 64
        for k, v in kw.items():
            setattr(d, k, v)
        return lib.egr_joint_layer_f32(C.byref(d), None)

    assert layer(lnp_g=None) == hip.EINVAL                                   # the tail reads post_norm's output
    assert layer(b_h0=None) == hip.ENULL and layer(h0_out=None) == hip.ENULL
    assert layer(C=128) == hip.EINVAL and layer(h0_n=32) == hip.EINVAL and layer(h0_out=P + 4) == hip.EINVAL
    assert layer(w_r0=P, b_r0=P, w_r2=P, b_r2=P, anchors3d=P, pred_out=P) == hip.EINVAL


def test_stem_pooled_batchnorm_entries_refuse_before_any_launch():
    """egr_bn_relu_maxpool_f32 / egr_bn_pool_backward_f32 (round 6): every refusal is a return code BEFORE a launch - raw C ABI, fake aligned
    pointers, no GPU."""
    from egorear_amd import hip
    from egorear_amd import hip_train as T          # binds the argtypes
    lib = hip.lib
    P = 0x10000
    fwd = lambda **kw: lib.egr_bn_relu_maxpool_f32(*[kw.get(k, d) for k, d in (("x", P), ("alpha", P), ("shift", P), ("y", P), ("slot", P), ("n", 4),    # noqa: E731
                                                     ("h", 8), ("w", 8), ("c", 64), ("groups", 2), ("k", 3), ("stride", 2), ("pad", 1), ("stream", None))])
    assert fwd(x=None) == hip.ENULL and fwd(slot=None) == hip.ENULL
    assert fwd(n=3) == hip.EINVAL                      # images not divisible by groups
    assert fwd(c=62) == hip.EINVAL and fwd(k=0) == hip.EINVAL and fwd(pad=2) == hip.EINVAL and fwd(x=P + 4) == hip.EINVAL
    bwd = lambda **kw: lib.egr_bn_pool_backward_f32(*[kw.get(k, d) for k, d in (("dpool", P), ("slot", P), ("x", P), ("mean", P), ("invstd", P),    # noqa: E731
                                                      ("alpha", P), ("shift", P), ("n", 4), ("h", 8), ("w", 8), ("c", 64), ("groups", 2), ("k", 3), ("stride", 2),
                                                      ("pad", 1), ("dgamma", P), ("dbeta", P), ("dx", P), ("ws", P), ("wsn", 1 << 20), ("xhat", None),
                                                      ("amax", None), ("stream", None))])
    assert bwd(dx=None) == hip.ENULL and bwd(ws=None) == hip.ENULL and bwd(shift=None) == hip.ENULL
    assert bwd(c=48) == hip.EINVAL                     # BatchNorm kernels: power-of-two channel counts from 64
    assert bwd(amax=P) == hip.EINVAL                   # a record without the batch extremes
    assert bwd(wsn=16) == hip.EWORKSPACE
    assert bwd(n=5) == hip.EINVAL


def test_launch_policy_from_environment_and_overrides():
    """hip.LaunchPolicy (round 6): the environment -> fields mapping of the process default, the thread-local override and the
    exact-arithmetic variant - no GPU involved."""
    import threading
    from egorear_amd import engine, hip
    p = hip.LaunchPolicy.from_env({})
    assert (p.w_format, p.h2, p.layer_h2, p.x6_min_rows, p.x6_min_flops, p.chain, p.chain_big_min_rows, p.wgrad_x6) == ("f16x2", True, True, 4096, 5e8, True, 65536, True)
    q = hip.LaunchPolicy.from_env({"EGR_W_FORMAT": "bf16x3", "EGR_X6_MIN_ROWS": "0", "EGR_CONV_CHAIN": "0", "EGR_FUSED_QUERY": "0"})
    assert (q.w_format, q.h2, q.layer_h2, q.x6_min_rows, q.chain, q.fused_query, q.wgrad_x6) == ("bf16x3", False, False, 0, False, False, True)
    assert hip.LaunchPolicy.from_env({"EGR_W_FORMAT": "f32"}).wgrad_x6 is False and hip.LaunchPolicy.from_env({"EGR_LAYER_H2": "0"}).layer_h2 is False
    e = p.exact()
    assert (e.w_format, e.h2, e.layer_h2) == ("bf16x3", False, False) and p.h2 is True          # a copy: the original is untouched
    default = hip.POLICY
    seen = {}

    def other_thread():
        seen["h2"], seen["fmt"] = hip.H2, engine.W_FORMAT          # another thread does not see this thread's override
    with hip.use_policy(e):
        assert hip.H2 is False and engine.W_FORMAT == "bf16x3" and engine.LAYER_H2 is False and hip.policy() is e
        t = threading.Thread(target=other_thread)
        t.start()
        t.join()
        with hip.use_policy(None):                                   # None: no change
            assert hip.policy() is e
    assert hip.policy() is default and (seen["h2"], seen["fmt"]) == (default.h2, default.w_format)
    saved = hip.X6_MIN_ROWS
    try:
        hip.X6_MIN_ROWS = 123                                        # assigning the historical attribute changes the process default
        assert hip.POLICY.x6_min_rows == 123 and hip.policy().x6_min_rows == 123
    finally:
        hip.X6_MIN_ROWS = saved
