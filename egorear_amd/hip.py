"""ctypes binding of libegorear_hip.so (C ABI: include/egorear_hip.h).

PyTorch is used here only as the owner of device memory and streams: every wrapper
takes torch tensors, checks device / dtype / layout on the host (a kernel fault can
reset the GPU), extracts raw pointers and the current HIP stream, and calls the C
entry point.  There is no fallback: if the library is missing, import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EGR_LIB") or os.path.join(_HERE, "csrc", "libegorear_hip.so")   # EGR_LIB: experiment builds (tools/build_variant.py)

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
RES_NONE, RES_BEFORE_ACT, RES_AFTER_ACT, RES_UP2_BEFORE_ACT = 0, 1, 2, 3


# --------------------------------------------------------------------------- launch policy

import contextlib
import dataclasses
import sys
import types


@dataclasses.dataclass
class LaunchPolicy:
    """Which kernel family a launch goes to - ONE object instead of module globals (round 6).  A module (engine.set_policy) or a
    caller (`with hip.use_policy(p):`, thread-local) can hold its own; everything else follows the process default `hip.POLICY`,
    which is made from the environment at import.  The historical module attributes (hip.H2, hip.X6_MIN_ROWS, hip.CHAIN_*,
    engine.W_FORMAT, engine.LAYER_H2, engine.FUSED_*) are properties onto it: reading one returns the ACTIVE policy's value,
    assigning one changes the process default (what the parity tests that sweep launch rules do)."""
    # weight operand of the implicit-GEMM launches: "f16x2" = bf16x3 image + fp16 companion, "bf16x3", "f32" (engine._w_operand)
    w_format: str = "f16x2"
    # the fp16 scheme for the launches whose input carries an abs-max record (EGR_W_FORMAT=bf16x3 / f32 switch it off)
    h2: bool = True
    # Below these sizes (all groups together) a launch is bound by latency or by streaming the weights once, not by the matrix
    # cores: it keeps the 4-byte weight format (measured: 3840 rows x K 4096 slower with the split, 8192 rows x N 512 x K 4608
    # faster).  Re-swept with the tap-sharing / streaming kernels in place (tools/x6_small_sweep.sh: 8192 rows / 4e9 flops ->
    # 4096 / 5e8: batch 2-16 latency -2 .. -7 %, batch 32 +1.9 % frames/s, batch 1 and 64 unchanged)
    x6_min_rows: int = 4096
    x6_min_flops: float = 5e8
    # the training step keeps the rule its gradient goldens were recorded under (tests/test_gpu_train_step.py; train.py passes x6_min)
    x6_train_min_rows: int = 8192
    x6_train_min_flops: float = 4e9
    # two 1x1 convolutions back to back as one launch (egr_conv1x1_chain_f32): EGR_CONV_CHAIN=0 keeps the pair of launches
    chain: bool = True
    chain_min_rows: int = 8192       # (measured: batch 1 1.46 -> 1.41 ms with the FPN level-1 chain; smaller chains lose)
    chain_big: bool = True           # the streamed 256 -> 256 -> 128 form (conv_pw2_kernel)
    chain_big_min_rows: int = 65536
    # weight gradients of large layers on the 16-bit matrix cores with operand splits (the kernel keeps small ones on fp32 MFMA)
    wgrad_x6: bool = True
    wgrad_force: bool = False        # tests: every weight-gradient launch on the split kernel, whatever its size
    # engine: the fused transformer layer's contractions in the fp16 scheme; one launch per layer / per token chain
    layer_h2: bool = True
    fused_layer: bool = True
    fused_query: bool = True

    @classmethod
    def from_env(cls, env=None) -> "LaunchPolicy":
        e = os.environ if env is None else env
        fmt = e.get("EGR_W_FORMAT", "f16x2")
        return cls(w_format=fmt, h2=fmt == "f16x2",
                   x6_min_rows=int(e.get("EGR_X6_MIN_ROWS", "4096")), x6_min_flops=float(e.get("EGR_X6_MIN_FLOPS", "5e8")),
                   x6_train_min_rows=int(e.get("EGR_X6_TRAIN_MIN_ROWS", "8192")), x6_train_min_flops=float(e.get("EGR_X6_TRAIN_MIN_FLOPS", "4e9")),
                   chain=e.get("EGR_CONV_CHAIN", "1") != "0", chain_min_rows=int(e.get("EGR_CONV_CHAIN_MIN_ROWS", "8192")),
                   chain_big=e.get("EGR_CONV_CHAIN_BIG", "1") != "0", chain_big_min_rows=int(e.get("EGR_CONV_CHAIN_BIG_MIN_ROWS", "65536")),
                   wgrad_x6=fmt != "f32", layer_h2=fmt == "f16x2" and e.get("EGR_LAYER_H2", "1") != "0",
                   fused_layer=e.get("EGR_FUSED_LAYER", "1") != "0", fused_query=e.get("EGR_FUSED_QUERY", "1") != "0")

    def replace(self, **kw) -> "LaunchPolicy":
        return dataclasses.replace(self, **kw)

    def exact(self) -> "LaunchPolicy":
        """The exact-operand arithmetic (EGR_W_FORMAT=bf16x3): three bf16 planes, six products, fused layers on the fp32 matrix cores."""
        return self.replace(w_format="bf16x3", h2=False, layer_h2=False)


POLICY = LaunchPolicy.from_env()       # the process default
_POLICY_TLS = threading.local()


def policy() -> LaunchPolicy:
    """The policy in force for the calling thread: the innermost `use_policy`, else the process default."""
    cur = getattr(_POLICY_TLS, "cur", None)
    return cur if cur is not None else POLICY


@contextlib.contextmanager
def use_policy(p: Optional[LaunchPolicy]):
    """Launches issued by this thread inside the block follow `p` (None: no change)."""
    if p is None:
        yield
        return
    old = getattr(_POLICY_TLS, "cur", None)
    _POLICY_TLS.cur = p
    try:
        yield
    finally:
        _POLICY_TLS.cur = old


def install_policy_properties(module_name: str, mapping: dict) -> None:
    """{MODULE_ATTRIBUTE: LaunchPolicy field}: the module's historical knobs become properties - read = the active policy, write =
    the process default."""
    def make(field):
        return property(lambda self: getattr(policy(), field), lambda self, v: setattr(POLICY, field, v))
    mod = sys.modules[module_name]
    mod.__class__ = type("_PolicyModule", (types.ModuleType,), {name: make(field) for name, field in mapping.items()})


install_policy_properties(__name__, {
    "H2": "h2", "X6_MIN_ROWS": "x6_min_rows", "X6_MIN_FLOPS": "x6_min_flops", "X6_TRAIN_MIN_ROWS": "x6_train_min_rows",
    "X6_TRAIN_MIN_FLOPS": "x6_train_min_flops", "CHAIN": "chain", "CHAIN_MIN_ROWS": "chain_min_rows", "CHAIN_BIG": "chain_big",
    "CHAIN_BIG_MIN_ROWS": "chain_big_min_rows", "WGRAD_X6": "wgrad_x6", "WGRAD_FORCE": "wgrad_force"})

EXPORTS = [
    "egr_conv2d_nhwc_f32", "egr_stem_conv7x7_f32", "egr_maxpool_nhwc_f32", "egr_upsample2x_nhwc_f32",
    "egr_avgpool_nhwc_f32", "egr_argmax_rows_f32", "egr_layernorm_f32", "egr_joint_mha_f32", "egr_msda_gather_f32",
    "egr_fisheye_project_f32", "egr_linear_smallk_f32", "egr_jqa_sum_f32", "egr_tokens_to_nhwc_f32", "egr_version",
    "egr_device_arch", "egr_conv_force_config", "egr_preprocess_u8_f32", "egr_pose_metrics_f32", "egr_gt_heatmap_f32", "egr_conv_debug_stamps", "egr_conv2d_wgrad_f32", "egr_conv2d_masked_f32", "egr_up2_relu_head_f32",
    "egr_msda_fwd_f32", "egr_msda_bwd_f32", "egr_w6_elems", "egr_pack_w6_f32", "egr_pack_w6_many_f32", "egr_joint_layer_f32",
    "egr_preprocess_fused_u8_f32", "egr_preprocess_band_rows", "egr_conv_set_persist", "egr_stem_conv7x7_pool_f32",
    "egr_stem_w6_bytes", "egr_pack_stem_w6_f32", "egr_stem_conv7x7_x6_f32", "egr_wgrad_last_kernel", "egr_conv_last_kernel", "egr_conv_set_tap", "egr_conv_set_tapx", "egr_conv_set_splitk_fused", "egr_fisheye_project2_f32", "egr_pack_layer_w_f32", "egr_pack_layer_wh2_f32",
    "egr_conv2d_nhwc_ex_f32", "egr_wh2_elems", "egr_pack_wh2_f32", "egr_absmax_f32", "egr_stem_conv7x7_x6_ex_f32", "egr_stem_wh2_bytes", "egr_pack_stem_wh2_f32", "egr_stem_conv7x7_h2_f32",
    "egr_pack_wh2_many_f32", "egr_conv2d_masked_ex_f32", "egr_conv2d_wgrad_ex_f32", "egr_wgrad_last_h2",
    "egr_wstream_image_bytes", "egr_pack_wstream_f32", "egr_linear_wstream_workspace_bytes", "egr_linear_wstream_f32", "egr_conv1x1_chain_f32",
    "egr_jqa_query_f32", "egr_pose_query_f32", "egr_layer_set_planes", "egr_head_set_persist",
]


class NMap(C.Structure):
    _fields_ = [("n_inner", C.c_int32), ("stride_inner", C.c_int64), ("stride_outer", C.c_int64)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("ho", C.c_int32), ("wo", C.c_int32), ("ldx", C.c_int32), ("ldy", C.c_int32), ("ldr", C.c_int32),
        ("xmap", NMap), ("ymap", NMap), ("rmap", NMap),
        ("act", C.c_int32), ("res_mode", C.c_int32), ("out_nchw", C.c_int32), ("split_k", C.c_int32),
        ("groups", C.c_int32),
        ("gx", C.c_int64), ("gw", C.c_int64), ("gp", C.c_int64), ("gy", C.c_int64), ("gr", C.c_int64),
        ("grs", C.c_int64), ("grm", C.c_int64),
        ("transposed", C.c_int32), ("w_format", C.c_int32),
    ]


class ConvAux(C.Structure):
    """egr_conv_aux of include/egorear_hip.h: side operands of the fp16 scheme (EGR_W_F16X2) and the abs-max record of the output."""
    _fields_ = [("w_descale", C.c_void_p), ("amax_in", C.c_void_p), ("amax_out", C.c_void_p),
                ("bn_partials", C.c_void_p), ("bn_tiles_out", C.c_void_p), ("bn_capacity", C.c_int64)]


class ChainAux(C.Structure):
    """egr_chain_aux of include/egorear_hip.h: the second conv of egr_conv1x1_chain_f32."""
    _fields_ = [("w2", C.c_void_p), ("w2_descale", C.c_void_p), ("shift1", C.c_void_p), ("cmid", C.c_int32), ("act1", C.c_int32),
                ("gw2", C.c_int64), ("gp1", C.c_int64)]


class LayerDesc(C.Structure):
    """egr_layer_desc of include/egorear_hip.h (field for field)."""
    _P = C.c_void_p
    _fields_ = ([(n, C.c_int32) for n in ("B", "J", "V", "C", "heads", "cf", "groups", "ffn_dim")] +
                [("eps", C.c_float), ("mha_scale", C.c_float)] +
                [(n, C.c_void_p) for n in ("x", "g", "e", "sigma", "rowmask",
                                           "w_fold", "c_fold", "w_out", "b_out", "w_fuse", "b_fuse", "ln1_g", "ln1_b",
                                           "w_qkv", "b_qkv", "w_mo", "b_mo", "ln2_g", "ln2_b", "w_f0", "b_f0", "w_f1", "b_f1", "ln3_g", "ln3_b",
                                           "x_out", "w_ol", "b_ol", "ol_out")] +
                [("ol_n", C.c_int32), ("w_packed", C.c_int32)] +
                [(n, C.c_void_p) for n in ("lnp_g", "lnp_b", "xn_out", "w_r0", "b_r0", "w_r2", "b_r2", "anchors3d", "pred_out",
                                           "w_h0", "b_h0", "h0_out", "amax_h0")] +
                [("h0_n", C.c_int32)])


class JqaQueryDesc(C.Structure):
    """egr_jqa_query_desc of include/egorear_hip.h (field for field)."""
    _fields_ = ([(n, C.c_int32) for n in ("B", "J", "C", "groups", "kb", "pool_hw", "ol_n", "w_packed")] +
                [(n, C.c_void_p) for n in ("t", "s32", "w_hp2", "b_hp2", "w_bfb", "b_bfb", "embed", "w_q", "b_q", "w_ol", "b_ol",
                                           "x_out", "ol_out")])


class PoseQueryDesc(C.Structure):
    """egr_pose_query_desc of include/egorear_hip.h (field for field)."""
    _fields_ = ([(n, C.c_int32) for n in ("B", "J", "C", "ol_n", "w_packed")] +
                [(n, C.c_void_p) for n in ("h1", "w_m2", "b_m2", "ctm", "cams", "w_qg0", "b_qg0", "w_qg2", "b_qg2", "w_qg4", "b_qg4",
                                           "w_ol", "b_ol", "pred_out", "anchors3d_out", "anchors2d_out", "valid_out", "x_out", "ol_out")])


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m egorear_amd.csrc.build` "
            "(or __graft_entry__.build()).  egorear_amd has no CPU / PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.egr_conv2d_nhwc_f32.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.egr_conv2d_nhwc_ex_f32.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, C.POINTER(ConvAux), vp]
    lib.egr_conv1x1_chain_f32.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, C.POINTER(ConvAux), C.POINTER(ChainAux), vp]
    lib.egr_fill_f32.argtypes = [vp, f32, i64, vp]      # (include/egorear_train.h; zeroes the abs-max records)
    lib.egr_fill_f32.restype = C.c_int
    lib.egr_absmax_f32.argtypes = [vp, i64, vp, vp]
    lib.egr_wh2_elems.restype = C.c_int64
    lib.egr_wh2_elems.argtypes = [i32, i32]
    lib.egr_pack_wh2_f32.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    lib.egr_stem_conv7x7_f32.argtypes = [vp, NMap, i32, i32, i32, vp, vp, vp, vp, i32, i64, vp]
    lib.egr_stem_conv7x7_pool_f32.argtypes = [vp, NMap, i32, i32, i32, vp, vp, vp, vp, i32, i64, vp]
    lib.egr_wgrad_last_kernel.argtypes = []
    lib.egr_conv_last_kernel.argtypes = []
    lib.egr_conv_set_tap.argtypes = [i32]
    lib.egr_conv_set_tapx.argtypes = [i32, i32, i32]
    lib.egr_conv_set_splitk_fused.argtypes = [i32]
    lib.egr_stem_w6_bytes.restype = C.c_int64
    lib.egr_stem_w6_bytes.argtypes = []
    lib.egr_pack_stem_w6_f32.argtypes = [vp, i32, vp, vp]
    lib.egr_stem_conv7x7_x6_f32.argtypes = [vp, NMap, i32, i32, i32, vp, vp, vp, vp, i32, i32, i64, vp]
    lib.egr_stem_conv7x7_x6_ex_f32.argtypes = [vp, NMap, i32, i32, i32, vp, vp, vp, vp, i32, i32, i64, vp, vp]
    lib.egr_stem_wh2_bytes.restype = C.c_int64
    lib.egr_stem_wh2_bytes.argtypes = []
    lib.egr_pack_stem_wh2_f32.argtypes = [vp, i32, vp, vp, vp]
    lib.egr_stem_conv7x7_h2_f32.argtypes = [vp, NMap, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, i64, vp, vp]
    lib.egr_wstream_image_bytes.restype = C.c_int64
    lib.egr_wstream_image_bytes.argtypes = [i32, i32]
    lib.egr_pack_wstream_f32.argtypes = [vp, i32, i32, vp, vp, vp]
    lib.egr_linear_wstream_workspace_bytes.restype = C.c_int64
    lib.egr_linear_wstream_workspace_bytes.argtypes = [i32, i32, i32]
    lib.egr_linear_wstream_f32.argtypes = [vp, i64, i32, i32, vp, vp, vp, i32, i32, vp, vp, i64, vp, vp, i64, vp]
    lib.egr_maxpool_nhwc_f32.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.egr_upsample2x_nhwc_f32.argtypes = [vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.egr_avgpool_nhwc_f32.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.egr_argmax_rows_f32.argtypes = [vp, i32, i32, i32, f32, vp, vp, vp, vp, vp]
    lib.egr_layernorm_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, f32, i32, vp]
    lib.egr_joint_mha_f32.argtypes = [vp, vp, i32, i32, i32, i32, f32, vp]
    lib.egr_msda_gather_f32.argtypes = [vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp]
    lib.egr_fisheye_project_f32.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp]
    lib.egr_fisheye_project2_f32.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]
    lib.egr_linear_smallk_f32.argtypes = [vp, i64, i64, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.egr_jqa_sum_f32.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.egr_tokens_to_nhwc_f32.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.egr_version.restype = C.c_char_p
    lib.egr_w6_elems.restype = C.c_int64
    lib.egr_device_arch.argtypes = [C.c_char_p, i32]
    lib.egr_conv_force_config.argtypes = [i32]
    lib.egr_conv_set_persist.argtypes = [i32, i32]
    lib.egr_up2_relu_head_f32.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32, vp, i32, i64, i64, i32, i64, vp]
    lib.egr_conv2d_masked_f32.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.egr_conv2d_masked_ex_f32.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, C.c_size_t, C.POINTER(ConvAux), vp]
    lib.egr_conv2d_wgrad_f32.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, C.c_size_t, i32, vp]
    lib.egr_conv2d_wgrad_ex_f32.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, C.c_size_t, i32, vp, vp, vp]
    lib.egr_gt_heatmap_f32.argtypes = [vp, i32, C.c_double, i32, i32, vp, vp, vp]
    lib.egr_pose_metrics_f32.argtypes = [vp, vp, i32, i32, f32, i32, vp, vp, vp]
    lib.egr_pack_w6_f32.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.egr_pack_w6_many_f32.argtypes = [vp, i32, i64, vp]
    lib.egr_pack_wh2_many_f32.argtypes = [vp, i32, i64, i64, vp]
    lib.egr_w6_elems.argtypes = [i32, i32]
    lib.egr_msda_fwd_f32.argtypes = [vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, i32, vp, vp]
    lib.egr_msda_bwd_f32.argtypes = [vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.egr_joint_layer_f32.argtypes = [C.POINTER(LayerDesc), vp]
    lib.egr_jqa_query_f32.argtypes = [C.POINTER(JqaQueryDesc), vp]
    lib.egr_pose_query_f32.argtypes = [C.POINTER(PoseQueryDesc), vp]
    lib.egr_pack_layer_w_f32.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.egr_pack_layer_wh2_f32.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.egr_preprocess_fused_u8_f32.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    lib.egr_preprocess_band_rows.argtypes = [vp, i32]
    lib.egr_preprocess_u8_f32.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    for name in EXPORTS:
        getattr(lib, name)  # fail at import if a symbol is missing
        if name not in ("egr_version", "egr_w6_elems", "egr_stem_w6_bytes", "egr_wh2_elems", "egr_stem_wh2_bytes", "egr_wstream_image_bytes",
                        "egr_linear_wstream_workspace_bytes"):
            getattr(lib, name).restype = C.c_int
    return lib


lib = _load()

_ERR = {-1: "EGR_EINVAL (unsupported shape/alignment)", -2: "EGR_ENULL (missing pointer)", -3: "EGR_EWORKSPACE"}


class LaunchError(RuntimeError):
    """A C-ABI entry point returned non-zero: `code` is its return value (-1 EGR_EINVAL, -2 EGR_ENULL, -3 EGR_EWORKSPACE, > 0 a
    hipError_t), so that a caller with a second way to run the launch can tell "this shape is not supported here" from a device fault."""

    def __init__(self, name: str, code: int):
        super().__init__(f"{name} failed: {_ERR.get(code, 'hipError_t %d' % code)}")
        self.name, self.code = name, code


EINVAL, ENULL, EWORKSPACE = -1, -2, -3


def _check(rc: int, name: str):
    if rc != 0:
        raise LaunchError(name, rc)


# When PROFILE is a list, every launch is bracketed by HIP events recorded on the launch stream (torch's current
# stream) and (kernel name, start, end, algorithmic work) is appended; bench.py uses this for the roofline leg.
PROFILE = None


# Device of the launch being assembled: set by the first tensor handed to _p(), checked for every further one, consumed by
# _launch().  The kernel goes to the current stream OF THAT DEVICE (not of torch's current device), and the C entry point is called
# with that device current - a module on cuda:1 works while torch.cuda.current_device() is 0 (two replicas in one process, a caller
# that never called set_device).
# Per Python thread (two replicas may be driven by two threads: the ctypes call releases the GIL between _p() and _launch());
# every path out of a launch - success, a host-side validation error, a failed call - leaves it cleared.
class _Dev(threading.local):
    idx = None


_DEVL = _Dev()


class _DevSlot:
    """_DEV[0] of earlier versions, now backed by the thread-local record."""

    def __getitem__(self, _i):
        return _DEVL.idx

    def __setitem__(self, _i, v):
        _DEVL.idx = v


_DEV = _DevSlot()


def _launch(name: str, cfunc, *args, flops: float = 0.0, nbytes: float = 0.0, tag: str = ""):
    idx, _DEV[0] = _DEV[0], None
    if idx is not None and idx != torch.cuda.current_device():
        with torch.cuda.device(idx):
            return _launch_here(name, cfunc, args, flops, nbytes, tag)
    return _launch_here(name, cfunc, args, flops, nbytes, tag)


def _launch_here(name, cfunc, args, flops, nbytes, tag):
    if PROFILE is None:
        rc = cfunc(*args)
    else:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = cfunc(*args)
        e.record()
        PROFILE.append((name, s, e, flops, nbytes, tag))
    _check(rc, name)


def _p(t: Optional[torch.Tensor], dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        _DEV[0] = None
        raise RuntimeError("egorear_amd: tensor is not on a HIP device (no CPU path exists)")
    if t.dtype != dtype:
        _DEV[0] = None
        raise RuntimeError(f"egorear_amd: expected {dtype}, got {t.dtype}")
    idx = t.device.index
    if _DEV[0] is None:
        _DEV[0] = idx
    elif _DEV[0] != idx:
        first, _DEV[0] = _DEV[0], None
        raise RuntimeError(f"egorear_amd: operands of one launch live on different devices (cuda:{first} and cuda:{idx})")
    return C.c_void_p(t.data_ptr())


def _pv(t: Optional[torch.Tensor]):
    """Raw device pointer of a tensor of any dtype (same device bookkeeping as _p)."""
    return None if t is None else _p(t, t.dtype)


def _stream():
    """Current stream of the device the launch's tensors live on (falls back to torch's current device for tensor-less calls)."""
    idx = _DEV[0] if _DEV[0] is not None else torch.cuda.current_device()
    return C.c_void_p(torch.cuda.current_stream(idx).cuda_stream)


def _cont(t: torch.Tensor, what: str):
    if not t.is_contiguous():
        raise RuntimeError(f"egorear_amd: {what} must be contiguous")
    return t


def version() -> str:
    return lib.egr_version().decode()


def conv_force_config(cfg: int):
    _check(lib.egr_conv_force_config(cfg), "egr_conv_force_config")


def conv_set_persist(slots: int = -1, max_ktiles: int = -1):
    _check(lib.egr_conv_set_persist(slots, max_ktiles), "egr_conv_set_persist")


def device_arch() -> str:
    buf = C.create_string_buffer(256)
    _check(lib.egr_device_arch(buf, 256), "egr_device_arch")
    return buf.value.decode()


# --------------------------------------------------------------------------- NHWC views

class Img:
    """A batch of channels-last images living (possibly as a channel slice) in a torch buffer.
    t: (N, H, W, C) with stride(3) == 1 and dense H/W (stride(1) == W*ld, stride(2) == ld)."""
    __slots__ = ("t", "n", "h", "w", "c", "ld", "nstride", "amax")

    def __init__(self, t: torch.Tensor, amax: Optional[torch.Tensor] = None):
        if t.dim() != 4 or t.stride(3) != 1:
            raise RuntimeError("egorear_amd: expected a channels-last (N,H,W,C) tensor")
        n, h, w, c = t.shape
        ld = t.stride(2) if w > 1 else (t.stride(1) if h > 1 else max(c, 1))
        if (w > 1 and h > 1 and t.stride(1) != w * ld):
            raise RuntimeError("egorear_amd: image rows must be dense")
        self.t, self.n, self.h, self.w, self.c, self.ld = t, n, h, w, c, ld
        self.nstride = t.stride(0) if n > 1 else h * w * ld
        # abs-max record of the tensor (64 uint32 slots, see egr_conv2d_nhwc_ex_f32) or None: left by the launch that produced it,
        # carried by the tensor object, read by a consuming EGR_W_F16X2 launch
        self.amax = amax if amax is not None else getattr(t, "_egr_amax", None)

    def tag(self, amax: Optional[torch.Tensor]):
        """Attach (or clear) the abs-max record: every launch that writes the tensor calls this."""
        self.amax = amax
        self.t._egr_amax = amax

    def nmap(self) -> NMap:
        return NMap(self.n, self.nstride, 0)


# --------------------------------------------------------------------------- ops

class W6:
    """Packed weights in the EGR_W_BF16X3 format (egr_pack_w6_f32): every fp32 weight as hi + mid + lo bf16, in fragment
    order.  Passed to conv2d in place of the packed fp32 matrix, it selects the bf16-matrix-core launch."""
    __slots__ = ("img", "npad", "K", "groups", "gstride", "f32", "used", "h2", "h2_ds", "h2_gstride", "h2_used")

    def __init__(self, img, npad, K, groups, gstride, f32=None):
        self.img, self.npad, self.K, self.groups, self.gstride = img, npad, K, groups, gstride
        self.f32 = f32      # the fp32 matrix it was made from: small launches stay on it (X6_MIN_ROWS / X6_MIN_FLOPS)
        self.used = False   # set by the first launch that takes the image (the training step only re-splits those)
        # EGR_W_F16X2 image of the same matrix (two fp16 planes of w * 2^k[co]) + the per-channel descale 2^-k[co]: taken by forward
        # launches whose input carries an abs-max record (add_wh2)
        self.h2 = self.h2_ds = None
        self.h2_gstride = 0
        self.h2_used = False    # set by the first launch that takes the fp16 image

    @property
    def shape(self):
        return (self.groups, self.npad, self.K) if self.groups > 1 else (self.npad, self.K)


def pack_w6(w: torch.Tensor) -> W6:
    """w: packed fp32 weights (npad, K) or (groups, npad, K) as conv2d takes them."""
    _cont(w, "packed weight")
    groups = w.shape[0] if w.dim() == 3 else 1
    npad, K = int(w.shape[-2]), int(w.shape[-1])
    n = int(lib.egr_w6_elems(npad, K))
    if n == 0:
        raise RuntimeError(f"egorear_amd.pack_w6: unsupported weight shape {tuple(w.shape)} (rows and K must be multiples of 32)")
    img = torch.empty(groups * n, device=w.device, dtype=torch.bfloat16)
    _launch("egr_pack_w6_f32", lib.egr_pack_w6_f32, _p(w), npad, K, groups, _p(img, torch.bfloat16), _stream())
    return W6(img, npad, K, groups, n, w)


def add_wh2(w6: W6) -> W6:
    """Give a W6 operand its EGR_W_F16X2 image (egr_pack_wh2_f32), made from the fp32 matrix it keeps."""
    if w6.f32 is None:
        raise RuntimeError("egorear_amd.add_wh2: the operand does not keep its fp32 matrix")
    n = int(lib.egr_wh2_elems(w6.npad, w6.K))
    img = torch.empty(w6.groups * n, device=w6.f32.device, dtype=torch.float16)
    ds = torch.empty(w6.groups * w6.npad, device=w6.f32.device, dtype=torch.float32)
    _launch("egr_pack_wh2_f32", lib.egr_pack_wh2_f32, _p(w6.f32), w6.npad, w6.K, w6.groups, _p(img, torch.float16), _p(ds), _stream())
    w6.h2, w6.h2_ds, w6.h2_gstride = img, ds, n
    return w6


ARENA_EXHAUSTED = 0      # over every AmaxArena of the process (bench.py reports it: a non-zero count means silent fallbacks)


class AmaxArena:
    """Abs-max records of one forward: 64 uint32 slots per recorded tensor in ONE buffer, zeroed by one launch when the forward
    begins (egr_fill_f32: all-zero bits), handed out in launch order.  A graph replay re-runs the fill and every producer."""

    def __init__(self, device, records: int = 256):
        self.buf = torch.zeros(records * 64, device=device, dtype=torch.int32)
        self.records, self.k = records, 0
        self.exhausted = 0        # requests refused since construction: each one is a launch that left the fp16 scheme

    def begin(self):
        self.k = 0
        _launch("egr_fill_f32", lib.egr_fill_f32, _p(self.buf, torch.int32), 0.0, self.buf.numel(), _stream())

    def new(self) -> Optional[torch.Tensor]:
        global ARENA_EXHAUSTED
        if self.k >= self.records:
            self.exhausted += 1
            ARENA_EXHAUSTED += 1
            return None
        self.k += 1
        return self.buf[(self.k - 1) * 64:self.k * 64]


def absmax_record(t: torch.Tensor, record: torch.Tensor) -> torch.Tensor:
    """Fold max |t| of a dense fp32 tensor into `record` (64 int32 slots, e.g. AmaxArena.new()) and tag the tensor with it: for
    tensors that enter the path from outside, so that their consumers can take the fp16 scheme."""
    _cont(t, "tensor")
    if record.numel() != 64 or record.dtype != torch.int32:
        raise RuntimeError("egorear_amd.absmax_record: the record is 64 int32 slots")
    _launch("egr_absmax_f32", lib.egr_absmax_f32, _p(t), t.numel(), _p(record, torch.int32), _stream(), nbytes=4.0 * t.numel())
    t._egr_amax = record
    return record


def pack_w6_into(w6: W6) -> None:
    """Refresh an image from its fp32 matrix (w6.f32) in place: the training step does this after every parameter update."""
    _launch("egr_pack_w6_f32", lib.egr_pack_w6_f32, _p(w6.f32), w6.npad, w6.K, w6.groups, _p(w6.img, torch.bfloat16), _stream())


class W6Job(C.Structure):
    _fields_ = [("w", C.c_void_p), ("img", C.c_void_p), ("npad", C.c_int32), ("k", C.c_int32), ("groups", C.c_int32), ("reserved", C.c_int32),
                ("first_block", C.c_int64)]


class W6Table:
    """Device table for egr_pack_w6_many_f32: every image of `images` re-split from its fp32 matrix in ONE launch."""

    def __init__(self, images):
        self.images = list(images)
        jobs = (W6Job * max(1, len(self.images)))()
        first = 0
        for i, w6 in enumerate(self.images):
            jobs[i] = W6Job(w6.f32.data_ptr(), w6.img.data_ptr(), w6.npad, w6.K, w6.groups, 0, first)
            first += w6.groups * ((w6.npad // 32 + 3) // 4 * 4) * (w6.K // 32)
        self.total = first
        raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).clone()
        self.dev = raw.to(self.images[0].img.device) if self.images else None

    def run(self):
        if self.images:
            _launch("egr_pack_w6_f32", lib.egr_pack_w6_many_f32, _p(self.dev, torch.uint8), len(self.images), self.total, _stream())


class WH2Job(C.Structure):
    _fields_ = [("w", C.c_void_p), ("img", C.c_void_p), ("descale", C.c_void_p), ("npad", C.c_int32), ("k", C.c_int32), ("groups", C.c_int32),
                ("reserved", C.c_int32), ("first_rblock", C.c_int64), ("first_pblock", C.c_int64)]


class WH2Table:
    """Device table for egr_pack_wh2_many_f32: the fp16 image (and descale) of every operand of `images` re-made from its fp32 matrix in
    two launches."""

    def __init__(self, images):
        self.images = list(images)
        jobs = (WH2Job * max(1, len(self.images)))()
        fr = fp = 0
        for i, w6 in enumerate(self.images):
            jobs[i] = WH2Job(w6.f32.data_ptr(), w6.h2.data_ptr(), w6.h2_ds.data_ptr(), w6.npad, w6.K, w6.groups, 0, fr, fp)
            fr += (w6.groups * w6.npad + 3) // 4
            fp += w6.groups * ((w6.npad // 32 + 3) // 4 * 4) * (w6.K // 32)
        self.rblocks, self.pblocks = fr, fp
        raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).clone()
        self.dev = raw.to(self.images[0].h2.device) if self.images else None

    def run(self):
        if self.images:
            _launch("egr_pack_wh2_f32", lib.egr_pack_wh2_many_f32, _p(self.dev, torch.uint8), len(self.images), self.rblocks, self.pblocks, _stream())


def pack_wh2_into(w6: W6) -> None:
    """Refresh the fp16 image of an operand from its fp32 matrix in place."""
    _launch("egr_pack_wh2_f32", lib.egr_pack_wh2_f32, _p(w6.f32), w6.npad, w6.K, w6.groups, _p(w6.h2, torch.float16), _p(w6.h2_ds), _stream())


def conv2d(x: Img, w, cout: int, kh: int, kw: int, stride: int, pad: int, *, scale=None, shift=None,
           act: int = ACT_NONE, res: Optional[Img] = None, res_mode: int = RES_NONE, rowscale=None, rowmask=None,
           out: Optional[Img] = None, out_nchw: Optional[torch.Tensor] = None, ymap: Optional[NMap] = None,
           xmap: Optional[NMap] = None, rmap: Optional[NMap] = None, workspace: Optional[torch.Tensor] = None,
           split_k: int = 1, groups: int = 1, gx: Optional[int] = None, gy: Optional[int] = None,
           gr: Optional[int] = None, grs: int = 0, grm: int = 0, transposed_out_hw: Optional[tuple] = None, x6_min: Optional[tuple] = None,
           mask: Optional[Img] = None, amax_out: Optional[torch.Tensor] = None, amax_arena: Optional["AmaxArena"] = None,
           bn_ws: Optional[torch.Tensor] = None, bn_slabs: Optional[list] = None) -> Optional[Img]:
    """Implicit-GEMM conv / linear.  Output goes to `out` (NHWC Img, maybe a channel slice), or to the raw
    tensor `out_nchw` (channel-major planes placed by `ymap`), or to a fresh NHWC tensor.

    groups > 1: `groups` same-shape problems in one launch.  w is (groups, cout_pad, K), scale/shift
    (groups, cout_pad).  x / out / res hold the images of all groups back to back (group stride = images per group x
    image stride) unless an explicit element stride gx / gy / gr is given, in which case they describe group 0."""
    pol = policy()
    x_full = x.t
    if groups > 1:
        if gx is None:
            if x.n % groups:
                raise RuntimeError("egorear_amd.conv2d: images not divisible by groups")
            ng = x.n // groups
            gx = ng * x.nstride
            x = Img(x.t[:ng], amax=x.amax)
        ng = x.n
    if transposed_out_hw is not None:   # data gradient: x is dy, the output is dx of the given spatial size
        ho, wo = transposed_out_hw
        if (ho + 2 * pad - kh) // stride + 1 != x.h or (wo + 2 * pad - kw) // stride + 1 != x.w:
            raise RuntimeError("egorear_amd.conv2d: transposed_out_hw inconsistent with dy's size")
    else:
        ho = (x.h + 2 * pad - kh) // stride + 1
        wo = (x.w + 2 * pad - kw) // stride + 1
    npad = (cout + 31) // 32 * 32
    K = kh * kw * x.c
    wshape = (groups, npad, K) if groups > 1 else (npad, K)
    if tuple(w.shape) != wshape:
        raise RuntimeError(f"egorear_amd.conv2d: packed weight shape {tuple(w.shape)} != {wshape}")
    if isinstance(w, W6) and w.f32 is not None:
        rows_all = x.n * groups * ho * wo
        # (the split kernel addresses one group's activations through a 2-GiB buffer window)
        x_bytes = 4 * ((x.n - 1) * x.nstride + (x.h * x.w + 2 * (kh * x.w + kw + 1)) * x.ld) + 64
        min_rows, min_flops = x6_min if x6_min is not None else (pol.x6_min_rows, pol.x6_min_flops)
        if rows_all < min_rows or 2.0 * rows_all * cout * K < min_flops or x_bytes >= (1 << 31):
            w = w.f32
    x6 = isinstance(w, W6)
    if x6 and pol.h2 and w.h2 is not None and x.amax is None and amax_arena is not None and x_full.is_contiguous():
        # an input without a record (it comes out of a launch that keeps none): one read of it makes one (the training step)
        rec = amax_arena.new()
        if rec is not None:
            absmax_record(x_full, rec)
            x.amax = rec
    # the fp16 scheme: launches whose input carries its abs-max record (the launch's pre-scale comes from it)
    h2 = x6 and pol.h2 and w.h2 is not None and x.amax is not None
    if x6 and not h2 and not w.used:
        if w.f32 is not None:
            pack_w6_into(w)     # an owner that only re-splits the images in use (the training step) may have left this one stale
        w.used = True
    if h2 and not w.h2_used:
        if w.f32 is not None:
            pack_wh2_into(w)
        w.h2_used = True
    wptr = (_p(w.h2, torch.float16) if h2 else _p(w.img, torch.bfloat16)) if x6 else _p(_cont(w, "packed weight"))
    d = ConvDesc()
    d.w_format = (4 if h2 else 1) if x6 else 0
    d.n, d.h, d.w, d.cin, d.cout = x.n, x.h, x.w, x.c, cout
    d.kh, d.kw, d.stride, d.pad, d.ho, d.wo = kh, kw, stride, pad, ho, wo
    d.ldx = x.ld
    d.xmap = xmap if xmap is not None else x.nmap()
    d.act, d.res_mode, d.split_k = act, res_mode, split_k
    d.transposed = 1 if transposed_out_hw is not None else 0
    d.groups, d.gx, d.gw, d.gp, d.grs, d.grm = groups, (gx or 0), (((w.h2_gstride if h2 else w.gstride) if x6 else npad * K) if groups > 1 else 0), (npad if groups > 1 else 0), grs, grm
    ret = None
    if out_nchw is not None:
        if ymap is None or (groups > 1 and gy is None):
            raise RuntimeError("egorear_amd.conv2d: out_nchw needs ymap (and gy when grouped)")
        d.out_nchw, d.ldy, d.ymap, d.gy = 1, 0, ymap, (gy or 0)
        yptr = _p(out_nchw)
    else:
        full = out
        if out is None:
            full = Img(torch.empty((groups * x.n, ho, wo, cout), device=x.t.device, dtype=torch.float32))
            out = full
        if groups > 1 and gy is None:
            if out.n != groups * x.n:
                raise RuntimeError("egorear_amd.conv2d: grouped output must hold groups x n images")
            gy = x.n * out.nstride
            out = Img(out.t[:x.n])
        if (out.n, out.h, out.w, out.c) != (x.n, ho, wo, cout) and ymap is None:
            raise RuntimeError(f"egorear_amd.conv2d: output shape {(out.n, out.h, out.w, out.c)} != {(x.n, ho, wo, cout)}")
        d.out_nchw, d.ldy, d.gy = 0, out.ld, (gy or 0)
        d.ymap = ymap if ymap is not None else out.nmap()
        yptr = _p(out.t)
        ret = full
    if res_mode != RES_NONE:
        if res is None:
            raise RuntimeError("egorear_amd.conv2d: res_mode set without res")
        if groups > 1 and gr is None:
            if res.n != groups * x.n:
                raise RuntimeError("egorear_amd.conv2d: grouped residual must hold groups x n images")
            gr = x.n * res.nstride
            res = Img(res.t[:x.n])
        if res_mode == RES_UP2_BEFORE_ACT and (2 * res.h != ho or 2 * res.w != wo or res.c < cout):
            raise RuntimeError("egorear_amd.conv2d: the upsampled residual must be exactly half the output resolution")
        d.ldr, d.gr = res.ld, (gr or 0)
        d.rmap = rmap if rmap is not None else res.nmap()
    else:
        d.rmap = NMap(1, 0, 0)
    for t, nm in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() < (groups * npad if groups > 1 else cout):
            raise RuntimeError(f"egorear_amd.conv2d: {nm} shorter than expected")
    M = x.n * ho * wo
    if rowscale is not None and rowscale.numel() < (groups - 1) * grs + M or rowmask is not None and rowmask.numel() < (groups - 1) * grm + M:
        raise RuntimeError("egorear_amd.conv2d: rowscale/rowmask shorter than M")
    ws_ptr, ws_n = (None, 0) if workspace is None else (_p(workspace), workspace.numel())
    if mask is not None:   # output *= [mask > 0]; mask has the output's dense layout
        if out_nchw is not None or scale is not None or shift is not None or rowscale is not None or rowmask is not None or act != ACT_NONE:
            raise RuntimeError("egorear_amd.conv2d: mask goes with a plain NHWC data gradient only")
        if (mask.n, mask.h, mask.w, mask.c) != (groups * x.n, ho, wo, cout) or not mask.t.is_contiguous() or ret is None or not ret.t.is_contiguous():
            raise RuntimeError("egorear_amd.conv2d: mask must be dense and shaped like the output")
        if amax_out is not None and (amax_out.numel() != 64 or amax_out.dtype != torch.int32 or not amax_out.is_contiguous()):
            raise RuntimeError("egorear_amd.conv2d: amax_out must be 64 contiguous int32 slots")
        aux = None
        if h2 or amax_out is not None:
            aux = ConvAux(_p(w.h2_ds).value if h2 else None, _p(x.amax, torch.int32).value if h2 else None,
                          _p(amax_out, torch.int32).value if amax_out is not None else None)
        ret.tag(amax_out)
        x6 = x6 and not h2     # (the tag below)
        _launch("egr_conv2d_nhwc_f32", lib.egr_conv2d_masked_ex_f32, C.byref(d), _p(x.t), wptr, _p(res.t) if res is not None else None,
                _p(mask.t), yptr, ws_ptr, ws_n, C.byref(aux) if aux is not None else None, _stream(),
                flops=2.0 * M * cout * K * groups / (stride * stride if transposed_out_hw is not None else 1),
                nbytes=4.0 * groups * (2 * M * cout + x.n * x.h * x.w * x.c + cout * K),
                tag=f"{'T ' if transposed_out_hw is not None else ''}{'h2 ' if h2 else ''}{'x6 ' if x6 else ''}masked G{groups} M{M} N{cout} K{K} k{kh}s{stride} cin{x.c}" if PROFILE is not None else "")
        return ret
    if out_nchw is not None:
        amax_out = None          # (the channel-major epilogue keeps no record)
    elif amax_out is not None and (amax_out.numel() != 64 or amax_out.dtype != torch.int32 or not amax_out.is_contiguous()):
        raise RuntimeError("egorear_amd.conv2d: amax_out must be 64 contiguous int32 slots")
    aux = None
    tiles = None
    if h2 or amax_out is not None or bn_ws is not None:
        aux = ConvAux(_p(w.h2_ds).value if h2 else None, _p(x.amax, torch.int32).value if h2 else None,
                      _p(amax_out, torch.int32).value if amax_out is not None else None, None, None, 0)
        if bn_ws is not None:
            # train-mode BatchNorm behind this conv: the launch leaves the per-tile channel statistics in `bn_ws` (float64 workspace of
            # hip_train.bn_workspace) and reports its slab count in bn_slabs[0] (egr_conv_aux.bn_partials)
            if bn_ws.dtype != torch.float64 or bn_slabs is None:
                raise RuntimeError("egorear_amd.conv2d: bn_ws is the float64 BatchNorm workspace, bn_slabs a list receiving the slab count")
            tiles = C.c_int32(0)
            aux.bn_partials = _p(bn_ws, torch.float64).value
            aux.bn_tiles_out = C.addressof(tiles)
            aux.bn_capacity = bn_ws.numel()
    fmt = "h2 " if h2 else ("x6 " if x6 else "")
    _launch("egr_conv2d_nhwc_f32", lib.egr_conv2d_nhwc_ex_f32, C.byref(d), _p(x.t), wptr, _p(scale), _p(shift),
            _p(res.t) if res is not None else None, _p(rowscale), _p(rowmask, torch.uint8), yptr, ws_ptr, ws_n,
            C.byref(aux) if aux is not None else None, _stream(),
            flops=2.0 * M * cout * K * groups / (stride * stride if transposed_out_hw is not None else 1), nbytes=4.0 * groups * (M * cout + x.n * x.h * x.w * x.c + cout * K),
            tag=f"{'T ' if transposed_out_hw is not None else ''}{fmt}G{groups} M{M} N{cout} K{K} k{kh}s{stride} cin{x.c}" if PROFILE is not None else "")
    if ret is not None:
        ret.tag(amax_out)      # (also clears a stale record when this launch keeps none)
    if tiles is not None:
        bn_slabs[:] = [int(tiles.value)]
    return ret


def chain_eligible(x: Img, w1, w2, cmid: int, cout: int, groups: int, scale1=None, scale2=None, res_mode: int = 0) -> bool:
    """Whether conv1x1_chain covers this pair: fp16 scheme on both operands, x carries its abs-max record, cin 64 / 128 -> 128 -> <= 128,
    bias-only convs, enough pixels for the streaming structure (the rule of the single streaming launches)."""
    pol = policy()
    if not (pol.chain and pol.h2 and isinstance(w1, W6) and isinstance(w2, W6) and w1.h2 is not None and w2.h2 is not None):
        return False
    if x.amax is None or scale1 is not None or scale2 is not None or cout > 128 or cout % 4:
        return False
    big = x.c == 256 and cmid == 256        # both matrices streamed through LDS (conv_pw2_kernel: the heat-map heads' 256 -> 256 -> 128)
    if not big and (x.c not in (64, 128) or cmid != 128):
        return False
    if big and (not pol.chain_big or res_mode != RES_NONE):
        return False
    if x.n % groups or w1.groups != groups or w2.groups != groups or w1.npad != cmid or w2.npad != 128 or w1.K != x.c or w2.K != cmid:
        return False
    # the entry's own layout checks (egr_conv1x1_chain_f32 returns EINVAL on them): 16-byte pixel rows and image strides, a 16-byte
    # aligned base - a channel slice at an odd offset stays on the single launches instead of aborting the forward
    if x.ld % 4 or x.nstride % 4 or x.t.data_ptr() % 16:
        return False
    return x.n * x.h * x.w >= (pol.chain_big_min_rows if big else pol.chain_min_rows)


def conv1x1_chain(x: Img, w1, w2, cmid: int, cout: int, *, shift1=None, shift2=None, act1: int = ACT_RELU, act2: int = ACT_NONE,
                  res: Optional[Img] = None, res_mode: int = RES_NONE, out: Optional[Img] = None, groups: int = 1,
                  amax_out: Optional[torch.Tensor] = None) -> Img:
    """y = act2(W2 . act1(W1 . x + shift1) + shift2 [+ res]) in one launch; the cmid-channel intermediate never reaches memory.
    x / out / res hold the images of all groups back to back (as hip.conv2d).  Call chain_eligible first."""
    if not chain_eligible(x, w1, w2, cmid, cout, groups, res_mode=res_mode):
        raise LaunchError("egr_conv1x1_chain_f32", EINVAL)
    ng = x.n // groups
    gx = ng * x.nstride if groups > 1 else 0
    xg = Img(x.t[:ng], amax=x.amax) if groups > 1 else x
    for w in (w1, w2):
        if not w.h2_used:
            if w.f32 is not None:
                pack_wh2_into(w)
            w.h2_used = True
    d = ConvDesc()
    d.w_format = 4
    d.n, d.h, d.w, d.cin, d.cout = xg.n, xg.h, xg.w, xg.c, cout
    d.kh, d.kw, d.stride, d.pad, d.ho, d.wo = 1, 1, 1, 0, xg.h, xg.w
    d.ldx, d.xmap = xg.ld, xg.nmap()
    d.act, d.res_mode, d.split_k, d.transposed, d.out_nchw = act2, res_mode, 1, 0, 0
    full = out
    if out is None:
        full = Img(torch.empty((x.n, x.h, x.w, cout), device=x.t.device, dtype=torch.float32))
        out = full
    if out.n != x.n or (out.h, out.w, out.c) != (x.h, x.w, cout):
        raise RuntimeError("egorear_amd.conv1x1_chain: output shape does not match")
    gy = ng * out.nstride if groups > 1 else 0
    og = Img(out.t[:ng]) if groups > 1 else out
    d.ldy, d.ymap = og.ld, og.nmap()
    gr = 0
    if res_mode != RES_NONE:
        if res is None or res.n != x.n:
            raise RuntimeError("egorear_amd.conv1x1_chain: res_mode set without a residual of groups x n images")
        if res_mode == RES_UP2_BEFORE_ACT and (2 * res.h != x.h or 2 * res.w != x.w or res.c < cout):
            raise RuntimeError("egorear_amd.conv1x1_chain: the upsampled residual must be exactly half the output resolution")
        gr = ng * res.nstride if groups > 1 else 0
        rg = Img(res.t[:ng]) if groups > 1 else res
        d.ldr, d.rmap = rg.ld, rg.nmap()
    else:
        d.rmap = NMap(1, 0, 0)
    d.groups, d.gx, d.gw, d.gp, d.gy, d.gr = groups, gx, (w1.h2_gstride if groups > 1 else 0), (128 if groups > 1 else 0), gy, gr
    for t, nm, need, gs in ((shift1, "shift1", cmid, w1.npad), (shift2, "shift2", cout, 128)):
        if t is not None and t.numel() < ((groups - 1) * gs + need):
            raise RuntimeError(f"egorear_amd.conv1x1_chain: {nm} shorter than expected")
    if amax_out is not None and (amax_out.numel() != 64 or amax_out.dtype != torch.int32 or not amax_out.is_contiguous()):
        raise RuntimeError("egorear_amd.conv1x1_chain: amax_out must be 64 contiguous int32 slots")
    aux = ConvAux(_p(w1.h2_ds).value, _p(x.amax, torch.int32).value, _p(amax_out, torch.int32).value if amax_out is not None else None, None, None, 0)
    ch = ChainAux(_p(w2.h2, torch.float16).value, _p(w2.h2_ds).value, _p(shift1).value if shift1 is not None else None, cmid, act1,
                  (w2.h2_gstride if groups > 1 else 0), (w1.npad if groups > 1 else 0))
    M = ng * x.h * x.w
    _launch("egr_conv1x1_chain_f32", lib.egr_conv1x1_chain_f32, C.byref(d), _p(xg.t), _p(w1.h2, torch.float16), _p(shift2),
            _p(res.t) if res is not None else None, _p(og.t), C.byref(aux), C.byref(ch), _stream(),
            flops=2.0 * M * groups * (cmid * x.c + cout * cmid), nbytes=4.0 * groups * (M * (cout + x.c) + cmid * x.c + cout * cmid),
            tag=f"h2 chain G{groups} M{M} N{cout} K{x.c} k1s1 cin{x.c} mid{cmid}" if PROFILE is not None else "")
    full.tag(amax_out)
    return full


def wgrad_is_split(x: Img, dy: Img, kh: int, kw: int, groups: int = 1) -> bool:
    """Whether conv2d_wgrad would run this problem on the split kernels (the size rule of egr_conv2d_wgrad_f32, or the forced mode)."""
    if not policy().wgrad_x6:
        return False
    if policy().wgrad_force:
        return True
    M = (x.n // groups) * dy.h * dy.w
    return M >= 1024 and 2.0 * M * dy.c * kh * kw * x.c * groups >= 4e9


def wgrad_records(x: Img, dy: Img, amax_arena: Optional["AmaxArena"]) -> None:
    """Abs-max records for the operands of a split weight-gradient launch that carry none (one read each), on the current stream."""
    if not policy().h2 or amax_arena is None:
        return
    for im in (x, dy):
        if im.amax is None and im.t.is_contiguous():
            rec = amax_arena.new()
            if rec is not None:
                absmax_record(im.t, rec)
                im.amax = rec


def conv2d_wgrad(x: Img, dy: Img, kh: int, kw: int, stride: int, pad: int, workspace: torch.Tensor, *, want_bias: bool = False,
                 dw: Optional[torch.Tensor] = None, db: Optional[torch.Tensor] = None, accumulate: bool = False, groups: int = 1,
                 x6: Optional[bool] = None, amax_arena: Optional["AmaxArena"] = None, gx: Optional[int] = None, gy: Optional[int] = None):
    """Weight (+ bias) gradient of the forward conv x -> y.  x, dy NHWC Imgs (all groups' images back to back when
    groups > 1).  Returns (dw, db): dw ([groups,] cout, kh*kw*cin) in the packed K order of conv2d
    (engine.unpack_conv_weight turns it back into OIHW), db ([groups,] cout)."""
    cin, cout = x.c, dy.c
    interleaved = gx is not None or gy is not None      # groups = channel slices of ONE batch (x, dy: group 0's slice; gx / gy: element strides)
    if interleaved and (gx is None or gy is None):
        raise RuntimeError("egorear_amd.conv2d_wgrad: gx and gy go together")
    if dy.n != x.n or (not interleaved and x.n % groups):
        raise RuntimeError("egorear_amd.conv2d_wgrad: image counts do not match / not divisible by groups")
    ng = x.n if interleaved else x.n // groups
    if ((x.h + 2 * pad - kh) // stride + 1, (x.w + 2 * pad - kw) // stride + 1) != (dy.h, dy.w):
        raise RuntimeError("egorear_amd.conv2d_wgrad: dy does not match the forward output geometry")
    d = ConvDesc()
    d.n, d.h, d.w, d.cin, d.cout = ng, x.h, x.w, cin, cout
    d.kh, d.kw, d.stride, d.pad, d.ho, d.wo = kh, kw, stride, pad, dy.h, dy.w
    d.ldx, d.ldy, d.rmap = x.ld, dy.ld, NMap(1, 0, 0)
    d.xmap, d.ymap = NMap(ng, x.nstride, 0), NMap(ng, dy.nstride, 0)
    K = kh * kw * cin
    d.groups, d.gx, d.gy, d.gw, d.gp = groups, (gx if interleaved else ng * x.nstride), (gy if interleaved else ng * dy.nstride), cout * K, cout
    if x6 is None and policy().wgrad_force and policy().wgrad_x6:
        x6 = "force"
    d.w_format = 3 if x6 == "force" else (1 if (policy().wgrad_x6 if x6 is None else x6) else 0)   # "force": the split kernel at any size
    # the fp16 scheme: split launches whose two operands carry abs-max records (made here with one read each when an arena is given)
    M = ng * dy.h * dy.w
    split = d.w_format == 3 or (d.w_format == 1 and M >= 1024 and 2.0 * M * cout * K * groups >= 4e9)
    if split:
        wgrad_records(x, dy, amax_arena)
    h2 = split and policy().h2 and x.amax is not None and dy.amax is not None
    if h2:
        d.w_format |= 4
    shape_w, shape_b = ((groups, cout, K), (groups, cout)) if groups > 1 else ((cout, K), (cout,))
    if dw is None:
        dw = torch.empty(shape_w, device=x.t.device, dtype=torch.float32)
    elif dw.numel() != groups * cout * K:
        raise RuntimeError("egorear_amd.conv2d_wgrad: dw has the wrong size")
    if want_bias and db is None:
        db = torch.empty(shape_b, device=x.t.device, dtype=torch.float32)
    _launch("egr_conv2d_wgrad_f32", lib.egr_conv2d_wgrad_ex_f32, C.byref(d), _p(x.t), _p(dy.t), _p(_cont(dw, "dw")), _p(db),
            _p(workspace), workspace.numel(), 1 if accumulate else 0, _p(x.amax, torch.int32) if h2 else None,
            _p(dy.amax, torch.int32) if h2 else None, _stream(), flops=2.0 * dy.n * dy.h * dy.w * cout * K,
            tag=f"{'h2 ' if h2 else ''}G{groups} M{ng * dy.h * dy.w} N{cout} K{K} k{kh}s{stride} cin{cin}" if PROFILE is not None else "")
    return dw, db


def stem(img: torch.Tensor, view0: int, nviews: int, wpack, scale, shift, groups: int = 1) -> Img:
    """img (B, V, 3, H, W) NCHW contiguous.  Group g handles views [view0 + g*nviews, view0 + (g+1)*nviews) with its own
    weights (wpack (groups,64,148), scale/shift (groups,64)); output NHWC (groups*nviews*B, H/2, W/2, 64), view-major."""
    B, V, Cc, H, W = img.shape
    if Cc != 3:
        raise RuntimeError("egorear_amd.stem: 3-channel input expected")
    _cont(img, "input image batch")
    raw = scale is None and shift is None          # training mode: bare convolution
    if view0 + groups * nviews > V or wpack.numel() != groups * 64 * 148 or (not raw and (scale.numel() != groups * 64 or shift.numel() != groups * 64)):
        raise RuntimeError("egorear_amd.stem: views / weights mismatch")
    n = nviews * B
    y = torch.empty((groups * n, H // 2, W // 2, 64), device=img.device, dtype=torch.float32)
    base = img.reshape(-1)[view0 * 3 * H * W:]
    xmap = NMap(B, V * 3 * H * W, 3 * H * W)  # n = v*B + b -> (b*V + v) image
    _launch("egr_stem_conv7x7_f32", lib.egr_stem_conv7x7_f32, _p(base), xmap, n, H, W, _p(_cont(wpack, "stem weight")), _p(scale),
            _p(shift), _p(y), groups, nviews * 3 * H * W, _stream(),
            flops=2.0 * groups * n * (H // 2) * (W // 2) * 64 * 147, nbytes=4.0 * groups * n * (3 * H * W + (H // 2) * (W // 2) * 64))
    return Img(y)


def stem_pool(img: torch.Tensor, view0: int, nviews: int, wpack, scale, shift, groups: int = 1) -> Img:
    """stem() + maxpool(3, 2, 1) in one pass (eval mode): output NHWC (groups*nviews*B, H/4, W/4, 64)."""
    B, V, Cc, H, W = img.shape
    if Cc != 3:
        raise RuntimeError("egorear_amd.stem_pool: 3-channel input expected")
    _cont(img, "input image batch")
    if scale is None or shift is None:
        raise RuntimeError("egorear_amd.stem_pool: eval-mode BatchNorm scale / shift required")
    if view0 + groups * nviews > V or wpack.numel() != groups * 64 * 148 or scale.numel() != groups * 64 or shift.numel() != groups * 64:
        raise RuntimeError("egorear_amd.stem_pool: views / weights mismatch")
    n = nviews * B
    y = torch.empty((groups * n, H // 4, W // 4, 64), device=img.device, dtype=torch.float32)
    base = img.reshape(-1)[view0 * 3 * H * W:]
    xmap = NMap(B, V * 3 * H * W, 3 * H * W)
    _launch("egr_stem_conv7x7_pool_f32", lib.egr_stem_conv7x7_pool_f32, _p(base), xmap, n, H, W, _p(_cont(wpack, "stem weight")),
            _p(scale), _p(shift), _p(y), groups, nviews * 3 * H * W, _stream(),
            flops=2.0 * groups * n * (H // 2) * (W // 2) * 64 * 147, nbytes=4.0 * groups * n * (3 * H * W + (H // 4) * (W // 4) * 64))
    return Img(y)


def pack_stem_w6(wpack: torch.Tensor) -> torch.Tensor:
    """(groups, 64, 148) fp32 stem filters -> the split-bf16 bank of stem_x6 ((groups, egr_stem_w6_bytes()) uint8)."""
    if wpack.dim() != 3 or wpack.shape[1:] != (64, 148) or wpack.dtype != torch.float32:
        raise RuntimeError("egorear_amd.pack_stem_w6: (groups, 64, 148) fp32 expected")
    img = torch.empty((wpack.shape[0], int(lib.egr_stem_w6_bytes())), device=wpack.device, dtype=torch.uint8)
    _launch("egr_pack_stem_w6_f32", lib.egr_pack_stem_w6_f32, _p(_cont(wpack, "stem weight")), wpack.shape[0], _p(img, torch.uint8), _stream())
    return img


def pack_wstream(w: torch.Tensor):
    """(n, k) fp32 weights of a few-rows Linear -> (the weight-stream image (4 n k bytes) uint8, descale (n,)) (egr_pack_wstream_f32)."""
    if w.dim() != 2 or w.dtype != torch.float32 or w.shape[0] % 64 or w.shape[1] % 256:
        raise RuntimeError("egorear_amd.pack_wstream: (n % 64 == 0, k % 256 == 0) fp32 expected")
    n, k = w.shape
    img = torch.empty((int(lib.egr_wstream_image_bytes(n, k)),), device=w.device, dtype=torch.uint8)
    ds = torch.empty((n,), device=w.device, dtype=torch.float32)
    _launch("egr_pack_wstream_f32", lib.egr_pack_wstream_f32, _p(_cont(w, "weight")), n, k, _p(img, torch.uint8), _p(ds), _stream())
    return img, ds


def linear_wstream(x: torch.Tensor, img: torch.Tensor, descale: torch.Tensor, bias: Optional[torch.Tensor], act: int, workspace: torch.Tensor,
                   amax_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = act(x w^T + bias) for x (rows, k) with its abs-max record (x._egr_amax), w as pack_wstream's image: the weight-stream
    launch (egr_linear_wstream_f32).  More than 64 rows run as chunks of 64 (the weights are streamed once per chunk)."""
    rec = getattr(x, "_egr_amax", None)
    if rec is None:
        raise RuntimeError("egorear_amd.linear_wstream: the rows carry no abs-max record")
    n = descale.numel()
    rows, k = x.shape
    if x.dtype != torch.float32 or x.stride(1) != 1 or img.numel() != 4 * n * k:
        raise RuntimeError("egorear_amd.linear_wstream: operands do not match")
    y = torch.empty((rows, n), device=x.device, dtype=torch.float32)
    for r0 in range(0, rows, 64):
        r = min(64, rows - r0)
        need = int(lib.egr_linear_wstream_workspace_bytes(r, n, k))
        if need < 0 or workspace.numel() * workspace.element_size() < need:
            raise RuntimeError(f"egorear_amd.linear_wstream: workspace of {need} bytes needed")
        xr, yr = x[r0:r0 + r], y[r0:r0 + r]
        _launch("egr_linear_wstream_f32", lib.egr_linear_wstream_f32, _p(xr), xr.stride(0), r, k, _p(img, torch.uint8), _p(descale),
                _p(bias), n, act, _p(rec, torch.int32), _p(yr), yr.stride(0), _p(amax_out, torch.int32), _p(workspace, workspace.dtype),
                workspace.numel() * workspace.element_size(), _stream(), flops=2.0 * r * n * k, nbytes=4.0 * n * k + 4.0 * r * (n + k),
                tag="h2 wstream")
    if amax_out is not None:
        y._egr_amax = amax_out
    return y


def pack_stem_wh2(wpack: torch.Tensor):
    """(groups, 64, 148) fp32 stem filters -> (the fp16-scheme bank (groups, egr_stem_wh2_bytes()) uint8, its descale (groups, 64))."""
    if wpack.dim() != 3 or wpack.shape[1:] != (64, 148) or wpack.dtype != torch.float32:
        raise RuntimeError("egorear_amd.pack_stem_wh2: (groups, 64, 148) fp32 expected")
    img = torch.empty((wpack.shape[0], int(lib.egr_stem_wh2_bytes())), device=wpack.device, dtype=torch.uint8)
    ds = torch.empty((wpack.shape[0], 64), device=wpack.device, dtype=torch.float32)
    _launch("egr_pack_stem_wh2_f32", lib.egr_pack_stem_wh2_f32, _p(_cont(wpack, "stem weight")), wpack.shape[0], _p(img, torch.uint8), _p(ds), _stream())
    return img, ds


def stem_x6(img: torch.Tensor, view0: int, nviews: int, w6: torch.Tensor, scale, shift, groups: int = 1, pool: bool = False,
            amax_out: Optional[torch.Tensor] = None, w_descale: Optional[torch.Tensor] = None) -> Img:
    """stem() on the bf16 matrix cores (w6 from pack_stem_w6); pool=True: + maxpool(3, 2, 1) in the same pass (eval mode),
    output (groups*nviews*B, H/4, W/4, 64); amax_out (pool=True): the output's abs-max record (64 zeroed int32 slots)."""
    B, V, Cc, H, W = img.shape
    if Cc != 3:
        raise RuntimeError("egorear_amd.stem_x6: 3-channel input expected")
    _cont(img, "input image batch")
    raw = scale is None and shift is None
    if pool and raw:
        raise RuntimeError("egorear_amd.stem_x6: the fused max-pool needs the eval-mode BatchNorm scale / shift")
    h2 = w_descale is not None      # w6 is then the fp16-scheme bank of pack_stem_wh2
    if (view0 + groups * nviews > V or w6.dtype != torch.uint8
            or w6.numel() != groups * int(lib.egr_stem_wh2_bytes() if h2 else lib.egr_stem_w6_bytes())
            or (h2 and w_descale.numel() != groups * 64)
            or (not raw and (scale.numel() != groups * 64 or shift.numel() != groups * 64))):
        raise RuntimeError("egorear_amd.stem_x6: views / weights mismatch")
    n = nviews * B
    d = 4 if pool else 2
    y = torch.empty((groups * n, H // d, W // d, 64), device=img.device, dtype=torch.float32)
    base = img.reshape(-1)[view0 * 3 * H * W:]
    xmap = NMap(B, V * 3 * H * W, 3 * H * W)
    if amax_out is not None and (not pool or amax_out.numel() != 64 or amax_out.dtype != torch.int32):
        raise RuntimeError("egorear_amd.stem_x6: the abs-max record (64 int32 slots) goes with pool=True")
    if h2:
        _launch("egr_stem_conv7x7_x6_f32", lib.egr_stem_conv7x7_h2_f32, _p(base), xmap, n, H, W, _p(_cont(w6, "stem weight"), torch.uint8),
                _p(_cont(w_descale, "stem descale")), _p(scale), _p(shift), _p(y), 1 if pool else 0, groups, nviews * 3 * H * W,
                _p(amax_out, torch.int32), _stream(),
                flops=2.0 * groups * n * (H // 2) * (W // 2) * 64 * 147, nbytes=4.0 * groups * n * (3 * H * W + (H // d) * (W // d) * 64),
                tag="h2 pool" if pool else "h2")
    else:
        _launch("egr_stem_conv7x7_x6_f32", lib.egr_stem_conv7x7_x6_ex_f32, _p(base), xmap, n, H, W, _p(_cont(w6, "stem weight"), torch.uint8), _p(scale),
                _p(shift), _p(y), 1 if pool else 0, groups, nviews * 3 * H * W, _p(amax_out, torch.int32), _stream(),
                flops=2.0 * groups * n * (H // 2) * (W // 2) * 64 * 147, nbytes=4.0 * groups * n * (3 * H * W + (H // d) * (W // d) * 64),
                tag="x6 pool" if pool else "x6")
    out = Img(y)
    out.tag(amax_out)
    return out


def maxpool(x: Img, k: int, stride: int, pad: int) -> Img:
    if not x.t.is_contiguous():
        raise RuntimeError("egorear_amd.maxpool: contiguous NHWC input expected")
    ho = (x.h + 2 * pad - k) // stride + 1
    wo = (x.w + 2 * pad - k) // stride + 1
    y = torch.empty((x.n, ho, wo, x.c), device=x.t.device, dtype=torch.float32)
    _launch("egr_maxpool_nhwc_f32", lib.egr_maxpool_nhwc_f32, _p(x.t), _p(y), x.n, x.h, x.w, x.c, k, stride, pad, _stream(),
            nbytes=4.0 * x.n * x.c * (x.h * x.w + ho * wo))
    out = Img(y)
    out.tag(x.amax)     # max |max-pool(x)| <= max |x|: the input's record bounds the output
    return out


def upsample2x(x: Img, out: Optional[Img] = None, relu: bool = False) -> Img:
    if x.n > 1 and x.nstride != x.h * x.w * x.ld:
        raise RuntimeError("egorear_amd.upsample2x: images must be densely stacked")
    if out is None:
        out = Img(torch.empty((x.n, 2 * x.h, 2 * x.w, x.c), device=x.t.device, dtype=torch.float32))
    if (out.n, out.h, out.w, out.c) != (x.n, 2 * x.h, 2 * x.w, x.c) or (out.n > 1 and out.nstride != out.h * out.w * out.ld):
        raise RuntimeError("egorear_amd.upsample2x: bad output view")
    _launch("egr_upsample2x_nhwc_f32", lib.egr_upsample2x_nhwc_f32, _p(x.t), x.ld, _p(out.t), out.ld, x.n, x.h, x.w, x.c, 1 if relu else 0, _stream(),
            nbytes=4.0 * 5 * x.n * x.h * x.w * x.c)
    out.tag(x.amax)     # bilinear interpolation (+ ReLU) is a convex combination: the input's record bounds the output
    return out


def up2_relu_head(lo: Img, wgt: torch.Tensor, bias: Optional[torch.Tensor], planes: torch.Tensor, ymap: NMap, gy: int, groups: int = 1):
    """Upsample x2 + ReLU + final 1x1 conv of a heat-map head, straight into channel-major planes (see egorear_hip.h).
    lo: dense (groups*n, h, w, cin); wgt (groups, cout, cin) contiguous; image i of group g lands at planes.flat[g*gy + ymap(i)]."""
    if not lo.t.is_contiguous() or lo.n % groups:
        raise RuntimeError("egorear_amd.up2_relu_head: dense NHWC input with groups | n expected")
    cout, cin = wgt.shape[-2], wgt.shape[-1]
    npg = lo.n // groups
    if cin != lo.c or wgt.numel() != groups * cout * cin or (bias is not None and bias.numel() != groups * cout):
        raise RuntimeError("egorear_amd.up2_relu_head: weight / bias shape")
    ho, wo = 2 * lo.h, 2 * lo.w
    last = (groups - 1) * gy + (npg - 1) // ymap.n_inner * ymap.stride_outer + min(npg - 1, ymap.n_inner - 1) * ymap.stride_inner + cout * ho * wo
    if not planes.is_contiguous() or last > planes.numel():
        raise RuntimeError("egorear_amd.up2_relu_head: output map runs outside the tensor")
    _launch("egr_up2_relu_head_f32", lib.egr_up2_relu_head_f32, _p(lo.t), lo.n, lo.h, lo.w, cin, _p(_cont(wgt, "weight")), _p(bias), cout,
            _p(planes), ymap.n_inner, ymap.stride_inner, ymap.stride_outer, groups, gy, _stream(),
            flops=2.0 * lo.n * ho * wo * cout * cin, nbytes=4.0 * lo.n * (lo.h * lo.w * cin + cout * ho * wo))


def avgpool(x: Img) -> torch.Tensor:
    if not x.t.is_contiguous():
        raise RuntimeError("egorear_amd.avgpool: contiguous input expected")
    y = torch.empty((x.n, x.c), device=x.t.device, dtype=torch.float32)
    _launch("egr_avgpool_nhwc_f32", lib.egr_avgpool_nhwc_f32, _p(x.t), _p(y), x.n, x.h * x.w, x.c, _stream(),
            nbytes=4.0 * x.n * x.c * (x.h * x.w + 1))
    return y


def argmax_rows(hm: torch.Tensor, thr: float):
    """hm (..., H, W) contiguous -> anchors (rows,2) f32, maxvals (rows,), valid (rows,) u8, index (rows,) i32."""
    _cont(hm, "heatmap")
    H, W = hm.shape[-2:]
    rows = hm.numel() // (H * W)
    dev = hm.device
    anchors = torch.empty((rows, 2), device=dev, dtype=torch.float32)
    maxvals = torch.empty((rows,), device=dev, dtype=torch.float32)
    valid = torch.empty((rows,), device=dev, dtype=torch.uint8)
    index = torch.empty((rows,), device=dev, dtype=torch.int32)
    _launch("egr_argmax_rows_f32", lib.egr_argmax_rows_f32, _p(hm), rows, H, W, float(thr), _p(anchors), _p(maxvals), _p(valid, torch.uint8),
                                   _p(index, torch.int32), _stream(), nbytes=4.0 * hm.numel() + 17.0 * rows)
    return anchors, maxvals, valid, index


def layernorm(x: torch.Tensor, gamma, beta, res: Optional[torch.Tensor] = None, eps: float = 1e-5, groups: int = 1) -> torch.Tensor:
    _cont(x, "layernorm input")
    c = x.shape[-1]
    rows = x.numel() // c
    if res is not None and (res.shape != x.shape or not res.is_contiguous()):
        raise RuntimeError("egorear_amd.layernorm: residual must match the input")
    if gamma.numel() != groups * c or beta.numel() != groups * c or rows % groups:
        raise RuntimeError("egorear_amd.layernorm: gamma/beta/groups mismatch")
    y = torch.empty_like(x)
    _launch("egr_layernorm_f32", lib.egr_layernorm_f32, _p(x), _p(res), _p(gamma), _p(beta), _p(y), rows, c, eps,
            rows // groups if groups > 1 else 0, _stream(), nbytes=4.0 * rows * c * (3 if res is not None else 2))
    return y


def joint_mha(qkv: torch.Tensor, b: int, j: int, heads: int, d: int, scale: float) -> torch.Tensor:
    _cont(qkv, "qkv")
    if qkv.numel() != b * j * 3 * heads * d:
        raise RuntimeError("egorear_amd.joint_mha: qkv size mismatch")
    out = torch.empty((b * j, heads * d), device=qkv.device, dtype=torch.float32)
    _launch("egr_joint_mha_f32", lib.egr_joint_mha_f32, _p(qkv), _p(out), b, j, heads, d, float(scale), _stream())
    return out


def msda_gather(feat: torch.Tensor, pos: Optional[torch.Tensor], offs_logits: torch.Tensor, anchors: torch.Tensor,
                valid: torch.Tensor, b: int, views: int, joints: int, heads: int, dh: int, hgt: int, wid: int, groups: int = 1):
    """feat (views, b, hgt*wid, cf) contiguous.  Returns g (G, rows, heads, cf), e (G, rows, heads*dh) | None,
    sigma (G, heads, rows), rowmask (rows,) with rows = (b, joint, view); G query sets (offs_logits (G, b*joints, .),
    pos (G, views, hgt*wid, heads*dh)) share feat / anchors / valid."""
    _cont(feat, "feature memory"); _cont(offs_logits, "offs_logits"); _cont(anchors, "anchors"); _cont(valid, "valid")
    cf = feat.shape[-1]
    if feat.numel() != views * b * hgt * wid * cf:
        raise RuntimeError("egorear_amd.msda_gather: feature memory size mismatch")
    if offs_logits.numel() != groups * b * joints * heads * 16 * 3:
        raise RuntimeError("egorear_amd.msda_gather: offs_logits size mismatch")
    if anchors.numel() != b * views * joints * 2 or valid.numel() != b * views * joints:
        raise RuntimeError("egorear_amd.msda_gather: anchors/valid size mismatch")
    if pos is not None:
        _cont(pos, "positional table")
        if pos.numel() != groups * views * hgt * wid * heads * dh:
            raise RuntimeError("egorear_amd.msda_gather: positional table size mismatch")
    rows = b * joints * views
    dev = feat.device
    g = torch.empty((groups, rows, heads, cf), device=dev, dtype=torch.float32)
    e = torch.empty((groups, rows, heads * dh), device=dev, dtype=torch.float32) if pos is not None else None
    sigma = torch.empty((groups, heads, rows), device=dev, dtype=torch.float32)
    rowmask = torch.empty((rows,), device=dev, dtype=torch.uint8)
    _launch("egr_msda_gather_f32", lib.egr_msda_gather_f32, _p(feat), cf, _p(pos), dh, _p(offs_logits), _p(anchors), _p(valid, torch.uint8), b, views,
                                   joints, heads, hgt, wid, _p(g), _p(e), _p(sigma), _p(rowmask, torch.uint8), groups, _stream(),
            # algorithmic bytes: 16 points x 4 bilinear corners x (cf + dh) floats per (query set, row, head) gathered (served
            # mostly by L2), the offsets / logits read, g / e / sigma written
            nbytes=4.0 * groups * rows * heads * (64.0 * (cf + (dh if pos is not None else 0)) + 48 + cf + (dh if pos is not None else 0) + 1),
            # unique bytes (what HBM must deliver at least: every operand once, every output once) for bench.py's roofline_hbm entry
            tag=("unique%d" % (4 * (feat.numel() + (pos.numel() if pos is not None else 0) + offs_logits.numel() + anchors.numel() + g.numel()
                                    + (e.numel() if e is not None else 0) + sigma.numel()) + valid.numel() + rowmask.numel())) if PROFILE is not None else "")
    return g, e, sigma, rowmask


def fisheye_project(pts: torch.Tensor, ctm: Optional[torch.Tensor], cams: torch.Tensor, out: Optional[torch.Tensor] = None):
    """pts (b, joints, 3) is updated IN PLACE in syn mode (ctm None), as the reference does - or, with `out`,
    left alone and the updated points (syn: mutated, rw: a copy) written there."""
    _cont(pts, "pts")
    b, joints = pts.shape[:2]
    if cams.numel() != 4 * 17:
        raise RuntimeError("egorear_amd.fisheye_project: cams must hold 4 records of 17 floats")
    if ctm is not None:
        _cont(ctm, "coord_trans_mat")
        if ctm.numel() != b * 4 * 16:
            raise RuntimeError("egorear_amd.fisheye_project: coord_trans_mat must be (b,4,4,4)")
    dev = pts.device
    anchors = torch.empty((b, 4, joints, 2), device=dev, dtype=torch.float32)
    valid = torch.empty((b, 4, joints), device=dev, dtype=torch.uint8)
    q4 = torch.empty((b * joints, 4), device=dev, dtype=torch.float32)
    if out is not None:
        if out.shape != pts.shape or not out.is_contiguous():
            raise RuntimeError("egorear_amd.fisheye_project: out must be a contiguous tensor of pts' shape")
        _launch("egr_fisheye_project_f32", lib.egr_fisheye_project2_f32, _p(pts), _p(out), _p(ctm), _p(cams), b, joints, _p(anchors),
                _p(valid, torch.uint8), _p(q4), _stream())
        return anchors, valid, q4
    _launch("egr_fisheye_project_f32", lib.egr_fisheye_project_f32, _p(pts), _p(ctm), _p(cams), b, joints, _p(anchors), _p(valid, torch.uint8), _p(q4),
                                       _stream())
    return anchors, valid, q4


def linear_smallk(x: torch.Tensor, sxm: int, sxk: int, w: torch.Tensor, bias, m: int, n: int, k: int, act: int,
                  groups: int = 1) -> torch.Tensor:
    _cont(w, "weight")
    if w.numel() != groups * n * k or m % groups or (m - 1) * sxm + (k - 1) * sxk >= x.numel():
        raise RuntimeError("egorear_amd.linear_smallk: size mismatch")
    y = torch.empty((m, n), device=x.device, dtype=torch.float32)
    _launch("egr_linear_smallk_f32", lib.egr_linear_smallk_f32, _p(x), sxm, sxk, _p(w), _p(bias), _p(y), m, n, k, act,
            m // groups if groups > 1 else 0, _stream())
    return y


def jqa_sum(hm_embed: torch.Tensor, embed: torch.Tensor, bfb: torch.Tensor, b: int, j: int, c: int, groups: int = 1) -> torch.Tensor:
    """b counts all frames of all groups (groups x frames per group); embed is (groups, j, c)."""
    _cont(hm_embed, "hm_embed"); _cont(embed, "embed"); _cont(bfb, "bfb")
    if hm_embed.numel() != b * j * c or embed.numel() != groups * j * c or bfb.numel() != b * c or b % groups:
        raise RuntimeError("egorear_amd.jqa_sum: size mismatch")
    y = torch.empty((b * j, c), device=hm_embed.device, dtype=torch.float32)
    _launch("egr_jqa_sum_f32", lib.egr_jqa_sum_f32, _p(hm_embed), _p(embed), _p(bfb), _p(y), b, j, c,
            b // groups if groups > 1 else 0, _stream())
    return y


def tokens_to_nhwc(x: torch.Tensor, b: int, j: int, hw: int) -> torch.Tensor:
    _cont(x, "tokens")
    if x.numel() != b * j * hw:
        raise RuntimeError("egorear_amd.tokens_to_nhwc: size mismatch")
    y = torch.empty((b, hw, j), device=x.device, dtype=torch.float32)
    _launch("egr_tokens_to_nhwc_f32", lib.egr_tokens_to_nhwc_f32, _p(x), _p(y), b, j, hw, _stream())
    return y


def pack_layer_w(w: torch.Tensor) -> torch.Tensor:
    """(..., rows, k) fp32 weight stack -> the same shape in egr_joint_layer_f32's fragment order (egr_pack_layer_w_f32)."""
    rows, k = w.shape[-2], w.shape[-1]
    if w.dtype != torch.float32 or rows % 16 or k % 128:
        raise RuntimeError("egorear_amd.pack_layer_w: fp32, rows % 16 == 0, k % 128 == 0 expected")
    w = _cont(w, "layer weight")
    out = torch.empty_like(w)
    _launch("egr_pack_layer_w_f32", lib.egr_pack_layer_w_f32, _p(w), w.numel() // (rows * k), rows, k, _p(out), _stream())
    return out


def pack_layer_wh2(w: torch.Tensor) -> torch.Tensor:
    """(..., rows, k) fp32 weight stack -> (..., rows * k + rows) floats: per matrix the fp16-scheme image egr_joint_layer_f32 reads with
    w_packed = 2 (two fp16 planes per weight in fragment order) and the rows' descales behind it (egr_pack_layer_wh2_f32)."""
    rows, k = w.shape[-2], w.shape[-1]
    if w.dtype != torch.float32 or rows % 16 or k % 128:
        raise RuntimeError("egorear_amd.pack_layer_wh2: fp32, rows % 16 == 0, k % 128 == 0 expected")
    w = _cont(w, "layer weight")
    out = torch.empty(w.shape[:-2] + (rows * k + rows,), device=w.device, dtype=torch.float32)
    _launch("egr_pack_layer_wh2_f32", lib.egr_pack_layer_wh2_f32, _p(w), w.numel() // (rows * k), rows, k, _p(out), _stream())
    return out


def joint_layer(x: torch.Tensor, g: torch.Tensor, e: Optional[torch.Tensor], sigma: torch.Tensor, rowmask: torch.Tensor, W: dict,
                B: int, J: int, V: int, Cdim: int, groups: int, *, ol: Optional[dict] = None, post: Optional[dict] = None,
                reg: Optional[dict] = None, want_xn: bool = False, head: Optional[dict] = None):
    """One transformer layer behind egr_msda_gather_f32 as one launch (egr_joint_layer_f32).  W: the layer's plain weight stacks
    (engine.pack_layer_fused).  ol = {"w", "b"}: also the next layer's offsets / logits; post = {"g", "b"}: also post_norm
    (want_xn: return it); reg = {"w0", "b0", "w2", "b2", "anchors"}: also the 3-D regression head; head = {"w" (groups, 64, J),
    "b" (groups, 64), "amax": record | None}: also the refiner's head offset (tokens as a 16 x 16 image -> 1x1 conv + ReLU -> up x2),
    returned as head["out"] (groups*B, 32, 32, 64).
    Returns (x_out, ol_out | None, xn | None, pred | None)."""
    rows = groups * B * J
    if x.shape != (rows, Cdim) or not x.is_contiguous():
        raise RuntimeError("egorear_amd.joint_layer: x must be (groups*B*J, C) contiguous")
    if g.numel() != groups * B * J * V * 4 * 128 or sigma.numel() != groups * 4 * B * J * V or rowmask.numel() != B * J * V:
        raise RuntimeError("egorear_amd.joint_layer: sampled operands do not match (B, J, V)")
    if e is not None and e.numel() != groups * B * J * V * Cdim:
        raise RuntimeError("egorear_amd.joint_layer: e does not match")
    packed = W.get("packed")
    packed = 2 if packed == 2 else (1 if packed else 0)     # 2: the fp16-scheme images (pack_layer_wh2), 1: fragment order (pack_layer_w)

    def msz(rows, k):                                       # floats of one weight matrix as passed
        return rows * k + (rows if packed == 2 else 0)
    need = {"w_fold": groups * msz(Cdim, 128), "c_fold": groups * Cdim, "w_out": groups * msz(Cdim, Cdim), "b_out": groups * Cdim,
            "w_fuse": groups * msz(Cdim, V * Cdim), "b_fuse": groups * Cdim, "ln1_g": groups * Cdim, "ln1_b": groups * Cdim,
            "w_qkv": groups * msz(3 * Cdim, Cdim), "b_qkv": groups * 3 * Cdim, "w_mo": groups * msz(Cdim, Cdim), "b_mo": groups * Cdim,
            "ln2_g": groups * Cdim, "ln2_b": groups * Cdim, "w_f0": groups * msz(512, Cdim), "b_f0": groups * 512,
            "w_f1": groups * msz(Cdim, 512), "b_f1": groups * Cdim, "ln3_g": groups * Cdim, "ln3_b": groups * Cdim}
    d = LayerDesc()
    d.B, d.J, d.V, d.C, d.heads, d.cf, d.groups, d.ffn_dim = B, J, V, Cdim, 4, 128, groups, 512
    d.eps, d.mha_scale = 1e-5, float((Cdim // 4) ** -0.5)
    d.x, d.g, d.e, d.sigma, d.rowmask = _p(x), _p(_cont(g, "g")), _p(e), _p(_cont(sigma, "sigma")), _p(rowmask, torch.uint8)
    for k, n in need.items():
        t = W[k]
        if t.numel() != n or not t.is_contiguous():
            raise RuntimeError(f"egorear_amd.joint_layer: weight {k} has {t.numel()} elements, expected {n}")
        setattr(d, k, _p(t))
    d.w_packed = packed                          # the matrices of W, ol and reg are then all in that order
    x_out = torch.empty_like(x)
    d.x_out = _p(x_out)
    ol_out = xn = pred = None
    if ol is not None:
        n = ol["b"].shape[-1]
        if ol["w"].numel() != groups * msz(n, Cdim) or ol["b"].numel() != groups * n or n % 16:
            raise RuntimeError("egorear_amd.joint_layer: offsets / logits weights do not match")
        ol_out = torch.empty((rows, n), device=x.device, dtype=torch.float32)
        d.w_ol, d.b_ol, d.ol_out, d.ol_n = _p(_cont(ol["w"], "w_ol")), _p(ol["b"]), _p(ol_out), n
    if post is not None:
        if post["g"].numel() != groups * Cdim or post["b"].numel() != groups * Cdim:
            raise RuntimeError("egorear_amd.joint_layer: post_norm parameters do not match")
        d.lnp_g, d.lnp_b = _p(post["g"]), _p(post["b"])
        if want_xn:
            xn = torch.empty_like(x)
            d.xn_out = _p(xn)
    if reg is not None:
        if post is None:
            raise RuntimeError("egorear_amd.joint_layer: the regression head sits behind post_norm")
        if reg["w0"].numel() != groups * msz(Cdim, Cdim) or reg["w2"].numel() != groups * 3 * Cdim or reg["anchors"].numel() != rows * 3:
            raise RuntimeError("egorear_amd.joint_layer: regression head operands do not match")
        pred = torch.empty((rows, 3), device=x.device, dtype=torch.float32)
        d.w_r0, d.b_r0, d.w_r2, d.b_r2 = _p(_cont(reg["w0"], "w_r0")), _p(reg["b0"]), _p(_cont(reg["w2"], "w_r2")), _p(reg["b2"])
        d.anchors3d, d.pred_out = _p(_cont(reg["anchors"], "anchors")), _p(pred)
    if head is not None:
        if post is None or reg is not None or Cdim != 256:
            raise RuntimeError("egorear_amd.joint_layer: the head offset sits behind post_norm of a 256-channel layer, without the regression head")
        hn = head["b"].shape[-1]
        if hn != 64 or head["w"].numel() != groups * hn * J or head["b"].numel() != groups * hn:
            raise RuntimeError("egorear_amd.joint_layer: head offset weights do not match")
        h0 = torch.empty((groups * B, 32, 32, hn), device=x.device, dtype=torch.float32)
        d.w_h0, d.b_h0, d.h0_out, d.h0_n = _p(_cont(head["w"], "w_h0")), _p(_cont(head["b"], "b_h0")), _p(h0), hn
        rec = head.get("amax")
        if rec is not None:
            d.amax_h0 = _p(rec, torch.int32)
            h0._egr_amax = rec
        head["out"] = h0
    flops = 2.0 * groups * B * (J * V * (Cdim * 128 + Cdim * Cdim) + J * (V * Cdim * Cdim + 3 * Cdim * Cdim + Cdim * Cdim + 2 * 512 * Cdim))
    _launch("egr_joint_layer_f32", lib.egr_joint_layer_f32, C.byref(d), _stream(), flops=flops)
    return x_out, ol_out, xn, pred


def jqa_query(t: torch.Tensor, s32: torch.Tensor, W: dict, B: int, J: int, Cdim: int, groups: int):
    """The JQA query of the refiners as one launch (egr_jqa_query_f32): t (groups*B*J, C) = ReLU(heatmap_proj[0](hm)), s32
    (groups*B, h, w, 512) NHWC.  W: {"w_hp2", "b_hp2", "w_bfb", "b_bfb", "embed", "w_q", "b_q", "w_ol", "b_ol", "packed"} - matrices in
    pack_layer_w / pack_layer_wh2 order.  Returns (x (groups*B*J, C), ol (groups*B*J, ol_n))."""
    rows = groups * B * J
    if t.shape != (rows, Cdim) or not t.is_contiguous() or Cdim != 256:
        raise RuntimeError("egorear_amd.jqa_query: t must be (groups*B*J, 256) contiguous")
    if s32.dim() != 4 or s32.shape[0] != groups * B or s32.shape[3] != 512 or not s32.is_contiguous():
        raise RuntimeError("egorear_amd.jqa_query: s32 must be (groups*B, h, w, 512) contiguous NHWC")
    packed = W.get("packed")
    packed = 2 if packed == 2 else (1 if packed else 0)

    def msz(r, k):
        return r * k + (r if packed == 2 else 0)
    n = W["b_ol"].shape[-1]
    need = {"w_hp2": groups * msz(Cdim, Cdim), "b_hp2": groups * Cdim, "w_bfb": groups * msz(Cdim, 512), "b_bfb": groups * Cdim,
            "embed": groups * J * Cdim, "w_q": groups * msz(Cdim, Cdim), "b_q": groups * Cdim, "w_ol": groups * msz(n, Cdim), "b_ol": groups * n}
    d = JqaQueryDesc()
    d.B, d.J, d.C, d.groups, d.kb, d.pool_hw, d.ol_n, d.w_packed = B, J, Cdim, groups, 512, s32.shape[1] * s32.shape[2], n, packed
    d.t, d.s32 = _p(t), _p(s32)
    for k, ne in need.items():
        w = W[k]
        if w.numel() != ne or not w.is_contiguous():
            raise RuntimeError(f"egorear_amd.jqa_query: weight {k} has {w.numel()} elements, expected {ne}")
        setattr(d, k, _p(w))
    x = torch.empty((rows, Cdim), device=t.device, dtype=torch.float32)
    ol = torch.empty((rows, n), device=t.device, dtype=torch.float32)
    d.x_out, d.ol_out = _p(x), _p(ol)
    _launch("egr_jqa_query_f32", lib.egr_jqa_query_f32, C.byref(d), _stream(),
            flops=2.0 * groups * B * (J * Cdim * (2 * Cdim + n) + Cdim * 512))
    return x, ol


def pose_query(h1: torch.Tensor, ctm: Optional[torch.Tensor], cams: torch.Tensor, W: dict, B: int, J: int, Cdim: int):
    """The lifting head between mlp_pred[1] and its first decoder layer as one launch (egr_pose_query_f32).  h1 (B, C).  W: {"w_m2",
    "b_m2", "w_qg0", "b_qg0", "w_qg2", "b_qg2", "w_qg4", "b_qg4", "w_ol", "b_ol", "packed"}.
    Returns (pred (B, J, 3), anchors_3d (B, J, 3), anchors_2d (B, 4, J, 2), valid (B, 4, J) uint8, x (B*J, C), ol (B*J, ol_n))."""
    if h1.shape != (B, Cdim) or not h1.is_contiguous() or Cdim != 128 or J != 16:
        raise RuntimeError("egorear_amd.pose_query: h1 must be (B, 128) contiguous, 16 joints")
    if ctm is not None and (ctm.shape != (B, 4, 4, 4) or not ctm.is_contiguous()):
        raise RuntimeError("egorear_amd.pose_query: coord_trans_mat must be (B, 4, 4, 4) contiguous")
    packed = W.get("packed")
    packed = 2 if packed == 2 else (1 if packed else 0)

    def msz(r, k):
        return r * k + (r if packed == 2 else 0)
    n = W["b_ol"].shape[-1]
    need = {"w_m2": msz(3 * J, Cdim), "b_m2": 3 * J, "w_qg0": Cdim * 4, "b_qg0": Cdim, "w_qg2": msz(Cdim, Cdim), "b_qg2": Cdim,
            "w_qg4": msz(Cdim, Cdim), "b_qg4": Cdim, "w_ol": msz(n, Cdim), "b_ol": n}
    d = PoseQueryDesc()
    d.B, d.J, d.C, d.ol_n, d.w_packed = B, J, Cdim, n, packed
    d.h1, d.ctm, d.cams = _p(h1), _p(ctm), _p(cams)
    for k, ne in need.items():
        w = W[k]
        if w.numel() != ne or not w.is_contiguous():
            raise RuntimeError(f"egorear_amd.pose_query: weight {k} has {w.numel()} elements, expected {ne}")
        setattr(d, k, _p(w))
    dev = h1.device
    pred = torch.empty((B, J, 3), device=dev, dtype=torch.float32)
    a3 = torch.empty((B, J, 3), device=dev, dtype=torch.float32)
    a2 = torch.empty((B, 4, J, 2), device=dev, dtype=torch.float32)
    valid = torch.empty((B, 4, J), device=dev, dtype=torch.uint8)
    x = torch.empty((B * J, Cdim), device=dev, dtype=torch.float32)
    ol = torch.empty((B * J, n), device=dev, dtype=torch.float32)
    d.pred_out, d.anchors3d_out, d.anchors2d_out, d.valid_out, d.x_out, d.ol_out = _p(pred), _p(a3), _p(a2), _p(valid, torch.uint8), _p(x), _p(ol)
    _launch("egr_pose_query_f32", lib.egr_pose_query_f32, C.byref(d), _stream(), flops=2.0 * B * (Cdim * 3 * J + J * Cdim * (2 * Cdim + n + 4)))
    return pred, a3, a2, valid, x, ol



# --------------------------------------------------------------------------- launch-device record hygiene
def _guarded(fn):
    """A wrapper whose host-side validation raises after its first _p() must not leave the thread's launch-device record set."""
    import functools

    @functools.wraps(fn)
    def run(*a, **k):
        try:
            return fn(*a, **k)
        except BaseException:
            _DEVL.idx = None
            raise
    return run


def _guard_module(ns):
    import types
    for _name, _fn in list(ns.items()):
        if isinstance(_fn, types.FunctionType) and not _name.startswith("_") and _fn.__module__ == ns.get("__name__"):
            ns[_name] = _guarded(_fn)


_guard_module(globals())
