"""Pre-processing row: the numpy oracle restates Pillow's resampling and is pinned by Pillow's
own outputs (committed golden + live where Pillow is importable); the HIP path must equal the oracle bit for bit."""
import os

import numpy as np
import pytest
import torch

from egorear_amd import synth
from oracle import preprocess_oracle as O


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "preprocess_pil.npz"))


def test_oracle_equals_pillow_golden(golden):
    frames = synth.synth_raw_frames(1, 4, seed=0).numpy()
    out = O.pil_bicubic_resize_u8(frames[0, 0])
    np.testing.assert_array_equal(out, golden["s0_v0_full"])                      # bit-exact uint8
    for seed in (0, 1):
        fr = synth.synth_raw_frames(1, 4, seed=seed).numpy()
        for v in (1, 3):
            o = O.pil_bicubic_resize_u8(fr[0, v])
            np.testing.assert_array_equal(o[::4, ::4], golden[f"s{seed}_v{v}_sl"])
            assert int(o.astype(np.int64).sum()) == int(golden[f"s{seed}_v{v}_sum"])
    small = golden["small_in"]
    np.testing.assert_array_equal(O.pil_bicubic_resize_u8(small, 37, 41), golden["small_out_37x41"])     # down, non-square
    np.testing.assert_array_equal(O.pil_bicubic_resize_u8(small, 128, 100), golden["small_out_128x100"])  # up


def test_oracle_equals_live_pillow_when_available():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    for (h, w, oh, ow) in ((872, 872, 256, 256), (33, 47, 256, 256), (300, 200, 64, 64)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize([ow, oh], Image.BICUBIC))
        np.testing.assert_array_equal(O.pil_bicubic_resize_u8(img, oh, ow), ref)


def test_host_tables_equal_oracle_tables():
    from egorear_amd.preprocess import resample_tables
    for (i, o) in ((872, 256), (100, 37), (64, 64), (50, 128)):
        b, c, k = resample_tables(i, o)
        b2, c2, k2 = O.precompute_coeffs(i, o)
        assert k == k2 and (b == b2).all() and (c == c2).all()
    assert resample_tables(872, 256)[2] == 15                   # 2 * ceil(2 * 872/256) + 1 taps


def test_normalize_is_totensor_then_normalize():
    img = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2)
    x = O.to_tensor_normalize(img)
    t = torch.from_numpy(img).permute(2, 0, 1).float().div(255)
    ref = (t - torch.tensor(O.MEAN).view(3, 1, 1)) / torch.tensor(O.STD).view(3, 1, 1)
    assert np.array_equal(x, ref.numpy())


@pytest.mark.gpu
def test_hip_preprocess_is_bit_exact(golden):
    from egorear_amd.preprocess import FramePreprocessor
    frames = synth.synth_raw_frames(2, 4, seed=0)
    pre = FramePreprocessor()
    x, u8 = pre(frames.cuda(), return_u8=True)
    assert x.shape == (2, 4, 3, 256, 256) and x.dtype == torch.float32
    np.testing.assert_array_equal(u8[0, 0].cpu().numpy(), golden["s0_v0_full"])   # == Pillow, bit for bit
    ref = O.preprocess_frames(frames.numpy())
    assert np.array_equal(x.cpu().numpy(), ref)                                    # floats too: same ops, same order
    # frames must be uint8 on the device
    with pytest.raises(RuntimeError):
        pre(frames)
    with pytest.raises(RuntimeError):
        pre(frames.cuda().float())


@pytest.mark.gpu
def test_hip_preprocess_feeds_the_network():
    """End to end from raw frames: pre-process on the GPU, run the full pipeline, compare with the oracle chain."""
    import copy
    from egorear_amd import configs
    from egorear_amd.estimator import EgoPoseFormerMVFEX
    from egorear_amd.preprocess import FramePreprocessor
    from oracle import egorear_oracle as E
    from conftest import CALIB
    frames = synth.synth_raw_frames(1, 4, seed=7)
    net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg())).eval()
    sd = synth.load_synth(net, 42)
    net = net.cuda()
    with torch.no_grad():
        preds, hms = net(FramePreprocessor()(frames.cuda()))
        img = torch.from_numpy(O.preprocess_frames(frames.numpy()))
        full = {k: v.cpu() for k, v in net.state_dict().items()}
        o_preds, o_hms, o_aux = E.mvfex_forward(full, E.make_cameras("ego4view_syn", CALIB), img)
    assert torch.equal(net.__dict__["_egr_last_aux"]["heatmap"]["argmax_idx"].cpu().long(), o_aux["heatmap"]["argmax_idx"])
    assert max(float((p.cpu() - q).abs().max()) for p, q in zip(preds, o_preds)) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("in_hw,out_hw", [((872, 872), (256, 256)), ((437, 500), (128, 160)), ((120, 96), (64, 64)), ((64, 1000), (32, 256))])
def test_hip_preprocess_other_sizes_match_the_oracle(in_hw, out_hw):
    """Every horizontal-pass kernel (LDS rows: heights that are multiples of 4 with 16-byte row groups; aligned-dword windows;
    byte loads) against the numpy restatement of Pillow, bit for bit."""
    from egorear_amd.preprocess import FramePreprocessor
    g = torch.Generator().manual_seed(in_hw[0] * 1000 + in_hw[1])
    frames = torch.randint(0, 256, (2, 3, in_hw[0], in_hw[1], 3), generator=g, dtype=torch.uint8)
    pre = FramePreprocessor(in_hw=in_hw, out_hw=out_hw)
    x = pre(frames.cuda())
    ref = O.preprocess_frames(frames.numpy(), out_hw[0], out_hw[1])
    assert x.shape == (2, 3, 3, out_hw[0], out_hw[1])
    assert np.array_equal(x.cpu().numpy(), ref)


@pytest.mark.gpu
def test_fused_and_two_pass_preprocessing_are_the_same_bytes():
    """egr_preprocess_fused_u8_f32 (Pillow's uint8 intermediate kept in LDS, one launch) against the two-pass kernels: identical
    fp32 output and identical uint8 image; shapes outside the fused kernel's limits fall back by themselves."""
    from egorear_amd.preprocess import FramePreprocessor
    frames = synth.synth_raw_frames(3, 4, seed=5).cuda()
    fused, two = FramePreprocessor(), FramePreprocessor()
    fused.fused, two.fused = True, False
    a, a8 = fused(frames, return_u8=True)
    b, b8 = two(frames, return_u8=True)
    assert fused.fused is True and fused.band_rows > 100          # 872 -> 256: the fused kernel took it (a ~121-row band in LDS)
    assert torch.equal(a, b) and torch.equal(a8, b8)
    small = FramePreprocessor(in_hw=(437, 500), out_hw=(128, 160))
    small.fused = True
    g = torch.Generator().manual_seed(3)
    f2 = torch.randint(0, 256, (1, 2, 437, 500, 3), generator=g, dtype=torch.uint8)
    x = small(f2.cuda())
    assert small.fused is False                                    # refused (EGR_EINVAL) -> two passes, same reference result
    assert np.array_equal(x.cpu().numpy(), O.preprocess_frames(f2.numpy(), 128, 160))

