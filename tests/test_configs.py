"""The 12 shipped YAMLs of the reference as data (tests/golden/yaml_model_cfgs.json, made by oracle/make_yaml_fixture.py from
/root/reference/configs): the presets of egorear_amd/configs.py must equal their `model_cfg`, the eight 4-view configurations
must construct from the UNCHANGED dicts, the four 2-view `*_stereo_front` MVFEx / pose3d ones are out of scope and
must refuse loudly - not build something else."""
import copy
import json
import os
import warnings

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "yaml_model_cfgs.json")) as f:
    YAMLS = json.load(f)

CLASS_OF = {"PoseHeatmapLightningModel": "EgoPoseFormerHeatmap", "PoseHeatmapMVFEXLightningModel": "EgoPoseFormerHeatmapMVFEX",
            "Pose3DMVFEXLightningModel": "EgoPoseFormerMVFEX"}   # pl_wrappers/egoposeformer/{heatmap,heatmap_mvf_ex,pose_3d_mvf_ex}.py
STEREO_ONLY = sorted(k for k in YAMLS if k.endswith("_stereo_front.yaml") and "heatmap_stereo" not in k)


def test_fixture_holds_the_twelve_configs():
    assert len(YAMLS) == 12 and len(STEREO_ONLY) == 4
    for name, y in YAMLS.items():
        assert y["class_path"].rsplit(".", 1)[-1] in CLASS_OF, name
        assert y["trainer"]["devices"] == 1 and y["trainer"]["precision"] == 32 and y["init_args"]["compile"] is True


@pytest.mark.parametrize("yaml_name,preset", [
    ("ego4view_syn_heatmap_stereo_front.yaml", ("heatmap_cfg",)),
    ("ego4view_syn_heatmap_stereo_back.yaml", ("heatmap_cfg",)),
    ("ego4view_rw_heatmap_stereo_front.yaml", ("heatmap_cfg",)),
    ("ego4view_rw_heatmap_stereo_back.yaml", ("heatmap_cfg",)),
    ("ego4view_syn_heatmap_mvfex-n1_jqa.yaml", ("heatmap_mvfex_cfg", "ego4view_syn")),
    ("ego4view_rw_heatmap_mvfex-n1_jqa.yaml", ("heatmap_mvfex_cfg", "ego4view_rw")),
    ("ego4view_syn_pose3d.yaml", ("pose3d_cfg", "ego4view_syn")),
    ("ego4view_rw_pose3d.yaml", ("pose3d_cfg", "ego4view_rw")),
])
def test_presets_equal_the_yaml_model_cfg(yaml_name, preset):
    from egorear_amd import configs
    mine = getattr(configs, preset[0])(*preset[1:])
    assert configs.set_imagenet_pretrain(copy.deepcopy(mine), True) == YAMLS[yaml_name]["model_cfg"]
    # the presets only differ in the flag that needs the ImageNet weights (benchmarks / tests load seeded weights instead)
    assert mine != YAMLS[yaml_name]["model_cfg"]


def _construct(yaml_name):
    from egorear_amd import estimator
    y = YAMLS[yaml_name]
    cls = getattr(estimator, CLASS_OF[y["class_path"].rsplit(".", 1)[-1]])
    with torch.device("meta"):                       # parameter shapes only: no 500 MB initialisation per config
        return cls(**copy.deepcopy(y["model_cfg"]))


@pytest.mark.parametrize("yaml_name", sorted(k for k in YAMLS if k not in STEREO_ONLY))
def test_four_view_configs_construct_from_the_unchanged_dict(yaml_name):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        net = _construct(yaml_name)
    # use_imagenet_pretrain: True is honoured: without torchvision's weights on this box the trunk says so loudly
    assert any(issubclass(x.category, RuntimeWarning) and "use_imagenet_pretrain" in str(x.message) for x in w)
    n = sum(p.numel() for p in net.parameters())
    expect = {"EgoPoseFormerHeatmap": 11_843_279, "EgoPoseFormerHeatmapMVFEX": 57_436_664, "EgoPoseFormerMVFEX": 126_047_857}
    assert n == expect[type(net).__name__]


@pytest.mark.parametrize("yaml_name", STEREO_ONLY)
def test_two_view_configs_are_refused(yaml_name):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(NotImplementedError):
            _construct(yaml_name)


def test_strict_pretrain_raises_instead_of_warning(monkeypatch):
    from egorear_amd import tree
    monkeypatch.setenv("EGR_STRICT_PRETRAIN", "1")
    monkeypatch.delenv("EGR_RESNET18_WEIGHTS", raising=False)
    with pytest.raises(RuntimeError, match="use_imagenet_pretrain"):
        tree.ResNet18Trunk(use_imagenet_pretrain=True)


def test_imagenet_weights_from_a_torchvision_state_dict_file(tmp_path, monkeypatch):
    """EGR_RESNET18_WEIGHTS: a torchvision resnet18 state_dict (conv1 / bn1 / layer1-4 / fc keys) lands in the reference's split
    of the trunk (resnet.py:14-21)."""
    from egorear_amd import tree
    src = tree.ResNet18Trunk()
    tv = {}
    for k, v in src.state_dict().items():
        for a, b in tree._TV_PREFIX:
            if k.startswith(b):
                tv[a + k[len(b):]] = (v + 0.5) if v.dtype.is_floating_point else v + 3
    tv["fc.weight"], tv["fc.bias"] = torch.zeros(1000, 512), torch.zeros(1000)
    path = tmp_path / "resnet18.pth"
    torch.save(tv, path)
    monkeypatch.setenv("EGR_RESNET18_WEIGHTS", str(path))
    with warnings.catch_warnings():
        warnings.simplefilter("error")               # no warning on this route
        got = tree.ResNet18Trunk(use_imagenet_pretrain=True)
    for k, v in src.state_dict().items():
        want = (v + 0.5) if v.dtype.is_floating_point else v + 3
        assert torch.equal(got.state_dict()[k], want), k


def test_load_model_cfg_returns_the_yaml_as_written(tmp_path):
    import yaml
    from egorear_amd import configs
    name = "ego4view_syn_pose3d.yaml"
    doc = {"model": {"class_path": YAMLS[name]["class_path"], "init_args": {"model_cfg": YAMLS[name]["model_cfg"], "compile": True}}}
    p = tmp_path / name
    p.write_text(yaml.safe_dump(doc))
    cfg = configs.load_model_cfg(str(p))
    assert cfg == YAMLS[name]["model_cfg"]
    assert cfg["heatmap_mvf_cfg"]["encoder_cfg"]["resnet_cfg"]["use_imagenet_pretrain"] is True     # not rewritten
