"""egorear_amd.run_reference: the reference's run.py on a PRISTINE checkout (VERDICT r4 missing #2 / next #6a).  Build-container test:
needs /root/reference (skipped where it is absent, e.g. on the GPU box); every check runs in a child interpreter so that the
sys.modules / sys.path / cwd changes of install() do not leak into the test session."""
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "run.py")), reason="reference checkout not present")


def _run(args, code=None, opt=False):
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable] + (["-O"] if opt else []) + ["-c", code] if code is not None else [sys.executable, "-m", "egorear_amd.run_reference"] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)


def test_check_mode_lists_the_three_classes_and_the_op():
    r = _run([REF, "--check"])
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    for name in ("EgoPoseFormerHeatmap", "EgoPoseFormerHeatmapMVFEX", "EgoPoseFormerMVFEX"):
        assert f"pose_estimation.models.estimator.{name} -> egorear_amd.estimator.{name}" in lines
    assert any(l.startswith("mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttnFunction -> egorear_amd.msda.") for l in lines)


def test_reference_imports_bind_to_this_package_without_touching_the_gpu_or_the_references_estimator_files():
    # (the child runs under -O: MSDeformAttn.forward's `assert ... .sum() == Len_in` (deform_attn.py:114) reads a tensor's value, which a
    # meta tensor does not have - every other statement of that forward is shape-only; the checks below therefore do not use `assert`)
    code = f"""
import sys, os, torch
def ck(c, msg=""):
    if not c:
        raise SystemExit("FAILED: " + str(msg))
from egorear_amd import run_reference, estimator, msda
run_reference.install({REF!r})
ck(os.getcwd() == {REF!r} and sys.path[0] == {REF!r})
# the wrappers' own import statements (pl_wrappers/egoposeformer/heatmap.py:21, heatmap_mvf_ex.py:23, pose_3d_mvf_ex.py:20)
from pose_estimation.models.estimator import EgoPoseFormerHeatmap
from pose_estimation.models.estimator import EgoPoseFormerHeatmapMVFEX
from pose_estimation.models.estimator import EgoPoseFormerMVFEX
import pose_estimation.models.estimator as E
ck(EgoPoseFormerHeatmap is estimator.EgoPoseFormerHeatmap and EgoPoseFormerHeatmapMVFEX is estimator.EgoPoseFormerHeatmapMVFEX)
ck(EgoPoseFormerMVFEX is estimator.EgoPoseFormerMVFEX and E.__egorear_amd__)
ck(not any(k.startswith("pose_estimation.models.estimator.") for k in sys.modules), [k for k in sys.modules if "estimator" in k])
# the reference's OWN deformable-attention module, imported from the checkout, now calls this package's op
# (loguru is a deployment dependency of the reference that this container lacks: a no-op logger stands in for it)
import types
try:
    import loguru
except ImportError:
    sys.modules["loguru"] = types.SimpleNamespace(logger=types.SimpleNamespace(info=print, warning=print, error=print, debug=print))
from pose_estimation.models.utils import deform_attn
ck(deform_attn.__file__.startswith({REF!r}))
ck(deform_attn.MultiScaleDeformableAttnFunction is msda.MultiScaleDeformableAttnFunction)
m = deform_attn.MSDeformAttn(d_model=64, n_levels=1, n_heads=4, n_points=4).to("meta")
shapes = torch.tensor([[8, 8]], device="meta"); starts = torch.tensor([0], device="meta")
q, ref, tok = torch.empty(2, 5, 64, device="meta"), torch.empty(2, 5, 1, 2, device="meta"), torch.empty(2, 64, 64, device="meta")
out = m(q, ref, tok, shapes, starts)
ck(out.shape == (2, 5, 64), out.shape)           # the reference's forward ran end to end on the op's Meta kernel
# (that the same statement sequence traces as ONE Dynamo graph with the operator in it: tests/test_msda_op_host.py)
ck(not torch.cuda.is_initialized())
# a YAML's model block constructs through the reference-path name
from egorear_amd import configs
net = EgoPoseFormerMVFEX(**configs.pose3d_cfg("ego4view_syn"))
ck(type(net).__module__ == "egorear_amd.estimator" and len(net.state_dict()) == 698)
print("OK")
"""
    r = _run(None, code, opt=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_install_refuses_when_the_references_estimator_is_already_imported():
    code = f"""
import sys, types
sys.modules["pose_estimation.models.estimator"] = types.ModuleType("pose_estimation.models.estimator")
from egorear_amd import run_reference
try:
    run_reference.install({REF!r})
except RuntimeError as e:
    assert "already imported" in str(e); print("OK")
"""
    r = _run(None, code)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_run_py_is_executed_in_process():
    """`fit|test ...` hands over to the checkout's run.py through runpy (same process, no exec).  Lightning is not installed in the
    build container, so run.py's first third-party import is what proves it ran."""
    try:
        import pytorch_lightning  # noqa: F401
        pytest.skip("pytorch_lightning present: run.py would start a real job")
    except ImportError:
        pass
    r = _run([REF, "test", "--config", "configs/ego4view_syn_pose3d.yaml"])
    assert r.returncode != 0
    assert "run.py" in r.stderr and "pytorch_lightning" in r.stderr, r.stderr[-2000:]


def test_not_a_checkout_is_refused(tmp_path):
    r = _run([str(tmp_path), "--check"])
    assert r.returncode != 0 and "not an EgoRear checkout" in r.stderr
