"""Run the reference's own `run.py` on the MI355X path with ZERO edits to the reference tree.

    python -m egorear_amd.run_reference <EgoRear checkout> fit|test|predict --config configs/....yaml [LightningCLI arguments]
    python -m egorear_amd.run_reference <EgoRear checkout> --check        # resolve the classes, print them, do not run run.py

What it does, in this order and in ONE process (no re-exec - `os.exec*` from a process that may have touched the GPU is not
allowed on the MI355X pool, and nothing here touches the GPU before `run.py` does):

  1. puts the checkout first on `sys.path` and makes it the working directory (the YAMLs' relative
     `camera_calib_file_dir_path: ./pose_estimation/utils/camera_calib_file/ego4view` and `--config configs/...` resolve as they
     do for `python run.py`);
  2. pre-seeds `sys.modules["pose_estimation.models.estimator"]` with a module that exports this package's three estimator
     classes under the names the reference's `pose_estimation/models/estimator/__init__.py:3-5` exports - the wrappers'
     `from pose_estimation.models.estimator import EgoPoseFormerHeatmap / ...HeatmapMVFEX / ...MVFEX`
     (`pl_wrappers/egoposeformer/heatmap.py:21`, `heatmap_mvf_ex.py:23`, `pose_3d_mvf_ex.py:20`) then bind to them; `pose_estimation`
     and `pose_estimation.models` are namespace packages (no `__init__.py`), so nothing of the reference's estimator package is
     executed;
  3. `egorear_amd.msda.install_mmcv_shim()` - `models/utils/deform_attn.py:9` imports mmcv's MSDA Function, which has no ROCm
     build; a maintainer's own modules that still use `MSDeformAttn` get `egr_msda_fwd/bwd_f32` (a `torch.library` op, so
     `run.py:7-9`'s `torch.compile(model.network)` traces through it);
  4. `runpy.run_path("<checkout>/run.py", run_name="__main__")` with `sys.argv = ["run.py", ...]`: `TorchCompileCLI()`
     (`run.py:11-25`) parses the same command line as always.

The alternative that edits one line of the reference (`INTEGRATION.md` §1) stays valid; this launcher is for a pristine checkout.
"""
from __future__ import annotations

import os
import runpy
import sys
import types

ESTIMATOR_MODULE = "pose_estimation.models.estimator"
CLASS_NAMES = ("EgoPoseFormerHeatmap", "EgoPoseFormerHeatmapMVFEX", "EgoPoseFormerMVFEX")


def install(checkout: str, chdir: bool = True) -> types.ModuleType:
    """Steps 1-3 above.  Returns the module now registered as `pose_estimation.models.estimator`.  Makes no GPU call."""
    checkout = os.path.abspath(checkout)
    if not os.path.isfile(os.path.join(checkout, "run.py")) or not os.path.isdir(os.path.join(checkout, "pose_estimation")):
        raise FileNotFoundError(f"egorear_amd.run_reference: {checkout!r} is not an EgoRear checkout (run.py and pose_estimation/ expected)")
    if ESTIMATOR_MODULE in sys.modules and not getattr(sys.modules[ESTIMATOR_MODULE], "__egorear_amd__", False):
        raise RuntimeError(f"egorear_amd.run_reference: {ESTIMATOR_MODULE} is already imported from {getattr(sys.modules[ESTIMATOR_MODULE], '__file__', '?')}; "
                           "install() must run before the reference's wrappers are imported")
    if checkout in sys.path:
        sys.path.remove(checkout)
    sys.path.insert(0, checkout)
    if chdir:
        os.chdir(checkout)
    from . import estimator, msda
    mod = types.ModuleType(ESTIMATOR_MODULE)
    mod.__doc__ = "egorear_amd.estimator under the reference's import path (egorear_amd.run_reference)"
    mod.__egorear_amd__ = True
    mod.__path__ = []          # a package without submodules: `...estimator.egoposeformer_heatmap` must not resolve to the reference's files
    for name in CLASS_NAMES:
        setattr(mod, name, getattr(estimator, name))
    mod.__all__ = list(CLASS_NAMES)
    sys.modules[ESTIMATOR_MODULE] = mod
    # `import pose_estimation.models.estimator as E` walks attributes: bind the module on its parent (a namespace package of the
    # checkout - importing it executes no reference code)
    import importlib
    parent = importlib.import_module(ESTIMATOR_MODULE.rsplit(".", 1)[0])
    if getattr(parent, "__file__", None) is not None:
        raise RuntimeError(f"egorear_amd.run_reference: {parent.__name__} has an __init__.py in this checkout; expected a namespace package")
    parent.estimator = mod
    msda.install_mmcv_shim()
    return mod


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 0 if argv else 2
    checkout, rest = argv[0], argv[1:]
    mod = install(checkout)
    if rest[:1] == ["--check"]:
        for name in CLASS_NAMES:
            cls = getattr(mod, name)
            print(f"{ESTIMATOR_MODULE}.{name} -> {cls.__module__}.{cls.__qualname__}")
        fn = sys.modules["mmcv.ops.multi_scale_deform_attn"].MultiScaleDeformableAttnFunction
        print(f"mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttnFunction -> {fn.__module__}.{fn.__qualname__}")
        return 0
    script = os.path.join(os.path.abspath(checkout), "run.py")
    sys.argv = ["run.py"] + rest
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
