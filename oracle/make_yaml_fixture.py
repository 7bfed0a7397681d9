#!/usr/bin/env python3
"""Fixture generator (test infrastructure): the `model` blocks of the reference's 12 shipped YAMLs as DATA.

    python -m oracle.make_yaml_fixture            # needs /root/reference; writes tests/golden/yaml_model_cfgs.json

For every configs/*.yaml: the Lightning class path, the constructor's `model_cfg` dict exactly as jsonargparse hands it to the
model class, and the scalar hyper-parameters next to it (batch size, lr, compile flag ...).  No source text is copied: the file
holds parsed values only.  tests/test_configs.py checks egorear_amd/configs.py's presets against it and constructs every model.
"""
import glob
import json
import os

import yaml

REF = "/root/reference/configs"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "yaml_model_cfgs.json")


def main():
    out = {}
    for path in sorted(glob.glob(os.path.join(REF, "*.yaml"))):
        with open(path) as f:
            y = yaml.safe_load(f)
        m = y["model"]
        args = dict(m["init_args"])
        cfg = args.pop("model_cfg")
        out[os.path.basename(path)] = {
            "class_path": m["class_path"],
            "model_cfg": cfg,
            "init_args": {k: v for k, v in args.items() if not isinstance(v, (dict, list)) or k in ("lr_decay_epochs",)},
            "trainer": {k: y.get("trainer", {}).get(k) for k in ("devices", "precision", "gradient_clip_val", "max_epochs", "benchmark")},
        }
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", OUT, len(out), "configs")


if __name__ == "__main__":
    main()
