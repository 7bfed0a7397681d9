"""Build libegorear_hip.so (the C-ABI library of include/egorear_hip.h) for gfx950, in-tree.

    python -m egorear_amd.csrc.build

hipcc cross-compiles without a GPU.  Objects go to egorear_amd/csrc/build/, the shared
library next to the sources so that it travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["egr_conv.hip", "egr_conv_tapx.hip", "egr_conv_chain.hip", "egr_stem.hip", "egr_stem_x6.hip", "egr_pointwise.hip", "egr_attn.hip", "egr_preprocess.hip", "egr_metrics.hip", "egr_wgrad.hip", "egr_train.hip", "egr_msda_bwd.hip", "egr_msda_op.hip", "egr_layer.hip", "egr_wstream.hip"]
LIB = os.path.join(HERE, "libegorear_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include"), "-I" + HERE]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    # every header of the source directory (egr_common.h, egr_stem_pool.h, ...): a change to any of them rebuilds every object
    headers = sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")) + [os.path.join(REPO, "include", "egorear_hip.h")]
    train_headers = [os.path.join(REPO, "include", "egorear_train.h")]
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers + (train_headers if src in ("egr_train.hip", "egr_msda_bwd.hip") else [])):
            jobs.append([hipcc] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(HERE, "build", s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
