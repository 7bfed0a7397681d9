#!/usr/bin/env python3
"""Experiment builds: egr_conv.hip (or another source) compiled with extra -D flags into egorear_amd/csrc/libegorear_hip_<name>.so
(the other objects are the regular build's).  EGR_LIB=<path> makes egorear_amd.hip load it.
    python tools/build_variant.py NAME [-DFOO -DBAR=1 ...] [--src egr_conv.hip]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egorear_amd.csrc import build as B

name = sys.argv[1]
src = "egr_conv.hip"
defs = []
it = iter(sys.argv[2:])
for a in it:
    if a == "--src":
        src = next(it)
    else:
        defs.append(a)
B.build(verbose=False)
o = os.path.join(B.HERE, "build", src.replace(".hip", f"_{name}.o"))
subprocess.run([B._hipcc()] + B.FLAGS + defs + ["-c", os.path.join(B.HERE, src), "-o", o], check=True)
objs = [o if s == src else os.path.join(B.HERE, "build", s.replace(".hip", ".o")) for s in B.SOURCES]
lib = os.path.join(B.HERE, f"libegorear_hip_{name}.so")
subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
print(lib)
