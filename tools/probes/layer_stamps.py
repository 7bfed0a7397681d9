#!/usr/bin/env python3
"""Phase times of joint_layer_kernel's workgroup 0 (the LAST layer launch of a forward: pose layer 3, C = 128, all tails).
    python tools/build_variant.py lstamps -DLAYER_STAMPS --src egr_layer.hip
    EGR_LIB=egorear_amd/csrc/libegorear_hip_lstamps.so python tools/probes/layer_stamps.py [batch]"""
import copy, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egorear_amd import configs, hip, synth
from egorear_amd.estimator import EgoPoseFormerMVFEX

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
REF = len(sys.argv) > 2 and sys.argv[2] == "ref"      # "ref": stop behind the heat-map estimator - the last layer launch is the refiners' (C = 256)
net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg())).eval(); synth.load_synth(net, 42); net = net.cuda()
img = synth.synth_images(B, 4, seed=1).cuda()
names = ["sampled rows: fold + output_proj (2 halves)", "fuse_mlp", "residual + norm_cross", "q/k/v", "attention", "out_proj + norm_spatial",
         "FFN + norm_ffn", "store + offsets tail", "post_norm + regression tail"]
with torch.no_grad():
    for rep in range(4):
        (net.heatmap_estimator if REF else net)(img); torch.cuda.synchronize()
        buf = (C.c_ulonglong * 64)()
        hip.lib.egr_layer_stamps.argtypes = [C.c_void_p]
        assert hip.lib.egr_layer_stamps(buf) == 0
        t = [buf[i] for i in range(10)]
        if rep == 3:
            for i, n in enumerate(names):
                print(f"{t[i + 1] - t[i]:8d} ticks {100.0 * (t[i + 1] - t[i]) / (t[9] - t[0]):5.1f} %  {n}")
            print(f"{t[9] - t[0]:8d} ticks  total (s_memtime)")
            print(f"    sampled rows, first half {buf[16] - t[0]} ticks, second half (the same code again) {buf[17] - buf[16]} ticks")
