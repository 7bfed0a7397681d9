#!/usr/bin/env python3
"""Why does bench.py's cpu_baseline move between runs?  Times the CPU oracle's forward (batch 8, 16 threads) per call:
(a) in a process that never touches the GPU, (b) after the HIP context exists and a forward ran, (c) with two captured lanes alive.
    python tools/probes/cpu_baseline_probe.py [a|b|c]"""
import copy, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egorear_amd import configs, synth
from egorear_amd.estimator import EgoPoseFormerMVFEX
from oracle import egorear_oracle as O

mode = sys.argv[1] if len(sys.argv) > 1 else "a"
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_syn"))).eval()
synth.load_synth(net, 42)
sd = {k: v.clone() for k, v in net.state_dict().items()}
cams = O.make_cameras("ego4view_syn", os.path.join(os.path.dirname(os.path.abspath(synth.__file__)), "calib", "ego4view"))
keep = []
if mode in ("b", "c"):
    net = net.to("cuda:0")
    img = synth.synth_images(64, 4, seed=1).to("cuda:0")
    with torch.no_grad():
        net(img)
        torch.cuda.synchronize()
        if mode == "c":
            from egorear_amd.runner import PipelinedForward
            pipe = PipelinedForward(net, lanes=2, copy_inputs=False)
            pipe.prime(img)
            for _ in range(20):
                pipe(img)
            pipe.wait()
            torch.cuda.synchronize()
            keep.append(pipe)
ts = []
with torch.no_grad():
    for i in range(12):
        x = synth.synth_images(8, 4, seed=100 + i)
        t0 = time.perf_counter()
        O.mvfex_forward(sd, cams, x)
        ts.append(time.perf_counter() - t0)
print(f"mode {mode}: threads {torch.get_num_threads()} affinity {len(os.sched_getaffinity(0))} loadavg {os.getloadavg()} "
      f"per-forward s: median {statistics.median(ts[2:]):.3f} min {min(ts[2:]):.3f} max {max(ts[2:]):.3f}  -> {8 / statistics.median(ts[2:]):.1f} frames/s")
