#!/usr/bin/env python3
"""egr_joint_layer_f32 alone at growing frame counts (refiner layer: C = 256, G = 4; lifting-head layer: C = 128, G = 1): one workgroup
per (query set, frame), so beyond 256 workgroups a CU runs several in turn - the second finds caches (instructions, weights) warm.
    python tools/probes/layer_scale.py"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egorear_amd import configs, engine, hip, synth
from egorear_amd.estimator import EgoPoseFormerMVFEX

net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg())).eval(); synth.load_synth(net, 42); net = net.cuda()
with torch.no_grad():
    net(synth.synth_images(2, 4, seed=1).cuda())
he = net.heatmap_estimator
dev = torch.device("cuda:0")
Pr = engine._state(he, dev).get(he.refiners()[0], lambda: None)
Pp = engine._state(net.pose3d_estimator, dev).get(net.pose3d_estimator, lambda: None)


def run(W, G, B, J, C, post=None, head=None, reps=20):
    V = 4
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(G * B * J, C, generator=gen).cuda()
    g = torch.randn(G * B * J * V, 4 * 128, generator=gen).cuda()
    sigma = torch.rand(G * 4 * B * J * V, generator=gen).cuda()
    rowmask = torch.ones(B * J * V, dtype=torch.uint8).cuda()
    for _ in range(3):
        hip.joint_layer(x, g, None, sigma, rowmask, W, B, J, V, C, G, post=post, want_xn=post is not None, head=dict(head) if head else None)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        hip.joint_layer(x, g, None, sigma, rowmask, W, B, J, V, C, G, post=post, want_xn=post is not None, head=dict(head) if head else None)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


post = {"g": Pr.post_norm[0], "b": Pr.post_norm[1]}
head = {"w": Pr.head0_w, "b": Pr.head0_b, "amax": None}
for B in (1, 16, 64, 128, 256):
    t0 = run(Pr.layer.fused, 4, B, 15, 256)
    t1 = run(Pr.layer.fused, 4, B, 15, 256, post=post, head=head)
    t2 = run(Pp.layers[0].fused, 1, B * 4, 16, 128)
    print(f"frames/set {B:4d}: refiner layer {t0:7.1f} us ({4 * B} workgroups), + post_norm + head offset {t1:7.1f} us; lifting layer x{4 * B}: {t2:7.1f} us", flush=True)
