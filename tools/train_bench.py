"""Training-step timing and per-kernel breakdown (config 5: ego4view_rw_pose3d, batch 32 per GPU).

    python tools/train_bench.py [--batch 32] [--steps 5]
"""
import argparse
import collections
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from egorear_amd import configs, hip, synth, train  # noqa: E402
from egorear_amd.estimator import EgoPoseFormerMVFEX  # noqa: E402
from egorear_amd.metrics import generate_target  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--graph", action="store_true", help="replay the step as one hipGraph")
    ap.add_argument("--dump", default="", help="write every launch of the instrumented step (order, us, name, tag) to this file")
    a = ap.parse_args()
    dev = "cuda:0"
    net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_rw")))
    synth.load_synth(net, 42)
    net = net.to(dev)
    tr = train.Trainer(net, use_graph=a.graph)
    B = a.batch
    img = synth.synth_images(B, 4, seed=1234).to(dev)
    ctm = synth.synth_coord_trans_mat(B).to(dev)
    gt_pose = synth.synth_gt_pose(B).to(dev)
    gt_hm = generate_target(synth.synth_joint_px(B).to(dev)).contiguous()
    for _ in range(a.warmup):
        terms, _ = tr.step(img, ctm, gt_pose, gt_hm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(a.steps):
        h0 = time.perf_counter()
        terms, _ = tr.step(img, ctm, gt_pose, gt_hm)
        host += time.perf_counter() - h0          # time the host needs to ENQUEUE a step (no sync inside)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"host enqueue time {host / a.steps * 1e3:.2f} ms/step")
    print(f"batch {B}: {dt * 1e3:.2f} ms/step, {B / dt:.1f} frames/s, loss {float(terms.sum()):.4f}, mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    # per-kernel breakdown of one step
    # eager and instrumented even when the timed steps were graph replays - and on ONE stream: with the two-stream reverse pass a
    # launch's event pair would also time whatever the other stream runs beside it
    overlap, train.OVERLAP = train.OVERLAP, False
    hip.PROFILE = []
    tr._run(img, ctm, gt_pose, gt_hm, update=True)
    torch.cuda.synchronize()
    prof, hip.PROFILE = hip.PROFILE, None
    train.OVERLAP = overlap
    if a.dump:
        with open(a.dump, "w") as f:
            for i, (name, s, e, fl, nb, tag) in enumerate(prof):
                us = s.elapsed_time(e) * 1e3
                f.write(f"{i:4d} {us:9.1f} us  {fl / us / 1e6 if fl > 0 else 0:7.1f} TF  {nb / us / 1e3 if nb > 0 else 0:7.0f} GB/s  {name:30s} {tag}\n")
    agg = collections.OrderedDict()
    for name, s, e, fl, nb, tag in prof:
        key = name
        if name == "egr_conv2d_nhwc_f32":
            key = name + (" [dgrad]" if False else "")
        ms = s.elapsed_time(e)
        cur = agg.setdefault(key, [0.0, 0, 0.0])
        cur[0] += ms
        cur[1] += 1
        cur[2] += fl
    tot = sum(v[0] for v in agg.values())
    print(f"instrumented step: {tot:.2f} ms inside {len(prof)} egr launches")
    for k, (ms, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        tf = f"{fl / ms / 1e9:7.1f} TF" if fl > 0 else ""
        print(f"  {k:32s} {ms:8.3f} ms  {n:5d} launches {tf}")
    # the slowest conv launches
    by_tag = collections.OrderedDict()
    for name, s, e, fl, nb, tag in prof:
        if "conv2d" in name:
            cur = by_tag.setdefault((name.replace("egr_", "").replace("_f32", ""), tag), [0.0, 0, 0.0])
            cur[0] += s.elapsed_time(e)
            cur[1] += 1
            cur[2] += fl
    for (name, tag), (ms, n, fl) in sorted(by_tag.items(), key=lambda kv: -kv[1][0])[:60]:
        print(f"    {ms:7.3f} ms x{n:3d} {fl / ms / 1e9 if ms > 0 else 0:7.1f} TF  {name:14s} {tag}")


if __name__ == "__main__":
    main()
