"""Parity census: the HIP path at the benchmarked batch size and arithmetic against the CPU oracle over MANY frames.

TEST INFRASTRUCTURE (oracle/ header rule): imported by tests/ and by bench.py's cpu_baseline leg only, as the checker.

Turns "bit-exact arg-max, 3-D joints within 1e-3 cm" into counts over hundreds of frames:
  * every arg-max of both heat-map sets (init + refined: 4 views x 15 joints x 2 per frame; utils/loss.py:122-142 via
    egoposeformer_heatmap_mvf_ex.py:128-143) and the flat index the anchors are made from,
  * the `valid` masks (maxvals >= 0.5) of the refiners' anchors and of the lifting head's reprojected anchors,
  * the largest deviation of all four pose sets (proposal + three decoder layers; egoposeformer_mvf_ex.py:309-322, 546-588) in cm,
  * the tie exposure: how many (frame, view, joint) maps have their two largest values closer than 1e-5 / 1e-6 (a rounding-level
    difference could then legitimately move the arg-max), and the smallest such gap.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Optional

import torch


def _gaps(hm: torch.Tensor) -> torch.Tensor:
    """(B, V, J, H, W) -> (B, V, J): largest minus second-largest value of every map."""
    top = hm.flatten(-2).topk(2, dim=-1).values
    return top[..., 0] - top[..., 1]


def compare(gpu: Dict[str, object], ora: Dict[str, object]) -> Dict[str, float]:
    """gpu / ora: {"preds": [4 x (n,16,3)], "hms": [2 x (n,4,15,64,64)], "argmax_idx" (n,4,15), "valid_h" (n,4,15), "maxvals" (n,4,15),
    "valid_p" (n,4,16) or None} on CPU.  -> additive counts + maxima for `n` frames."""
    out: Dict[str, float] = {"frames": int(ora["hms"][0].shape[0])}
    mism = 0
    compared = 0
    for g, o in zip(gpu["hms"], ora["hms"]):
        ag, ao = g.flatten(-2).argmax(-1), o.flatten(-2).argmax(-1)
        mism += int((ag != ao).sum())
        compared += ao.numel()
    out["argmax_compared"] = compared
    out["argmax_mismatches"] = mism
    out["anchor_index_mismatches"] = int((gpu["argmax_idx"].long() != ora["argmax_idx"].long()).sum())
    out["valid_mask_mismatches"] = int((gpu["valid_h"].bool() != ora["valid_h"].bool()).sum())
    if gpu.get("valid_p") is not None and ora.get("valid_p") is not None:
        out["valid_mask_mismatches"] += int((gpu["valid_p"].bool() != ora["valid_p"].bool()).sum())
    out["valid_true"] = int(ora["valid_h"].bool().sum())
    out["valid_false"] = int((~ora["valid_h"].bool()).sum())
    out["max_joint_err_cm"] = max(float((g - o).abs().max()) for g, o in zip(gpu["preds"], ora["preds"]))
    out["max_heatmap_err"] = max(float((g - o).abs().max()) for g, o in zip(gpu["hms"], ora["hms"]))
    gaps = torch.cat([_gaps(o).flatten() for o in ora["hms"]])
    out["top2_gap_below_1e-5"] = int((gaps < 1e-5).sum())
    out["top2_gap_below_1e-6"] = int((gaps < 1e-6).sum())
    out["top2_gap_min"] = float(gaps.min())
    # how close the maxima come to the 0.5 threshold of the valid mask (a rounding-level difference could flip a mask there)
    out["maxval_to_threshold_min"] = float((ora["maxvals"] - 0.5).abs().min())
    return out


def merge(acc: Optional[Dict[str, float]], c: Dict[str, float]) -> Dict[str, float]:
    if acc is None:
        return dict(c)
    for k, v in c.items():
        if k.startswith("max_"):
            acc[k] = max(acc[k], v)
        elif k.endswith("_min"):
            acc[k] = min(acc[k], v)
        else:
            acc[k] += v
    return acc


def gpu_outputs(net, img_dev: torch.Tensor) -> Dict[str, object]:
    """One forward of the drop-in EgoPoseFormerMVFEX on the device; everything the census compares, on the CPU."""
    with torch.no_grad():
        preds, hms = net(img_dev)
    aux = net.__dict__["_egr_last_aux"]
    vp = aux["pose3d"].get("anchors_valid")
    return {"preds": [p.cpu() for p in preds], "hms": [h.cpu() for h in hms], "argmax_idx": aux["heatmap"]["argmax_idx"].cpu(),
            "valid_h": aux["heatmap"]["anchors_valid"].cpu(), "maxvals": aux["heatmap"]["maxvals"].cpu(),
            "valid_p": vp.cpu() if vp is not None else None}


def oracle_outputs(sd, cams, img: torch.Tensor, O) -> Dict[str, object]:
    with torch.no_grad():
        preds, hms, aux = O.mvfex_forward(sd, cams, img)
    vp = aux["pose3d"].get("anchors_valid")
    return {"preds": preds, "hms": hms, "argmax_idx": aux["heatmap"]["argmax_idx"], "valid_h": aux["heatmap"]["anchors_valid"],
            "maxvals": aux["heatmap"]["maxvals"], "valid_p": vp}


def _slice(d: Dict[str, object], lo: int, hi: int) -> Dict[str, object]:
    return {k: ([t[lo:hi] for t in v] if isinstance(v, list) else (v[lo:hi] if v is not None else None)) for k, v in d.items()}


def run(net, sd, cams, O, batches: List[torch.Tensor], dev, oracle_batch: int = 8, log: Optional[Callable[[str], None]] = None,
        times: Optional[List[float]] = None) -> Dict[str, float]:
    """Each element of `batches` is one device batch (the benchmarked size, e.g. 64 frames): ONE HIP forward per batch under the shipped
    launch policy, the oracle over the same frames in chunks of `oracle_batch`.  `times` collects the oracle's per-chunk seconds."""
    acc = None
    for bi, img in enumerate(batches):
        g = gpu_outputs(net, img.to(dev))
        for lo in range(0, img.shape[0], oracle_batch):
            hi = min(lo + oracle_batch, img.shape[0])
            t0 = time.perf_counter()
            o = oracle_outputs(sd, cams, img[lo:hi], O)
            if times is not None:
                times.append(time.perf_counter() - t0)
            acc = merge(acc, compare(_slice(g, lo, hi), o))
        if log:
            log(f"census: batch {bi + 1}/{len(batches)} done, {acc['frames']} frames, {acc['argmax_mismatches']} arg-max mismatches, "
                f"max joint err {acc['max_joint_err_cm']:.2e} cm")
    return acc
