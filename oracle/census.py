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


# two candidates of one map closer than this in the ORACLE's values are a tie within fp32 rounding of the reference's own conv stack
# (heat-map values are O(0.1-1): 1e-6 is ~8-16 ulp; the HIP path's largest heat-map deviation from the oracle is 3e-6)
ROUNDING_GAP = 1e-6


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
    worst_gap, outside, detail = 0.0, 0, []
    for si, (g, o) in enumerate(zip(gpu["hms"], ora["hms"])):
        gf, of = g.flatten(-2), o.flatten(-2)
        ag, ao = gf.argmax(-1), of.argmax(-1)
        bad = (ag != ao).nonzero()
        mism += int(bad.shape[0])
        compared += ao.numel()
        for b, v, j in bad.tolist():
            # what separates the two candidates IN THE ORACLE'S OWN MAP: a gap at rounding level means the reference's arg-max itself
            # depends on the summation order of its convolutions (another BLAS / thread count would move it too)
            gap = float(of[b, v, j, ao[b, v, j]] - of[b, v, j, ag[b, v, j]])
            gap_hip = float(gf[b, v, j, ag[b, v, j]] - gf[b, v, j, ao[b, v, j]])
            worst_gap = max(worst_gap, gap)
            outside += int(gap > ROUNDING_GAP)
            detail.append({"frame": b, "set": si, "view": v, "joint": j, "oracle_gap": gap, "hip_gap": gap_hip, "oracle_max": float(of[b, v, j, ao[b, v, j]]),
                           "idx_oracle": int(ao[b, v, j]), "idx_hip": int(ag[b, v, j])})
    out["argmax_compared"] = compared
    out["argmax_mismatches"] = mism
    out["argmax_mismatches_outside_rounding"] = outside       # oracle-side gap above ROUNDING_GAP: a real disagreement
    out["max_mismatch_oracle_gap"] = worst_gap
    out["mismatch_detail"] = detail
    out["argmax_mismatches_fp64_sides_with_hip"] = 0          # filled in by _referee for chunks that have mismatches
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
        if isinstance(v, list):
            acc[k] = acc[k] + v
        elif k.startswith("max_"):
            acc[k] = max(acc[k], v)
        elif k.endswith("_min"):
            acc[k] = min(acc[k], v)
        else:
            acc[k] += v
    return acc


def _referee(c: Dict[str, object], sd, img: torch.Tensor, O, ora) -> None:
    """A mismatch is put before the SAME oracle evaluated in float64 (weights and input are float32 numbers, every operation behind
    them in double): which of the two candidate positions does exact-ish arithmetic pick?  If float64 sides with the HIP path, the
    float32 reference's own rounding moved its arg-max."""
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    frames = sorted({d["frame"] for d in c["mismatch_detail"]})
    with torch.no_grad():
        for b in frames:
            hms64, _, _ = O.heatmap_mvfex_forward(sd64, "heatmap_estimator", img[b:b + 1].double(), 0.5)
            for d in c["mismatch_detail"]:
                if d["frame"] != b:
                    continue
                m = hms64[d["set"]][0, d["view"], d["joint"]].flatten()
                i64 = int(m.argmax())
                d["idx_fp64"] = i64
                d["fp64_gap_oracle_minus_hip"] = float(m[d["idx_oracle"]] - m[d["idx_hip"]])
                d["fp64_sides_with"] = "hip" if i64 == d["idx_hip"] else ("oracle" if i64 == d["idx_oracle"] else "neither")
    c["argmax_mismatches_fp64_sides_with_hip"] = sum(d.get("fp64_sides_with") == "hip" for d in c["mismatch_detail"])


def compare_chunk(g, o, sd, img: torch.Tensor, O, first_frame: int = 0) -> Dict[str, object]:
    """compare() + the float64 referee for whatever disagrees; `first_frame` = index of the chunk's first frame in the whole sample."""
    c = compare(g, o)
    if c["mismatch_detail"]:
        _referee(c, sd, img, O, ora=o)
        for d in c["mismatch_detail"]:
            d["frame_in_sample"] = first_frame + d["frame"]
    return c


def gpu_outputs(net, img_dev: torch.Tensor, ctm_dev: Optional[torch.Tensor] = None) -> Dict[str, object]:
    """One forward of the drop-in EgoPoseFormerMVFEX on the device; everything the census compares, on the CPU."""
    with torch.no_grad():
        preds, hms = net(img_dev) if ctm_dev is None else net(img_dev, ctm_dev)
    aux = net.__dict__["_egr_last_aux"]
    vp = aux["pose3d"].get("anchors_valid")
    return {"preds": [p.cpu() for p in preds], "hms": [h.cpu() for h in hms], "argmax_idx": aux["heatmap"]["argmax_idx"].cpu(),
            "valid_h": aux["heatmap"]["anchors_valid"].cpu(), "maxvals": aux["heatmap"]["maxvals"].cpu(),
            "valid_p": vp.cpu() if vp is not None else None}


def oracle_outputs(sd, cams, img: torch.Tensor, O, ctm: Optional[torch.Tensor] = None) -> Dict[str, object]:
    with torch.no_grad():
        preds, hms, aux = O.mvfex_forward(sd, cams, img, ctm)
    vp = aux["pose3d"].get("anchors_valid")
    return {"preds": preds, "hms": hms, "argmax_idx": aux["heatmap"]["argmax_idx"], "valid_h": aux["heatmap"]["anchors_valid"],
            "maxvals": aux["heatmap"]["maxvals"], "valid_p": vp}


def _slice(d: Dict[str, object], lo: int, hi: int) -> Dict[str, object]:
    return {k: ([t[lo:hi] for t in v] if isinstance(v, list) else (v[lo:hi] if v is not None else None)) for k, v in d.items()}


def run(net, sd, cams, O, batches: List[torch.Tensor], dev, oracle_batch: int = 8, log: Optional[Callable[[str], None]] = None,
        times: Optional[List[float]] = None, ctms: Optional[List[torch.Tensor]] = None) -> Dict[str, float]:
    """Each element of `batches` is one device batch (the benchmarked size, e.g. 64 frames): ONE HIP forward per batch under the shipped
    launch policy, the oracle over the same frames in chunks of `oracle_batch`.  `times` collects the oracle's per-chunk seconds."""
    acc = None
    for bi, img in enumerate(batches):
        ctm = ctms[bi] if ctms is not None else None       # (ego4view_rw: per-frame coord_trans_mat)
        g = gpu_outputs(net, img.to(dev), ctm.to(dev) if ctm is not None else None)
        for lo in range(0, img.shape[0], oracle_batch):
            hi = min(lo + oracle_batch, img.shape[0])
            t0 = time.perf_counter()
            o = oracle_outputs(sd, cams, img[lo:hi], O, ctm[lo:hi] if ctm is not None else None)
            if times is not None:
                times.append(time.perf_counter() - t0)
            acc = merge(acc, compare_chunk(_slice(g, lo, hi), o, sd, img[lo:hi], O, first_frame=bi * img.shape[0] + lo))
        if log:
            log(f"census: batch {bi + 1}/{len(batches)} done, {acc['frames']} frames, {acc['argmax_mismatches']} arg-max mismatches, "
                f"max joint err {acc['max_joint_err_cm']:.2e} cm")
    return acc
