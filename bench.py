#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: 4-view frames/sec of the full inference hot path
(heatmap encoders -> MVFEx/JQA refiners -> 2D-to-3D lifting), config `ego4view_syn_pose3d`,
batch 64 per GPU, synthetic 256x256x4-view input resident in HBM, random-init weights (seeded).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-graph] [--no-cpu-baseline]

N > 1 is launched by the driver with torch.distributed.run (one rank per GPU); inference shards the
batch-of-frames axis with no data-path collective ("weak" scaling: B frames per GPU).  One JSON line
is printed by rank 0.  A "step" is one forward of B frames.

Extra objects on the line:
  roofline     — dominant kernel (the implicit-GEMM conv/linear entry point; "[f16x2]" = its fp16-scheme launches, which are
                 the bulk; the chained-1x1 launches of round 5 are booked with it): FLOPs of its launches / their summed
                 duration, measured with HIP events on the launch stream in an instrumented pass right after the timed region.
                 fp16-scheme launches execute 3 fp16 MFMA products per fp32 product: achieved = 3 x algorithmic rate against the
                 16-bit dense peak (2516.6 TFLOP/s); the algorithmic rate is given beside it (`algorithmic_tflops`).
                 `families` prices the 3x3 launches (MFMA-bound) and the 1x1 launches (HBM-bound) each against its own roof;
                 the same figures as scalars: frac_conv3x3, conv3x3_ms, conv1x1_GBps, frac_conv1x1_hbm, conv1x1_ms;
                 traffic = HBM bytes per launch MEASURED IN THIS RUN (round 6; traffic_live true): before this process touches the
                 GPU it runs itself twice as a child under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` (--pmc-child:
                 three eager forwards of the benchmarked batch; separate passes, FETCH x 2 on gfx950; ~10 s each); the committed
                 summary's figure rides along as traffic_replayed (+ traffic_replayed_commit) and takes over (traffic_commit /
                 traffic_file) when rocprofv3 is unavailable, the passes fail, or --no-live-pmc / EGR_BENCH_LIVE_PMC=0 is given.
                 mfma_busy_frac / mfma_busy_frac_conv3x3: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) of the whole family /
                 of the role-split 3x3 launches, from a third child pass of the same kind.
  exact_leg    — the same forward in exact operand arithmetic (bf16x3, six products), 50 steps: prices the 22-bit trade.
  one_lane     — the same forward as ONE captured graph replayed back to back (what the second lane's overlap is worth).
  cpu_baseline — the CPU oracle (oracle/egorear_oracle.py, a PyTorch-CPU port of the reference path) timed on this box's host
                 cores on a bounded sample of the same workload: distinct frames, value = batch / MEDIAN forward time after two
                 warm-ups; torch_num_threads, host_loadavg_1min and value_least_disturbed say how loaded the (shared) host was.
  parity_vs_cpu_oracle — the metric's "MPJPE vs ref" half as a CENSUS over that same sample (oracle/census.py): the HIP path at the
                 benchmarked batch size and launch policy against every oracle forward - all arg-maxes of both heat-map sets, the
                 valid masks, the four pose sets, the tie exposure (top-2 gaps), every mismatch with the oracle-side gap and a
                 float64 referee; MPJPE of both against the seeded synthetic ground truth.
  scalars      — every leg once more as scalar keys (exact_fps, steady_fps, one_lane_fps, cfg2_fps, cfg3_fps, train_ms_per_step,
                 train_frac, parity_frames, argmax_mismatches[_outside_rounding], max_joint_err_cm, frac_conv3x3, ...): at the top
                 level, last in the line, and at the front of `roofline` (the driver's record keeps roofline / cpu_baseline / config).
  ranks        — world size, backend, and one record per rank (device index, PCI address / UUID, host, own frames/s): an N > 1 line
                 can be checked for one distinct GPU per rank (the run refuses to start otherwise unless EGR_ALLOW_SHARED_GPU=1).
  configs      — the other single-GPU configurations BASELINE.json lists, timed separately at N = 1 (never part of `value`):
                 config 2 (two stereo heat-map estimators, batch 32), config 3 (HeatmapMVFEX, batch 32), each with its own
                 parity object against the CPU oracle on a small sample, and the CPU leg of config 1 (batch 1, two views).
  roofline_hbm — the HBM-bound kernels of the forward: algorithmic bytes / measured time against 8 TB/s.
  preprocess, train — the SURVEY.md §8(f) legs (raw 872x872 uint8 -> model input; the config-5 optimisation step),
                 timed separately, never part of `value`.  preprocess.from_raw_frames = the whole chain from raw
                 872x872x4-view uint8 frames in HBM (pre-processing + forward of the same B frames).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GFLOP_PER_FRAME = 64.36      # algorithmic conv+matmul FLOPs per 4-view frame
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2516.6  # v_mfma_f32_32x32x16_bf16: 32 cycles per 32x32x16 on 1024 SIMDs at 2.4 GHz (guide: "~2.5 PF dense")
X6_TERMS = 6                   # bf16 products per fp32 product in the split-bf16 launches
H2_TERMS = 3                   # fp16 products per fp32 product in the fp16-scheme launches; v_mfma_f32_32x32x16_f16 has the bf16 form's rate
W_FORMAT = os.environ.get("EGR_W_FORMAT", "f16x2")
PEAK_HBM_GBS = 8000.0
# the arithmetic type the path computes in (fp32 tensors and accumulators; what the large contractions multiply in follows EGR_W_FORMAT)
DTYPE_NOTE = {"f16x2": "f32 (fp16x2 split, 22-bit operands, fp32 accumulate)", "bf16x3": "f32 (bf16x3 split, exact operands, fp32 accumulate)"}.get(
    os.environ.get("EGR_W_FORMAT", "f16x2"), "f32")
ARITHMETIC = {
    "f16x2": ("fp32 tensors, fp32 accumulation; conv / linear contractions with >= 4096 rows run on the fp16 matrix cores: both operands as two "
              "fp16 planes of the value times an exact power of two (22 significant bits; pre-scale from the producing launch's abs-max record), "
              "the three products (l,h) (h,l) (h,h) kept - measured as close to the exact sum as an fp32 fma chain (tests/test_gpu_conv_h2.py, "
              "tools/proto/f16x3_accuracy.hip); inference and the training step's forward, data-gradient and weight-gradient launches alike (training: "
              "a record is made with one extra read where the producing launch keeps none); launches whose input has no record use the exact "
              "three-way bf16 split (six products); "
              "EGR_W_FORMAT=bf16x3 / f32 select the bf16 scheme / the fp32 matrix cores everywhere"),
    "bf16x3": ("fp32 tensors, fp32 accumulation; conv / linear contractions with >= 4096 rows run on the bf16 matrix cores with every fp32 "
               "operand split exactly into three bf16 and the six products of order <= 2 kept (as close to the exact sum as an fp32 fma "
               "chain: tests/test_gpu_conv_x6.py, tools/proto/gemm_bf16x6.hip)"),
}.get(W_FORMAT, "fp32 tensors, fp32 matrix cores (EGR_W_FORMAT=f32)")


def _kernel_key(name: str, tag: str) -> str:
    """Profile key of a launch: the implicit-GEMM entry point runs three kernel families - the fp16 scheme (tag "h2"), the split-bf16
    one ("x6") and the fp32-matrix-core one.  egr_conv1x1_chain_f32 (two 1x1 launches of the fp16 scheme fused into one, round 5) is
    booked with that entry point: same family, same roofs - its algorithmic bytes no longer contain the intermediate tensor."""
    if name == "egr_conv1x1_chain_f32":
        name = "egr_conv2d_nhwc_f32"
    if tag.startswith("h2 ") or " h2 " in tag:
        return name + "[f16x2]"
    if tag.startswith("x6 ") or " x6 " in tag:
        return name + "[bf16x3]"
    return name


def _family(key: str, tag: str):
    """The implicit-GEMM entry point's split launches fall into two families with different roofs: the 3x3 convolutions (matrix-core
    bound: conv_tapx_kernel / conv_igemm_tap[2]_kernel) and the 1x1 convolutions / linears (HBM bound at these channel counts:
    conv_pw_x6_kernel and the tiled kernels)."""
    if not (key.endswith("[f16x2]") or key.endswith("[bf16x3]")):
        return None
    fmt = key[key.index("["):]
    if " k3s" in tag:
        return "conv3x3" + fmt
    if " k1s" in tag:
        return "conv1x1" + fmt
    return "other" + fmt


def _family_roofs(fams: dict) -> dict:
    """Each family of the dominant entry point against ITS OWN roof, from the same instrumented forward as `roofline`."""
    out = {}
    for name, k in sorted(fams.items()):
        if k["ms"] <= 0:
            continue
        terms = H2_TERMS if name.endswith("[f16x2]") else X6_TERMS
        alg = k["flops"] / (k["ms"] * 1e-3) / 1e12
        gbs = k["bytes"] / (k["ms"] * 1e-3) / 1e9
        mfma = {"achieved": round(alg * terms, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(alg * terms / PEAK_BF16_MFMA_TFLOPS, 4),
                "algorithmic_tflops": round(alg, 1), "frac_algorithmic": round(alg / PEAK_BF16_MFMA_TFLOPS, 4)}
        hbm = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
               "what": "algorithmic bytes (input once, output once, weights once) / measured time"}
        bound = "mfma" if name.startswith("conv3x3") else "hbm"
        out[name] = {"bound": bound, **(mfma if bound == "mfma" else hbm), "other_roof": hbm if bound == "mfma" else mfma,
                     "launches_per_step": k["launches"], "kernel_ms_per_step": round(k["ms"], 3)}
    return out


def exact_leg(net, img, B: int, lanes: int, steps: int = 50):
    """The 22-bit trade of the headline, priced in the same run: frames/s of the SAME forward (same weights, batch, lanes, hipGraph
    replay) in exact operand arithmetic - every fp32 operand of the large contractions as the exact sum of three bf16 numbers, six
    products (EGR_W_FORMAT=bf16x3), the fused transformer layers on the fp32 matrix cores.  Never `value`."""
    import torch
    from egorear_amd import engine, hip
    from egorear_amd.runner import PipelinedForward
    # a second LaunchPolicy held by the module for the duration of this leg (engine.set_policy): nothing process-global is flipped,
    # the packs of this module alone are rebuilt for the other weight format
    try:
        engine.set_policy(net, hip.policy().exact())
        with torch.no_grad():
            net(img)
            torch.cuda.synchronize()
            if lanes > 1:
                pipe = PipelinedForward(net, lanes=lanes, copy_inputs=False)
                pipe.prime(img)
                ms = _gpu_time_ms(lambda: pipe(img), 5, steps, join=pipe.wait)
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    net(img)
                ms = _gpu_time_ms(g.replay, 5, steps)
        return {"value": round(B / ms * 1e3, 2), "unit": "frames/s", "ms_per_step": round(ms, 3), "steps": steps, "batch_per_gpu": B,
                "dtype": "f32 (bf16x3 split, exact operands, fp32 accumulate)",
                "what": "same forward, weights, batch and launch mode as `value` with EGR_W_FORMAT=bf16x3 arithmetic: what the fp16 scheme's "
                        "22-bit operands buy (parity of this leg: tests/test_gpu_bf16x3_leg.py)"}
    finally:
        engine.set_policy(net, None)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step (weak scaling: fixed as N grows)")
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: frames per step over ALL GPUs (split evenly over the ranks); "
                                                               "0 = weak scaling with --batch frames per GPU")
    ap.add_argument("--steady-steps", type=int, default=400, help="steps of the steady-state leg behind the timed burst (0 = skip)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("EGR_BENCH_LANES", "2")),
                    help="captured forwards in flight (runner.PipelinedForward): consecutive steps overlap on that many streams")
    ap.add_argument("--pmc-child", action="store_true", help="(internal) the profiled workload of the live PMC passes: --steps eager forwards, nothing else")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not measure roofline.traffic in this run (two rocprofv3 child passes, ~40 s); "
                                                               "replay the committed PMC summary instead")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-iters", type=int, default=30)      # + 2 warm-ups = 32 forwards of 8 = 256 frames = four GPU batches of 64; ~15 s on the GPU box's 16 host threads
    ap.add_argument("--no-train", action="store_true", help="skip the training-step leg (config 5)")
    ap.add_argument("--no-configs", action="store_true", help="skip the config 1-3 legs")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-arithmetic (bf16x3) leg")
    ap.add_argument("--train-batch", type=int, default=32, help="frames per GPU per optimisation step (config 5: 256 / 8)")
    ap.add_argument("--train-steps", type=int, default=10)
    return ap.parse_args()


def _host_cores() -> int:
    """CPU share of this process (the GPU box exposes a slice of a large host: use the affinity mask, cap at 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("EGR_CPU_THREADS", "16"))))


def _cpu_model() -> str:
    """Model name of the host CPU the baselines run on ("core count and CPU model printed")."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def _natural(path: str):
    """Sort key that orders r02_v10 after r02_v9 (digits compared as numbers)."""
    import re
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(path))]


def _pmc_traffic(batch: int, fmt: str = "", launches_per_step: int = 0):
    """HBM bytes per launch of the conv kernel from the committed rocprofv3 PMC passes (profiles/*pmc_traffic.json,
    produced by tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs of this same command, with the
    gfx950 x2 correction on FETCH_SIZE).  Counters cannot be collected from inside the timed run; null if the file
    is absent, was measured at another batch size, or counted another number of launches of this kernel per forward than
    this run makes (a stale file: the launch rule or the kernels changed since the passes were collected).
    Returns (bytes per launch, provenance) - the provenance names the file and commit the figure is replayed from."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*pmc_traffic.json")), key=_natural)
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            t = json.load(f)
        if t.get("batch") != batch:
            return None, None
        e = t
        if fmt and t.get("by_format"):      # per weight format of the implicit-GEMM kernel ("f16x2" / "bf16x3" / "f32")
            e = t["by_format"].get(fmt)
        if not e:
            return None, None
        if launches_per_step and e.get("launches_per_forward") not in (None, launches_per_step):
            return None, None
        # provenance: PMC counters cannot be read inside the timed run - the value is REPLAYED from the committed passes of this command
        return e["hbm_bytes_per_launch"], {"replayed_from": os.path.relpath(files[-1], REPO), "commit": t.get("commit"),
                                           "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (FETCH x 2 on gfx950); null when the launch count of this run differs"}
    except Exception:
        return None, None


def _pmc_family(kernel_name: str) -> str:
    """Which entry of the traffic summary a kernel belongs to (the rule of tools/pmc_traffic.py)."""
    import re
    nm = kernel_name
    tapx = "conv_tapx_kernel" in nm or "conv_pw_chain_kernel" in nm or "conv_pw2_kernel" in nm
    split = "conv_igemm_x6" in nm or "conv_igemm_tap" in nm or "conv_pw_x6" in nm or tapx
    h2 = tapx or (split and re.search(r",\s*2>\(", nm.replace(") ", ")")) is not None)
    if "conv_igemm" in nm or "conv_pw_x6" in nm or tapx:
        return ("f16x2" if h2 else "bf16x3") if split else "f32"
    return "other"


def _pmc_is_tapx3x3(kernel_name: str) -> bool:
    """conv_tapx_kernel<WM, WN, FN, STRIDE, ...> with STRIDE 1 / 2: the role-split 3x3 launches (STRIDE 0 = its wide 1x1 mode)."""
    import re
    m = re.search(r"conv_tapx_kernel<\s*\d+\s*,\s*\d+\s*,\s*\d+\s*,\s*(\d+)", kernel_name)
    return bool(m) and m.group(1) != "0"


def live_pmc_traffic(batch: int, forwards: int = 3, timeout_s: int = 150):
    """HBM bytes per launch and matrix-pipe utilisation of the implicit-GEMM kernels MEASURED IN THIS RUN: three child processes of this
    same script under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` / `SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` (separate
    passes, as MI355X_MICROARCH.md prescribes; FETCH_SIZE x 2 on gfx950, both in KB; MFMA-busy cycles are summed over the chip's 1024
    SIMDs, GRBM_GUI_ACTIVE over its 8 XCDs), each running `forwards` eager forwards of the benchmarked batch and nothing else
    (--pmc-child).  Called BEFORE this process touches the GPU.  Returns ({family: {"hbm_bytes_per_launch", "launches_per_forward",
    "mfma_busy_frac", ...}}, provenance) or (None, reason); family "tapx3x3" = the role-split 3x3 launches alone."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None, "rocprofv3 not found"
    try:
        root = tempfile.mkdtemp(prefix="egr_pmc_", dir="/tmp")
    except OSError as exc:
        return None, f"no scratch directory: {exc}"
    env = dict(os.environ, TMPDIR="/tmp")
    sums = collections.defaultdict(lambda: collections.defaultdict(float))      # counter -> family -> sum
    seen = collections.defaultdict(lambda: collections.defaultdict(set))        # counter -> family -> dispatch ids
    try:
        for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")):
            out = os.path.join(root, counters[0].lower())
            cmd = [rp, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--batch", str(batch), "--steps", str(forwards)]
            _log(f"live PMC pass {' '.join(counters)}: rocprofv3 --kernel-trace --pmc ... -- python3 bench.py --pmc-child")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
            if r.returncode != 0 or not files:
                if counters[0] == "SQ_VALU_MFMA_BUSY_CYCLES":
                    break                      # (the traffic passes stand on their own)
                return None, f"{counters[0]} pass failed (rc {r.returncode}): {r.stdout.decode(errors='replace')[-300:]}"
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    c = row["Counter_Name"]
                    if c not in counters:
                        continue
                    fams = [_pmc_family(row["Kernel_Name"])]
                    if _pmc_is_tapx3x3(row["Kernel_Name"]):
                        fams.append("tapx3x3")
                    for k in fams:
                        sums[c][k] += float(row["Counter_Value"])
                        seen[c][k].add(row["Dispatch_Id"])
    except Exception as exc:       # (timeout, unreadable CSV ...): the replayed figure stays in charge
        return None, f"{type(exc).__name__}: {exc}"
    finally:
        shutil.rmtree(root, ignore_errors=True)
    res = {}
    for fam, ids in seen["FETCH_SIZE"].items():
        n, nw = len(ids), len(seen["WRITE_SIZE"].get(fam, ()))
        if fam == "other" or n == 0 or nw == 0:
            continue
        e = {"hbm_bytes_per_launch": 2.0 * sums["FETCH_SIZE"][fam] * 1024 / n + sums["WRITE_SIZE"][fam] * 1024 / nw,
             "launches_per_forward": n / forwards, "launches_counted": n}
        gui = sums["GRBM_GUI_ACTIVE"].get(fam, 0.0)
        if gui > 0:
            e["mfma_busy_frac"] = sums["SQ_VALU_MFMA_BUSY_CYCLES"][fam] / (gui / 8.0 * 1024.0)
        res[fam] = e
    prov = {"measured_in_this_run": True, "forwards_profiled": forwards,
            "note": "child processes of bench.py (--pmc-child: eager forwards of the benchmarked batch only) under rocprofv3 --kernel-trace --pmc "
                    "FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, separate passes, FETCH x 2 on gfx950, before this process touched the GPU"}
    return res, prov


def pmc_child(args):
    """--pmc-child: `--steps` eager forwards of `--batch` frames, nothing else (the workload of live_pmc_traffic's profiled passes)."""
    import torch
    from egorear_amd import configs, synth
    from egorear_amd.estimator import EgoPoseFormerMVFEX
    dev = torch.device("cuda", 0)
    net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_syn"))).eval()
    synth.load_synth(net, 42)
    net = net.to(dev)
    img = synth.synth_images(args.batch, 4, seed=1234).to(dev)
    with torch.no_grad():
        for _ in range(args.steps):
            net(img)
    torch.cuda.synchronize()


def _pmc_traffic_train(batch: int, key: str):
    """HBM bytes per launch of a training-step kernel from the committed PMC passes (profiles/*pmc_traffic_train.json, produced by
    tools/pmc_traffic_train.py from separate FETCH_SIZE / WRITE_SIZE runs of tools/train_bench.py); null if absent / other batch.
    Returns (bytes per launch, provenance)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*pmc_traffic_train.json")), key=_natural)
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            t = json.load(f)
        if t.get("batch") != batch:
            return None, None
        name = {"egr_conv2d_nhwc_f32[f16x2]": "conv_igemm_f16x2", "egr_conv2d_nhwc_f32[bf16x3]": "conv_igemm_bf16x3",
                "egr_conv2d_nhwc_f32": "conv_igemm_f32", "egr_conv2d_wgrad_f32[f16x2]": "conv_wgrad_f16x2",
                "egr_conv2d_wgrad_f32": "conv_wgrad_bf16x3"}.get(key)
        e = t["kernels"].get(name) if name else None
        if not e:
            return None, None
        return e["hbm_bytes_per_launch"], {"replayed_from": os.path.relpath(files[-1], REPO), "commit": t.get("commit"),
                                           "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/train_bench.py (FETCH x 2 on gfx950)"}
    except Exception:
        return None, None


def _roofline(key: str, k: dict, traffic, traffic_source=None):
    """Roofline object of one profiled kernel.  Split launches execute X6_TERMS bf16 (H2_TERMS fp16) MFMA products per algorithmic
    fp32 product: `achieved` is the executed 16-bit matrix-core rate against the bf16 / fp16 dense peak; the algorithmic
    (fp32-equivalent) rate and the fp32-matrix-core peak it would otherwise be priced against are given beside it."""
    alg = k["flops"] / (k["ms"] * 1e-3) / 1e12
    x6, h2 = key.endswith("[bf16x3]"), key.endswith("[f16x2]")
    terms = X6_TERMS if x6 else (H2_TERMS if h2 else 1)
    achieved, peak = (alg * terms, PEAK_BF16_MFMA_TFLOPS) if (x6 or h2) else (alg, PEAK_F32_MFMA_TFLOPS)
    r = {"bound": "mfma", "kernel": key, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
         "traffic": traffic, "launches_per_step": k["launches"], "avg_launch_us": round(1e3 * k["ms"] / k["launches"], 2),
         "flops_per_launch": round(k["flops"] / k["launches"], 1), "algorithmic_bytes_per_launch": round(k["bytes"] / k["launches"], 1),
         "kernel_ms_per_step": round(k["ms"], 3), "algorithmic_tflops": round(alg, 2),
         "frac_algorithmic": round(alg / (PEAK_BF16_MFMA_TFLOPS if (x6 or h2) else PEAK_F32_MFMA_TFLOPS), 4),
         "traffic_source": traffic_source if traffic is not None else None,
         # the same computation priced against the roof it faced before (and still faces with EGR_W_FORMAT=f32)
         "algorithmic_frac_of_f32_mfma_peak": round(alg / PEAK_F32_MFMA_TFLOPS, 4)}
    if x6:
        # measured with tools/proto/mfma_ceiling (profiles/r01_v17_mfma_ceiling_and_ring.txt): register operands only, random data
        r["bare_mfma_loop_tflops_measured"] = 1700.0
        r["matrix_core_path"] = (f"fp32 operands as exact sums of three bf16; {X6_TERMS} bf16 MFMA products per fp32 product, fp32 accumulate; "
                                 f"achieved = {X6_TERMS} x algorithmic rate; the fp32 matrix cores peak at {PEAK_F32_MFMA_TFLOPS} TFLOP/s")
    if h2:
        # measured with tools/proto/f16x3_accuracy (profiles/r03_v1_proto_f16x3_accuracy.txt): register operands only, random data
        r["bare_mfma_loop_tflops_measured"] = 1595.0
        r["matrix_core_path"] = (f"fp32 operands as two fp16 planes of the power-of-two-scaled value (22 significant bits); {H2_TERMS} fp16 MFMA products "
                                 f"per fp32 product (v_mfma_f32_32x32x16_f16, the bf16 form's rate), fp32 accumulate; achieved = {H2_TERMS} x algorithmic "
                                 f"rate; the fp32 matrix cores peak at {PEAK_F32_MFMA_TFLOPS} TFLOP/s")
    return r


def _log(msg: str):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(state_dict, net, dev, batch: int, iters: int, gpu_batch: int = 64, budget_s: float = 22.0):
    """The CPU oracle (kind 'port') timed on a bounded sample AND used as the checker of the metric's "MPJPE vs ref" half, in one
    pass: `iters` + 2 forwards of `batch` DISTINCT seeded frames each (the first two are warm-ups: compared, not timed).  The same
    frames go through the HIP path in batches of `gpu_batch` (the benchmarked batch size and launch policy) and every oracle forward
    is compared with its slice (oracle/census.py: arg-max of both heat-map sets, valid masks, all four pose sets, top-2 gap census).
    `value` = batch / MEDIAN per-forward time (a wall-clock total moved 2x between boxes: one slow forward of a noisy neighbour)."""
    import statistics

    import torch
    from egorear_amd import synth
    from oracle import census
    from oracle import egorear_oracle as O
    cores = _host_cores()
    torch.set_num_threads(cores)
    cams = O.make_cameras("ego4view_syn", os.path.join(REPO, "egorear_amd", "calib", "ego4view"))
    n_fwd = iters + 2
    frames = n_fwd * batch
    n_gpu = -(-frames // gpu_batch)
    # alternating image scales 1.0 / 0.35 (the reference goldens' two settings: maxima on both sides of the 0.5 threshold)
    batches = [synth.synth_images(gpu_batch, 4, seed=1234 + i, scale=(1.0 if i % 2 == 0 else 0.35)) for i in range(n_gpu)]
    flat = torch.cat(batches)[:frames]
    left = frames - (n_gpu - 1) * gpu_batch
    batches[-1] = batches[-1][:left] if left < gpu_batch else batches[-1]
    gt = synth.synth_gt_pose(frames, seed=1235)
    times, acc = [], None
    mp_hip = mp_cpu = 0.0
    _log(f"cpu baseline + parity census: {n_fwd} oracle forwards of batch {batch} ({frames} frames), {cores} threads (torch.get_num_threads() = {torch.get_num_threads()})")
    with torch.no_grad():
        pos = 0
        for bi, imgb in enumerate(batches):
            g = census.gpu_outputs(net, imgb.to(dev))
            for lo in range(0, imgb.shape[0], batch):
                hi = min(lo + batch, imgb.shape[0])
                t0 = time.perf_counter()
                o = census.oracle_outputs(state_dict, cams, imgb[lo:hi], O)
                times.append(time.perf_counter() - t0)
                gs = census._slice(g, lo, hi)
                acc = census.merge(acc, census.compare_chunk(gs, o, state_dict, imgb[lo:hi], O, first_frame=pos))
                mp_hip += float(O.compute_mpjpe_batch(gs["preds"][-1], gt[pos:pos + hi - lo]).sum())
                mp_cpu += float(O.compute_mpjpe_batch(o["preds"][-1], gt[pos:pos + hi - lo]).sum())
                pos += hi - lo
            _log(f"  batch {bi + 1}/{n_gpu}: {acc['frames']} frames, {acc['argmax_mismatches']} arg-max mismatches, max joint err {acc['max_joint_err_cm']:.2e} cm")
            if sum(times[2:]) > budget_s and len(times) >= 10:
                break          # bounded sample: a loaded host (the box shares its CPU: see loadavg) gets fewer frames, not a longer run
    frames = pos
    timed = times[2:]
    med = statistics.median(timed)
    parity = {"frames": acc["frames"], "gpu_batch": gpu_batch, "argmax_equal": acc["argmax_mismatches"] == 0 and acc["anchor_index_mismatches"] == 0,
              **{k: v for k, v in acc.items() if k != "frames"},
              "mpjpe_mm_hip": mp_hip / frames * 10, "mpjpe_mm_cpu_oracle": mp_cpu / frames * 10, "tolerance_cm": 1e-3,
              "what": "HIP path at the benchmarked batch size / launch policy vs the CPU oracle on the same frames (oracle/census.py); "
                      "argmax_compared = both heat-map sets x 4 views x 15 joints per frame; image scales 1.0 and 0.35 alternate per GPU batch"}
    return {"value": round(batch / med, 3), "unit": "frames/s", "cores": cores, "kind": "port", "cpu_model": _cpu_model(),
            "torch_num_threads": torch.get_num_threads(), "host_loadavg_1min": round(os.getloadavg()[0], 1), "host_cpus_visible": len(os.sched_getaffinity(0)),
            "value_least_disturbed": round(batch / min(timed), 3), "forward_s_median": round(med, 4), "forward_s_min": round(min(timed), 4),
            "forward_s_max": round(max(timed), 4),
            "sample": f"{len(timed)} forwards of batch {batch}, distinct frames (config ego4view_syn_pose3d, eval/no_grad, torch-CPU fp32, "
                      f"{cores} threads) after 2 warm-up forwards; value = batch / median forward time; {sum(timed):.1f} s timed; the box's host CPU is shared "
                      f"(host_loadavg_1min): value_least_disturbed = batch / fastest forward"}, parity


def cpu_train_baseline(batch: int = 4, iters: int = 7):
    """The training oracle (oracle/train_oracle.py: the reference's step restated on PyTorch-CPU autograd, kind 'port') timed on
    the host cores on a bounded sample: `iters` forward + backward + clip + AdamW steps of `batch` frames (after one warm-up)."""
    import torch
    from egorear_amd import configs, synth
    from egorear_amd.estimator import EgoPoseFormerMVFEX
    from oracle import egorear_oracle as O
    from oracle import train_oracle as TO
    cores = _host_cores()
    torch.set_num_threads(cores)
    net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_rw")))
    sd = synth.load_synth(net, 42)
    names = [k for k, _ in net.named_parameters()]
    ref = TO.OracleTrainer(sd, names, O.make_cameras("ego4view_rw", os.path.join(REPO, "egorear_amd", "calib", "ego4view")))
    args_ = (synth.synth_images(batch, 4, seed=5), synth.synth_coord_trans_mat(batch), synth.synth_gt_pose(batch), TO.synth_gt_heatmap(batch))
    _log(f"cpu training baseline: warm-up step, batch {batch}, {cores} threads")
    ref.step(*args_)
    _log("cpu training baseline: timing")
    t0 = time.perf_counter()
    for _ in range(iters):
        ref.step(*args_)
    dt = time.perf_counter() - t0
    return {"value": round(batch * iters / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port", "cpu_model": _cpu_model(),
            "sample": f"{iters} optimisation steps of batch {batch} (train-mode forward, autograd backward, clip, torch AdamW; torch-CPU fp32, {cores} threads), {dt:.1f} s"}


HBM_KERNELS = ("egr_maxpool_nhwc_f32", "egr_upsample2x_nhwc_f32", "egr_argmax_rows_f32", "egr_msda_gather_f32", "egr_avgpool_nhwc_f32",
               "egr_layernorm_f32", "egr_up2_relu_head_f32", "egr_stem_conv7x7_f32", "egr_stem_conv7x7_x6_f32[bf16x3]",
               "egr_stem_conv7x7_x6_f32[f16x2]", "egr_linear_wstream_f32[f16x2]", "egr_absmax_f32")


def roofline_hbm(kernels: dict, pre_leg) -> list:
    """bytes / time of the memory-bound kernels of one forward against the 8 TB/s HBM peak (algorithmic bytes: inputs read once +
    outputs written once; the gather kernel's figure counts every sampled corner, most of which L2 serves)."""
    out = []
    for name in HBM_KERNELS:
        k = kernels.get(name)
        if k and k["bytes"] > 0 and k["ms"] > 0:
            gbs = k["bytes"] / k["ms"] / 1e6
            e = {"kernel": name, "launches_per_step": k["launches"], "ms_per_step": round(k["ms"], 3), "algorithmic_GBps": round(gbs, 1),
                 "peak": PEAK_HBM_GBS, "frac": round(gbs / PEAK_HBM_GBS, 4)}
            if name == "egr_msda_gather_f32":
                # Priced on UNIQUE bytes (what HBM has to deliver: every feature map once + the per-row operands and outputs); the
                # sampled-corner rate (every corner of every sample counted) is an L2 gather rate and is reported beside it, against
                # no HBM peak.  k["unique_bytes"] is filled in from the launch tags by the caller.
                ub = k.get("unique_bytes", 0.0)
                e["l2_gather_GBps"] = e.pop("algorithmic_GBps")
                e["unique_bytes_GBps_upper_bound"] = round(ub / k["ms"] / 1e6, 1)
                e["frac"] = None                 # not an HBM-bound kernel: no fraction of the HBM roof is claimed for it
                e["bound"] = "l2 gather"
                e["note"] = ("unique_bytes_GBps_upper_bound: an UPPER bound of the unique bytes (every feature map and positional table once + offsets / logits / "
                             "anchors + sampled rows out; the samples touch only part of each map, and the maps were written by the launches just before - "
                             "the 256-MB infinity cache serves most of them) against the HBM peak - the kernel is not HBM-bound; l2_gather_GBps: sampled-corner bytes (a 64x64x128 map, 2 MB per view, is re-read 60-64 "
                             "times per frame out of L2) - an L2 gather rate, priced against no HBM roof")
            if name in ("egr_stem_conv7x7_f32", "egr_stem_conv7x7_x6_f32[bf16x3]", "egr_stem_conv7x7_x6_f32[f16x2]", "egr_up2_relu_head_f32"):
                e["note"] = "matrix-core kernel with a memory-bound output side; its MFMA rate is in kernel_ms / DESIGN.md"
            out.append(e)
    if pre_leg:
        out.append({"kernel": "egr_preprocess_u8_f32", "ms_per_step": pre_leg["ms_per_batch"], "algorithmic_GBps": pre_leg["algorithmic_GBps"],
                    "peak": PEAK_HBM_GBS, "frac": pre_leg["frac_of_8TBps"]})
    return out


def _gpu_time_ms(fn, warmup: int, iters: int, join=None) -> float:
    """HIP events around `iters` calls on the current stream; `join` (PipelinedForward.wait) makes the current stream wait for the
    lanes' streams before the closing event."""
    import torch
    for _ in range(warmup):
        fn()
    if join is not None:
        join()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    if join is not None:
        join()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def config_legs(args, dev, with_cpu: bool):
    """BASELINE.json configs 2 and 3 on this GPU (batch 32, hipGraph replay like the headline, HIP events) and the CPU leg of
    config 1.  Never part of `value`.  Each GPU leg carries its own parity object: the CPU oracle on a small sample of the same
    seeded frames."""
    import torch
    from egorear_amd import configs, synth
    from egorear_amd.estimator import EgoPoseFormerHeatmapMVFEX
    from egorear_amd.runner import GraphedForward
    from oracle import egorear_oracle as O
    B, PB = 32, 2
    net = EgoPoseFormerHeatmapMVFEX(**copy.deepcopy(configs.heatmap_mvfex_cfg("ego4view_syn"))).eval()
    synth.load_synth(net, 42)
    cpu_sd = {k: v.clone() for k, v in net.state_dict().items()} if with_cpu else None
    net = net.to(dev)
    img = synth.synth_images(B, 4, seed=1234).to(dev)
    front, back = net.heatmap_estimator_stereo_front, net.heatmap_estimator_stereo_back
    img_f, img_b = img[:, 0:2].contiguous(), img[:, 2:4].contiguous()
    legs = {}
    with torch.no_grad():
        lanes = max(1, args.lanes)
        if lanes > 1:      # like the headline: consecutive steps overlap on `lanes` streams
            from egorear_amd.runner import PipelinedForward
            g_front, g_back, g_net = (PipelinedForward(m, lanes=lanes, copy_inputs=False) for m in (front, back, net))
            g_front.prime(img_f); g_back.prime(img_b); g_net.prime(img)
            ms2 = _gpu_time_ms(lambda: (g_front(img_f), g_back(img_b)), 3, 20, join=lambda: (g_front.wait(), g_back.wait()))
            ms3 = _gpu_time_ms(lambda: g_net(img), 3, 20, join=g_net.wait)
        else:
            g_front, g_back, g_net = GraphedForward(front), GraphedForward(back), GraphedForward(net)
            ms2 = _gpu_time_ms(lambda: (g_front(img_f), g_back(img_b)), 3, 20)
            ms3 = _gpu_time_ms(lambda: g_net(img), 3, 20)
        launch_note = "hipGraph replay" if lanes <= 1 else f"{lanes} captured forwards in flight on {lanes} streams"
        legs["config2_heatmap_4view"] = {
            "workload": "ego4view_syn_heatmap_stereo_front + stereo_back: two EgoPoseFormerHeatmap estimators (ResNet18 + FPN + 1x1 head), views 0-1 / 2-3, eval/no_grad",
            "value": round(B / ms2 * 1e3, 1), "unit": "frames/s", "ms_per_step": round(ms2, 3), "batch": B, "launch": launch_note + " (one graph per estimator)",
            "algorithmic_gflop_per_frame": 27.27, "path_tflops": round(B / ms2 * 27.27, 2)}
        legs["config3_heatmap_mvfex"] = {
            "workload": "ego4view_syn_heatmap_mvfex-n1_jqa: EgoPoseFormerHeatmapMVFEX (2 encoders, init heads, 4 MVFEx/JQA refiners), eval/no_grad",
            "value": round(B / ms3 * 1e3, 1), "unit": "frames/s", "ms_per_step": round(ms3, 3), "batch": B, "launch": launch_note,
            "algorithmic_gflop_per_frame": 61.11, "path_tflops": round(B / ms3 * 61.11, 2)}
        if with_cpu:
            cores = _host_cores()
            torch.set_num_threads(cores)
            sample = synth.synth_images(PB, 4, seed=1234)
            g_f, g_b = front(sample[:, 0:2].contiguous().to(dev)).cpu(), back(sample[:, 2:4].contiguous().to(dev)).cpu()
            g_hms, _ = net(sample.to(dev))
            o_f = O.heatmap_forward(cpu_sd, "heatmap_estimator_stereo_front", sample[:, 0:2])
            o_b = O.heatmap_forward(cpu_sd, "heatmap_estimator_stereo_back", sample[:, 2:4])
            o_hms, _, _ = O.heatmap_mvfex_forward(cpu_sd, "", sample)

            def am(t):
                return t.flatten(-2).argmax(-1)
            legs["config2_heatmap_4view"]["parity_vs_cpu_oracle"] = {
                "frames": PB, "argmax_equal": bool((am(g_f) == am(o_f)).all() and (am(g_b) == am(o_b)).all()),
                "argmax_compared": int(am(o_f).numel() + am(o_b).numel()),
                "max_heatmap_err": float(max((g_f - o_f).abs().max(), (g_b - o_b).abs().max()))}
            legs["config3_heatmap_mvfex"]["parity_vs_cpu_oracle"] = {
                "frames": PB, "argmax_equal": bool(all((am(a.cpu()) == am(b)).all() for a, b in zip(g_hms, o_hms))),
                "argmax_compared": int(sum(am(b).numel() for b in o_hms)),
                "max_heatmap_err": float(max((a.cpu() - b).abs().max() for a, b in zip(g_hms, o_hms)))}
            # config 1: the reference's own CPU-runnable case - one stereo estimator, batch 1, two views
            one = sample[:1, 0:2]
            O.heatmap_forward(cpu_sd, "heatmap_estimator_stereo_front", one)
            n_it = 20
            t0 = time.perf_counter()
            for _ in range(n_it):
                O.heatmap_forward(cpu_sd, "heatmap_estimator_stereo_front", one)
            dt = time.perf_counter() - t0
            ms1 = _gpu_time_ms(lambda: front(img_f[:1]), 3, 20)
            legs["config1_heatmap_stereo_front_b1"] = {
                "workload": "ego4view_syn_heatmap_stereo_front, batch 1, 2 views (13.63 GFLOP per two-view frame)",
                "cpu_baseline": {"value": round(n_it / dt, 2), "unit": "two-view frames/s", "cores": cores, "kind": "port", "cpu_model": _cpu_model(),
                                 "sample": f"{n_it} forwards of batch 1 (torch-CPU fp32, {cores} threads), {dt:.2f} s"},
                "gpu": {"value": round(1e3 / ms1, 1), "unit": "two-view frames/s", "ms_per_step": round(ms1, 3), "launch": "eager"}}
    del net, g_front, g_back, g_net
    torch.cuda.empty_cache()
    return legs


def _rccl_debug_setup(world: int):
    """Ask RCCL for its topology / transport log (per-process file) so that the N > 1 line can say what the gradient exchange ran
    over (xGMI peer-to-peer, shared memory, network).  EGR_RCCL_SUMMARY=0 or a caller-set NCCL_DEBUG leave the environment alone."""
    if world > 1 and os.environ.get("EGR_RCCL_SUMMARY", "1") != "0" and "NCCL_DEBUG" not in os.environ:
        os.environ["NCCL_DEBUG"] = "INFO"
        os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,GRAPH"
        os.environ["NCCL_DEBUG_FILE"] = f"/tmp/egr_rccl_{os.getpid()}.log"
        return os.environ["NCCL_DEBUG_FILE"]
    return None


def _rccl_summary(path):
    """Transport summary of this rank's RCCL log: channel connections by transport, whether xGMI links were detected."""
    if not path or not os.path.exists(path):
        return None
    import re
    via, xgmi, rings, channels = {}, 0, 0, None
    try:
        with open(path, errors="replace") as f:
            for line in f:
                m = re.search(r" via ([A-Za-z0-9_/]+)", line)
                if m and "->" in line:
                    via[m.group(1)] = via.get(m.group(1), 0) + 1
                if "xgmi" in line.lower():
                    xgmi += 1
                if "Connected all rings" in line or "Connected all trees" in line:
                    rings += 1
                m = re.search(r"(\d+) coll channels", line)
                if m:
                    channels = int(m.group(1))
    except Exception:
        return None
    return {"connections_by_transport": via, "log_lines_mentioning_xgmi": xgmi, "rings_or_trees_connected": rings, "coll_channels": channels,
            "source": "NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,GRAPH log of rank 0"}


def _exchange_leg(tr, world: int, backend: str, dev):
    """The gradient exchange by itself (every rank calls this): one standalone SUM all-reduce per gradient stage bucket of the flat
    gradient buffer, 5 repetitions each, HIP events on the current stream behind a barrier.  busbw = 2 (N - 1) / N x bytes / time."""
    import torch
    import torch.distributed as dist
    from egorear_amd.dist import allreduce_gradients_
    stages = []
    names = ["lifting head", "refiners", "initial heat-map heads", "encoders"]
    for st, (b, e) in sorted(tr.opt.stage_range.items()):      # {gradient stage: [begin, end)} of the flat gradient buffer
        buf = tr.opt.flat_g[b:e]
        for _ in range(2):
            allreduce_gradients_(buf, tr.opt.pg)
        if buf.is_cuda:
            torch.cuda.synchronize()
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                allreduce_gradients_(buf, tr.opt.pg)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
        else:                                            # host tensors (the gloo tests): wall clock
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(5):
                allreduce_gradients_(buf, tr.opt.pg)
            ms = max((time.perf_counter() - t0) * 1e3 / 5, 1e-6)
        nbytes = 4 * (e - b)
        stages.append({"stage": names[st] if st < len(names) else str(st), "bytes": nbytes, "allreduce_ms": round(ms, 3),
                       "busbw_GBps": round(2.0 * (world - 1) / world * nbytes / ms / 1e6, 1)})
    return {"collective": "SUM all-reduce of the flat fp32 gradient buffer, one bucket per gradient stage, started when the stage's gradients are "
                          "complete and overlapped with the rest of the reverse pass",
            "communicator_size": world, "backend": backend + (" (RCCL)" if backend == "nccl" else ""), "stages": stages,
            "bytes_per_step": sum(x["bytes"] for x in stages), "standalone_ms_per_step": round(sum(x["allreduce_ms"] for x in stages), 3)}


def frames_per_rank(batch: int, global_batch: int, world: int) -> int:
    """Frames per rank and step: weak scaling = `batch` on every rank; strong scaling (--global-batch G) = G / world, refused when the
    ranks would get unequal shares (value = world x B x steps / time assumes equal ones)."""
    if global_batch > 0:
        if global_batch % world:
            raise SystemExit(f"bench.py: --global-batch {global_batch} is not divisible by {world} ranks")
        return global_batch // world        # strong scaling: the total work is fixed, every rank takes an equal share of the frames
    return batch


def init_distributed(world: int, rank: int, backend: str, dev):
    """The process group of an N > 1 run -> (backend in use, backend note for the line, rccl_ok).  "nccl" (= RCCL on ROCm) is tried
    first with one eager collective; if it fails the run falls back to gloo for the barrier / max-over-ranks - and the fallback must
    be UNANIMOUS: every rank publishes its outcome and a mixed outcome is refused (tests/test_dist_gloo.py rehearses this with eight
    ranks on the CPU, where the RCCL attempt fails on every rank)."""
    import torch
    import torch.distributed as dist
    backend_note, rccl_ok = "", None
    if world <= 1:
        return backend, backend_note, rccl_ok
    if backend == "nccl":
        try:
            dist.init_process_group("nccl", device_id=dev)  # inference: only the barrier and the max-over-ranks; training: the gradient all-reduce
            probe = torch.zeros(1, device=dev)
            dist.all_reduce(probe)                             # the communicator's first collective, before anything is timed
            torch.cuda.synchronize()
            rccl_ok = True
        except Exception as exc:
            rccl_ok = False
            # RCCL unusable on this node (every rank fails the same way: the eager connect is collective).  The inference line needs
            # no data-path collective, so it is still measured, with gloo for the barrier / timing; the JSON says so and the
            # training leg is then skipped instead of measuring a host-staged exchange.
            backend_note = f"gloo (RCCL failed: {type(exc).__name__}: {str(exc)[:200]})"
            if os.environ.get("EGR_REQUIRE_RCCL") == "1":
                raise SystemExit(f"bench.py: RCCL was required (EGR_REQUIRE_RCCL=1) and failed on rank {rank}: {exc}")
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group("gloo")
            backend = "gloo"
        # The fallback must be unanimous: a rank that fell back while its peers run RCCL would sit in a different rendezvous.  Every
        # rank publishes its outcome in the launcher's store-backed gloo group / the RCCL group it ended up in; a mixed outcome
        # cannot complete this all-gather and ends in the process group's timeout with the reason on stderr instead of a silent hang.
        flags = [None] * world
        dist.all_gather_object(flags, bool(rccl_ok))
        check_unanimous(flags)
    else:
        dist.init_process_group(backend)
    return backend, backend_note, rccl_ok


def check_unanimous(flags) -> None:
    if any(flags) != all(flags):
        raise SystemExit(f"bench.py: RCCL came up on some ranks only ({list(flags)}); refusing to mix backends")


def train_leg(args, dev, rank: int, world: int, backend: str):
    """SURVEY.md §8(f) rank 2 / BASELINE.json config 5, timed separately (never part of `value`): one optimisation step =
    training-mode forward + wrapper losses + backward + gradient all-reduce (RCCL, world > 1) + clip + AdamW, all on the
    HIP kernels (egorear_amd.train.Trainer).  Every rank runs it (the all-reduce is collective); rank 0 reports."""
    import torch
    from egorear_amd import configs, hip, synth, train
    from egorear_amd.dist import timed_steps
    from egorear_amd.estimator import EgoPoseFormerMVFEX
    from egorear_amd.metrics import generate_target
    B = args.train_batch
    net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_rw")))
    synth.load_synth(net, 42)
    net = net.to(dev)
    # after two eager steps the step replays as one hipGraph (one process) or as one hipGraph per gradient stage with the
    # all-reduces issued between the replays (several processes)
    tr = train.Trainer(net, use_graph=True)
    img = synth.synth_images(B, 4, seed=4321 + rank).to(dev)
    ctm = synth.synth_coord_trans_mat(B, seed=77 + rank).to(dev)
    gt_pose = synth.synth_gt_pose(B, seed=99 + rank).to(dev)
    gt_hm = generate_target(synth.synth_joint_px(B, seed=55 + rank).to(dev)).contiguous()
    state = {}

    def run():
        state["terms"], _ = tr.step(img, ctm, gt_pose, gt_hm)

    elapsed = timed_steps(run, args.train_steps, 3, torch.cuda.synchronize, dev if backend == "nccl" else None)
    exchange = None
    if world > 1:
        try:
            exchange = _exchange_leg(tr, world, backend, dev)      # collective: every rank takes part
        except Exception as exc:
            exchange = {"error": f"{type(exc).__name__}: {exc}"}
    if rank != 0:
        return None
    kernels = {}
    if world == 1:   # per-kernel breakdown of one more (eager, instrumented) step; single process only: no collective inside
        # (on ONE stream: with the two-stream reverse pass a launch's event pair would also time whatever the other stream runs beside it)
        overlap, train.OVERLAP = train.OVERLAP, False
        try:
            hip.PROFILE = []
            tr._run(img, ctm, gt_pose, gt_hm, update=True)
            torch.cuda.synchronize()
            prof, hip.PROFILE = hip.PROFILE, None
        finally:
            train.OVERLAP = overlap
        for name, s, e, flops, nbytes, tag in prof:
            key = _kernel_key(name, tag)
            k = kernels.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            k["launches"] += 1
            k["ms"] += s.elapsed_time(e)
            k["flops"] += flops
            k["bytes"] += nbytes
    leg = {"metric": "training 4-view frames/sec (fwd + bwd + grad all-reduce + clip + AdamW)",
           "value": round(world * B * args.train_steps / elapsed, 2), "unit": "frames/s", "ms_per_step": round(1e3 * elapsed / args.train_steps, 3),
           "steps": args.train_steps, "batch_per_gpu": B, "global_batch": B * world, "dtype": DTYPE_NOTE, "data": "synthetic",
           "workload": "ego4view_rw_pose3d fine-tune step (config 5): train-mode BatchNorm, MPJPE x4 + heat-map row-norm x2 losses, "
                       "all 126 M parameters, gradient-norm clip 5.0, AdamW(1e-3, wd 5e-4, two groups)",
           "parallelism": f"dp{world}: frames sharded, stage-bucketed gradient all-reduce overlapped with backward" if world > 1 else "single GPU",
           "launch": ("eager" if tr.graph is None else "hipGraph replay" if not isinstance(tr.graph, list)
                      else f"{len(tr.graph)} hipGraph segments per step, gradient all-reduces between them"), "arithmetic": ARITHMETIC,
           "loss_total": round(float(state["terms"].sum()), 4)}
    if exchange is not None:
        leg["gradient_exchange"] = exchange
    if world == 1 and not args.no_cpu_baseline:
        leg["cpu_baseline"] = cpu_train_baseline()
    if kernels:
        leg["launches_per_step"] = sum(v["launches"] for v in kernels.values())      # launches through the C ABI in one step (torch's own few copies not counted)
        leg["kernel_ms"] = {n: round(v["ms"], 3) for n, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])[:12]}
        dom = max(kernels, key=lambda n: kernels[n]["ms"])
        k = kernels[dom]
        if k["flops"] > 0:
            leg["roofline"] = _roofline(dom, k, *_pmc_traffic_train(B, dom))
    return leg


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    if args.pmc_child:
        return pmc_child(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    live_traffic, live_prov = None, None
    if rank == 0 and world == 1 and not args.no_live_pmc and not args.no_graph and os.environ.get("EGR_BENCH_LIVE_PMC", "1") != "0":
        # (before anything here touches the GPU; --no-graph runs are the profiled ones themselves: no nesting)
        live_traffic, live_prov = live_pmc_traffic(args.batch if args.global_batch <= 0 else args.global_batch)
        if live_traffic is None:
            _log(f"live PMC passes unavailable ({live_prov}); roofline.traffic is replayed from the committed summary")
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    from egorear_amd.dist import check_distinct_devices, claim_device, device_record, gather_rank_records
    allow_shared = os.environ.get("EGR_ALLOW_SHARED_GPU") == "1"
    dev_index = claim_device(local_rank, allow_shared)           # one distinct GPU per rank, or the run refuses to start
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("EGR_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" only for rehearsals on one GPU
    backend_note = ""
    rccl_log = _rccl_debug_setup(world) if backend == "nccl" else None
    backend, backend_note, rccl_ok = init_distributed(world, rank, backend, dev)

    from egorear_amd import configs, hip, synth
    from egorear_amd.estimator import EgoPoseFormerMVFEX

    strong = args.global_batch > 0
    B = frames_per_rank(args.batch, args.global_batch, world)
    net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_syn"))).eval()
    synth.load_synth(net, 42)
    cpu_sd = {k: v.clone() for k, v in net.state_dict().items()} if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    net = net.to(dev)
    img = synth.synth_images(B, 4, seed=1234 + rank).to(dev)  # each rank: its own shard of frames, resident in HBM

    def step():
        return net(img)

    use_graph = not args.no_graph
    graph = None
    with torch.no_grad():
        _log("first step (packs weights)")
        step()  # packs weights, warms the allocator
        torch.cuda.synchronize()
        _log("first step done")
        pipe = None
        if use_graph and args.lanes > 1:
            from egorear_amd.runner import PipelinedForward
            _log(f"capturing {args.lanes} hipGraphs (one per lane)")
            try:
                pipe = PipelinedForward(net, lanes=args.lanes, copy_inputs=False)
                pipe.prime(img)
                _log("captured")
            except Exception as exc:     # (e.g. no memory for a second lane): one captured forward at a time, said so in config.launch
                _log(f"lanes: {type(exc).__name__}: {exc} - falling back to one lane")
                pipe, args.lanes = None, 1
                torch.cuda.synchronize()
        if pipe is not None:
            def run():
                pipe(img)
        elif use_graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            _log("capturing hipGraph")
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = step()
            _log("captured")
            run = graph.replay
        else:
            run = step

        from egorear_amd.dist import timed_steps
        own = {}
        elapsed = timed_steps(run, args.steps, args.warmup, torch.cuda.synchronize, dev if backend == "nccl" else None, detail=own)
        # steady state: the same step over a few hundred iterations (clocks and temperature settled); never `value`
        steady = None
        if args.steady_steps > 0:
            el2 = timed_steps(run, args.steady_steps, 0, torch.cuda.synchronize, dev if backend == "nccl" else None)
            steady = {"steps": args.steady_steps, "value": round(world * B * args.steady_steps / el2, 2), "unit": "frames/s",
                      "ms_per_step": round(1e3 * el2 / args.steady_steps, 3)}
        # one lane: the same forward as ONE captured graph replayed back to back (what the second lane's overlap is worth; never `value`)
        one_lane = None
        if rank == 0 and world == 1 and pipe is not None and args.steady_steps > 0:
            try:
                g1 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1):
                    step()
                ms1 = _gpu_time_ms(g1.replay, 5, 60)
                one_lane = {"value": round(B / ms1 * 1e3, 2), "unit": "frames/s", "ms_per_step": round(ms1, 3), "steps": 60}
                del g1
            except Exception as exc:
                one_lane = {"error": f"{type(exc).__name__}: {exc}"}
        import socket
        rec = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), **device_record(dev_index),
               "frames_per_s": round(B * args.steps / own["own_s"], 2)}
        records = gather_rank_records(rec)                       # collective: every rank takes part
        ranks_obj = {"rccl_ok": rccl_ok, "rccl_transport": _rccl_summary(rccl_log) if rank == 0 else None,
                     "world_size": dist.get_world_size() if world > 1 else 1,
                     "backend": (backend_note or (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else ""))) if world > 1 else "none (single process)",
                     "distinct_devices": check_distinct_devices(records, allow_shared), "per_rank": records}

        # ---- roofline leg: per-launch HIP-event timing of one instrumented (eager) step
        roof = None
        kernels = {}
        _log(f"timed region done: {elapsed:.3f} s")
        if rank == 0:
            hip.PROFILE = []
            step()
            torch.cuda.synchronize()
            prof, hip.PROFILE = hip.PROFILE, None
            fams = {}
            for name, s, e, flops, nbytes, tag in prof:
                key = _kernel_key(name, tag)
                ms_l = s.elapsed_time(e)
                for dst, kk in ((kernels, key), (fams, _family(key, tag))):
                    if kk is None:
                        continue
                    k = dst.setdefault(kk, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
                    k["launches"] += 1
                    k["ms"] += ms_l
                    k["flops"] += flops
                    k["bytes"] += nbytes
                    if tag.startswith("unique") and tag[6:].isdigit():
                        k["unique_bytes"] = k.get("unique_bytes", 0.0) + float(tag[6:])
            dom = max(kernels, key=lambda n: kernels[n]["ms"])
            fam = "f16x2" if dom.endswith("[f16x2]") else ("bf16x3" if dom.endswith("[bf16x3]") else "f32")
            replayed, replay_src = _pmc_traffic(B, fam, kernels[dom]["launches"])
            lt = (live_traffic or {}).get(fam)
            if lt is not None and abs(lt["launches_per_forward"] - kernels[dom]["launches"]) < 0.5:
                # measured in THIS run; the committed summary's figure rides along for comparison
                roof = _roofline(dom, kernels[dom], lt["hbm_bytes_per_launch"], dict(live_prov, launches_counted=lt["launches_counted"]))
                roof["traffic_replayed"], roof["traffic_replayed_commit"] = replayed, (replay_src or {}).get("commit")
                if "mfma_busy_frac" in lt:      # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), same child passes
                    roof["mfma_busy_frac"] = round(lt["mfma_busy_frac"], 4)
                t3 = (live_traffic or {}).get("tapx3x3")
                if t3 and "mfma_busy_frac" in t3:
                    roof["mfma_busy_frac_conv3x3"] = round(t3["mfma_busy_frac"], 4)
            else:
                roof = _roofline(dom, kernels[dom], replayed, replay_src)
                if live_traffic is None and live_prov:
                    roof["traffic_live_unavailable"] = str(live_prov)[:300]
            roof["families"] = _family_roofs(fams)
            # the same per-family figures as SCALAR keys (a parser that keeps scalars only loses `families`)
            f3, f1 = roof["families"].get("conv3x3[f16x2]"), roof["families"].get("conv1x1[f16x2]")
            if f3:
                roof["frac_conv3x3"] = f3["frac"]                          # executed fp16 MFMA rate of the 3x3 launches / 2516.6 TFLOP/s
                roof["conv3x3_algorithmic_tflops"] = f3["algorithmic_tflops"]
                roof["conv3x3_ms"] = f3["kernel_ms_per_step"]
            if f1:
                roof["conv1x1_GBps"] = f1["achieved"]                      # algorithmic bytes / time of the 1x1 launches (HBM roof 8000)
                roof["frac_conv1x1_hbm"] = f1["frac"]
                roof["conv1x1_ms"] = f1["kernel_ms_per_step"]
            ts = roof.get("traffic_source") or {}
            roof["traffic_live"] = bool(ts.get("measured_in_this_run"))    # true: the PMC passes ran as child processes of this very run
            roof["traffic_commit"] = ts.get("commit")                      # commit the replayed PMC passes were collected at (null: live, or no figure)
            roof["traffic_file"] = ts.get("replayed_from")
            roof["all_kernels_ms_per_step"] = round(sum(v["ms"] for v in kernels.values()), 3)
            roof["note"] = ("kernel times come from ONE instrumented eager forward after the timed region (a HIP event pair around every launch, one stream): "
                            "their sum can exceed ms_per_step, which is a hipGraph replay without the events and without host launch gaps"
                            + (f" and, with {args.lanes} lanes, overlaps the low-occupancy tail of one step with the convolutions of the next "
                               "(measured on one box in round 4: 6168 frames/s with one lane, 6511-6516 with two, 6391 with three, 6506 with four)" if (use_graph and args.lanes > 1) else ""))

    cpu_line = parity_line = None
    if cpu_sd is not None:   # BEFORE the other legs mutate the launch policy / release the network: the checker sees the benchmarked state
        try:
            cpu_line, parity_line = cpu_baseline(cpu_sd, net, dev, args.cpu_batch, args.cpu_iters, B)
        except Exception as exc:
            cpu_line = {"error": f"{type(exc).__name__}: {exc}"}
    pre_leg = None
    if rank == 0:
        # §8(f) rank 1, timed separately (never part of `value`): raw uint8 872x872 frames -> model input, on the GPU
        from egorear_amd.preprocess import FramePreprocessor
        pre = FramePreprocessor(device=dev)
        nfr = min(B, 16)
        raw = synth.synth_raw_frames(nfr, 4, seed=1234).to(dev)
        with torch.no_grad():
            for _ in range(3):
                pre(raw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                pre(raw)
            e1.record()
            torch.cuda.synchronize()
        pms = e0.elapsed_time(e1) / 10
        # algorithmic bytes: the frame read once + the fp32 tensor written once (the fused kernel keeps Pillow's uint8 intermediate
        # in LDS; the two-pass fallback adds its write + read)
        pbytes = nfr * 4 * (872 * 872 * 3 + (0 if pre.fused else 2 * 872 * 256 * 3) + 256 * 256 * 12)
        pre_leg = {"frames_per_s": round(nfr / pms * 1e3, 1), "ms_per_batch": round(pms, 3), "batch": nfr,
                   "algorithmic_GBps": round(pbytes / pms / 1e6, 1), "bound": "hbm", "frac_of_8TBps": round(pbytes / pms / 1e6 / PEAK_HBM_GBS, 4),
                   "what": "uint8 (B,4,872,872,3) -> PIL-exact bicubic 256x256 + /255 + ImageNet normalise -> fp32 (B,4,3,256,256)",
                   "launches": 1 if pre.fused else 2}
        # the whole chain from raw 872 x 872 x 4-view uint8 frames resident in HBM (never `value` either: SURVEY.md 8d times the network
        # on the pre-processed tensor): pre-processing + forward of B frames, eager launches, HIP events
        try:
            reps = -(-B // nfr)
            raw_b = raw.repeat(reps, 1, 1, 1, 1)[:B].contiguous()
            with torch.no_grad():
                for _ in range(2):
                    net(pre(raw_b))
                torch.cuda.synchronize()
                e0.record()
                for _ in range(5):
                    net(pre(raw_b))
                e1.record()
                torch.cuda.synchronize()
            rms = e0.elapsed_time(e1) / 5
            pre_leg["from_raw_frames"] = {"frames_per_s": round(B / rms * 1e3, 1), "ms_per_step": round(rms, 3), "batch": B,
                                          "what": "raw uint8 (B,4,872,872,3) in HBM -> pre-processing -> full forward (eager launches)"}
            del raw_b
        except Exception as exc:  # never at the expense of the main line
            pre_leg["from_raw_frames"] = {"error": f"{type(exc).__name__}: {exc}"}
    exact = None
    if rank == 0 and world == 1 and W_FORMAT == "f16x2" and use_graph and not args.no_exact:
        _log("exact-arithmetic leg (bf16x3)")
        try:
            exact = exact_leg(net, img, B, args.lanes)
        except Exception as exc:  # never at the expense of the main line
            exact = {"error": f"{type(exc).__name__}: {exc}"}
    cfg_legs = None
    if rank == 0 and world == 1 and not args.no_configs:
        _log("config 2 / 3 legs (+ config 1 on the CPU)")
        try:
            cfg_legs = config_legs(args, dev, with_cpu=not args.no_cpu_baseline)
        except Exception as exc:  # never at the expense of the main line
            cfg_legs = {"error": f"{type(exc).__name__}: {exc}"}
    train = None
    if not args.no_train and world > 1 and rccl_ok is False:
        # the step's gradient exchange is designed for RCCL over xGMI; a host-staged gloo all-reduce would be a different measurement
        train = {"skipped": "RCCL did not come up on this node (ranks.rccl_ok = false): the training leg's gradient all-reduce is not measured through gloo"}
    elif not args.no_train:
        _log("training-step leg")
        try:
            import gc
            del graph, run, step, net, img, pipe     # hand the inference state (graph pool, weights, inputs) back before the step allocates
            gc.collect()
            torch.cuda.empty_cache()
            train = train_leg(args, dev, rank, world, backend)
        except Exception as exc:  # the inference line must survive a failure here
            train = {"error": f"{type(exc).__name__}: {exc}"}
        _log("training-step leg done")
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        fps = world * B * args.steps / elapsed
        line = {
            "metric": "4-view frames/sec (heatmap+MVFEx+3D lift)", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": DTYPE_NOTE, "data": "synthetic",
            "config": {"workload": "ego4view_syn_pose3d full pipeline (2x ResNet18+FPN encoders, 4 MVFEx/JQA refiners, "
                                   "3D lifting head), 4 views x 256x256 fp32 per frame, eval/no_grad",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"frames sharded over {world} GPU(s), no collective",
                       "launch": ("eager" if not use_graph else "hipGraph replay" if args.lanes <= 1 else
                                  f"{args.lanes} hipGraphs (one captured forward per lane) replayed round-robin on {args.lanes} streams: consecutive steps overlap"),
                       "arithmetic": ARITHMETIC},
            "steady_state": steady,
            "exact_leg": exact,
            "path_tflops_per_gpu": round(fps / world * GFLOP_PER_FRAME / 1e3, 2),
            "path_frac_of_f32_mfma_peak": round(fps / world * GFLOP_PER_FRAME / 1e3 / PEAK_F32_MFMA_TFLOPS, 4),  # > 1 is possible: most contractions run on the bf16 matrix cores
            "roofline": roof,
            "roofline_hbm": roofline_hbm(kernels, pre_leg),
            "ranks": ranks_obj,
            "configs": cfg_legs,
            "preprocess": pre_leg,
            "train": train,
            # abs-max records refused because an arena ran out (each one a launch that silently left the fp16 scheme): 0 when healthy
            "amax_arena_exhausted": int(hip.ARENA_EXHAUSTED),
        }
        if kernels:
            line["kernel_ms"] = {n: round(v["ms"], 3) for n, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])}
        line["one_lane"] = one_lane
        line["cpu_baseline"], line["parity_vs_cpu_oracle"] = cpu_line, parity_line
        # ---- everything a scalar-only parser must see, as SCALAR keys: at the top level (last, so that the tail of the line shows them)
        # and - because the driver's record keeps `roofline` / `cpu_baseline` / `config` only - once more at the front of `roofline`
        def _get(d, *path):
            for k in path:
                d = d.get(k) if isinstance(d, dict) else None
            return d
        tr_roof = _get(train, "roofline") or {}
        scal = {      # (ordered by what a record that keeps only the first ~two dozen scalars of `roofline` must see)
            "traffic_live": _get(roof, "traffic_live"), "mfma_busy_frac": _get(roof, "mfma_busy_frac"),
            "mfma_busy_frac_conv3x3": _get(roof, "mfma_busy_frac_conv3x3"),
            "frac_conv3x3": _get(roof, "frac_conv3x3"), "conv3x3_ms": _get(roof, "conv3x3_ms"),
            "frac_conv1x1_hbm": _get(roof, "frac_conv1x1_hbm"), "conv1x1_ms": _get(roof, "conv1x1_ms"),
            "exact_fps": _get(exact, "value"), "steady_fps": _get(steady, "value"), "one_lane_fps": _get(one_lane, "value"),
            "train_ms_per_step": _get(train, "ms_per_step"), "train_frac": tr_roof.get("frac"),
            "parity_frames": _get(parity_line, "frames"), "argmax_mismatches": _get(parity_line, "argmax_mismatches"),
            "argmax_mismatches_outside_rounding": _get(parity_line, "argmax_mismatches_outside_rounding"),
            "max_joint_err_cm": _get(parity_line, "max_joint_err_cm"),
            "cfg2_fps": _get(cfg_legs, "config2_heatmap_4view", "value"), "cfg3_fps": _get(cfg_legs, "config3_heatmap_mvfex", "value"),
            "train_fps": _get(train, "value"), "train_dominant_kernel": tr_roof.get("kernel"), "train_launches_per_step": _get(train, "launches_per_step"),
            "argmax_compared": _get(parity_line, "argmax_compared"), "valid_mask_mismatches": _get(parity_line, "valid_mask_mismatches"),
            "argmax_mismatches_fp64_sides_with_hip": _get(parity_line, "argmax_mismatches_fp64_sides_with_hip"),
            "top2_gap_below_1e-5": _get(parity_line, "top2_gap_below_1e-5"),
            "all_kernels_ms_per_step": _get(roof, "all_kernels_ms_per_step"), "launches_per_forward": len(prof) if rank == 0 else None,
        }
        if roof is not None:
            head = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic")
            line["roofline"] = {**{k: roof[k] for k in head if k in roof}, **scal, **{k: v for k, v in roof.items() if k not in head and k not in scal}}
        line.update(scal)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
