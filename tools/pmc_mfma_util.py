#!/usr/bin/env python3
"""MFMA utilisation of the conv kernel from a rocprofv3 PMC pass:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d DIR -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline
    python tools/pmc_mfma_util.py DIR out.json
SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (64 per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_32x32x16_bf16), summed over the chip's
1024 SIMDs; conv_igemm_bf16x3 = the launches of conv_igemm_x6_kernel (split-bf16 operands);
GRBM_GUI_ACTIVE is summed over the 8 XCDs (guide: effective clock = GRBM_GUI_ACTIVE / 8 / wall time)."""
import collections, csv, glob, json, re, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); seen = set()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    x6 = "conv_igemm_x6" in n or "conv_igemm_tap" in n or "conv_pw_x6" in n
    h2 = x6 and re.search(r",\s*2>\(", n) is not None        # last template argument: planes per operand (2 = the fp16 scheme)
    x6w = "conv_wgrad_x6" in n or "conv_wgrad3_x6" in n
    h2w = x6w and re.search(r",\s*2>\(", n) is not None
    h2s = "stem_x6_kernel" in n and re.search(r",\s*2>\(", n) is not None
    # conv_tapx_kernel<WM, WN, FN, STRIDE, RES> (fp16 scheme only): its own rows - 3x3 (STRIDE 1 / 2) and the wide 1x1 mode (STRIDE 0) -
    # AND part of the conv_igemm_f16x2 total (the key bench.py's roofline object carries)
    if "conv_tapx_kernel" in n:
        m = re.search(r"conv_tapx_kernel<\s*\d+\s*,\s*\d+\s*,\s*\d+\s*,\s*(\d+)", n)
        kk = "conv_tapx_1x1_f16x2" if (m and m.group(1) == "0") else "conv_tapx_3x3_f16x2"
        per[kk][r["Counter_Name"]] += float(r["Counter_Value"])
        if ("t", r["Dispatch_Id"]) not in seen:
            seen.add(("t", r["Dispatch_Id"])); dur[kk] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    chain = "conv_pw_chain_kernel" in n or "conv_pw2_kernel" in n          # chained 1x1 pairs (fp16 scheme only): booked with the streaming / tiled kernels
    if "conv_tapx_kernel" not in n and ((("conv_igemm_tap" in n or "conv_pw_x6" in n or "conv_igemm_x6" in n) and h2) or chain):
        kk = "conv_other_split_f16x2"      # what is left on the tap-sharing / streaming / tiled kernels
        per[kk][r["Counter_Name"]] += float(r["Counter_Value"])
        if ("o", r["Dispatch_Id"]) not in seen:
            seen.add(("o", r["Dispatch_Id"])); dur[kk] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    h2 = h2 or "conv_tapx_kernel" in n or chain
    k = (("conv_igemm_f16x2" if h2 else ("conv_igemm_bf16x3" if x6 else "conv_igemm")) if ("conv_igemm" in n or "conv_pw_x6" in n or "conv_tapx_kernel" in n or chain) else
         (("conv_wgrad_f16x2" if h2w else ("conv_wgrad_bf16x3" if x6w else "conv_wgrad")) if "conv_wgrad" in n else
          (("stem_f16x2" if h2s else "stem_bf16x3") if "stem_x6_kernel" in n else ("stem" if "stem_kernel" in n else ("wstream_f16x2" if "ws_stream_kernel" in n else "other")))))
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen:
        seen.add(r["Dispatch_Id"]); dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = {}
for k in ("conv_igemm_f16x2", "conv_tapx_3x3_f16x2", "conv_tapx_1x1_f16x2", "conv_other_split_f16x2", "conv_igemm_bf16x3", "conv_igemm", "conv_wgrad_f16x2", "conv_wgrad_bf16x3", "conv_wgrad", "stem_f16x2", "stem_bf16x3", "stem",
          "wstream_f16x2"):
    if k not in per:
        continue
    busy, gui = per[k]["SQ_VALU_MFMA_BUSY_CYCLES"], per[k]["GRBM_GUI_ACTIVE"]
    cycles = gui / 8.0
    out[k] = {"mfma_busy_simd_cycles": busy, "elapsed_cycles": cycles, "mfma_util": busy / (cycles * 1024.0),
              "effective_clock_GHz": cycles / max(dur[k], 1.0), "kernel_time_ms": dur[k] / 1e6}
json.dump(out, open(sys.argv[2], "w"), indent=1); print(json.dumps(out))
