#!/bin/bash
# Evidence for profiles/: rocprofv3 kernel stats + the PMC passes of the inference bench (run on the GPU box, from the repo root).
#   tools/collect_profiles.sh TAG [COMMIT]  -> gpurun_out/prof_TAG/{stats,fetch,write,mfma}/...
set -e
tag=$1
commit=${2:-unknown}
out=gpurun_out/prof_$tag
mkdir -p $out
CMD="python3 bench.py --steps 5 --warmup 2 --steady-steps 0 --no-graph --no-cpu-baseline --no-train --no-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $CMD > $out/stats.log 2>&1
PM="python3 bench.py --steps 2 --warmup 1 --steady-steps 0 --no-graph --no-cpu-baseline --no-train --no-configs"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $PM > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- $PM > $out/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -- $PM > $out/mfma.log 2>&1
python3 tools/pmc_traffic.py $out/fetch $out/write $out/pmc_traffic.json 64 12 $commit > /dev/null
python3 tools/launch_list.py --batch 64 > $out/launch_list.txt 2>/dev/null
python3 tools/pmc_traffic_per_launch.py $out/fetch $out/write $out/launch_list.txt 12 > $out/pmc_traffic_per_launch.txt || true
python3 tools/pmc_mfma_util.py $out/mfma $out/pmc_mfma_util.json > /dev/null
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
grep -h '"metric"' $out/stats.log | tail -1 > $out/bench_profiled.json || true
echo done $tag
[ "$3" = "notrain" ] && { rm -rf $out/stats/*/*kernel_trace.csv $out/mfma; exit 0; }     # (fetch / write kept: ~20 MB of CSV for re-reductions)
# training step (config 5, batch 32): per-kernel breakdown of the eager step + the PMC passes of the same command
TR="python3 tools/train_bench.py --batch 32 --steps 2 --warmup 1"
python3 tools/train_bench.py --batch 32 --graph > $out/train_step_breakdown.txt 2>&1 || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/tfetch -- $TR > $out/tfetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/twrite -- $TR > $out/twrite.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/tmfma -- $TR > $out/tmfma.log 2>&1
python3 tools/pmc_traffic_train.py $out/tfetch $out/twrite $out/pmc_traffic_train.json 32 $commit > /dev/null
python3 tools/pmc_mfma_util.py $out/tmfma $out/pmc_train_mfma_util.json > /dev/null
rm -rf $out/stats/*/*kernel_trace.csv $out/fetch $out/write $out/mfma $out/tfetch $out/twrite $out/tmfma
echo done-train $tag
