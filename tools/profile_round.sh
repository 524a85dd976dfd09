#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): bench line + rocprofv3 kernel stats + the PMC passes of one round.
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r02'
# Outputs land in gpurun_out/<tag>_*; summarise afterwards (in the build container) with
#   tools/kernel_stats_summary.py, tools/traffic_summary.py, tools/mfma_busy_summary.py
set -e -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SAMPLING="--no-cpu-baseline --no-split-leg --no-train-mode-leg --no-train-step-leg --no-autocast-leg --no-cfg2-leg --no-vae-train-leg --no-slices-check"
python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.log 2>&1
tail -1 $OUT/${TAG}_bench.log | cut -c1-300
rm -rf $OUT/${TAG}_autocast $OUT/${TAG}_stats $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_mfma $OUT/${TAG}_train_f32 $OUT/${TAG}_train_bf16
# the sampling hot path alone (headline roofline): same command as the bench's timed region, one pass
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o ${TAG} -- python3 $ROOT/bench.py --steps 1 --warmup 1 $SAMPLING > $OUT/${TAG}_stats.log 2>&1
echo "stats done"
# the opt-in bf16 (autocast) sampling + decode pass
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_autocast -o ${TAG} -- python3 $ROOT/tools/autocast_bench.py --passes 1 > $OUT/${TAG}_autocast.log 2>&1
echo "autocast stats done"
# the training step (BASELINE cfg 5 per-GPU shape), both operand precisions
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_train_f32 -o ${TAG} -- python3 $ROOT/tools/train_bench.py --batch 128 --latent 64 --steps 2 --precision f32 > $OUT/${TAG}_train_f32.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_train_bf16 -o ${TAG} -- python3 $ROOT/tools/train_bench.py --batch 128 --latent 64 --steps 2 --precision bf16 > $OUT/${TAG}_train_bf16.log 2>&1
echo "train stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o ${TAG} -- python3 $ROOT/bench.py --steps 1 --warmup 0 --num-steps 2 $SAMPLING > $OUT/${TAG}_pmc_fetch.log 2>&1
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o ${TAG} -- python3 $ROOT/bench.py --steps 1 --warmup 0 --num-steps 2 $SAMPLING > $OUT/${TAG}_pmc_write.log 2>&1
echo "write done"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_mfma -o ${TAG} -- python3 $ROOT/bench.py --steps 1 --warmup 0 --num-steps 2 $SAMPLING > $OUT/${TAG}_pmc_mfma.log 2>&1
echo "mfma done"
# the raw kernel-trace CSVs are large; keep only what the summaries need
find $OUT/${TAG}_stats $OUT/${TAG}_autocast $OUT/${TAG}_train_f32 $OUT/${TAG}_train_bf16 -name "*kernel_trace.csv" -delete
du -sh $OUT/${TAG}_* | tail -12
