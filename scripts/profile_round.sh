#!/bin/bash
# Round profile on the GPU box: kernel trace + PMC passes of the bench workload (summaries are copied to profiles/ by hand).
#   bash scripts/profile_round.sh r03            (config #2)
#   bash scripts/profile_round.sh r03_vitl --config vitl     (extra bench.py arguments after the tag; the PMC file is then named per config)
set -e
TAG=${1:-r03}
shift || true
EXTRA="$*"
CFG=$(echo "$EXTRA" | sed -n 's/.*--config \([a-z]*\).*/\1/p')
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
OUT=$ROOT/gpurun_out/${TAG}_prof
mkdir -p $OUT
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/mfma -o mfma -- $BENCH > $OUT/mfma.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/valu -o valu -- $BENCH > $OUT/valu.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $BENCH > $OUT/write.log 2>&1
find $OUT -name "*.csv" | xargs ls -la
# keep what merges back small: the per-dispatch traces are summarised here
python3 $ROOT/scripts/summarize_round.py $OUT $ROOT/gpurun_out/${TAG}_summary.json || true
python3 $ROOT/scripts/summarize_pmc.py $(find $OUT/fetch -name "*counter_collection.csv" | head -n 1) $(find $OUT/write -name "*counter_collection.csv" | head -n 1) $ROOT/gpurun_out/${TAG}_pmc_nt256${CFG:+_$CFG}.json || true
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -n 1) $ROOT/gpurun_out/${TAG}_kernel_stats.csv || true
find $OUT -name "*kernel_trace.csv" -size +8M -delete
