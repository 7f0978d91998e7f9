#!/bin/bash
# Developer tool (GPU box): the four rocprofv3 passes of one bench command that tools/pmc_summary.py folds.
#   tools/profile_bench.sh <outdir under gpurun_out> [bench args ...]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mfma-leg --no-extra-legs $*"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_stats.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/sq -o q -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_sq.log 2>&1
echo "sq pass done"
# keep only what the summary needs (the merged-back scratch is capped at 64 MiB): the per-launch traces of the counter passes are
# folded on the box by tools/pmc_summary.py, the csv files themselves stay there
find $OUT -name "*agent_info*" -delete
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT $PMC_ARCH $PMC_BATCH $PMC_DTYPE > $OUT/pmc.json
cp $OUT/stats/s_kernel_stats.csv $OUT/kernel_stats.csv
rm -rf $OUT/fetch $OUT/write $OUT/sq $OUT/stats/s_kernel_trace.csv
ls -la $OUT
