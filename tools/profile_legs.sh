#!/bin/bash
# Developer tool (GPU box): rocprofv3 --kernel-trace --stats of the other bench configurations (one pass each, no counters).
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, args...
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mfma-leg --no-extra-legs "$@" > $OUT/bench_$n.log 2>&1
  cp $OUT/$n/s_kernel_stats.csv $OUT/kernel_stats_$n.csv; rm -rf $OUT/$n; echo "$n done"
}
run vitb_f32 --arch ViT-B/32 --batch-per-gpu 512
run vitb_f16 --arch ViT-B/32 --dtype f16 --batch-per-gpu 512
run rn50_f16 --dtype f16
run vitl_f16 --arch ViT-L/14@336px --dtype f16 --batch-per-gpu 256
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/adapter -o s -- python3 $GRAFT_REPO_ROOT/tools/bench_adapter_step.py 256 1024 200 > $OUT/bench_adapter_bs256.log 2>&1
cp $OUT/adapter/s_kernel_stats.csv $OUT/kernel_stats_adapter_bs256.csv; rm -rf $OUT/adapter
python3 $GRAFT_REPO_ROOT/tools/bench_adapter_step.py 256 1024 1000 > $OUT/adapter_step_timing.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/bench_adapter_step.py 1024 1024 500 >> $OUT/adapter_step_timing.log 2>&1
ls $OUT
