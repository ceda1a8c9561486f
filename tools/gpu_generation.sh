#!/bin/bash
# generation pipeline (main_tip_finetune.py:759-824): timing per batch size and a kernel profile of the default batching
ROOT=$GRAFT_REPO_ROOT; mkdir -p $ROOT/gpurun_out; cd $ROOT
ITERS=104 BATCH_ITERS=${BATCH_ITERS:-0} timeout 600 python tools/bench_generation.py 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp && rm -rf $ROOT/gpurun_out/prof_gen && ITERS=26 BATCH_ITERS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_gen -- python3 $ROOT/tools/bench_generation.py > $ROOT/gpurun_out/prof_gen.log 2>&1
cd $ROOT; f=$(find gpurun_out/prof_gen -name "*kernel_stats.csv" | head -1); head -24 $f | cut -c1-150
