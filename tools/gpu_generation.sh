#!/bin/bash
ROOT=$GRAFT_REPO_ROOT; mkdir -p $ROOT/gpurun_out; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "generation or text or prompt" 2>&1 | tail -3
ITERS=48 timeout 600 python tools/bench_generation.py 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp && rm -rf $ROOT/gpurun_out/prof_gen && ITERS=16 BATCH_ITERS=4 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_gen -- python3 $ROOT/tools/bench_generation.py > $ROOT/gpurun_out/prof_gen.log 2>&1
cd $ROOT; f=$(find gpurun_out/prof_gen -name "*kernel_stats.csv" | head -1); head -22 $f | cut -c1-150
