#!/bin/bash
# Round-2 GPU check (run through gpurun): new scale / chunk tests, bench line with the per-kernel table, kernel trace.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_scale.py -m gpu -x -q > gpurun_out/scale_test.log 2>&1; tail -5 gpurun_out/scale_test.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r2.log 2>&1; tail -1 gpurun_out/bench_r2.log | cut -c1-6000
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_r2 && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > $R/gpurun_out/bench_prof_r2.log 2>&1
cd $R; f=$(find gpurun_out/prof_r2 -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-160
