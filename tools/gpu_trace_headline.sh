#!/bin/bash
# kernel trace of the all-rows headline configuration ALONE (no class-rows-only region, no config 3 / 4)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_head && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_head -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-class-rows > $R/gpurun_out/bench_head.log 2>&1
cd $R; tail -1 gpurun_out/bench_head.log | cut -c1-300; f=$(ls -t gpurun_out/prof_head/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-150
