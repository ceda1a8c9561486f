#!/bin/bash
# kernel trace of variant C (batch 256, 14 prior tokens) next to variant A: tools/bench_variant_c.py under rocprofv3
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_c
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c -- python3 $R/tools/bench_variant_c.py > $R/gpurun_out/variant_c_prof.log 2>&1
cd $R; grep '^{' gpurun_out/variant_c_prof.log
f=$(ls -t gpurun_out/prof_c/*/*kernel_stats.csv | head -1); test -n "$f" && head -40 "$f" | cut -c1-200
