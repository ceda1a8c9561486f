#!/bin/bash
mkdir -p gpurun_out
python tools/bench_variant_c.py 2>&1 | tail -1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_vc && REPS=3 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vc -- python3 $R/tools/bench_variant_c.py > $R/gpurun_out/vc_prof.log 2>&1
cd $R; f=$(find gpurun_out/prof_vc -name "*kernel_stats.csv" | head -1); head -22 $f | cut -c1-150
