#!/bin/bash
# end-of-round collection (round 3 onwards; replaces gpu_r2_final.sh): full bench line, kernel trace of the default run, kernel trace of the all-rows headline alone, variant C
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
python bench.py > gpurun_out/bench_final.log 2>&1; tail -1 gpurun_out/bench_final.log | cut -c1-400
bash tools/gpu_trace_headline.sh > gpurun_out/trace_headline.log 2>&1; tail -9 gpurun_out/trace_headline.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_r3 && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > $R/gpurun_out/bench_prof_r3.log 2>&1
cd $R; f=$(find gpurun_out/prof_r3 -name "*kernel_stats.csv" | head -1); head -10 $f | cut -c1-160
python tools/bench_variant_c.py 2>&1 | tail -1
