#!/bin/bash
# round-5 collection: full bench line, kernel trace of the all-rows headline alone, kernel trace of the default bench run, PMC traffic
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
python bench.py > gpurun_out/bench_final.log 2>&1; tail -1 gpurun_out/bench_final.log | cut -c1-300
bash tools/gpu_trace_headline.sh > gpurun_out/trace_headline.log 2>&1; tail -9 gpurun_out/trace_headline.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_full && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_full -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_prof_full.log 2>&1
cd $R; f=$(ls -t gpurun_out/prof_full/*/*kernel_stats.csv | head -1); test -n "$f" && head -12 "$f" | cut -c1-160
CONFIGS="r05:-" AB_STEPS=30 bash tools/gpu_energy_ab.sh 2>&1 | grep -E "bench|traffic" | cut -c1-220
