#!/bin/bash
# round 4: GPU test suite + full bench line + kernel trace of the all-rows headline (start-of-session baseline)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputests.log 2>&1; tail -3 gpurun_out/r4_gputests.log
python bench.py > gpurun_out/r4_bench.log 2>&1; tail -1 gpurun_out/r4_bench.log | cut -c1-600
bash tools/gpu_trace_headline.sh > gpurun_out/trace_headline.log 2>&1; tail -14 gpurun_out/trace_headline.log | cut -c1-200
