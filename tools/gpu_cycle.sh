#!/bin/bash
# GPU iteration helper (run through gpurun): GEMM unit tests, bench line, rocprofv3 kernel stats.
mkdir -p gpurun_out
if [ "$1" != "nobtest" ]; then
timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -x -q > gpurun_out/gemm_test.log 2>&1; tail -3 gpurun_out/gemm_test.log
fi
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_cur.log 2>&1; tail -1 gpurun_out/bench_cur.log | cut -c1-2000
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_cur && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cur -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_prof_cur.log 2>&1
cd $R; f=$(find gpurun_out/prof_cur -name "*kernel_stats.csv" | head -1); head -9 $f | cut -c1-140
