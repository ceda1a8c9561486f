#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -x -q > gpurun_out/gemm_test.log 2>&1; tail -4 gpurun_out/gemm_test.log
timeout 600 python tools/gemm_ab.py 2>&1 | tee gpurun_out/gemm_ab.log
