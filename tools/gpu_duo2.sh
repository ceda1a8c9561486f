#!/bin/bash
mkdir -p gpurun_out
export SHAPES="qkv outproj" KERNELS="3" ROUNDS=4
echo "default"; python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids
echo "grid 1/CU"; HG_DUO_GRID=1 python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids
echo "lds cut 2048"; HG_DUO_LDS_CUT=2048 python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids
echo "lds cut 16384"; HG_DUO_LDS_CUT=16384 python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids
