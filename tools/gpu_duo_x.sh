#!/bin/bash
export HG_LIB_PATH=$GRAFT_REPO_ROOT/ab/lib_exp.so KERNELS=3 ROUNDS=4 SHAPES="outproj cproj qkv"
for m in 0 1 2 3; do echo "xmode $m"; HG_DUO_XMODE=$m python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids; done
