#!/bin/bash
# stamped build of the duo kernel into ab/ (in-kernel s_memtime shares per K-tile); run via gpurun after building here
export HG_LIB_PATH=$GRAFT_REPO_ROOT/ab/lib_stamps.so HG_STAMPS=1 KERNELS=3 ROUNDS=1
SHAPES="${SHAPES:-outproj cproj qkv}" python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids
