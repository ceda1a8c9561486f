#!/bin/bash
# time stamps around a tile boundary of the 256x256 ring (tools/build_variant.sh trace -DHG_TRACE hg_gemm_ring)
export HG_LIB_PATH=$PWD/ab/trace.so HG_TRACE=1 KERNELS=2 ROUNDS=1
SHAPES="${SHAPES:-qkv cfc}" python tools/gemm_ab.py 2>&1 | grep "trace" | sort -u | head -40
