#!/bin/bash
# Knock-out timing of the 256x256 ring kernel (experiments build ab/exp.so = tools/build_variant.sh exp -DHG_EXPERIMENTS):
# HG_RING_MODE bits 2 no MFMA, 4 no epilogue, 32 no fragment reads, 64 no operand DMA (results are wrong by design).
export HG_LIB_PATH=$PWD/ab/exp.so
for m in ${MODES:-0 4 2 32 64 34 66 96 98 100 102}; do
  echo "HG_RING_MODE=$m: $(HG_RING_MODE=$m SHAPES="${SHAPES:-qkv cfc}" KERNELS=2 ROUNDS=4 python tools/gemm_ab.py 2>&1 | sed 's/ M=50432//; s/equal=.*//' | tr '\n' ' ')"
done
