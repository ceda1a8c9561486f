#!/bin/bash
# fixed vs per-round cost of the ring kernel: QKV shape at several M, full kernel (mode 0) and empty skeleton (mode 102)
export HG_LIB_PATH=$PWD/ab/exp.so
for m in 102 100 0; do for M in 7168 14336 28672 57344 114688; do
  echo "mode $m M=$M rounds=$(python3 -c "print(round(($M/256)*9/256,2))"): $(HG_RING_MODE=$m M=$M SHAPES=qkv KERNELS=2 ROUNDS=4 python tools/gemm_ab.py 2>&1 | grep qkv | sed 's/equal=.*//' | cut -c30-100)"
done; done
