#!/bin/bash
# knock-outs of the row-owner GEMM (ab/exp.so = tools/build_variant.sh exp -DHG_EXPERIMENTS hg_gemm_rows): HG_ROWS_MODE bits
# 1 no epilogue, 2 no MFMA, 4 no W DMA, 8 no A DMA
echo "product: $(KERNELS='2 5' python tools/rows_time.py 2>&1 | tail -1)"
export HG_LIB_PATH=$PWD/ab/exp.so
for m in ${MODES:-0 1 3 5 9 13 15 2 12}; do echo "HG_ROWS_MODE=$m: $(HG_ROWS_MODE=$m python tools/rows_time.py 2>&1 | tail -1)"; done
