#!/bin/bash
# knock-outs of the fused in_proj + attention kernel (ab/qa_exp.so = -DHG_EXPERIMENTS build; HG_QA_MODE bits: 1 no attention
# phases, 2 no MFMA in the K loop, 4 no epilogue / attention at all, 8 no operand DMA)
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_attention.py -x -q -k "fused" 2>&1 | tail -5
for m in ${MODES:-0 1 4 6 12 14}; do
  echo -n "HG_QA_MODE=$m: "; HG_LIB_PATH=$R/ab/qa_exp.so HG_QA_MODE=$m ROUNDS=2 GSZ="0" timeout 300 python tools/qkv_attn_time.py 2>&1 | tail -1
done
