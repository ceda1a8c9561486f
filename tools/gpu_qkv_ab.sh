#!/bin/bash
# fused in_proj + attention kernel: A/B of library builds on one box (LIBS="base w128" -> ab/<name>.so), kernel tests first
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_attention.py -x -q -k "fused_qkv" 2>&1 | tail -3
for pass in 1 2 3; do
for l in ${LIBS:-base w128}; do
  echo -n "$l: "; HG_LIB_PATH=$R/ab/$l.so ROUNDS=2 GSZ="0" timeout 300 python tools/qkv_attn_time.py 2>&1 | tail -1 | sed 's/.*|\( fused.*\)/\1/'
done
done
