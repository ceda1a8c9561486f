#!/bin/bash
# softmax-arithmetic knock-outs of the fused kernel's attention phases (ab/xatt<bits>.so = -DQA_XATT=<bits>, hg_qkv_attn_body.inc):
# what a cheaper softmax could buy at most; two passes over the variants on one box
R=$GRAFT_REPO_ROOT; cd $R
for pass in 1 2; do
for x in ${XATT:-0 1 9 11 15}; do
  echo -n "QA_XATT=$x: "; HG_LIB_PATH=$R/ab/xatt$x.so ROUNDS=2 GSZ="0" timeout 300 python tools/qkv_attn_time.py 2>&1 | tail -1 | sed 's/.*|\( fused.*\)/\1/'
done
done
