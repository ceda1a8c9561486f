#!/bin/bash
# Variant C with in_proj + attention as one kernel (K = D + 64): kernel-level and model-level parity, then the A/B by option.
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_attention.py -x -q -k "fused_qkv" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "variant_c" 2>&1 | tail -5
for i in 1 2; do
  HG_QKV_ATTN_C=0 python tools/bench_variant_c.py
  HG_QKV_ATTN_C=1 python tools/bench_variant_c.py
done
} > gpurun_out/qkv_attn_c.log 2>&1
tail -30 gpurun_out/qkv_attn_c.log
