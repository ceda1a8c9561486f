#!/bin/bash
# one-kernel VAE (hg_vae_fused.hip): parity tests, then config 4 timed on the three dispatch modes (HG_VAE_FUSED = 0 GEMM path, 1 default, 2 every row)
ROOT=$GRAFT_REPO_ROOT; mkdir -p $ROOT/gpurun_out; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_vae_fused.py tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -x -q -k "vae or generation or config4" 2>&1 | tail -5
for m in 0 1 2; do echo "HG_VAE_FUSED=$m"; HG_VAE_FUSED=$m ITERS=10 timeout 300 python tools/vae_time.py 2>&1 | grep -v amdgpu.ids; done
