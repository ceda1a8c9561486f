#!/bin/bash
# is the one-kernel VAE bound by power (clock) or by its instruction stream?  the same launch on all-zero operands
ROOT=$GRAFT_REPO_ROOT; cd $ROOT
for z in "" 1; do for m in 0 2; do echo "ZERO=$z HG_VAE_FUSED=$m"; ZERO=$z HG_VAE_FUSED=$m R=98304 ITERS=10 timeout 300 python tools/vae_time.py 2>&1 | grep -v amdgpu.ids; done; done
