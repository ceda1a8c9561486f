#!/bin/bash
# knock-outs of the lock-step flow loop (tools/experiments/round3) on 16 workgroups (clock at its maximum: time ~ cycles) and on 256:
# us per launch of the plain residual GEMM, M = 8192 / 50432, N = 768, K = 3072
for v in ${VARIANTS:-flow flow_NO_DMA flow_NO_READ flow_NO_MFMA flow_NO_BARRIER flow_MFMA_ONLY flow_DMA_ONLY}; do
for g in 16 256; do
HG_LIB_PATH=/root/repo/ab/$v.so HG_RING2_GRID=$g V=$v python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -1
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from hoigen_amd import _lib
ctx = _lib.ctx(0)
g = int(os.environ["HG_RING2_GRID"])
M, N, K = (128 * 64 if g == 16 else 197 * 256), 768, 3072
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; b = torch.randn(N, device="cuda")
out = torch.zeros(M, N, device="cuda")
call = lambda: _lib.lib().hg_test_gemm(ctx, a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 3, 4, None)
for it in range(3): call()
torch.cuda.synchronize()
_, recs = _lib.profile(ctx, _lib.HG_PROF_ALL, 16, lambda: [call() for _ in range(6)])
us = sum(r[4] for r in recs) / len(recs) * 1e3
tiles = (M // 128) * 3; rounds = -(-tiles // g)
print("%-18s grid %3d: %7.1f us  = %.3f us per K-tile-round" % (os.environ["V"], g, us, us / rounds / 48))
PY
done; done
