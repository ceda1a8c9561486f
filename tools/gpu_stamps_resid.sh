#!/bin/bash
# in-kernel s_memtime shares per K-tile of the residual GEMMs (ab/stamps.so = -DHG_STAMPS build): ring2<10> and, with HG_RING_BIG=1, ring<4,10>
export HG_LIB_PATH=/root/repo/ab/stamps.so
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys
os.environ["HG_STAMPS"] = "1"
sys.path.insert(0, "/root/repo")
import torch
from hoigen_amd import _lib
ctx = _lib.ctx(0)
M = 197 * 256
for (N, K, epi) in [(768, 3072, 10), (768, 768, 10), (3072, 768, 1), (2304, 768, 0)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda")
    for it in range(2):
        rc = _lib.lib().hg_test_gemm(ctx, a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, epi, 2, None)
        assert rc == 0, rc
        torch.cuda.synchronize()
PY
