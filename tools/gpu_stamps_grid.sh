export HG_LIB_PATH=/root/repo/ab/stamps.so
for g in 256 240 128 64 16; do
echo "grid $g"; HG_RING_GRID=$g python - <<'PY' 2>&1 | grep -v amdgpu.ids | grep "wave 0" | tail -1
import os, sys
os.environ["HG_STAMPS"] = "1"
sys.path.insert(0, "/root/repo")
import torch
from hoigen_amd import _lib
ctx = _lib.ctx(0)
M = 197 * 256
for (N, K, epi) in [(3072, 768, 1)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda")
    for it in range(3):
        rc = _lib.lib().hg_test_gemm(ctx, a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, epi, 2, None)
        assert rc == 0, rc
        torch.cuda.synchronize()
PY
done
