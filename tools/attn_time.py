"""Diagnostics: the ViT-B/16 attention shape (256 sequences x 197 tokens x 12 heads) through hg_test_attention, a few
launches; run under `rocprofv3 --kernel-trace` and read the kernel durations (tools/gpu_attn_time.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hoigen_amd import _lib
ctx = _lib.ctx(0)
n_seq, L, heads = int(os.environ.get("NSEQ", 256)), int(os.environ.get("L", 197)), int(os.environ.get("HEADS", 12))
qkv = torch.randn(n_seq * L, 3 * heads * 64, device="cuda")
out = torch.empty(n_seq * L, heads * 64, device="cuda")
for it in range(int(os.environ.get("ITERS", 8))):
    rc = _lib.lib().hg_test_attention(ctx, qkv.data_ptr(), None, None, n_seq, L, heads, int(os.environ.get("CAUSAL", 0)),
                                      out.data_ptr(), None)
    assert rc == 0, rc
torch.cuda.synchronize()
