"""Fused in_proj + attention kernel of the library in HG_LIB_PATH against fp32 PyTorch on the fp16-rounded operands and against the two
kernels it replaces (GEMM + attention_kernel of the same library): rel-L2 and max abs error; ViT-B/16 shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hoigen_amd import _lib
ctx = _lib.ctx(0); L_ = _lib.lib()
n_seq, L, heads = int(os.environ.get("NSEQ", 64)), int(os.environ.get("L", 197)), 12
D = heads * 64
g = torch.Generator(device="cuda").manual_seed(3)
a = torch.randn(n_seq * L, D, device="cuda", generator=g)
w = torch.randn(3 * D, D, device="cuda", generator=g) * D ** -0.5 * float(os.environ.get("WSCALE", 1.0))
bias = torch.randn(3 * D, device="cuda", generator=g) * 0.3
cs = w.half().float().sum(1)
mr = torch.stack([torch.randn(n_seq * L, device="cuda", generator=g) * 0.05, torch.rand(n_seq * L, device="cuda", generator=g) + 0.5], 1).contiguous()
def run(fused):
    out = torch.empty(n_seq * L, D, device="cuda")
    rc = L_.hg_test_qkv_attn(ctx, a.data_ptr(), w.data_ptr(), bias.data_ptr(), cs.data_ptr(), mr.data_ptr(), n_seq, L, heads, fused, out.data_ptr(), None)
    assert rc == 0, L_.hg_last_error(ctx)
    torch.cuda.synchronize(); return out
f, s = run(1), run(0)
qkv = ((a.half().float() @ w.half().float().t() - mr[:, :1] * cs[None]) * mr[:, 1:] + bias[None]).half().float()
q, k, v = [t.view(n_seq, L, heads, 64).permute(0, 2, 1, 3) for t in qkv.split(D, dim=1)]
ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(n_seq * L, D)
rel = lambda x, y: ((x - y).norm() / y.norm()).item()
print(f"fused vs fp32 reference: rel-L2 {rel(f, ref):.3e} max abs {(f - ref).abs().max().item():.3e} | separate vs reference: {rel(s, ref):.3e} {(s - ref).abs().max().item():.3e} | "
      f"fused vs separate: {rel(f, s):.3e} max abs {(f - s).abs().max().item():.3e} equal {torch.equal(f, s)} | repeat equal {torch.equal(f, run(1))}")
