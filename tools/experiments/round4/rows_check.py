"""Row-owner residual GEMM (hg_gemm_rows.hip, kernel 5 of hg_test_gemm_ln) against ring2 (kernel 2): bits of the stream and
its fp16 copy, statistics within tolerance, and time per launch (hipEvents through hg_profile) at the two ViT-B/16 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hoigen_amd import _lib
ctx = _lib.ctx(0)
lib = _lib.lib()

def run(a, w, bias, kernel, mu, x0):
    M, K = a.shape; N = w.shape[0]
    out = x0.clone(); out2 = torch.empty(M, N, device="cuda"); mr = torch.empty(M, 2, device="cuda"); muo = torch.empty(M, device="cuda")
    rc = lib.hg_test_gemm_ln(ctx, a.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, 10, kernel, None, None,
                             mu.data_ptr(), None, out2.data_ptr(), mr.data_ptr(), muo.data_ptr(), None)
    assert rc == 0, lib.hg_last_error(ctx)
    torch.cuda.synchronize()
    return out, out2, mr, muo

shapes = [(197 * 12 + 5, 768, 768), (4096 + 77, 768, 3072), (128 * 37, 768, 832), (197 * 256, 768, 768), (197 * 256, 768, 3072), (77 * 600, 512, 512), (77*600, 512, 2048)]
if os.environ.get("SMALL"): shapes = shapes[:3]
for M, N, K in shapes:
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g); w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    x0 = torch.randn(M, N, device="cuda", generator=g) * 2 + torch.randn(M, 1, device="cuda", generator=g)
    mu = x0.mean(1) + 0.05 * torch.randn(M, device="cuda", generator=g)
    xr = x0 + a.half().float() @ w.half().float().t() + bias
    mean = xr.mean(1); rstd = 1.0 / torch.sqrt(xr.var(1, unbiased=False) + 1e-5)
    ref = run(a, w, bias, 2, mu, x0) if N == 768 else None
    got = run(a, w, bias, 5, mu, x0)
    again = run(a, w, bias, 5, mu, x0)
    sc = xr.abs().max().item()
    print(f"{M}x{N}x{K}: x err {(got[0]-xr).abs().max().item()/sc:.2e} x16 err {(got[1]-(xr-mu[:,None]).half().float()).abs().max().item():.2e} "
          f"mean err {(got[3]-mean).abs().max().item()/sc:.2e} dm err {(got[2][:,0]-(mean-mu)).abs().max().item()/sc:.2e} rstd rel {((got[2][:,1]-rstd).abs()/rstd).max().item():.2e} "
          f"repeat-equal {all(torch.equal(u, v) for u, v in zip(got, again))}"
          + (f" bits==ring2: x {torch.equal(got[0], ref[0])} x16 {torch.equal(got[1], ref[1])}" if ref else ""), flush=True)
    if M >= 40000:
        for kernel in ((2, 5) if N == 768 else (5,)):
            def f():
                for _ in range(5): run(a, w, bias, kernel, mu, x0)
            _, recs = _lib.profile(ctx, 10, 64, f)
            ts = sorted(r[4] for r in recs if r[1] == M)
            print(f"   kernel {kernel}: {len(ts)} launches, median {ts[len(ts)//2]*1e3:.1f} us, min {ts[0]*1e3:.1f} us", flush=True)
