"""Measurement reference only (not part of the product path): what the vendor GEMM (hipBLASLt via torch.matmul)
reaches on the ViT-B/16 GEMM shapes on this GPU, to judge the headroom of the hand-written kernels."""
import torch, time
M = 197 * 256
for (N, K, name) in [(2304, 768, "qkv"), (3072, 768, "c_fc"), (768, 3072, "c_proj"), (768, 768, "out_proj")]:
    a = torch.randn(M, K, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.02
    b = torch.randn(N, device="cuda", dtype=torch.float16)
    for fn, label in ((lambda: a @ w.t(), "matmul"), (lambda: torch.addmm(b, a, w.t()), "addmm")):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:9s} {label:7s} M={M} N={N} K={K}: {ms*1e3:7.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
