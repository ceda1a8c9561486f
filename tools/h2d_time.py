#!/usr/bin/env python3
"""Host -> device staging of one headline batch (256 x 3 x 224 x 224 fp32 = 154 MB), pinned and pageable: what a caller that
holds its crops in host memory adds in front of hg_encode_image (the C ABI takes device pointers; bench.py's `value` starts with
the batch resident in HBM).  One JSON line."""
import json, time, torch
x = torch.randn(256, 3, 224, 224)
xp = x.pin_memory()
d = torch.empty_like(x, device="cuda:0")
def t(src, n=10):
    for _ in range(2): d.copy_(src, non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): d.copy_(src, non_blocking=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
mp, mg = t(xp), t(x)
mb = x.numel() * 4 / 1e6
print(json.dumps({"batch_mb": round(mb, 1), "pinned_ms": round(mp, 3), "pinned_gbs": round(mb / mp, 1), "pageable_ms": round(mg, 3), "pageable_gbs": round(mb / mg, 1)}))
