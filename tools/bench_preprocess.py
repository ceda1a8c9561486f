#!/usr/bin/env python3
"""Measurement for SURVEY.md §8f-2 (crop pre-processing in front of encode_image): 256 boxes of one 480x640 image
-> [256,3,224,224] fp32.  Prints one JSON line: crops/s on the GPU (inputs resident), achieved HBM GB/s against the
8 TB/s roofline (algorithmic bytes = crop pixels read once + fp32 output written once), and Pillow on the host
cores as the CPU baseline (the third-party library the reference calls)."""
import json
import os
import sys
import time

import numpy as np
import torch
torch.set_grad_enabled(False)   # inference only

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoigen_amd import preprocess  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(0)
    H, W, n = 480, 640, 256
    img = rng.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
    boxes = []
    for _ in range(n):
        x0, y0 = rng.randint(0, W - 60), rng.randint(0, H - 60)
        boxes.append((x0, y0, min(W, x0 + rng.randint(40, 320)), min(H, y0 + rng.randint(40, 280))))
    boxes = np.asarray(boxes, np.int32)
    pre = preprocess.CropPreprocessor(224)
    img_d = torch.from_numpy(img).to(dev)
    for _ in range(3):
        out = pre(img_d, boxes)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        out = pre(img_d, boxes)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    bytes_alg = int(sum((b[2] - b[0]) * (b[3] - b[1]) * 3 for b in boxes) + n * 3 * 224 * 224 * 4)
    # CPU baseline: Pillow on a bounded sample
    from PIL import Image
    pil = Image.fromarray(img)
    m = 64
    t0 = time.perf_counter()
    for b in boxes[:m]:
        c = pil.crop(tuple(int(v) for v in b))
        w, h = c.size
        nw, nh = (224, int(224 * h / w)) if w <= h else (int(224 * w / h), 224)
        r = c.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - 224) / 2.0)), int(round((nh - 224) / 2.0))
        a = np.asarray(r.crop((left, top, left + 224, top + 224)), np.float32) / 255.0
    cpu = m / (time.perf_counter() - t0)
    print(json.dumps({"what": "crop + bicubic resize 224 + normalise, 256 boxes of a 480x640 image (SURVEY 8f-2)",
                      "ms": round(dt * 1e3, 3), "crops_per_s": round(n / dt, 1),
                      "roofline": {"bound": "hbm", "achieved": round(bytes_alg / dt / 1e9, 1), "peak": 8000.0,
                                   "unit": "GB/s", "frac": round(bytes_alg / dt / 8e12, 4),
                                   "algorithmic_bytes": bytes_alg},
                      "cpu_baseline": {"value": round(cpu, 1), "unit": "crops/s", "cores": 1, "kind": "reference",
                                       "sample": f"Pillow crop+resize+centre-crop of the first {m} boxes"}}))


if __name__ == "__main__":
    main()
