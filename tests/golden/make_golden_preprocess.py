#!/usr/bin/env python3
"""Generates tests/golden/g6_preprocess.npz with Pillow itself (the third-party library whose resampling the
reference's crop pre-processing calls: pre_images/crop_images.py:204-219 `image.crop`, clipnet/clip.py:75-82
Resize(BICUBIC)/CenterCrop/ToTensor/Normalize through torchvision's PIL path).  torchvision is not installed
here; its two size rules (short side -> n_px, long side -> int(n_px * long / short); CenterCrop offset
int(round((size - n_px) / 2.0))) are applied explicitly.  Run in the build container:  python tests/golden/make_golden_preprocess.py
"""
import os

import numpy as np
from PIL import Image

MEAN = np.asarray((0.48145466, 0.4578275, 0.40821073), np.float32)
STD = np.asarray((0.26862954, 0.26130258, 0.27577711), np.float32)


def expand2square(pil_img, background_color):      # utils_tip_cache_and_union_finetune.py:201-212 (behaviour)
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    res = Image.new(pil_img.mode, (side, side), background_color)
    res.paste(pil_img, (0, (w - h) // 2) if w > h else ((h - w) // 2, 0))
    return res


def transform(pil_img, n=224):
    w, h = pil_img.size
    nw, nh = (n, int(n * h / w)) if w <= h else (int(n * w / h), n)
    r = pil_img.resize((nw, nh), Image.BICUBIC)
    left, top = int(round((nw - n) / 2.0)), int(round((nh - n) / 2.0))
    return np.asarray(r.crop((left, top, left + n, top + n)).convert("RGB"), np.uint8)


def main():
    rng = np.random.RandomState(20241022)
    out = {}
    cases = [(240, 320), (97, 61)]
    for ci, (H, W) in enumerate(cases):
        # smooth + noisy content so that both interpolation and clamping paths are exercised
        yy, xx = np.mgrid[0:H, 0:W]
        base = np.stack([128 + 120 * np.sin(xx / 7.0 + c) * np.cos(yy / 11.0 - c) for c in range(3)], -1)
        img = np.clip(base + rng.randint(-12, 12, size=(H, W, 3)), 0, 255).astype(np.uint8)
        img[rng.rand(H, W) < 0.01] = 255
        img[rng.rand(H, W) < 0.01] = 0
        boxes = []
        for _ in range(4):
            x0, x1 = sorted(rng.randint(-20, W + 20, size=2)); y0, y1 = sorted(rng.randint(-20, H + 20, size=2))
            if x1 - x0 < 8: x1 = x0 + 8 + rng.randint(0, 50)
            if y1 - y0 < 8: y1 = y0 + 8 + rng.randint(0, 50)
            boxes.append((x0, y0, x1, y1))
        boxes.append((0, 0, W, H))                        # whole image
        boxes.append((W // 8, H // 8, W // 8 + 200, H // 8 + 200) if min(H, W) > 220 else (3, 5, 3 + 30, 5 + 30))  # square
        boxes.append((5, 7, 5 + 40, 7 + 12))              # strong upscale, wide
        boxes = np.asarray(boxes, np.int32)
        pil = Image.fromarray(img)
        u8, u8_sq = [], []
        for bi, b in enumerate(boxes):
            c = pil.crop(tuple(int(v) for v in b))
            u8.append(transform(c))
            if bi < 2:                                    # square padding: first two boxes only (fixture size)
                u8_sq.append(transform(expand2square(c, (0, 0, 0))))
        u8, u8_sq = np.stack(u8), np.stack(u8_sq)
        out[f"img{ci}"] = img
        out[f"boxes{ci}"] = boxes
        out[f"u8_{ci}"] = u8
        out[f"u8_sq_{ci}"] = u8_sq
    # the detector's CLIP view: IResize([224, 224]) of the whole image and of one box, Pillow's resize to both sides
    for ci in range(len(cases)):
        pil = Image.fromarray(out[f"img{ci}"])
        W, H = pil.size
        bs = [(0, 0, W, H), tuple(int(v) for v in out[f"boxes{ci}"][0])]
        out[f"stretch_boxes{ci}"] = np.asarray(bs, np.int32)
        out[f"stretch_u8_{ci}"] = np.stack([np.asarray(pil.crop(b).resize((224, 224), Image.BICUBIC).convert("RGB"), np.uint8)
                                           for b in bs])
    # normalised float output for one case (ToTensor + Normalize)
    x = out["u8_0"].astype(np.float32) / np.float32(255.0)
    out["norm_0"] = np.ascontiguousarray(((x - MEAN) / STD).transpose(0, 3, 1, 2))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g6_preprocess.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
