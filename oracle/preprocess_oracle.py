"""TEST INFRASTRUCTURE ONLY - CPU restatement of the crop pre-processing in front of `encode_image`
(SURVEY.md §8f-2).  Nothing under hoigen_amd/ imports this file.

Path restated (reference file:line):
  * box crop:     `image.crop((x0, y0, x1, y1))`                      pre_images/crop_images.py:204-219
                  (PIL semantics: integer box, pixels outside the image are 0)
  * square pad:   `expand2square(pil_img, background_color)`           utils_tip_cache_and_union_finetune.py:201-212
                  (optional; the crop is pasted centred, `(side - short) // 2` from the top/left)
  * CLIP transform `_transform(n_px)`: Resize(n_px, BICUBIC) -> CenterCrop(n_px) -> ToTensor -> Normalize
                                                                       clipnet/clip.py:75-82
The arithmetic of Resize lives in a third-party dependency that is not vendored in the reference: Pillow's
`ImagingResample` (src/libImaging/Resample.c; Pillow 12.2.0 in this image) called through torchvision's
`Resize` on PIL images.  Its published algorithm, restated here:
  - separable, horizontal pass first, then vertical, each pass rounding to uint8;
  - per output index: centre = (i + 0.5) * scale, support = 2 * max(scale, 1), taps [xmin, xmax) =
    [int(centre - support + 0.5), int(centre + support + 0.5)) clamped to the image, weights = bicubic
    (a = -0.5) of (x - centre + 0.5) / max(scale, 1), normalised to sum 1 in double precision;
  - 8-bit path: weights to fixed point with 22 fractional bits (round half away from zero), accumulate in
    int32 starting from 1 << 21, arithmetic shift by 22, clamp to 0..255.
torchvision's size rule: short side -> n_px, long side -> int(n_px * long / short); CenterCrop offsets are
int(round((size - n_px) / 2.0)).

Pinned: tests/golden/g6_preprocess.npz holds outputs of Pillow itself (tests/golden/make_golden_preprocess.py)
for random images and boxes, including boxes that leave the image; tests/test_oracle_vs_golden.py compares
this restatement bit-exactly (uint8) with them.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the box (0, in_size).
    Returns (bounds [out,2] int32 = (xmin, count), kk [out, ksize] int32)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        for x, w in enumerate(k):
            kk[xx, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One pass of ImagingResample (8 bits per channel) along `axis` (0 = vertical, 1 = horizontal)."""
    in_size = img.shape[axis]
    bounds, kk = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for i in range(out_size):
        xmin, cnt = bounds[i]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for t in range(cnt):
            acc += src[xmin + t] * int(kk[i, t])
        out[i] = _clip8(acc)
    return np.moveaxis(out, 0, axis)


def resize_bicubic(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """PIL `Image.resize((out_w, out_h), BICUBIC)` for a uint8 [H, W, C] image: horizontal, then vertical."""
    h, w = img.shape[:2]
    if w != out_w:
        img = resample_axis(img, out_w, 1)
    if h != out_h:
        img = resample_axis(img, out_h, 0)
    return img


def crop_box(img: np.ndarray, box) -> np.ndarray:
    """PIL `Image.crop(box)`: integer (x0, y0, x1, y1), zeros outside the image (crop_images.py:204-219)."""
    x0, y0, x1, y1 = (int(v) for v in box)
    h, w = img.shape[:2]
    out = np.zeros((max(y1 - y0, 0), max(x1 - x0, 0), img.shape[2]), np.uint8)
    sx0, sy0, sx1, sy1 = max(x0, 0), max(y0, 0), min(x1, w), min(y1, h)
    if sx1 > sx0 and sy1 > sy0:
        out[sy0 - y0:sy1 - y0, sx0 - x0:sx1 - x0] = img[sy0:sy1, sx0:sx1]
    return out


def expand2square(img: np.ndarray, background=(0, 0, 0)) -> np.ndarray:
    """utils_tip_cache_and_union_finetune.py:201-212."""
    h, w = img.shape[:2]
    if w == h:
        return img
    side = max(w, h)
    out = np.empty((side, side, img.shape[2]), np.uint8)
    out[:] = np.asarray(background, np.uint8)
    if w > h:
        out[(w - h) // 2:(w - h) // 2 + h, :] = img
    else:
        out[:, (h - w) // 2:(h - w) // 2 + w] = img
    return out


def resized_size(w: int, h: int, n_px: int):
    """torchvision Resize(n_px) on a PIL image: (new_w, new_h)."""
    if w <= h:
        return n_px, int(n_px * h / w)
    return int(n_px * w / h), n_px


def clip_transform_u8(img: np.ndarray, n_px: int = 224) -> np.ndarray:
    """Resize(n_px, BICUBIC) + CenterCrop(n_px) on a uint8 [H, W, 3] image -> uint8 [n_px, n_px, 3]."""
    h, w = img.shape[:2]
    nw, nh = resized_size(w, h, n_px)
    r = resize_bicubic(img, nw, nh)
    left, top = int(round((nw - n_px) / 2.0)), int(round((nh - n_px) / 2.0))
    return r[top:top + n_px, left:left + n_px]


def normalize(u8: np.ndarray) -> np.ndarray:
    """ToTensor + Normalize (clipnet/clip.py:80-81): uint8 [H, W, 3] -> float32 [3, H, W]."""
    x = u8.astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(CLIP_MEAN, np.float32)) / np.asarray(CLIP_STD, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def preprocess_boxes(img: np.ndarray, boxes, n_px: int = 224, pad_square: bool = False, background=(0, 0, 0),
                     stretch: bool = False, imagenet_norm: bool = False):
    """image uint8 [H, W, 3] + boxes [n, 4] -> (uint8 [n, n_px, n_px, 3], float32 [n, 3, n_px, n_px]).
    ``stretch``: the detector's CLIP view, ``IResize([n_px, n_px])`` = ``F.resize(img, (n_px, n_px), BICUBIC)``
    (detr/datasets/transforms_clip.py:139-171, :279-288), no centre crop; ``imagenet_norm``: the constants of
    utils_tip_cache_and_union_finetune.py:86-89."""
    u8 = []
    for b in boxes:
        c = crop_box(img, b)
        if pad_square:
            c = expand2square(c, background)
        u8.append(resize_bicubic(c, n_px, n_px) if stretch else clip_transform_u8(c, n_px))
    u8 = np.stack(u8)
    if imagenet_norm:
        m, sd = np.asarray(IMAGENET_MEAN, np.float32), np.asarray(IMAGENET_STD, np.float32)
        fl = [np.ascontiguousarray(((u.astype(np.float32) / np.float32(255.0) - m) / sd).transpose(2, 0, 1)) for u in u8]
        return u8, np.stack(fl)
    return u8, np.stack([normalize(u) for u in u8])
