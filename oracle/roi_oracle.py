"""TEST INFRASTRUCTURE ONLY - CPU restatement of RoI-align over the local feature map.
Nothing under hoigen_amd/ imports this file.

Reference call sites (upt_tip_cache_model_free_finetune_distill3.py:1026-1037):
    spatial_scale = 1 / (image_size[0,0] / local_features.shape[1])
    f = torchvision.ops.roi_align(local_features.unsqueeze(0), [boxes], output_size=(7, 7),
                                  spatial_scale=spatial_scale, aligned=True)          # [n, C, 7, 7]
    f = f.flatten(2).mean(-1)                                                          # [n, C]
The arithmetic lives in a third-party dependency that is NOT installed in this image (torchvision; the reference
does not pin a version).  Its published algorithm (torchvision/csrc/ops/cpu/roi_align_kernel.cpp and
roi_align_common.h, unchanged since 0.7) is restated here:
  * aligned=True: offset 0.5; roi_start = box * spatial_scale - 0.5, roi size = end - start (not clamped);
  * sampling_ratio = -1 (default): per-bin grid = ceil(roi_h / pooled_h) x ceil(roi_w / pooled_w), count =
    max(grid_h * grid_w, 1); samples at start + ph * bin + (iy + .5) * bin / grid;
  * bilinear interpolation with the boundary rule: outside [-1, H] x [-1, W] -> 0; coordinates <= 0 -> 0;
    low index >= size - 1 -> both indices = size - 1 (weight on it 1); float32 arithmetic.
PARITY UNPINNED against the real library: no torchvision here to generate vectors from (and no network to get it).
tests/test_roi_align.py anchors this restatement on (a) analytic properties (constant and affine feature maps, for
which bilinear sampling is exact) and (b) hand-derived known-answer vectors (tests/golden/roi_known_answers.py) that
separate the aligned=True half-pixel offset, the adaptive ceil(roi/P) sampling grid and the boundary rules from their
alternatives.
"""
import math

import numpy as np


def roi_align(feat, boxes, output_size=7, spatial_scale=1.0, aligned=True):
    """feat [C, H, W] float32, boxes [n, 4] (x1, y1, x2, y2) -> [n, C, P, P] float32."""
    feat = np.asarray(feat, np.float32)
    C, H, W = feat.shape
    P = int(output_size)
    boxes = np.asarray(boxes, np.float32).reshape(-1, 4)
    out = np.zeros((boxes.shape[0], C, P, P), np.float32)
    f32 = np.float32
    off = f32(0.5) if aligned else f32(0.0)
    for n, b in enumerate(boxes):
        sw, sh = b[0] * f32(spatial_scale) - off, b[1] * f32(spatial_scale) - off
        ew, eh = b[2] * f32(spatial_scale) - off, b[3] * f32(spatial_scale) - off
        rw, rh = ew - sw, eh - sh
        if not aligned:
            rw, rh = max(rw, f32(1.0)), max(rh, f32(1.0))
        bh, bw = f32(rh) / f32(P), f32(rw) / f32(P)
        gh, gw = int(math.ceil(float(rh) / P)), int(math.ceil(float(rw) / P))
        count = f32(max(gh * gw, 1))
        for ph in range(P):
            for pw in range(P):
                acc = np.zeros(C, np.float32)
                for iy in range(gh):
                    y = f32(sh) + f32(ph) * bh + (f32(iy) + f32(0.5)) * bh / f32(gh)
                    for ix in range(gw):
                        x = f32(sw) + f32(pw) * bw + (f32(ix) + f32(0.5)) * bw / f32(gw)
                        acc += _bilinear(feat, H, W, y, x)
                out[n, :, ph, pw] = acc / count
    return out


def _bilinear(feat, H, W, y, x):
    f32 = np.float32
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return np.zeros(feat.shape[0], np.float32)
    y, x = max(f32(y), f32(0.0)), max(f32(x), f32(0.0))
    yl, xl = int(y), int(x)
    if yl >= H - 1:
        yh = yl = H - 1
        y = f32(yl)
    else:
        yh = yl + 1
    if xl >= W - 1:
        xh = xl = W - 1
        x = f32(xl)
    else:
        xh = xl + 1
    ly, lx = f32(y) - f32(yl), f32(x) - f32(xl)
    hy, hx = f32(1.0) - ly, f32(1.0) - lx
    return (hy * hx) * feat[:, yl, xl] + (hy * lx) * feat[:, yl, xh] + (ly * hx) * feat[:, yh, xl] + (ly * lx) * feat[:, yh, xh]


def roi_align_mean(feat, boxes, output_size=7, spatial_scale=1.0):
    """roi_align(...).flatten(2).mean(-1) (upt…:1034-1037) -> [n, C]."""
    r = roi_align(feat, boxes, output_size, spatial_scale, True)
    return r.reshape(r.shape[0], r.shape[1], -1).mean(-1, dtype=np.float32)
