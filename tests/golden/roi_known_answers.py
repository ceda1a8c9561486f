"""Hand-derived known-answer vectors for torchvision.ops.roi_align(..., aligned=True, sampling_ratio=-1)
(upt_tip_cache_model_free_finetune_distill3.py:1027-1037).  torchvision is NOT installed in this image (and cannot be
installed: no network), so these are not outputs of the real library; they are worked out by hand from its published
rules (roi_align_common.h) and chosen so that each one separates the rule it pins from the plausible alternatives.
Each case: (feature map [C,H,W], box (x1,y1,x2,y2), spatial_scale, output size P, expected [C,P,P], what it pins).
"""
import numpy as np


def cases():
    out = []
    # --- A: the half-pixel offset of aligned=True.  f[y][x] = 4y + x on 4x4, box (0.5,0.5)-(2.5,2.5), P = 2.
    # aligned=True: start = 0.5 - 0.5 = 0, RoI 2x2, bins 1x1, grid ceil(2/2) = 1 -> one sample per bin at the bin
    # centre (y, x) in {0.5, 1.5}^2; bilinear interpolation of an affine map is exact: 4y + x.
    # (aligned=False would sample at {1, 2}^2 and give 5, 6, 9, 10.)
    f = (4 * np.arange(4)[:, None] + np.arange(4)[None, :]).astype(np.float32)[None]
    out.append((f, (0.5, 0.5, 2.5, 2.5), 1.0, 2, np.array([[[2.5, 3.5], [6.5, 7.5]]], np.float32),
                "aligned=True subtracts 0.5 after scaling"))
    # --- B: the adaptive sampling grid ceil(roi / P).  f[y][x] = x^2 on 2x8 (rows equal), box (0.5,0.5)-(6.5,1.5), P = 2.
    # start_x = 0, RoI width 6 -> bin 3, grid_w = ceil(6/2) = 3: samples at x = 0.5, 1.5, 2.5 | 3.5, 4.5, 5.5.
    # Linear interpolation of x^2 between integers at k + 0.5 is (k^2 + (k+1)^2) / 2 = 0.5, 2.5, 6.5 | 12.5, 20.5, 30.5
    # -> bin means 9.5/3 and 63.5/3.  RoI height 1 -> bin 0.5, grid_h = ceil(1/2) = 1; rows are equal, so y is irrelevant.
    # (a fixed sampling_ratio = 2 would sample x = 0.75, 2.25 -> (0.75 + 5.25) / 2 = 3.0.)
    f = np.tile((np.arange(8, dtype=np.float32) ** 2)[None, :], (2, 1))[None]
    e = np.array([[[9.5 / 3, 63.5 / 3], [9.5 / 3, 63.5 / 3]]], np.float32)
    out.append((f, (0.5, 0.5, 6.5, 1.5), 1.0, 2, e, "sampling grid = ceil(roi_size / output_size) per axis"))
    # --- C: boundary rules.  f = [[1,2],[3,4]], P = 2.
    # box (-1,-1)-(3,3): start -1.5, RoI 4, bin 2, grid 2: samples at -1, 0 | 1, 2 on both axes.  Coordinates in [-1, 0]
    # are clamped to 0 (pixel 0 with weight 1); a coordinate equal to W = 2 is still inside ("> W" is the cut) and, its
    # low index being >= W - 1, reads pixel 1 with weight 1 -> every bin sees a single pixel: [[1,2],[3,4]].
    f = np.array([[[1, 2], [3, 4]]], np.float32)
    out.append((f, (-1, -1, 3, 3), 1.0, 2, np.array([[[1, 2], [3, 4]]], np.float32),
                "coordinates in [-1, 0] clamp to 0; coordinate == size is inside and reads the last pixel"))
    # box (-3,-3)-(1,1): start -3.5, samples at -3, -2 | -1, 0: the first bin of each axis lies below -1 -> contributes 0
    # (and still counts in the divisor); the second bin reads pixel (0,0) four times -> [[0,0],[0,1]].
    out.append((f, (-3, -3, 1, 1), 1.0, 2, np.array([[[0, 0], [0, 1]]], np.float32),
                "samples below -1 are zero but stay in the average"))
    # --- D: spatial_scale is applied before the offset.  f[y][x] = x on 1x... 2x4 map, box (16,0)-(48,32) in a 64-pixel
    # image, scale 4/64: x: 16/16 - 0.5 = 0.5 .. 48/16 - 0.5 = 2.5 -> RoI width 2, P = 2, bins 1, grid 1: samples at
    # x = 1.0, 2.0 -> 1, 2.  y: 0 - 0.5 = -0.5 .. 1.5: bins 1, samples at y = 0, 1: rows equal -> same values.
    f = np.tile(np.arange(4, dtype=np.float32)[None, :], (2, 1))[None]
    out.append((f, (16, 0, 48, 32), 4.0 / 64.0, 2, np.array([[[1, 2], [1, 2]]], np.float32),
                "box * spatial_scale - 0.5"))
    return out
