"""Device-side crop pre-processing in front of ``encode_image``.

Counterpart of the reference's PIL pipeline
    ``image.crop(box)``                               pre_images/crop_images.py:204-219
    ``expand2square(crop, background)`` (optional)    utils_tip_cache_and_union_finetune.py:201-212
    ``_transform(n_px)`` = Resize(BICUBIC) / CenterCrop / ToTensor / Normalize    clipnet/clip.py:75-82
for all boxes of one image in three kernel launches (``hg_preprocess_crops``), producing the fp32
``[n, 3, n_px, n_px]`` batch ``encode_image`` consumes.  The resampling reproduces Pillow's 8-bit
``ImagingResample`` bit for bit: the native side works out the crop geometry on the host, a kernel fills the
fixed-point weight tables (IEEE double, the formulas of Pillow's ``precompute_coeffs`` /
``normalize_coeffs_8bpc``) and two kernels do the integer accumulation.  There is no CPU fallback: CPU tensors
raise ``RuntimeError``.
"""
from typing import Sequence

import numpy as np
import torch

from . import _lib


class CropPreprocessor:
    """``CropPreprocessor(n_px)(image_u8, boxes) -> float32 [n, 3, n_px, n_px]`` on the image's device.

    ``image_u8``: uint8 tensor [H, W, 3] (RGB) on a HIP device; ``boxes``: integer (x0, y0, x1, y1) per crop in
    PIL convention (may leave the image: the outside is 0).  ``pad_square=True`` applies ``expand2square`` with
    ``background`` before the resize.  ``return_u8=True`` also returns the resized uint8 crops [n, n_px, n_px, 3].
    """

    def __init__(self, n_px: int = 224, pad_square: bool = False, background: Sequence[int] = (0, 0, 0),
                 stretch: bool = False, imagenet_norm: bool = False):
        """``stretch``: ``IResize([n_px, n_px])`` of the detector's CLIP view (both sides to n_px, no centre crop;
        detr/datasets/transforms_clip.py:279-288); ``imagenet_norm``: ImageNet mean/std
        (utils_tip_cache_and_union_finetune.py:86-89) instead of CLIP's (clipnet/clip.py:81)."""
        self.n_px, self.pad_square, self.background = int(n_px), bool(pad_square), tuple(background)
        self.flags = (1 if pad_square else 0) | (2 if stretch else 0) | (4 if imagenet_norm else 0)

    def __call__(self, image_u8: torch.Tensor, boxes, return_u8: bool = False):
        if not isinstance(image_u8, torch.Tensor) or image_u8.dtype != torch.uint8 or image_u8.dim() != 3 \
                or image_u8.shape[2] != 3:
            raise TypeError("image must be a uint8 tensor [H, W, 3]")
        if image_u8.device.type != "cuda":
            raise RuntimeError("hoigen_amd: crop pre-processing runs only on a HIP device (there is no CPU fallback)")
        boxes = np.asarray(boxes.cpu() if isinstance(boxes, torch.Tensor) else boxes)
        boxes = boxes.reshape(-1, 4)
        n, npx = boxes.shape[0], self.n_px
        dev = image_u8.device
        out = torch.empty(n, 3, npx, npx, dtype=torch.float32, device=dev)
        u8 = torch.empty(n, npx, npx, 3, dtype=torch.uint8, device=dev) if return_u8 else None
        if n:
            if ((boxes[:, 2] <= boxes[:, 0]) | (boxes[:, 3] <= boxes[:, 1])).any():
                raise ValueError("empty crop box")
            bx = np.ascontiguousarray(boxes, dtype=np.int32)
            bg = (int(self.background[0]) & 255) | ((int(self.background[1]) & 255) << 8) | \
                 ((int(self.background[2]) & 255) << 16)
            img = image_u8.contiguous()
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
            with torch.cuda.device(dev):
                rc = _lib.lib().hg_preprocess_crops(_lib.ctx(idx), img.data_ptr(), img.shape[0], img.shape[1],
                                                    bx.ctypes.data, n, npx, self.flags, bg,
                                                    out.data_ptr(), u8.data_ptr() if return_u8 else None,
                                                    torch.cuda.current_stream().cuda_stream)
            _lib.check(idx, rc, "hg_preprocess_crops")
        return (out, u8) if return_u8 else out
