"""RoI-align over the variant-C local feature map.

Counterpart of upt_tip_cache_model_free_finetune_distill3.py:1026-1037:
    spatial_scale = 1 / (image_size[0, 0] / local_features.shape[1])
    f = torchvision.ops.roi_align(local_features.unsqueeze(0), [boxes], output_size=(7, 7),
                                  spatial_scale=spatial_scale, aligned=True)
    f = f.flatten(2).mean(-1)
as one kernel (``hg_roi_align``).  CPU tensors raise ``RuntimeError`` (there is no CPU fallback).
"""
import torch

from . import _lib


def roi_align(local_features: torch.Tensor, boxes: torch.Tensor, output_size=(7, 7), spatial_scale: float = 1.0,
              aligned: bool = True, reduce_mean: bool = False) -> torch.Tensor:
    """``local_features`` [C,H,W] or [1,C,H,W] fp32, ``boxes`` [n,4] (x1,y1,x2,y2) -> [n,C,P,P]
    (``reduce_mean=True``: ``.flatten(2).mean(-1)`` -> [n,C] without materialising the pooled map)."""
    if local_features.device.type != "cuda":
        raise RuntimeError("hoigen_amd: roi_align runs only on a HIP device (there is no CPU fallback)")
    if not aligned:
        raise NotImplementedError("the reference only uses aligned=True (upt…:1027)")
    if isinstance(output_size, int):
        output_size = (output_size, output_size)
    if output_size[0] != output_size[1]:
        raise NotImplementedError("square output only")
    f = local_features
    if f.dim() == 4:
        if f.shape[0] != 1:
            raise ValueError("one image per call (the reference passes local_features.unsqueeze(0))")
        f = f[0]
    f = f.detach().float().contiguous()
    b = boxes.detach().to(f.device).float().contiguous().reshape(-1, 4)
    C, H, W = f.shape
    n, P = b.shape[0], int(output_size[0])
    dev = f.device.index if f.device.index is not None else torch.cuda.current_device()
    pooled = None if reduce_mean else torch.empty(n, C, P, P, dtype=torch.float32, device=f.device)
    mean = torch.empty(n, C, dtype=torch.float32, device=f.device) if reduce_mean else None
    with torch.cuda.device(dev):
        rc = _lib.lib().hg_roi_align(_lib.ctx(dev), f.data_ptr(), C, H, W, b.data_ptr(), n, float(spatial_scale), P,
                                     pooled.data_ptr() if pooled is not None else None,
                                     mean.data_ptr() if mean is not None else None,
                                     torch.cuda.current_stream().cuda_stream)
    _lib.check(dev, rc, "hg_roi_align")
    return mean if reduce_mean else pooled
