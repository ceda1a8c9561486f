"""Feature-regeneration records: the producer of the ``union_embeddings_cachemodel_*.p`` entries that
``UPT.load_cache_model`` reads (/root/reference/upt_tip_cache_model_free_finetune_distill3.py:636-688):

    annotation[file_name] = {'boxes_h': [n,4], 'boxes_o': [n,4], 'verbs': [n], 'objects': [n],
                             'union_features': [n,512], 'object_features': [n,512], 'huamn_features': [n,512]}

(``huamn_features`` is the reference's spelling, :680.)  The reference ships only the consumer; the producer that
made its pickles is not in the tree.  Here every human-object pair of one image is turned into three crops - human
box, object box and their union box (min/max corners, pre_images/crop_images.py:207-209) - pre-processed on the
device (``CropPreprocessor``: PIL crop + CLIP transform, bit-exact with Pillow) and encoded with ``encode_image`` in
ONE batch of 3n crops; features are returned un-normalised (the consumer divides by the norm itself, :678-680).
BASELINE.json config 5 ("pre-extracted-feature regeneration") is this loop over a dataset with the batch sharded
over the GPUs (hoigen_amd.distributed).
"""
from typing import Dict

import numpy as np
import torch

from .preprocess import CropPreprocessor


def union_boxes(boxes_h: np.ndarray, boxes_o: np.ndarray) -> np.ndarray:
    """[min(x1), min(y1), max(x2), max(y2)] of every pair (pre_images/crop_images.py:207-209)."""
    bh, bo = np.asarray(boxes_h).reshape(-1, 4), np.asarray(boxes_o).reshape(-1, 4)
    return np.concatenate([np.minimum(bh[:, :2], bo[:, :2]), np.maximum(bh[:, 2:], bo[:, 2:])], axis=1)


def pil_box(boxes: np.ndarray) -> np.ndarray:
    """Float boxes -> the integer box PIL's ``image.crop`` uses: each coordinate rounded to the nearest integer
    (ImageCrop: ``int(round(x))``)."""
    return np.rint(np.asarray(boxes, np.float64)).astype(np.int32)


@torch.no_grad()
def emit_record(clip_model, image_u8: torch.Tensor, boxes_h, boxes_o, verbs, objects, n_px: int = 224,
                preprocessor: CropPreprocessor = None) -> Dict[str, np.ndarray]:
    """One image -> one record.  ``image_u8``: uint8 [H,W,3] on the HIP device; boxes (x1,y1,x2,y2) in pixels."""
    bh = np.asarray(boxes_h, np.float32).reshape(-1, 4)
    bo = np.asarray(boxes_o, np.float32).reshape(-1, 4)
    n = bh.shape[0]
    if bo.shape[0] != n or len(verbs) != n or len(objects) != n:
        raise ValueError("boxes_h, boxes_o, verbs and objects must describe the same pairs")
    pre = preprocessor or CropPreprocessor(n_px)
    rec = {"boxes_h": bh, "boxes_o": bo, "verbs": np.asarray(verbs, np.int64), "objects": np.asarray(objects, np.int64)}
    if n == 0:
        z = np.zeros((0, clip_model.visual.output_dim), np.float32)
        rec.update(union_features=z, object_features=z.copy(), huamn_features=z.copy())
        return rec
    crops = np.concatenate([pil_box(union_boxes(bh, bo)), pil_box(bo), pil_box(bh)], axis=0)
    feats = clip_model.encode_image(pre(image_u8, crops)).float().cpu().numpy()
    rec.update(union_features=feats[:n], object_features=feats[n:2 * n], huamn_features=feats[2 * n:])
    return rec
