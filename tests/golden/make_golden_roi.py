"""g11_roi.npz: outputs of the REAL torchvision.ops.roi_align (CPU kernel) for the call the reference makes
(upt_tip_cache_model_free_finetune_distill3.py:1026-1037):

    f = torchvision.ops.roi_align(local_features.unsqueeze(0), [boxes], output_size=(7, 7),
                                  spatial_scale=spatial_scale, aligned=True)
    f = f.flatten(2).mean(-1)

torchvision is a third-party dependency of the reference (un-vendored, unpinned); it is not installed in the build
container, so this script is meant to run wherever `import torchvision` works (the GPU boxes run the same PyTorch-ROCm
image: `gpurun -- python tests/golden/make_golden_roi.py gpurun_out/g11_roi.npz`, then copy the file here).  When the
import fails the script records the error text in gpurun_out/g11_roi_import_error.txt and exits 3.

Inputs are regenerated from seeds by `roi_cases()` below (the fixture stores them too, so the tests do not depend on
numpy's generator staying bit-stable): a seeded [64, 14, 14] map (512 channels add nothing: channels are independent),
random boxes reaching outside the 224 x 224 image, and the degenerate / out-of-map / whole-image boxes of
tests/test_roi_align.py; a second, non-square map [8, 9, 20] with its own scale; aligned=False once (the alternative
the reference does not use) so that the oracle's handling of the flag is pinned as well."""
import os
import sys

import numpy as np


def roi_cases():
    rng = np.random.RandomState(11)
    feat = rng.randn(64, 14, 14).astype(np.float32)
    boxes = []
    for _ in range(40):
        x1, x2 = sorted(rng.uniform(-30, 260, size=2))
        y1, y2 = sorted(rng.uniform(-30, 260, size=2))
        boxes.append((x1, y1, x2 + 1, y2 + 1))
    boxes += [(0, 0, 224, 224), (10, 10, 10.5, 10.5), (300, 300, 400, 400), (-50, -50, 5, 5), (16, 16, 16, 16),
              (0, 0, 16, 16), (208, 208, 224, 224), (100, 0, 101, 224), (0, 100, 224, 101), (223.9, 223.9, 224, 224)]
    boxes = np.asarray(boxes, np.float32)
    feat2 = rng.randn(8, 9, 20).astype(np.float32)
    boxes2 = np.asarray([(0, 0, 320, 144), (13.7, 5.2, 200.1, 99.9), (-20, -20, 50, 50), (250, 100, 400, 200),
                         (5, 5, 6, 6)], np.float32)
    return [("sq14", feat, boxes, 7, 14.0 / 224.0, True), ("rect", feat2, boxes2, 7, 1.0 / 16.0, True),
            ("sq14_unaligned", feat, boxes[:12], 7, 14.0 / 224.0, False), ("sq14_p3", feat, boxes[:12], 3, 14.0 / 224.0, True)]


def main(out_path):
    try:
        import torch
        import torchvision
        from torchvision.ops import roi_align
    except Exception as e:  # noqa: BLE001 - the text of whatever went wrong is the record
        err = os.path.join(os.path.dirname(out_path) or ".", "g11_roi_import_error.txt")
        with open(err, "w") as f:
            f.write(f"{type(e).__name__}: {e}\n")
        print(f"torchvision not importable: {type(e).__name__}: {e}")
        return 3
    data = {"torch_version": np.array(torch.__version__), "torchvision_version": np.array(torchvision.__version__)}
    for name, feat, boxes, P, scale, aligned in roi_cases():
        f = torch.from_numpy(feat).unsqueeze(0)
        out = roi_align(f, [torch.from_numpy(boxes)], output_size=(P, P), spatial_scale=scale, aligned=aligned)
        data[f"{name}_feat"] = feat
        data[f"{name}_boxes"] = boxes
        data[f"{name}_meta"] = np.array([P, scale, 1.0 if aligned else 0.0], np.float64)
        data[f"{name}_pooled"] = out.numpy().astype(np.float32)
        data[f"{name}_mean"] = out.flatten(2).mean(-1).numpy().astype(np.float32)
    np.savez_compressed(out_path, **data)
    print(f"wrote {out_path}: torchvision {torchvision.__version__}, torch {torch.__version__}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "g11_roi.npz")))
