"""Crop pre-processing: oracle vs Pillow goldens, host tables vs oracle, HIP vs goldens/oracle."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hoigen_amd import preprocess  # noqa: E402
from oracle import preprocess_oracle as po  # noqa: E402

G = os.path.join(ROOT, "tests", "golden", "g6_preprocess.npz")


@pytest.fixture(scope="module")
def g6():
    return dict(np.load(G))


def test_oracle_matches_pillow_goldens(g6):
    """The restatement of Pillow's ImagingResample + crop/pad/centre-crop is bit-exact on the committed vectors."""
    for ci in range(2):
        u8, nrm = po.preprocess_boxes(g6[f"img{ci}"], g6[f"boxes{ci}"])
        assert np.array_equal(u8, g6[f"u8_{ci}"])
        u8s, _ = po.preprocess_boxes(g6[f"img{ci}"], g6[f"boxes{ci}"][:2], pad_square=True)
        assert np.array_equal(u8s, g6[f"u8_sq_{ci}"])
        if ci == 0:
            assert np.array_equal(nrm, g6["norm_0"])


def test_facade_errors_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="HIP device"):
        preprocess.CropPreprocessor()(torch.zeros(8, 8, 3, dtype=torch.uint8), [(0, 0, 8, 8)])
    with pytest.raises(TypeError):
        preprocess.CropPreprocessor()(torch.zeros(8, 8, 3), [(0, 0, 8, 8)])


@pytest.mark.gpu
def test_hip_preprocess_bit_exact_vs_pillow_goldens(g6):
    dev = torch.device("cuda:0")
    for ci in range(2):
        img = torch.from_numpy(g6[f"img{ci}"]).to(dev)
        out, u8 = preprocess.CropPreprocessor(224)(img, g6[f"boxes{ci}"], return_u8=True)
        assert np.array_equal(u8.cpu().numpy(), g6[f"u8_{ci}"]), "uint8 crops must be bit-exact"
        if ci == 0:
            assert np.abs(out.cpu().numpy() - g6["norm_0"]).max() <= 1e-6
        _, u8s = preprocess.CropPreprocessor(224, pad_square=True)(img, g6[f"boxes{ci}"][:2], return_u8=True)
        assert np.array_equal(u8s.cpu().numpy(), g6[f"u8_sq_{ci}"])
    img0 = torch.from_numpy(g6["img0"]).to(dev)
    assert preprocess.CropPreprocessor()(img0, np.zeros((0, 4), np.int32)).shape == (0, 3, 224, 224)
    with pytest.raises(ValueError):
        preprocess.CropPreprocessor()(img0, [(10, 10, 10, 20)])


@pytest.mark.gpu
def test_hip_preprocess_vs_oracle_random_boxes_and_pipeline():
    """Random image/boxes (incl. out-of-image, coloured square padding) vs the oracle, bit-exact; and the output
    feeds encode_image directly."""
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(7)
    img = rng.randint(0, 256, size=(301, 457, 3)).astype(np.uint8)
    boxes = []
    for _ in range(12):
        x0, x1 = sorted(rng.randint(-30, 487, size=2)); y0, y1 = sorted(rng.randint(-30, 331, size=2))
        boxes.append((x0, y0, max(x1, x0 + 5), max(y1, y0 + 5)))
    boxes += [(3, 4, 4, 300), (0, 0, 457, 2), (100, 100, 325, 324)]      # 1-pixel wide, 2 rows high, 225 x 224
    boxes = np.asarray(boxes, np.int32)
    for pad, bg in ((False, (0, 0, 0)), (True, (122, 116, 104))):
        want_u8, want = po.preprocess_boxes(img, boxes, 224, pad, bg)
        out, u8 = preprocess.CropPreprocessor(224, pad, bg)(torch.from_numpy(img).to(dev), boxes, return_u8=True)
        assert np.array_equal(u8.cpu().numpy(), want_u8)
        assert np.abs(out.cpu().numpy() - want).max() <= 1e-6
    from hoigen_amd import synth
    from hoigen_amd.model import build_model
    m = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(dev)
    emb = m.encode_image(out)
    assert emb.shape == (15, 512) and torch.isfinite(emb).all()
