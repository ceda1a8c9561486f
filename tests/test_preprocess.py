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
        # the detector's CLIP view: IResize([224, 224]) = resize of both sides, no centre crop
        u8r, nrm_r = po.preprocess_boxes(g6[f"img{ci}"], g6[f"stretch_boxes{ci}"], stretch=True, imagenet_norm=True)
        assert np.array_equal(u8r, g6[f"stretch_u8_{ci}"])
        want = (g6[f"stretch_u8_{ci}"].astype(np.float32) / 255.0 - np.float32([0.485, 0.456, 0.406])) / np.float32([0.229, 0.224, 0.225])
        assert np.abs(nrm_r - want.transpose(0, 3, 1, 2)).max() <= 1e-6


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
        # IResize([224, 224]) + ImageNet normalisation (the detector's CLIP view, utils_tip...:86-89,105-114)
        pre = preprocess.CropPreprocessor(224, stretch=True, imagenet_norm=True)
        outr, u8r = pre(img, g6[f"stretch_boxes{ci}"], return_u8=True)
        assert np.array_equal(u8r.cpu().numpy(), g6[f"stretch_u8_{ci}"])
        _, want = po.preprocess_boxes(g6[f"img{ci}"], g6[f"stretch_boxes{ci}"], stretch=True, imagenet_norm=True)
        assert np.abs(outr.cpu().numpy() - want).max() <= 1e-6
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


@pytest.mark.gpu
def test_record_emission_vs_oracle_chain():
    """hoigen_amd.records.emit_record (the producer of the entries upt...:636-688 reads): union / object / human crops of
    every pair -> Pillow-exact pre-processing -> encode_image, against the oracle chain on the same boxes."""
    from hoigen_amd import records, synth
    from hoigen_amd.model import build_model
    from oracle import clip_oracle as co
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(11)
    img = rng.randint(0, 256, size=(240, 320, 3)).astype(np.uint8)
    bh = np.array([[20.4, 30.6, 120.5, 200.2], [150, 10, 300, 230]], np.float32)
    bo = np.array([[100.0, 90.0, 210.7, 160.1], [140, 120, 200, 180]], np.float32)
    sd_np = synth.clip_state_dict(synth.VIT_B16, 0)
    m = build_model(synth.to_torch(sd_np)).float().to(dev)
    rec = records.emit_record(m, torch.from_numpy(img).to(dev), bh, bo, verbs=[3, 57], objects=[1, 40])
    assert sorted(rec) == ["boxes_h", "boxes_o", "huamn_features", "object_features", "objects", "union_features", "verbs"]
    assert rec["union_features"].shape == (2, 512) and rec["verbs"].tolist() == [3, 57]
    ub = records.union_boxes(bh, bo)
    assert np.allclose(ub, [[20.4, 30.6, 210.7, 200.2], [140, 10, 300, 230]])
    sd = co.reference_weight_rounding(sd_np)
    for key, boxes in (("union_features", ub), ("object_features", bo), ("huamn_features", bh)):
        _, x = po.preprocess_boxes(img, records.pil_box(boxes))
        want = co.encode_image(sd, torch.from_numpy(x)).numpy()
        err = np.linalg.norm(rec[key] - want, axis=1) / np.linalg.norm(want, axis=1)
        assert err.max() <= 1e-3, (key, err)
    empty = records.emit_record(m, torch.from_numpy(img).to(dev), np.zeros((0, 4)), np.zeros((0, 4)), [], [])
    assert empty["union_features"].shape == (0, 512)
