"""RoI-align over the local feature map.  torchvision is not installed (here or on the GPU boxes: the two
*_torchvision_fixture tests below run once tests/golden/make_golden_roi.py could be run somewhere), so the oracle is anchored on
analytic properties and on hand-derived known-answer vectors (tests/golden/roi_known_answers.py) that separate the
aligned=True half-pixel offset, the adaptive ceil(roi/P) grid and the boundary rules from their alternatives - not on
outputs of the real library (see oracle/roi_oracle.py); the HIP kernel is compared with both."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import roi_oracle as ro  # noqa: E402

SCALE = 14.0 / 224.0
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from roi_known_answers import cases as roi_cases  # noqa: E402


def test_oracle_known_answers():
    for feat, box, scale, P, want, what in roi_cases():
        got = ro.roi_align(feat, np.array([box], np.float32), P, scale)[0]
        assert np.abs(got - want).max() <= 1e-6, (what, got, want)
    # the alternatives the vectors are meant to exclude really differ
    feat, box, scale, P, want, _ = roi_cases()[0]
    assert np.abs(ro.roi_align(feat, np.array([box], np.float32), P, scale, aligned=False)[0] - want).max() >= 2.0



def affine_maps(H=14, W=14):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    return np.stack([np.ones((H, W), np.float32), yy, xx, 2 * yy - 3 * xx + 1]).astype(np.float32)


def test_oracle_analytic_properties():
    """Bilinear sampling is exact on affine maps and the sample grid is symmetric about the RoI centre, so the mean
    over the 7x7 bins equals the map evaluated at the RoI centre (feature coordinates c * scale - 0.5) for RoIs
    whose samples stay inside the map; a constant map gives the constant for any RoI that stays inside."""
    feat = affine_maps()
    boxes = np.array([[32, 48, 160, 200], [16, 16, 208, 208], [40, 40, 56, 72], [100, 20, 130, 200]], np.float32)
    m = ro.roi_align_mean(feat, boxes, 7, SCALE)
    cx = (boxes[:, 0] + boxes[:, 2]) / 2 * SCALE - 0.5
    cy = (boxes[:, 1] + boxes[:, 3]) / 2 * SCALE - 0.5
    want = np.stack([np.ones_like(cx), cy, cx, 2 * cy - 3 * cx + 1], 1)
    assert np.abs(m - want).max() <= 2e-5
    # per-bin: every bin of an interior RoI on an affine map equals the map at the bin centre
    p = ro.roi_align(feat, boxes[:1], 7, SCALE)[0]
    b = boxes[0]
    bx = (b[0] * SCALE - 0.5) + (np.arange(7) + 0.5) * (b[2] - b[0]) * SCALE / 7
    by = (b[1] * SCALE - 0.5) + (np.arange(7) + 0.5) * (b[3] - b[1]) * SCALE / 7
    assert np.abs(p[1] - by[:, None]).max() <= 2e-5 and np.abs(p[2] - bx[None, :]).max() <= 2e-5
    # outside the map everything is zero; shape contract
    assert ro.roi_align(feat, np.array([[400, 400, 500, 500]], np.float32), 7, SCALE).max() == 0.0
    assert ro.roi_align(feat, np.zeros((0, 4), np.float32), 7, SCALE).shape == (0, 4, 7, 7)


G11 = os.path.join(ROOT, "tests", "golden", "g11_roi.npz")
NO_G11 = ("tests/golden/g11_roi.npz absent: torchvision is importable neither in the build container nor on the MI355X pool "
          "boxes (tests/golden/README.md records the ImportError of tests/golden/make_golden_roi.py)")


def g11_cases():
    d = np.load(G11)
    names = sorted({k[:-len("_pooled")] for k in d.files if k.endswith("_pooled")})
    return [(n, d[f"{n}_feat"], d[f"{n}_boxes"], int(d[f"{n}_meta"][0]), float(d[f"{n}_meta"][1]), bool(d[f"{n}_meta"][2]),
             d[f"{n}_pooled"], d[f"{n}_mean"]) for n in names]


@pytest.mark.skipif(not os.path.exists(G11), reason=NO_G11)
def test_oracle_vs_torchvision_fixture():
    """The oracle against outputs of the real torchvision.ops.roi_align (make_golden_roi.py): pins SURVEY.md 8f-4."""
    for name, feat, boxes, P, scale, aligned, pooled, mean in g11_cases():
        got = ro.roi_align(feat, boxes, P, scale, aligned=aligned)
        tol = 1e-5 * max(1.0, float(np.abs(pooled).max()))
        assert np.abs(got - pooled).max() <= tol, name
        assert np.abs(got.reshape(got.shape[0], got.shape[1], -1).mean(-1, dtype=np.float32) - mean).max() <= tol, name


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(G11), reason=NO_G11)
def test_hip_vs_torchvision_fixture():
    from hoigen_amd.roi import roi_align
    dev = torch.device("cuda:0")
    for name, feat, boxes, P, scale, aligned, pooled, mean in g11_cases():
        if not aligned:
            continue      # the kernel implements the reference's call (aligned=True) only
        f, b = torch.from_numpy(feat).to(dev), torch.from_numpy(boxes).to(dev)
        tol = 1e-5 * max(1.0, float(np.abs(pooled).max()))
        assert np.abs(roi_align(f, b, P, scale).cpu().numpy() - pooled).max() <= tol, name
        assert np.abs(roi_align(f, b, P, scale, reduce_mean=True).cpu().numpy() - mean).max() <= tol, name


def test_facade_errors_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hoigen_amd.roi import roi_align
    with pytest.raises(RuntimeError, match="HIP device"):
        roi_align(torch.zeros(4, 14, 14), torch.zeros(1, 4))


@pytest.mark.gpu
def test_hip_roi_align_vs_oracle():
    from hoigen_amd.roi import roi_align
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(5)
    feat = rng.randn(512, 14, 14).astype(np.float32)
    boxes = []
    for _ in range(40):
        x1, x2 = sorted(rng.uniform(-30, 260, size=2)); y1, y2 = sorted(rng.uniform(-30, 260, size=2))
        boxes.append((x1, y1, x2 + 1, y2 + 1))
    boxes += [(0, 0, 224, 224), (10, 10, 10.5, 10.5), (300, 300, 400, 400), (-50, -50, 5, 5)]
    boxes = np.asarray(boxes, np.float32)
    want = ro.roi_align(feat, boxes, 7, SCALE)
    f, b = torch.from_numpy(feat).to(dev), torch.from_numpy(boxes).to(dev)
    got = roi_align(f.unsqueeze(0), b, (7, 7), SCALE, aligned=True)
    assert got.shape == (44, 512, 7, 7) and got.dtype == torch.float32
    assert np.abs(got.cpu().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    mean = roi_align(f, b, 7, SCALE, reduce_mean=True)
    want_mean = want.reshape(44, 512, -1).mean(-1, dtype=np.float32)
    assert np.abs(mean.cpu().numpy() - want_mean).max() <= 1e-5
    assert roi_align(f, torch.zeros(0, 4, device=dev), 7, SCALE).shape == (0, 512, 7, 7)
    # hand-derived known answers on the device (half-pixel offset, adaptive grid, boundary rules)
    for feat_k, box, scale, P, want_k, what in roi_cases():
        got_k = roi_align(torch.from_numpy(feat_k).to(dev), torch.tensor([box], dtype=torch.float32, device=dev), P, scale)
        assert np.abs(got_k.cpu().numpy()[0] - want_k).max() <= 1e-6, what
    # analytic anchor on the device as well
    aff = torch.from_numpy(affine_maps()).to(dev)
    m = roi_align(aff, torch.tensor([[32., 48., 160., 200.]], device=dev), 7, SCALE, reduce_mean=True).cpu().numpy()[0]
    assert np.abs(m - np.array([1.0, 7.25, 5.5, -1.0], np.float32)).max() <= 2e-5


@pytest.mark.gpu
def test_hip_roi_align_on_variant_c_local_map():
    """End of the chain: visual(x, prior) -> local [B,512,14,14] -> roi_align -> [n,512] (upt…:1615,1026-1037)."""
    from hoigen_amd import synth
    from hoigen_amd.model import build_model
    from hoigen_amd.roi import roi_align
    dev = torch.device("cuda:0")
    sd = synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))
    sd.update(synth.to_torch(synth.adapter_state_dict(synth.VIT_B16, 1)))
    m = build_model(sd, use_adapter=True).to(dev)
    img = torch.from_numpy(synth.crops(1, 224, seed=1234)).to(dev)
    _, local = m.visual(img, None)
    boxes = torch.tensor([[10., 20., 120., 200.], [0., 0., 224., 224.]], device=dev)
    out = roi_align(local[0], boxes, 7, SCALE, reduce_mean=True)
    want = ro.roi_align_mean(local[0].cpu().numpy(), boxes.cpu().numpy(), 7, SCALE)
    assert out.shape == (2, 512) and np.abs(out.cpu().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
