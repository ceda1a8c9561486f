"""CoOp-VAE Encoder -> reparameterise -> Generator as ONE kernel (hoigen_amd/csrc/hg_vae_fused.hip; SURVEY.md 2.2 K10, north_star's
"fused VAE encoder / decoder / reparameterise kernel"; reference: main_coop_vae.py:261-296,444-448) through the C ABI
(hg_vae_forward / hg_generator with option vae_fused = 2: every row on the one-kernel path).

Evidence: the reference's own outputs (tests/golden/g4_vae.npz), the CPU oracle on ragged row counts (1 row ... several work items,
tails that leave waves and lanes empty), the GEMM path of the same library (same fp16 operand roundings, different fp32 summation
order), row independence (a row's bits do not depend on its neighbours or its position), repeated launches.  Tolerance vs reference /
oracle: relative L2 <= 1e-3 per tensor and per row (north_star)."""
import os

import numpy as np
import pytest
import torch

from hoigen_amd import synth, vae

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-3
NAMES = ("mean", "log_var", "z", "bias")


def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def rel_l2(a, b):
    a = a.detach().float().cpu().numpy().astype(np.float64)
    b = b.detach().float().cpu().numpy().astype(np.float64) if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    assert a.shape == b.shape and np.isfinite(a).all()
    whole = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
    rows = np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-30)
    return whole, rows.max()


def check(a, b, tol=TOL, what=""):
    whole, worst = rel_l2(a, b)
    assert whole <= tol and worst <= tol, f"{what}: rel-L2 {whole:.3e}, worst row {worst:.3e} > {tol}"
    return whole


@pytest.fixture()
def nets():
    d = dev()
    E, Gn = vae.Encoder().to(d), vae.Generator().to(d)
    E.load_state_dict(synth.to_torch(synth.encoder_state_dict(2)))
    Gn.load_state_dict(synth.to_torch(synth.generator_state_dict(3)))
    yield E, Gn
    vae.set_option("vae_fused", 1, d)


def test_one_kernel_vae_vs_reference_fixture(nets):
    """The reference's own Encoder / Generator outputs on seeded weights and inputs (tests/golden/make_golden.py -> g4)."""
    E, Gn = nets
    d = dev()
    g = dict(np.load(f"{G}/g4_vae.npz"))
    vae.set_option("vae_fused", 2, d)
    x = vae.l2_normalize(torch.from_numpy(synth.hg_normal((160, 512), 30)).to(d))
    eps = torch.from_numpy(synth.hg_normal((160, 512), 31)).to(d)
    got = vae.VAE(E, Gn)(x, eps)
    for a, n in zip(got, NAMES):
        e = check(a, g[n], what=f"{n} (one kernel) vs reference")
        print(f"\none-kernel VAE {n}: rel-L2 vs reference {e:.3e}")
    check(Gn(torch.from_numpy(synth.hg_normal((64, 512), 33)).to(d)), g["gen_from_z"], what="Generator(z), one kernel, vs reference")
    mean, lv = E(x)
    check(mean, g["mean"], what="Encoder mean (encoder-only call)")
    check(lv, g["log_var"], what="Encoder log_var (encoder-only call)")


@pytest.mark.parametrize("R", [1, 31, 33, 129, 1000, 4097])
def test_one_kernel_vae_vs_oracle_ragged_rows(nets, R):
    from oracle import clip_oracle as co, vae_oracle as vo
    E, Gn = nets
    d = dev()
    se, sg = synth.encoder_state_dict(2), synth.generator_state_dict(3)
    x = co.l2_normalize(torch.from_numpy(synth.hg_normal((R, 512), 150 + R)))
    eps = torch.from_numpy(synth.hg_normal((R, 512), 160 + R))
    ref = vo.vae_forward(co.as_tensors(se), co.as_tensors(sg), x, eps)
    vae.set_option("vae_fused", 2, d)
    got = vae.VAE(E, Gn)(x.to(d), eps.to(d))
    vae.set_option("vae_fused", 0, d)
    gemm = vae.VAE(E, Gn)(x.to(d), eps.to(d))
    for a, b, c, n in zip(got, ref, gemm, NAMES):
        check(a, b, what=f"{n} R={R} one kernel vs oracle")
        check(a, c, tol=4e-4, what=f"{n} R={R} one kernel vs GEMM path")      # (two realisations of the same fp16 operand roundings)
        assert not torch.equal(a, c), "the option did not change the executed path"


def test_one_kernel_vae_nonzero_biases_and_large_activations(nets):
    """The reference initialises every bias to 0 (main_coop_vae.py:32-39) and the synthetic fixtures inherit that with small weights:
    a trained checkpoint has neither.  Non-zero biases in all five Linear layers and inputs well outside the unit sphere, against the
    oracle: exercises the bias table of the first layers, the epilogue biases and exp(0.5 log_var) away from 1."""
    from oracle import clip_oracle as co, vae_oracle as vo
    d = dev()
    se, sg = synth.encoder_state_dict(2), synth.generator_state_dict(3)
    rng = np.random.default_rng(7)
    for sd in (se, sg):
        for k in sd:
            if k.endswith("bias"):
                sd[k] = (rng.standard_normal(sd[k].shape) * 0.3).astype(np.float32)
            else:
                sd[k] = (sd[k] * 2.5).astype(np.float32)
    E, Gn = vae.Encoder().to(d), vae.Generator().to(d)
    E.load_state_dict(synth.to_torch(se)); Gn.load_state_dict(synth.to_torch(sg))
    R = 777
    x = torch.from_numpy(synth.hg_normal((R, 512), 5)) * 0.7
    eps = torch.from_numpy(synth.hg_normal((R, 512), 6))
    ref = vo.vae_forward(co.as_tensors(se), co.as_tensors(sg), x, eps)
    vae.set_option("vae_fused", 2, d)
    got = vae.VAE(E, Gn)(x.to(d), eps.to(d))
    for a, b, n in zip(got, ref, NAMES):
        e = check(a, b, what=f"{n} with biases, one kernel vs oracle")
        print(f"\none-kernel VAE with non-zero biases, {n}: {e:.3e}")
    z = torch.from_numpy(synth.hg_normal((300, 512), 8)) * 1.3
    check(Gn(z.to(d)), vo.generator(co.as_tensors(sg), z), what="Generator with biases, one kernel vs oracle")


def test_one_kernel_vae_rows_are_independent_and_launches_repeat(nets):
    """A row's bits depend on nothing but the row: sub-ranges (not aligned to the 128-row items or the 32-row waves) equal the
    whole call, a non-finite row stays alone, three launches agree (a race in the ring's counted waits would show as a flaky
    mismatch)."""
    E, Gn = nets
    d = dev()
    vae.set_option("vae_fused", 2, d)
    gen = torch.Generator(device=d).manual_seed(9)
    R = 70_000          # 547 items: two rounds and a partly filled third on 256 CUs
    x = vae.l2_normalize(torch.randn(R, 512, device=d, generator=gen))
    eps = torch.randn(R, 512, device=d, generator=gen)
    V = vae.VAE(E, Gn)
    whole = V(x, eps)
    for _ in range(2):
        again = V(x, eps)
        for a, b, n in zip(again, whole, NAMES):
            assert torch.equal(a, b), f"{n}: repeated launch differs"
    for lo, hi in ((0, 1), (5, 6), (127, 129), (1000, 1777), (33333, 66001), (R - 1, R)):
        part = V(x[lo:hi], eps[lo:hi])
        for a, b, n in zip(part, whole, NAMES):
            assert torch.equal(a, b[lo:hi]), f"{n} rows [{lo},{hi}) depend on their position"
    bad = x.clone()
    bad[12345] = float("nan")
    out = V(bad, eps)
    keep = torch.ones(R, dtype=torch.bool, device=d)
    keep[12345] = False
    # (what the NaN row itself yields is not asserted: relu is v_max_f32, which returns 0 for NaN where torch.relu returns NaN -
    # on this path and on the GEMM path's epilogue alike)
    for a, b, n in zip(out, whole, NAMES):
        assert torch.equal(a[keep], b[keep]), f"{n}: a NaN row leaked into its neighbours"
    zz = torch.randn(R, 512, device=d, generator=gen)
    gw = Gn(zz)
    assert torch.equal(Gn(zz[40000:55001]), gw[40000:55001])


def test_vae_fused_option_is_validated(nets):
    d = dev()
    with pytest.raises(RuntimeError):
        vae.set_option("vae_fused", 3, d)
    vae.set_option("vae_fused", 1, d)
