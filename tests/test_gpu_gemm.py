"""GEMM kernels (simple 128x128 and persistent ring) against a plain PyTorch fp32 matmul of the
fp16-rounded operands, every epilogue, ragged M, and ring == simple bit for bit (same k order)."""
import ctypes as C

import pytest
import torch

from hoigen_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    h = _lib.lib().hg_create(0)
    assert h
    yield h
    _lib.lib().hg_destroy(h)


def run(ctx, a, w, bias, epi, kernel, out0=None):
    M, K = a.shape
    N = w.shape[0]
    out = out0.clone() if out0 is not None else torch.empty(M, N, device="cuda")
    rc = _lib.lib().hg_test_gemm(ctx, a.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None,
                                 out.data_ptr(), M, N, K, epi, kernel, None)
    assert rc == 0, _lib.lib().hg_last_error(ctx)
    torch.cuda.synchronize()
    return out


def ref(a, w, bias, epi, out0):
    y = a.half().float() @ w.half().float().t()
    if bias is not None:
        y = y + bias
    if epi == 1:
        y = y * torch.sigmoid(1.702 * y)
    if epi in (2, 6):
        y = torch.relu(y)
    if epi in (0, 1, 2):
        y = y.half().float()
    if epi == 3:
        y = out0 + y
    return y


SHAPES = [(1024, 256, 256), (2048, 768, 768), (197 * 16, 2304, 768), (197 * 12 + 5, 768, 3072), (513, 512, 1024),
          (4096, 3072, 768),
          # K-tile counts 5 and 6 (the K loop is instantiated per K-tile position: first / middle / last three), a
          # 256-row-tile case with a ragged last tile, and one with more than 256 tiles of 256x256
          (777, 512, 320), (2048, 768, 384), (256 * 33 + 100, 2048, 256),
          # exactly / almost one full round of 256x256 tiles -> the two-phase 256-row kernel, K-tile counts 4 and 5
          (8192, 2048, 256), (8192 - 60, 2048, 320)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4, 6])
def test_gemm_kernels(ctx, M, N, K, epi):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N + K + epi)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    out0 = torch.randn(M, N, device="cuda", generator=g) if epi == 3 else None
    want = ref(a, w, bias, epi, out0)
    scale = want.abs().max().item()
    got1 = run(ctx, a, w, bias, epi, 1, out0)
    tol = (3e-3 if epi in (0, 1, 2) else 2e-5) * scale          # fp16 outputs: 1 ulp of fp16 at the top of the range
    assert (got1 - want).abs().max().item() <= tol, "simple kernel"
    got2 = run(ctx, a, w, bias, epi, 2, out0)
    assert (got2 - want).abs().max().item() <= tol, "ring kernel"
    assert torch.equal(got1, got2), "ring and simple kernels accumulate in the same order"
    if N % 256 == 0 and K >= 256 and M >= 512:      # two-workgroups-per-CU kernel (hg_gemm_duo.hip): same bits
        assert torch.equal(got1, run(ctx, a, w, bias, epi, 3, out0)), "duo and simple kernels accumulate in the same order"
        assert torch.equal(run(ctx, a, w, None, epi, 1, out0), run(ctx, a, w, None, epi, 3, out0)), "duo without bias"
    got_nb = run(ctx, a, w, None, epi, 2, out0)
    want_nb = ref(a, w, None, epi, out0)
    assert (got_nb - want_nb).abs().max().item() <= tol, "ring kernel without bias"


def test_ring_many_tiles_per_workgroup(ctx):
    """More tiles than CUs (persistent stream across tile boundaries, epilogue/vmcnt bookkeeping)."""
    M, N, K = 197 * 256, 768, 768
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    bias = torch.randn(N, device="cuda", generator=g)
    for epi in (0, 3, 4):
        out0 = torch.randn(M, N, device="cuda", generator=g) if epi == 3 else None
        g1 = run(ctx, a, w, bias, epi, 1, out0)
        for _ in range(3):                      # repeated launches: races would show as flaky mismatches
            g2 = run(ctx, a, w, bias, epi, 2, out0)
            assert torch.equal(g1, g2)
            assert torch.equal(g1, run(ctx, a, w, bias, epi, 3, out0)), "duo kernel"
    want = ref(a, w, bias, 4, None)
    assert (g1 - want).abs().max().item() <= 2e-5 * want.abs().max().item()
