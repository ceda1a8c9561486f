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


# ---- the epilogues the vision tower actually runs: LayerNorm folded into the consumer GEMM (8, 9), residual update that
# ---- emits the centred fp16 copy and the row statistics (10), and the scaled variant behind the adapter (12)
def run_ln(ctx, a, w, bias, epi, kernel, *, cs=None, mr=None, mu=None, scale=None, x0=None):
    M, K = a.shape
    N = w.shape[0]
    p = lambda t: t.data_ptr() if t is not None else None
    if epi in (8, 9):
        out = torch.empty(M, N, device="cuda")
        rc = _lib.lib().hg_test_gemm_ln(ctx, p(a), p(w), p(bias), p(out), M, N, K, epi, kernel, p(cs), p(mr), None, None,
                                        None, None, None, None)
        assert rc == 0, _lib.lib().hg_last_error(ctx)
        torch.cuda.synchronize()
        return out
    out = x0.clone()
    out2 = torch.empty(M, N, device="cuda")
    mr_out = torch.empty(M, 2, device="cuda")
    mu_out = torch.empty(M, device="cuda")
    rc = _lib.lib().hg_test_gemm_ln(ctx, p(a), p(w), p(bias), p(out), M, N, K, epi, kernel, None, None, p(mu), p(scale),
                                    p(out2), p(mr_out), p(mu_out), None)
    assert rc == 0, _lib.lib().hg_last_error(ctx)
    torch.cuda.synchronize()
    return out, out2, mr_out, mu_out


def _operands(M, N, K, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    return g, a, w, bias


# ragged M, K-tile counts 4 / 5 / 6 / 12 / 48; (8192, 2048, .) fills exactly one round of 256x256 tiles (two-phase loop),
# the others run the 128-row variants
LN_SHAPES = [(197 * 12 + 5, 768, 768), (256 * 33 + 100, 768, 256), (2048 + 37, 768, 320), (1024 + 3, 2304, 384),
             (4096 + 77, 768, 3072), (8192, 2048, 256), (8192 - 60, 2048, 768), (197 * 12 + 5, 2304, 832)]


@pytest.mark.parametrize("M,N,K", LN_SHAPES)
@pytest.mark.parametrize("epi", [8, 9])
def test_layernorm_folded_epilogues(ctx, M, N, K, epi):
    """rstd * (x16 W'^T - mean * cs) + b'  against the same expression in fp32 PyTorch."""
    g, a, w, bias = _operands(M, N, K, M + 3 * N + K + epi)
    mean = torch.randn(M, device="cuda", generator=g) * 0.3
    rstd = torch.rand(M, device="cuda", generator=g) * 1.5 + 0.5
    mr = torch.stack([mean, rstd], 1).contiguous()
    cs = w.half().float().sum(1)
    acc = a.half().float() @ w.half().float().t()
    want = rstd[:, None] * (acc - mean[:, None] * cs[None]) + bias
    if epi == 9:
        want = want * torch.sigmoid(1.702 * want)
    want = want.half().float()
    tol = 3e-3 * want.abs().max().item()
    first = None
    for it in range(3):                                        # repeated launches: a race shows as a flaky mismatch
        got = run_ln(ctx, a, w, bias, epi, 0, cs=cs, mr=mr)
        assert torch.isfinite(got).all()
        assert (got - want).abs().max().item() <= tol, f"launch {it}"
        first = got if first is None else first
        assert torch.equal(got, first), "same inputs, different bits"


# more residual shapes: K-tile counts 12, 13, 48 on full 128-row tiles; fewer tiles than CUs, several tiles per workgroup
RESID_SHAPES = LN_SHAPES[:5] + [(128 * 40, 768, 768), (128 * 37, 768, 832), (128 * 41, 768, 3072), (128 * 300, 768, 768),
                                (128 * 394, 768, 768), (128 * 24, 2304, 1024)]


@pytest.mark.parametrize("M,N,K", RESID_SHAPES)
def test_residual_epilogue_emits_copy_and_statistics(ctx, M, N, K):
    """EPI_RESID_LN_F32: x += acc + b; x16 = fp16(x - mu); (sum, M2) groups -> finalize_stats -> (mean - mu, rstd), mu = mean.
    ring2 (128x256, the product path) and the duo kernel must agree bit for bit on the stream and its copy."""
    g, a, w, bias = _operands(M, N, K, 5 * M + N + K)
    x0 = torch.randn(M, N, device="cuda", generator=g) * 2 + torch.randn(M, 1, device="cuda", generator=g)   # rows with offsets
    mu = x0.mean(1) + 0.05 * torch.randn(M, device="cuda", generator=g)      # "previous mean": close to, not equal to, the new one
    xr = x0 + a.half().float() @ w.half().float().t() + bias
    mean = xr.mean(1)
    rstd = 1.0 / torch.sqrt(xr.var(1, unbiased=False) + 1e-5)
    want16 = (xr - mu[:, None]).half().float()
    scale = xr.abs().max().item()
    ref_bits = None
    for kernel in (2, 3):
        for it in range(3):
            x, x16, mr_out, mu_out = run_ln(ctx, a, w, bias, 10, kernel, mu=mu, x0=x0)
            assert (x - xr).abs().max().item() <= 2e-5 * scale, (kernel, it)
            assert (x16 - want16).abs().max().item() <= 2e-3 * (xr - mu[:, None]).abs().max().item(), (kernel, it)
            assert (mu_out - mean).abs().max().item() <= 1e-5 * scale
            assert ((mr_out[:, 0] - (mean - mu)).abs().max().item()) <= 1e-5 * scale
            assert ((mr_out[:, 1] - rstd).abs() / rstd).max().item() <= 1e-4
            bits = (x, x16, mr_out, mu_out)
            if it == 0:
                first = bits
            for u, v in zip(bits, first):
                assert torch.equal(u, v), f"kernel {kernel} launch {it}: same inputs, different bits"
            if ref_bits is None:
                ref_bits = bits
            # the two kernels accumulate in the same order: stream and fp16 copy agree bit for bit (their row-statistics
            # reductions group the lanes differently: equal within the tolerances above only)
            assert torch.equal(x, ref_bits[0]) and torch.equal(x16, ref_bits[1]), f"kernel {kernel}: bits differ from ring2's"


@pytest.mark.parametrize("M,N,K", [(197 * 12 + 5, 768, 64), (256 * 33 + 100, 768, 64), (2048 + 37, 768, 128), (1024 + 3, 768, 192),
                                   (4096 + 77, 768, 256)])
def test_scaled_residual_epilogue_duo(ctx, M, N, K):
    """EPI_SCALE_RESID_LN_F32 (adapter up_proj, K = 64 in production: one K-tile per tile, the path that skips the
    vmcnt wait after a residual epilogue on every tile after the first)."""
    g, a, w, bias = _operands(M, N, K, 7 * M + N + K)
    x0 = torch.randn(M, N, device="cuda", generator=g)
    sc = torch.randn(N, device="cuda", generator=g) * 0.5
    mu = x0.mean(1)
    xr = x0 + (a.half().float() @ w.half().float().t() + bias) * sc
    mean = xr.mean(1)
    rstd = 1.0 / torch.sqrt(xr.var(1, unbiased=False) + 1e-5)
    scale = xr.abs().max().item()
    first = None
    for it in range(3):
        x, x16, mr_out, mu_out = run_ln(ctx, a, w, bias, 12, 0, mu=mu, scale=sc, x0=x0)
        assert (x - xr).abs().max().item() <= 2e-5 * scale
        assert (x16 - (xr - mu[:, None]).half().float()).abs().max().item() <= 2e-3 * (xr - mu[:, None]).abs().max().item()
        assert (mu_out - mean).abs().max().item() <= 1e-5 * scale
        assert ((mr_out[:, 1] - rstd).abs() / rstd).max().item() <= 1e-4
        first = (x, x16, mr_out) if first is None else first
        assert all(torch.equal(u, v) for u, v in zip((x, x16, mr_out), first))


@pytest.mark.parametrize("M,N,K", [(1024, 768, 64), (777, 512, 128), (2048 + 9, 256, 192)])
@pytest.mark.parametrize("epi", [0, 3, 4])
def test_duo_short_k(ctx, M, N, K, epi):
    """The duo kernel admits K >= 64: K-tile counts 1, 2, 3 against the simple kernel, bit for bit."""
    g, a, w, bias = _operands(M, N, K, M + N + K + epi)
    out0 = torch.randn(M, N, device="cuda", generator=g) if epi == 3 else None
    want = ref(a, w, bias, epi, out0)
    got1 = run(ctx, a, w, bias, epi, 1, out0)
    assert (got1 - want).abs().max().item() <= (3e-3 if epi == 0 else 2e-5) * want.abs().max().item()
    for _ in range(3):
        assert torch.equal(got1, run(ctx, a, w, bias, epi, 3, out0))


@pytest.mark.parametrize("M,N,K", [(197 * 12 + 5, 768, 768), (128 * 41, 768, 3072), (256 * 33 + 100, 768, 256), (128 * 394, 768, 768)])
def test_residual_stream_as_centre_hi_lo(ctx, M, N, K):
    """GemmArgs::hl: five residual updates in a row with the stream held as centre + hi + lo between them (the first
    reads fp32, the last writes fp32) against the same five updates on an fp32 stream: the final rows agree to what the halves
    carry (fp16 + fp16: 22 bits of x - centre; fp16 + bf8, the default build: 13-14 bits), the last copy and statistics agree,
    repeated runs are bit-identical; rows with a large common offset and 30x outlier columns included."""
    g, a, w, bias = _operands(M, N, K, 13 * M + N + K)
    x0 = torch.randn(M, N, device="cuda", generator=g) * 2 + 8 * torch.randn(M, 1, device="cuda", generator=g)
    x0[:, 5::97] *= 30.0
    mu0 = x0.mean(1) + 0.05 * torch.randn(M, device="cuda", generator=g)
    p = lambda t: t.data_ptr()

    def run(hilo):
        x, mu = x0.clone(), mu0.clone()
        out2, mr = torch.empty(M, N, device="cuda"), torch.empty(M, 2, device="cuda")
        rc = _lib.lib().hg_test_gemm_hilo(ctx, p(a), p(w), p(bias), p(x), M, N, K, 5, hilo, p(mu), p(out2), p(mr), None)
        assert rc == 0, _lib.lib().hg_last_error(ctx)
        torch.cuda.synchronize()
        return x, mu, out2, mr

    f32 = run(0)
    upd = a.half().float() @ w.half().float().t() + bias
    want = x0 + 5 * upd
    scale = want.abs().max().item()
    assert (f32[0] - want).abs().max().item() <= 1e-5 * scale
    hl = run(1)
    again = run(1)
    assert all(torch.equal(u, v) for u, v in zip(hl, again)), "same inputs, different bits"
    assert torch.isfinite(hl[0]).all()
    # against the fp32 stream: per element relative to the row's spread about its centre (what the halves are scaled to)
    spread = (want - want.mean(1, keepdim=True)).abs().amax(1, keepdim=True)
    err = ((hl[0] - f32[0]).abs() / spread).max().item()
    print(f"\\nhi/lo stream vs fp32 stream after 5 updates: max |diff| / row spread = {err:.2e}")
    bits = C.c_int32(0)
    assert _lib.lib().hg_get_option(ctx, b"stream_lo_bits", C.byref(bits)) == 0
    # lo as fp16: 22 bits of x - centre.  lo as bf8 (the default build, HG_LO8): the remainder (<= 2^-11 of the element) keeps two
    # mantissa bits, <= 2^-14 of the element per update, and the updates' errors add: 5 x 6.1e-5 of the row spread at most
    assert err <= (4e-6 if bits.value == 16 else 5 * 2.0 ** -14 * 1.02)
    assert (hl[1] - f32[1]).abs().max().item() <= 1e-5 * scale                      # mean
    assert ((hl[3][:, 1] - f32[3][:, 1]).abs() / f32[3][:, 1]).max().item() <= 1e-4   # rstd
    assert (hl[2] - f32[2]).abs().max().item() <= 2e-3 * spread.max().item()          # last centred copy (fp16 grid)
