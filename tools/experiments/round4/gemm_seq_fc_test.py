"""Parity test of the sequence-tile c_fc / in_proj GEMM (hg_gemm_seq_fc.hip) as it ran inside tests/test_gpu_gemm.py (8 cases green on
MI355X, bit-identical to the 256 x 256 ring): needs the `kernel >= 1000` branch of hg_test_gemm_ln (gemm_seq_fc_api.patch)."""
# ---- c_fc + QuickGELU (and the plain folded projection) on sequence tiles (hg_gemm_seq.hip): one (sequence, 384-column panel) item per tile
@pytest.mark.parametrize("n_seq,L,N,K,epi", [(3, 197, 3072, 768, 9), (41, 197, 3072, 768, 9), (256, 197, 3072, 768, 9), (300, 197, 3072, 768, 9),
                                             (7, 193, 3072, 768, 9), (5, 208, 3072, 768, 9), (33, 197, 2304, 768, 8), (9, 200, 1536, 384, 9)])
def test_sequence_tile_folded_gemm_equals_ring(ctx, n_seq, L, N, K, epi):
    """hg_gemm_seq.hip against the 256 x 256 ring's LayerNorm-folded fp16 epilogues (kinds 8, 9) on the same operands: BIT-IDENTICAL
    (same MFMA, same k order, same epilogue expressions) over ragged item counts (fewer items than CUs, not a multiple of the grid,
    more than eight per CU), the shortest / longest sequence of a row tile, N = 2304 (six panels) and K = 384 (six K-tiles); repeated
    launches agree (a race in the counted waits would show as a flaky mismatch); against the fp32 expression; every XCD group size."""
    M = n_seq * L
    g, a, w, bias = _operands(M, N, K, 11 * M + N + K + epi)
    cs = w.half().float().sum(1)
    mr = torch.stack([torch.randn(M, device="cuda", generator=g) * 0.05, torch.rand(M, device="cuda", generator=g) + 0.5], 1).contiguous()
    got = run_ln(ctx, a, w, bias, epi, 1000 + L, cs=cs, mr=mr)
    for _ in range(3):
        assert torch.equal(got, run_ln(ctx, a, w, bias, epi, 1000 + L, cs=cs, mr=mr))
    if N % 256 == 0 and K >= 256 and M >= 512:
        assert torch.equal(got, run_ln(ctx, a, w, bias, epi, 2, cs=cs, mr=mr)), "ring kernel"
    v = (a.half().float() @ w.half().float().t() - mr[:, :1] * cs[None]) * mr[:, 1:] + bias[None]
    want = v * torch.sigmoid(1.702 * v) if epi == 9 else v
    assert (got - want).abs().max().item() <= 2e-3 * want.abs().max().item()
    for gsz in (1, 2, 4, 0):
        if (N // 384) % max(gsz, 1) == 0:
            assert _lib.lib().hg_set_option(ctx, b"seq_fc_gsz", gsz) == 0
            assert torch.equal(got, run_ln(ctx, a, w, bias, epi, 1000 + L, cs=cs, mr=mr)), gsz
    assert _lib.lib().hg_set_option(ctx, b"seq_fc_gsz", 0) == 0
