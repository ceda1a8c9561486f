"""Parity test of the sequence-tile residual GEMM (hg_gemm_seq_rln.hip) as it ran inside tests/test_gpu_gemm.py (8 cases green on MI355X):
needs the hg_test_gemm_seq hook of gemm_seq_rln_hook.txt built into the library."""
# ---- the residual GEMMs of the vision tower on sequence tiles (hg_gemm_seq.hip): one (sequence, 384-column panel) item per tile
@pytest.mark.parametrize("n_seq,L,K", [(3, 197, 768), (41, 197, 768), (256, 197, 768), (19, 197, 3072), (256, 197, 3072),
                                       (7, 193, 768), (5, 208, 768), (300, 197, 384)])
def test_sequence_tile_residual_gemm_equals_ring2(ctx, n_seq, L, K):
    """hg_gemm_seq.hip against gemm_ring2's LayerNorm-emitting residual GEMM (hg_test_gemm_hilo) on the same operands, five chained
    updates: with the fp32 stream the rows are BIT-IDENTICAL (same MFMA, same k order, same x + (acc + bias)) and so is the fp16
    copy wherever the two centres agree; the statistics (16 column groups of 48 instead of 12 of 64) agree to fp32 rounding; with
    the stream as centre + hi + lo the final rows agree with the fp32 run to fp32 rounding.  Ragged item counts (fewer items than
    CUs, not a multiple of the grid, more than two per CU), the shortest / longest sequence of a row tile, K = 384 (six K-tiles),
    768 and 3072; repeated runs are bit-identical (a race in the counted waits would show as a flaky mismatch)."""
    N, M = 768, n_seq * L
    g, a, w, bias = _operands(M, N, K, 7 * M + K)
    x0 = torch.randn(M, N, device="cuda", generator=g) * 2 + 8 * torch.randn(M, 1, device="cuda", generator=g)
    x0[:, 5::97] *= 30.0
    mu0 = x0.mean(1) + 0.05 * torch.randn(M, device="cuda", generator=g)
    p = lambda t: t.data_ptr()

    def run(kernel, hilo, steps=5):
        x, mu = x0.clone(), mu0.clone()
        out2, mr = torch.empty(M, N, device="cuda"), torch.empty(M, 2, device="cuda")
        if kernel == "seq":
            rc = _lib.lib().hg_test_gemm_seq(ctx, p(a), p(w), p(bias), p(x), n_seq, L, N, K, steps, hilo, p(mu), p(out2), p(mr), None)
        else:
            rc = _lib.lib().hg_test_gemm_hilo(ctx, p(a), p(w), p(bias), p(x), M, N, K, steps, hilo, p(mu), p(out2), p(mr), None)
        assert rc == 0, _lib.lib().hg_last_error(ctx)
        torch.cuda.synchronize()
        return x, mu, out2, mr

    seq = run("seq", 0)
    for _ in range(2):
        assert all(torch.equal(u, v) for u, v in zip(seq, run("seq", 0))), "same inputs, different bits"
    upd = a.half().float() @ w.half().float().t() + bias
    want = x0 + 5 * upd
    scale = want.abs().max().item()
    assert (seq[0] - want).abs().max().item() <= 1e-5 * scale
    if K >= 256 and M >= 512:          # (gemm_ring2's own eligibility)
        ring = run("ring2", 0)
        assert torch.equal(seq[0], ring[0]), "fp32 stream"
        assert (seq[1] - ring[1]).abs().max().item() <= 1e-5 * scale, "centres"
        assert (seq[3] - ring[3]).abs().max().item() <= 2e-5 * max(1.0, ring[3].abs().max().item()), "statistics"
        one_s, one_r = run("seq", 0, 1), run("ring2", 0, 1)      # one update: both copies are centred on mu0
        assert torch.equal(one_s[2], one_r[2]), "fp16 copy"
    hl = run("seq", 1)
    assert all(torch.equal(u, v) for u, v in zip(hl, run("seq", 1))), "same inputs, different bits (hi / lo)"
    assert (hl[0] - seq[0]).abs().max().item() <= 4e-6 * scale
    # the copy is fp16(x - mean): against the fp32 run's copy one fp16 ulp of the centred value where a rounding boundary is crossed
    assert (hl[2] - seq[2]).abs().max().item() <= 2.0 ** -10 * seq[2].abs().max().item()
    assert (hl[3] - seq[3]).abs().max().item() <= 1e-4 * max(1.0, seq[3].abs().max().item())
