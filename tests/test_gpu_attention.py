"""Attention kernels (hg_attn.hip) against a plain PyTorch fp32 softmax(QK^T/8)V of the fp16-rounded inputs:
full attention (ViT L = 197, text L = 77 causal, short / odd / maximum lengths) and the one-row-per-sequence variant
used by the last block (class token / EOT token), which must reproduce the full kernel's row bit for bit."""
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


def ref_attention(qkv, n_seq, L, heads, causal):
    D = heads * 64
    x = qkv.half().float().view(n_seq, L, 3, heads, 64)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))          # [n, h, L, 64]
    s = q @ k.transpose(-1, -2) * 0.125
    if causal:
        s = s + torch.full((L, L), float("-inf"), device=s.device).triu(1)
    return (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(n_seq * L, D)


def run_full(ctx, qkv, n_seq, L, heads, causal):
    out = torch.empty(n_seq * L, heads * 64, device="cuda")
    rc = _lib.lib().hg_test_attention(ctx, qkv.data_ptr(), None, None, n_seq, L, heads, int(causal), out.data_ptr(), None)
    assert rc == 0, _lib.lib().hg_last_error(ctx)
    torch.cuda.synchronize()
    return out


def run_rows(ctx, qkv, q0, sel, n_seq, L, heads, causal):
    out = torch.empty(n_seq, heads * 64, device="cuda")
    rc = _lib.lib().hg_test_attention(ctx, qkv.data_ptr(), q0.data_ptr(), sel.data_ptr() if sel is not None else None,
                                      n_seq, L, heads, int(causal), out.data_ptr(), None)
    assert rc == 0, _lib.lib().hg_last_error(ctx)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("n_seq,L,heads,causal", [(9, 197, 12, False), (7, 77, 8, True), (5, 13, 8, True),
                                                   (3, 1, 2, False), (4, 33, 4, True), (2, 224, 3, False),
                                                   (3, 224, 2, True), (6, 50, 12, False), (4, 16, 4, True)])
def test_attention_vs_fp32_reference(ctx, n_seq, L, heads, causal):
    g = torch.Generator(device="cuda").manual_seed(L * 131 + heads)
    qkv = torch.randn(n_seq * L, 3 * heads * 64, device="cuda", generator=g) * 1.5
    want = ref_attention(qkv, n_seq, L, heads, causal)
    got = run_full(ctx, qkv, n_seq, L, heads, causal)
    # fp16 probabilities and fp16 output: 2 ulp of fp16 at the top of the range
    assert (got - want).abs().max().item() <= 2e-3 * want.abs().max().item()
    assert torch.equal(got, run_full(ctx, qkv, n_seq, L, heads, causal)), "deterministic"


@pytest.mark.parametrize("n_seq,L,heads,causal", [(5, 13, 8, True), (301, 14, 8, True), (7, 16, 3, False), (9, 17, 8, True),
                                                   (33, 32, 2, True), (3, 1, 2, False)])
def test_short_sequences_packed_four_to_a_workgroup_are_bit_identical(ctx, n_seq, L, heads, causal):
    """L <= 32 is one query tile = one wave per (sequence, head): the launcher runs four such items per workgroup (the generation
    pipeline's 14-token prompts were bound by the workgroup dispatch rate).  Same instruction sequence per wave: bit-identical to the
    one-workgroup-per-item launch (test hook: causal bit 1), ragged item counts (not a multiple of four) included; and against fp32."""
    g = torch.Generator(device="cuda").manual_seed(L * 7 + heads + n_seq)
    qkv = torch.randn(n_seq * L, 3 * heads * 64, device="cuda", generator=g) * 1.5
    packed = run_full(ctx, qkv, n_seq, L, heads, int(causal))
    single = run_full(ctx, qkv, n_seq, L, heads, int(causal) | 2)
    assert torch.equal(packed, single)
    want = ref_attention(qkv, n_seq, L, heads, causal)
    assert (packed - want).abs().max().item() <= 2e-3 * want.abs().max().item()


@pytest.mark.parametrize("n_seq,L,heads,causal", [(9, 197, 12, False), (7, 77, 8, True), (5, 13, 8, True),
                                                   (3, 224, 2, True), (4, 33, 4, False)])
def test_one_row_variant_is_the_full_kernels_row(ctx, n_seq, L, heads, causal):
    D = heads * 64
    g = torch.Generator(device="cuda").manual_seed(L * 17 + heads)
    qkv = torch.randn(n_seq * L, 3 * D, device="cuda", generator=g)
    full = run_full(ctx, qkv, n_seq, L, heads, causal).view(n_seq, L, D)
    for sel in (None, torch.randint(0, L, (n_seq,), device="cuda", generator=g, dtype=torch.int32),
                torch.full((n_seq,), L - 1, device="cuda", dtype=torch.int32)):
        idx = sel.long() if sel is not None else torch.zeros(n_seq, dtype=torch.long, device="cuda")
        q0 = qkv.view(n_seq, L, 3 * D)[torch.arange(n_seq, device="cuda"), idx, :D].contiguous()
        rows = run_rows(ctx, qkv, q0, sel, n_seq, L, heads, causal)
        assert torch.equal(rows, full[torch.arange(n_seq, device="cuda"), idx]), "same instruction sequence, same bits"


# ---- in_proj + attention as ONE kernel (hg_qkv_attn.hip): q, k, v never leave the chip -------------------------------------
def _qkv_attn_operands(n_seq, L, heads, seed, extra=0):
    D = heads * 64
    K = D + extra                                                             # (extra = 64: [x16 | e] of a block with a folded adapter)
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(n_seq * L, K, device="cuda", generator=g)                 # centred fp16 copy of the stream
    w = torch.randn(3 * D, K, device="cuda", generator=g) * D ** -0.5         # LayerNorm-folded in_proj weight
    bias = torch.randn(3 * D, device="cuda", generator=g) * 0.3
    cs = w.half().float().sum(1)
    mr = torch.stack([torch.randn(n_seq * L, device="cuda", generator=g) * 0.05,
                      torch.rand(n_seq * L, device="cuda", generator=g) + 0.5], 1).contiguous()
    return a, w, bias, cs, mr


def run_qkv_attn(ctx, ops, n_seq, L, heads, fused):
    a, w, bias, cs, mr = ops
    out = torch.empty(n_seq * L, heads * 64, device="cuda")
    rc = _lib.lib().hg_test_qkv_attn(ctx, a.data_ptr(), w.data_ptr(), bias.data_ptr(), cs.data_ptr(), mr.data_ptr(), n_seq, L,
                                     heads, int(fused), out.data_ptr(), None)
    assert rc == 0, _lib.lib().hg_last_error(ctx)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("n_seq,L,heads", [(3, 197, 12), (8, 197, 12), (41, 197, 12), (256, 197, 12), (300, 197, 12),
                                           (5, 193, 12), (4, 208, 12), (7, 200, 6), (19, 197, 6)])
def test_fused_qkv_attention_equals_gemm_then_attention(ctx, n_seq, L, heads):
    """The fused kernel against the two kernels it replaces (LayerNorm-folded in_proj GEMM, attention_kernel) on the same
    operands: bit for bit, over ragged item counts (fewer items than CUs, not a multiple of the grid, several rounds per CU),
    the shortest / longest sequence a row tile holds, and D = 384 (six K-tiles: every K-tile kind back to back); repeated
    launches must agree (a race in the counted waits would show as a flaky mismatch); and against fp32 PyTorch."""
    ops = _qkv_attn_operands(n_seq, L, heads, 1000 * L + 10 * n_seq + heads)
    got = run_qkv_attn(ctx, ops, n_seq, L, heads, True)
    if heads * 192 % 256 == 0:      # (the folded ring GEMM wants N = 3 D to be a multiple of 256: D = 384 only has the fp32 reference)
        want = run_qkv_attn(ctx, ops, n_seq, L, heads, False)
        assert torch.equal(got, want), f"max abs diff {(got - want).abs().max().item():.3e}"
    for _ in range(3):
        assert torch.equal(got, run_qkv_attn(ctx, ops, n_seq, L, heads, True))
    a, w, bias, cs, mr = ops
    D = heads * 64
    qkv = ((a.half().float() @ w.half().float().t() - mr[:, :1] * cs[None]) * mr[:, 1:] + bias[None])
    ref = ref_attention(qkv, n_seq, L, heads, False)
    assert (got - ref).abs().max().item() <= 3e-3 * ref.abs().max().item()


@pytest.mark.parametrize("n_seq,L,heads", [(3, 197, 12), (41, 197, 12), (128, 197, 12), (300, 197, 12), (5, 193, 12), (4, 208, 12),
                                           (7, 200, 6)])
def test_fused_qkv_attention_over_d_plus_64_columns(ctx, n_seq, L, heads):
    """The block with a folded instance adapter (variant C): in_proj sums over D + 64 columns = 13 (7 for D = 384) K-tiles, the
    3 m + 1 schedule of the K loop (hg_seq_kloop_run.inc, second kernel instance).  Same evidence as above: bit-identical to the
    K = D + 64 GEMM followed by attention, repeated launches agree, and fp32 PyTorch."""
    ops = _qkv_attn_operands(n_seq, L, heads, 2000 * L + 10 * n_seq + heads, extra=64)
    got = run_qkv_attn(ctx, ops, n_seq, L, heads, 3)
    if heads * 192 % 256 == 0:
        want = run_qkv_attn(ctx, ops, n_seq, L, heads, 2)
        assert torch.equal(got, want), f"max abs diff {(got - want).abs().max().item():.3e}"
    for _ in range(3):
        assert torch.equal(got, run_qkv_attn(ctx, ops, n_seq, L, heads, 3))
    a, w, bias, cs, mr = ops
    qkv = ((a.half().float() @ w.half().float().t() - mr[:, :1] * cs[None]) * mr[:, 1:] + bias[None])
    ref = ref_attention(qkv, n_seq, L, heads, False)
    assert (got - ref).abs().max().item() <= 3e-3 * ref.abs().max().item()
    # the D-column kernel instance still runs unharmed beside it
    ops0 = _qkv_attn_operands(n_seq, L, heads, 5)
    if heads * 192 % 256 == 0:
        assert torch.equal(run_qkv_attn(ctx, ops0, n_seq, L, heads, 1), run_qkv_attn(ctx, ops0, n_seq, L, heads, 0))


def test_fused_qkv_attention_under_xcd_group_sizes(ctx):
    """The order in which an XCD walks its (sequence, head pair) items (option qkv_attn_gsz) never changes a result."""
    n_seq, L, heads = 96, 197, 12
    ops = _qkv_attn_operands(n_seq, L, heads, 77)
    want = run_qkv_attn(ctx, ops, n_seq, L, heads, False)
    for gsz in (1, 2, 3, 6, 0):
        assert _lib.lib().hg_set_option(ctx, b"qkv_attn_gsz", gsz) == 0
        assert torch.equal(run_qkv_attn(ctx, ops, n_seq, L, heads, True), want), gsz
