"""Full-size and chunk-boundary properties of the HIP path (BASELINE.json configs 4 and 5 on one GPU) and the
round-1 advisor findings that only show on a device.

Rows (crops, prompts, VAE rows) are independent, and the native library processes large calls in chunks
(`max_chunk_img` 256 crops, `max_chunk_txt` 640 prompts, `max_chunk_rows` 32 768 rows): a call that crosses a
chunk boundary must return, row for row, the bits of separate calls on the pieces.  Nothing here needs the
oracle at these sizes; parity of the pieces is pinned in test_gpu_parity.py.
"""
import gc
import json
import os

import numpy as np
import pytest
import torch

from hoigen_amd import synth, vae
from hoigen_amd.model import build_model

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def g0():
    return json.load(open(f"{G}/g0_tokens.json"))


@pytest.fixture(scope="module")
def fullA():
    return build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(dev())


def test_config5_batch2048_equals_eight_chunks_of_256(fullA):
    """BASELINE config 5's global batch on ONE GPU: encode_image(2048 crops) == the eight 256-crop shards the
    8-GPU job encodes, bit for bit (the all-gather only concatenates)."""
    d = dev()
    gen = torch.Generator(device=d).manual_seed(2048)
    crops = torch.randn(2048, 3, 224, 224, device=d, generator=gen)
    golden4 = torch.from_numpy(synth.crops(4, 224, seed=1234)).to(d)
    crops[1000:1004] = golden4
    whole = fullA.encode_image(crops)
    assert whole.shape == (2048, 512) and torch.isfinite(whole).all()
    for r in range(8):
        part = fullA.encode_image(crops[256 * r: 256 * (r + 1)])
        assert torch.equal(part, whole[256 * r: 256 * (r + 1)]), f"shard {r} differs from the single-call result"
    ref = np.load(f"{G}/g2_vitb16_image.npz")["encode_image"]
    got = whole[1000:1004].float().cpu().numpy()
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-3      # reference crops inside the big batch



def test_fused_in_proj_attention_is_bit_identical_inside_encode_image(fullA):
    """Option qkv_attn (hg_qkv_attn.hip: in_proj + attention as one kernel, q / k / v in LDS): encode_image with it forced on (2),
    chosen by batch (1, the default) and off (0) gives the SAME bits at batch 256 (six full rounds of items), 171 (the default
    picks the two kernels there) and 40 (fewer items than CUs), with every row of the last block computed and with the class rows
    only; the reference's golden crops ride inside the batch."""
    d = dev()
    gen = torch.Generator(device=d).manual_seed(256)
    crops = torch.randn(256, 3, 224, 224, device=d, generator=gen)
    crops[100:104] = torch.from_numpy(synth.crops(4, 224, seed=1234)).to(d)
    ref = np.load(f"{G}/g2_vitb16_image.npz")["encode_image"]
    try:
        for row0 in (1, 0):
            fullA.visual.set_option("last_block_row0", row0)
            for n in (256, 171, 40):
                outs = []
                for mode in (2, 1, 0):
                    fullA.visual.set_option("qkv_attn", mode)
                    outs.append(fullA.encode_image(crops[256 - n:]))
                assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[2]), (row0, n)
        fullA.visual.set_option("qkv_attn", 2)
        got = fullA.encode_image(crops)[100:104].float().cpu().numpy()
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-3
    finally:
        fullA.visual.set_option("qkv_attn", 1)
        fullA.visual.set_option("last_block_row0", 1)

def test_mlp_pair_one_launch_is_bit_identical_inside_encode_image(fullA):
    """Option mlp_pair (hg_mlp_pair.hip: c_fc -> QuickGELU -> c_proj of a block as ONE persistent launch, the c_fc tiles publishing
    per-row-panel ready counters their c_proj tiles wait for; with value 2 the LayerNorm statistics of the updated rows are combined in the
    launch's tail by the last of a row half's column-tile workgroups): the same tiles, K loops, epilogues and statistics in another order and on other
    workgroups, so encode_image gives the SAME bits as the two launches + finalize_stats - at batch 256, 171 (ragged last panels) and 40 (fewer
    c_proj tiles than two rounds), with several chunk sizes of the c_fc tile order and with 30 or 24 of an XCD's 32 workgroups running
    c_fc tiles (the others really wait for their first panels), every row of the last block or the class rows only; repeated launches
    agree (a race would show up as a difference); the reference's golden crops ride inside the batch."""
    d = dev()
    gen = torch.Generator(device=d).manual_seed(77)
    crops = torch.randn(256, 3, 224, 224, device=d, generator=gen)
    crops[60:64] = torch.from_numpy(synth.crops(4, 224, seed=1234)).to(d)
    ref = np.load(f"{G}/g2_vitb16_image.npz")["encode_image"]
    try:
        for row0 in (0, 1):
            fullA.visual.set_option("last_block_row0", row0)
            for n in (256, 171, 40):
                fullA.visual.set_option("mlp_pair", 0)
                want = fullA.encode_image(crops[:n])
                for pair, chunk, slots in ((1, 32, 32), (2, 32, 32), (1, 8, 30), (1, 3, 24), (2, 25, 30)):
                    fullA.visual.set_option("mlp_pair", pair)      # (1: finalize_stats in a launch of its own; 2: its work in the pair launch's tail)
                    fullA.visual.set_option("mlp_pair_chunk", chunk)
                    fullA.visual.set_option("mlp_pair_fc_slots", slots)
                    for rep in range(3 if n == 256 else 1):
                        got = fullA.encode_image(crops[:n])
                        assert torch.equal(got, want), (row0, n, pair, chunk, slots, rep, float((got - want).abs().max()))
        got = fullA.encode_image(crops)[60:64].float().cpu().numpy()
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-3
    finally:
        fullA.visual.set_option("mlp_pair", 1)
        fullA.visual.set_option("mlp_pair_chunk", 32)
        fullA.visual.set_option("mlp_pair_fc_slots", 32)
        fullA.visual.set_option("last_block_row0", 1)


def test_generator_above_a_million_rows():
    """hg_generator / hg_vae_forward on more than 2^20 rows (the one-kernel path addresses its [R, 512] tensors with 32-bit byte
    offsets, i.e. 2^20 rows per launch - the default dispatch used to hand it 1 081 344 rows of a 1.1 M-row call and fail): the launcher
    slices at 2^20 rows.  Rows on both sides of the slice boundary equal the same rows of a short call bit for bit (option vae_fused = 2:
    every row through the one kernel), and the default dispatch agrees with it to fp32 summation order."""
    d = dev()
    Gn = vae.Generator().to(d)
    Gn.load_state_dict(synth.to_torch(synth.generator_state_dict(3)))
    R = 1_100_000
    gen = torch.Generator(device=d).manual_seed(11)
    z = torch.randn(R, 512, device=d, generator=gen)
    lo, hi = (1 << 20) - 300, (1 << 20) + 300
    try:
        vae.set_option("vae_fused", 2, d)
        whole = Gn(z)
        assert torch.isfinite(whole[::4099]).all()
        part = Gn(z[lo:hi].contiguous())
        assert torch.equal(whole[lo:hi], part)
        assert torch.equal(whole[:128], Gn(z[:128].contiguous()))
        vae.set_option("vae_fused", 1, d)
        dflt = Gn(z)
        rel = float((dflt[lo:hi].double() - whole[lo:hi].double()).norm() / whole[lo:hi].double().norm())
        assert rel < 1e-5, rel
    finally:
        vae.set_option("vae_fused", 1, d)


def test_mlp_pair_wait_that_cannot_end_is_an_error_not_a_hang(fullA):
    """The c_proj tiles of the MLP pair launch wait for other workgroups of the same launch (hg_mlp_pair.hip).  Option mlp_pair_fault = 1
    sends that launch out one workgroup short: a work slot of one XCD stays untaken, its c_fc tiles are never produced, the row panels
    they belong to never become complete.  What must happen: the waits give up at their bound (~0.6 s for the first, the others as soon as
    they see the error word), the call comes back - with wrong rows, it cannot be failed without a synchronisation -, the NEXT call
    fails with HG_ERR_HIP (RuntimeError) and consumes the report; with the option back at 0 the same context gives the same bits as
    before.  Never a hang, never a silent wrong answer."""
    import time
    d = dev()
    x = torch.randn(40, 3, 224, 224, device=d, generator=torch.Generator(device=d).manual_seed(31))
    v = fullA.visual
    want = fullA.encode_image(x).clone()
    torch.cuda.synchronize()
    v.set_option("mlp_pair_fault", 1)
    try:
        t0 = time.time()
        fullA.encode_image(x)
        torch.cuda.synchronize()
        took = time.time() - t0
        print(f"\nencode_image with one workgroup of every pair launch missing came back after {took:.1f} s")
        assert took < 120, "the bounded waits did not end"
        with pytest.raises(RuntimeError, match="hand-off wait inside the MLP pair kernel"):
            fullA.encode_image(x)
    finally:
        v.set_option("mlp_pair_fault", 0)
    torch.cuda.synchronize()
    again = fullA.encode_image(x)
    assert torch.equal(again, want), "the context did not recover after the reported fault"


def test_mlp_pair_launches_from_two_streams_do_not_wait_for_each_other(fullA, g0):
    """Both towers run the MLP pair launch, whose 256 workgroups wait for each other: if encode_image on one stream and encode_text on
    another (two contexts) ever held part of the CUs each, both would sit out their bounds.  The library therefore serialises pair
    launches of a device across streams from the moment a second stream shows up (hg_api.hip PairGate; tools/two_stream_pair_stress.py
    is the long version of this test and ran clean with and without the gate - the gate closes a window, it does not fix a failure
    that was seen): both calls give the bits of the single-stream calls, no wait gives up (the next calls do not raise), and the two
    streams together take no longer than about the two calls one after the other."""
    import time
    from hoigen_amd import clip
    d = dev()
    x = torch.randn(64, 3, 224, 224, device=d, generator=torch.Generator(device=d).manual_seed(32))
    ids = clip.tokenize(g0["hoi600"]["text"][:300]).to(d)
    want_i, want_t = fullA.encode_image(x).clone(), fullA.encode_text(ids).clone()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        fullA.encode_image(x); fullA.encode_text(ids)
    torch.cuda.synchronize()
    serial = (time.time() - t0) / 3
    sa, sb = torch.cuda.Stream(device=d), torch.cuda.Stream(device=d)
    sa.wait_stream(torch.cuda.current_stream(d)); sb.wait_stream(torch.cuda.current_stream(d))
    outs = []
    t0 = time.time()
    for _ in range(6):
        with torch.cuda.stream(sa):
            a = fullA.encode_image(x)
        with torch.cuda.stream(sb):
            t = fullA.encode_text(ids)
        outs.append((a, t))
    torch.cuda.synchronize()
    both = (time.time() - t0) / 6
    print(f"\nencode_image(64) + encode_text(300): one stream {serial * 1e3:.2f} ms, two streams {both * 1e3:.2f} ms per pair of calls")
    for a, t in outs:
        assert torch.equal(a, want_i) and torch.equal(t, want_t)
    assert both < 3 * serial + 0.05, "the two streams held each other up"
    fullA.encode_image(x); fullA.encode_text(ids)      # (a wait that gave up would make these raise)
    torch.cuda.synchronize()


def test_mlp_pair_in_the_text_tower_is_bit_identical(fullA):
    """The same one-launch MLP in the text tower (width 512: c_fc 8 column tiles, c_proj 2; the next LayerNorm's weight rides in the
    activation copy - the kernel's gamma instances; 46 200 rows at 77 tokens: a ragged last panel with one 128-row half) and in the
    truncated tower the generation pipeline runs (13-16 tokens): encode_text equals the two launches bit for bit, for all three
    text_ln_fold settings (0: separate LayerNorm kernels - no pair there, the switch must be harmless)."""
    d = dev()
    g0 = json.load(open(f"{G}/g0_tokens.json"))
    rows = g0["hoi600"]["ids"]
    ids = np.zeros((len(rows), 77), np.int64)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = r
    ids = torch.from_numpy(ids).to(d)
    prev_fold, prev_trunc = fullA.get_option("text_ln_fold"), fullA.truncate_text
    try:
        for fold in (1, 2, 0):
            fullA.set_option("text_ln_fold", fold)
            for trunc in (False, True):
                fullA.truncate_text = trunc
                fullA.set_option("mlp_pair", 0)
                want = fullA.encode_text(ids)
                for pair in (1, 2):
                    fullA.set_option("mlp_pair", pair)
                    got = fullA.encode_text(ids)
                    assert torch.equal(got, want), (fold, trunc, pair, float((got.float() - want.float()).abs().max()))
    finally:
        fullA.set_option("mlp_pair", 1)
        fullA.set_option("text_ln_fold", prev_fold)
        fullA.truncate_text = prev_trunc


def test_config4_vae_100k_rows_equals_chunks():
    """BASELINE config 4: 100 000 rows through Encoder -> reparameterise -> Generator in one call == separate calls on the row
    ranges, ragged ranges included - on the GEMM path (option vae_fused = 0: crosses the 32 768-row chunk boundary three times) and on
    the one-kernel path (vae_fused = 2: 782 work items of 128 rows), bit for bit each: a row's result depends on its path, never on its
    neighbours or its position.  The default (vae_fused = 1) keeps the Encoder on the GEMM path and gives the Generator of the leading
    98 304 rows (three whole rounds of items over 256 CUs) to the one kernel, the last 1 696 rows to the GEMMs; the paths agree to fp32
    summation order."""
    d = dev()
    E, Gn = vae.Encoder().to(d), vae.Generator().to(d)
    E.load_state_dict(synth.to_torch(synth.encoder_state_dict(2)))
    Gn.load_state_dict(synth.to_torch(synth.generator_state_dict(3)))
    V = vae.VAE(E, Gn)
    gen = torch.Generator(device=d).manual_seed(4)
    R = 100_000
    x = vae.l2_normalize(torch.randn(R, 512, device=d, generator=gen))
    eps = torch.randn(R, 512, device=d, generator=gen)
    names = ("mean", "log_var", "z", "bias")
    whole = {}
    try:
        for mode in (0, 2):
            vae.set_option("vae_fused", mode, d)
            whole[mode] = V(x, eps)
            assert all(t.shape == (R, 512) and torch.isfinite(t).all() for t in whole[mode])
            for lo, hi in ((0, 32768), (32768, 65536), (65536, 98304), (98304, R), (32000, 33111), (99999, R)):
                part = V(x[lo:hi], eps[lo:hi])
                for a, b, n in zip(part, whole[mode], names):
                    assert torch.equal(a, b[lo:hi]), f"vae_fused={mode}: {n} rows [{lo},{hi}) differ from the single-call result"
        vae.set_option("vae_fused", 1, d)
        auto = V(x, eps)
        n_cu = torch.cuda.get_device_properties(d).multi_processor_count
        items = (R + 127) // 128
        full = items // n_cu * n_cu
        rf = min(R, (full + (items - full if (items - full) * 100 >= n_cu * 70 else 0)) * 128)
        for a, f, g, n in zip(auto, whole[2], whole[0], names):
            err = float((f.double() - g.double()).norm() / g.double().norm())
            assert err < 2e-4, f"{n}: one-kernel path vs GEMM path {err:.2e}"
            if n != "bias":      # default dispatch: the Encoder and the reparameterisation stay on the GEMM path ...
                assert torch.equal(a, g), f"{n}: the default dispatch moved the Encoder off the GEMM path"
            else:                # ... the Generator of the leading whole rounds of items is the one kernel, on the GEMM path's fp16 z
                assert torch.equal(a[rf:], g[rf:]) and rf > 0 and not torch.equal(a[:rf], g[:rf])
                e2 = float((a[:rf].double() - g[:rf].double()).norm() / g[:rf].double().norm())
                assert e2 < 2e-4, f"bias: one-kernel Generator vs GEMM Generator on the same z {e2:.2e}"
    finally:
        vae.set_option("vae_fused", 1, d)
    # generator-only sampling path (main_tip_finetune.py:779-781)
    z = torch.randn(R, 512, device=d, generator=gen)
    gw = Gn(z)
    assert torch.equal(Gn(z[40000:70001]), gw[40000:70001])
    # mlp_net over the same row count (finetune_ship.py:302-314)
    M = vae.mlp_net(512, 512, 512).to(d)
    M.load_state_dict(synth.to_torch(synth.mlp_net_state_dict(4)))
    mw = M(x)
    assert torch.equal(M(x[32760:32780]), mw[32760:32780])


def test_variant_c_600_crops_with_priors_equals_chunks():
    """Variant C (adapters + priors, CLIP_models_adapter_prior2.py:489-506) across the 256-crop chunk boundary:
    priors and masks must be offset with the crops."""
    d = dev()
    sd = synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))
    sd.update(synth.to_torch(synth.adapter_state_dict(synth.VIT_B16, 21)))
    m = build_model(sd, use_adapter=True, adapter_pos="all").to(d)
    B, N = 600, 14
    gen = torch.Generator(device=d).manual_seed(600)
    img = torch.randn(B, 3, 224, 224, device=d, generator=gen)
    pri = torch.randn(B, N, 64, device=d, generator=gen)
    mask = torch.zeros(B, N, dtype=torch.bool, device=d)
    mask[:, N - 4:] = True
    mask[::3, N - 6:] = True                       # crops differ in their number of padded prior tokens
    glob, loc = m.visual(img, (pri, mask))
    assert glob.shape == (B, 512) and loc.shape == (B, 512, 14, 14) and torch.isfinite(loc).all()
    for lo, hi in ((0, 256), (256, 512), (512, 600), (250, 262)):
        g2, l2 = m.visual(img[lo:hi], (pri[lo:hi], mask[lo:hi]))
        assert torch.equal(g2, glob[lo:hi]) and torch.equal(l2, loc[lo:hi]), f"crops [{lo},{hi})"
    # and the prior matters (a shifted prior changes the rows it belongs to)
    g3, _ = m.visual(img[256:260], (pri[0:4], mask[0:4]))
    assert not torch.equal(g3, glob[256:260])


def test_text_700_prompts_equals_chunks(fullA, g0):
    """encode_text across a pass boundary of the text tower (65 536 rows per pass, equal passes): 1 000 prompts x 77 tokens = two passes
    of 500 (the 600 HOI + 81 object + 117 verb + 202 CoOp prompts); truncated to their 13-16 tokens they are one pass.  A prompt's bits
    depend on the PATH its call takes, never on its neighbours: with the LayerNorms folded (text_ln_fold 1, the default, and 2) every
    call of at least 512 rows takes the folded path and smaller calls the separate kernels - as in the vision tower - so pieces of >= 512
    rows equal the whole call bit for bit, a smaller piece equals the separate-kernel result bit for bit and the folded one within the
    parity tolerance; with text_ln_fold = 0 every piece equals the whole call."""
    rows = g0["hoi600"]["ids"] + g0["obj81"]["ids"] + g0["verb117"]["ids"] + g0["coop_hoi600"]["ids"][:202]
    ids = np.zeros((len(rows), 77), np.int64)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = r
    ids = torch.from_numpy(ids).to(dev())
    assert ids.shape[0] == 1000
    longest = int(ids.argmax(dim=-1).argmax())

    def piece(lo, hi, trunc):
        # truncation length = max(EOT)+1 over the CALL: give the pieces the same length by keeping the longest prompt of the set in each
        sel = torch.arange(lo, hi, device=ids.device)
        if trunc and not (lo <= longest < hi):
            sel = torch.cat([sel, torch.tensor([longest], device=ids.device)])
        return fullA.encode_text(ids[sel])[: hi - lo]

    # (option mlp_fused lives in the separate-LayerNorm path: a row's MLP runs as the one kernel or as two GEMMs depending on where it
    # falls in its pass - each path is held to bit-equality on its own, the default against both)
    wholes = {}
    try:
        for fold, mlp in ((0, 0), (0, 2), (0, 1), (1, 0), (2, 0)):
            fullA.set_option("text_ln_fold", fold)
            fullA.set_option("mlp_fused", mlp)
            for trunc in (False, True):
                fullA.truncate_text = trunc
                whole = wholes[fold, mlp, trunc] = fullA.encode_text(ids)
                if (fold, mlp) == (0, 1):
                    for other in (0, 2):
                        ref = wholes[0, other, trunc].float()
                        err = float((whole.float() - ref).norm() / ref.norm())
                        assert err < 9e-4, f"mlp_fused=1 vs {other}, truncate={trunc}: {err:.2e}"      # (two realisations of 12 blocks of fp16 roundings, each ~6.5e-4 from the reference)
                    continue
                for lo, hi in ((0, 500), (500, 1000), (480, 530), (0, 640)):      # (50 prompts x 13 tokens = 650 rows: the folded path)
                    assert torch.equal(piece(lo, hi, trunc), whole[lo:hi]), f"prompts [{lo},{hi}) truncate={trunc} text_ln_fold={fold} mlp_fused={mlp}"
                small = piece(490, 510, trunc)      # 20 prompts: 273 rows truncated (the separate kernels whatever the option), 1 540 at 77 tokens
                if fold == 0 or not trunc:
                    assert torch.equal(small, whole[490:510]), f"prompts [490,510) truncate={trunc} text_ln_fold={fold} mlp_fused={mlp}"
                else:
                    assert torch.equal(small, wholes[0, 0, trunc][490:510]), "a call below 512 rows must take the separate-LayerNorm path"
                    err = float((small.float() - whole[490:510].float()).norm() / whole[490:510].float().norm())
                    assert err < 1e-3, f"small call vs folded whole call: {err:.2e}"
        assert not torch.equal(wholes[0, 0, False], wholes[0, 2, False]), "option mlp_fused did not change the executed path"
        assert not torch.equal(wholes[0, 0, False], wholes[1, 0, False]) and not torch.equal(wholes[1, 0, False], wholes[2, 0, False])
    finally:
        fullA.set_option("mlp_fused", 0)
        fullA.set_option("text_ln_fold", 1)
        fullA.truncate_text = True


def test_prompt_learner_fp16_buffers_and_renamed_classes(fullA, g0):
    """with an fp16 CLIP (the real case after clip.load on a GPU) the learner's token_prefix /
    token_suffix / ctx are fp16 and were converted to fp32 TEMPORARIES whose storage was free before the kernel ran.
    Prompts must equal the exact fp32 concatenation of the fp16-rounded operands (main_coop_vae.py:119-128), call
    after call, with allocator churn in between, and after get_prefix_suffix_token with new class names
    (main_tip_finetune.py:564 does this every step)."""
    d = dev()
    assert fullA.dtype == torch.float16
    names_a = g0["_classnames"]["hoi"][:40]
    names_b = g0["_classnames"]["obj"][:25]
    pl = vae.PromptLearner_hoi(names_a, fullA)
    assert pl.ctx.dtype == torch.float16 and pl.token_prefix.dtype == torch.float16
    with torch.no_grad():
        pl.ctx.copy_(torch.from_numpy(synth.hg_normal((5, 512), 40, 0.02)).to(d))

    def expect(bias, target):
        pre, suf, ctx = (t.detach().float() for t in (pl.token_prefix, pl.token_suffix, pl.ctx))
        return torch.cat([pre[target], ctx[None] + bias[:, None, :], suf[target]], dim=1)

    gen = torch.Generator(device=d).manual_seed(9)
    for names in (names_a, names_b, names_a):
        pl.get_prefix_suffix_token(names, fullA)
        assert pl.token_prefix.dtype == torch.float16 and pl.token_prefix.shape[0] == len(names)
        for it in range(6):
            R = 7 + 13 * it
            bias = torch.randn(R, 512, device=d, generator=gen) * 0.1
            target = torch.randint(0, len(names), (R,), device=d, generator=gen)
            # churn the caching allocator with blocks of the sizes the temporaries had
            junk = [torch.full((n,), float("nan"), device=d) for n in (5 * 512, len(names) * 512, len(names) * 71 * 512)]
            del junk
            out = pl(bias, target)
            junk = [torch.full((n,), float("nan"), device=d) for n in (5 * 512, len(names) * 512, len(names) * 71 * 512)]
            torch.cuda.synchronize()
            assert torch.equal(out, expect(bias, target)), f"prompts differ ({len(names)} names, call {it})"
            del junk
    te = vae.TextEncoder(fullA)
    feats = te(pl(bias, target), pl.tokenized_prompts[target])
    assert feats.shape == (R, 512) and torch.isfinite(feats).all()


def test_cache_logits_slots_are_not_silently_reused():
    """the 9th live CacheLogits must raise instead of overwriting slot 0; released slots return."""
    from hoigen_amd import cache_model as cm
    d = dev()
    gen = torch.Generator(device=d).manual_seed(3)
    ws = [torch.randn(128, 512, device=d, generator=gen) for _ in range(cm.HG_MAX_CACHE_SLOTS + 1)]
    f = torch.randn(16, 512, device=d, generator=gen)
    before = len(cm._free)
    live = [cm.CacheLogits(w) for w in ws[:before]]
    first = live[0](f)
    with pytest.raises(RuntimeError, match="slots are in use"):
        cm.CacheLogits(ws[-1])
    assert torch.equal(live[0](f), first), "an older object's weights were overwritten"
    live[3].close()
    again = cm.CacheLogits(ws[-1])
    ref = (f.half().float() @ ws[-1].half().float().T)
    assert (again(f) - ref).norm() / ref.norm() < 1e-3
    with pytest.raises(RuntimeError, match="closed"):
        live[3](f)
    del live, again
    gc.collect()
    assert len(cm._free) == before


def test_entry_points_leave_the_current_device_alone(fullA):
    """native entry points run on their context's device and restore the caller's."""
    if torch.cuda.device_count() < 2:
        cur = torch.cuda.current_device()
        fullA.encode_image(torch.randn(2, 3, 224, 224, device=dev()))
        assert torch.cuda.current_device() == cur
        return
    with torch.cuda.device(1):
        fullA.encode_image(torch.randn(2, 3, 224, 224, device=dev()))
        assert torch.cuda.current_device() == 1


_RCCL_CHILD = r"""
import os, sys
sys.path.insert(0, {repo!r})
import torch, torch.distributed as dist
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)   # RCCL
from hoigen_amd import synth
from hoigen_amd.distributed import ShardedEncoder
from hoigen_amd.model import build_model
model = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(dev)
B = 8
enc = ShardedEncoder(model.visual.encode_into, B, 512, dev, force_comm=True)
assert enc.comm is not None and enc.world == 1
gen = torch.Generator(device=dev).manual_seed(7)
batches = [torch.randn(B, 3, 224, 224, device=dev, generator=gen) for _ in range(3)]
side = torch.cuda.Stream(dev)
outs = []
for x in batches:                      # three steps: both buffers are reused once, through the done-event chain
    buf, ev = enc.step(x)
    assert ev is not None
    with torch.cuda.stream(side):      # a consumer on another stream waits for the gather's event
        side.wait_event(ev)
        outs.append(buf.clone())
enc.finish()
torch.cuda.synchronize()
for x, o in zip(batches, outs):
    ref = model.visual(x)                      # model.dtype (fp16): the same fp32 embeddings, cast
    assert torch.equal(o.to(ref.dtype), ref), "gathered buffer differs from encode_image"
dist.barrier()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
"""


def test_sharded_encoder_rccl_leg_on_one_rank():
    """the RCCL leg of ShardedEncoder (side stream, event chain, in-place all_gather_into_tensor on an
    `nccl` group created with device_id) executed on the one GPU of this box, in a fresh child process that creates the
    process group before anything else touches the GPU; the gathered rows equal encode_image's."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", _RCCL_CHILD.format(repo=repo, port=port)], capture_output=True, text=True,
                       env=env, timeout=900)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_variant_c_adapter_folded_into_the_block_gemms_equals_separate_up_proj():
    """The two ways the adapter's update reaches the stream (option adapter_fold): 0 = up_proj GEMM with a scaled
    residual epilogue (the fallback for shapes the folded path does not take), 1 (default) = no up_proj at all - QKV and
    out-proj take 64 more K columns, ln_1's statistics come from the decoder.  CLIP_models_adapter_prior2.py:184-203,456.
    Same function, different fp16 roundings: within the parity tolerance of each other, with and without priors, one and two
    decoder layers."""
    d = dev()
    for layers, seed in ((1, 21), (2, 22)):
        sd = synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))
        sd.update(synth.to_torch(synth.adapter_state_dict(synth.VIT_B16, seed, num_layers=layers)))
        m = build_model(sd, use_adapter=True, adapter_pos="all", adapter_num_layers=layers).to(d)
        g = torch.Generator(device=d).manual_seed(7)
        img = torch.randn(6, 3, 224, 224, device=d, generator=g)
        pri = torch.randn(6, 14, 64, device=d, generator=g)
        mask = torch.zeros(6, 14, dtype=torch.bool, device=d)
        mask[:, 10:] = True
        res = {}
        for mode in (1, 0):
            m.set_option("adapter_fold", mode)
            res[mode] = [t.clone() for t in m.visual(img, (pri, mask))] + [t.clone() for t in m.visual(img, None)]
        for a, b, name in zip(res[1], res[0], ("global", "local", "global no prior", "local no prior")):
            assert torch.isfinite(a).all()
            e = ((a - b).norm() / b.norm()).item()
            print(f"adapter folded vs separate up_proj, {layers} layer(s), {name}: rel-L2 {e:.2e}")
            assert 0 < e < 1e-3, f"{layers} layer(s), {name}: rel-L2 {e:.2e}"      # (> 0: the switch changed the executed path)
        assert not torch.equal(res[1][0], res[1][2])      # the prior matters
        del m


def test_vae_family_modules_share_one_context_per_device():
    """Encoder / Generator / VAE / mlp_net objects take weight slots of ONE native context per device (one workspace however many
    modules: the three-branch sampler of main_tip_finetune.py:759-824 holds nine), give them back when they die, and say so when
    a seventeenth is asked for."""
    import gc
    from hoigen_amd import _lib
    d = dev()
    gc.collect()
    x = vae.l2_normalize(torch.randn(64, 512, device=d))
    mods, outs = [], []
    for k in range(5):                                   # five encoders with different weights, alive together
        E = vae.Encoder().to(d).eval()
        E.load_state_dict(synth.to_torch(synth.encoder_state_dict(50 + k)))
        mods.append(E)
        outs.append(E(x)[0].clone())
    handles = {m._slot.get(d)[1] for m in mods}
    assert len(handles) == 1, "one shared context"
    assert len({m._slot.slot for m in mods}) == 5, "five different slots"
    for k, E in enumerate(mods):                          # each still answers with its own weights
        assert torch.equal(E(x)[0], outs[k])
    assert not torch.equal(outs[0], outs[1])
    # exhaustion is loud, and dead modules free their slots
    extra = []
    with pytest.raises(RuntimeError, match="live vae modules"):
        for _ in range(_lib.HG_MAX_SLOTS + 1):
            G = vae.Generator().to(d).eval()
            G(torch.randn(4, 512, device=d))
            extra.append(G)
    del extra, G
    gc.collect()
    G = vae.Generator().to(d).eval()
    assert G(torch.randn(4, 512, device=d)).shape == (4, 512)


def test_vae_family_modules_on_different_streams_and_threads_serialise_on_the_shared_context():
    """all VAE-family modules of a device share one native context and workspace.  Modules driven from different
    streams (and threads) must still give their own answers: the façade reserves the context per call and makes a new stream
    wait for the previous user's work (vae._Session)."""
    import threading
    d = dev()
    x = vae.l2_normalize(torch.randn(20000, 512, device=d))
    gens = []
    for k in range(3):
        G = vae.Generator().to(d).eval()
        G.load_state_dict(synth.to_torch(synth.generator_state_dict(60 + k)))
        gens.append(G)
    want = [G(x).clone() for G in gens]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=d) for _ in gens]
    got = [None] * len(gens)

    def work(i):
        with torch.no_grad(), torch.cuda.stream(streams[i]):
            for _ in range(4):
                got[i] = gens[i](x)

    for rounds in range(2):
        ts = [threading.Thread(target=work, args=(i,)) for i in range(len(gens))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        for i in range(len(gens)):
            assert torch.equal(got[i], want[i]), f"round {rounds}, generator {i}: another stream's launch ran over the shared workspace"
