"""No kernel of the product build may spill registers or use scratch memory (spills had crept back into
two residual instantiations).  Compiles every HIP source for gfx950 with hipcc's resource-usage remarks - no GPU needed."""
import concurrent.futures as cf
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "hoigen_amd", "csrc")
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _usage(src: str, out_dir: str):
    cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-Rpass-analysis=kernel-resource-usage",
           "-c", src, "-o", os.path.join(out_dir, os.path.basename(src) + ".o")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rows, name = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"(VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)", line)
        if m and name:
            rows.append((os.path.basename(src), name, m.group(1), int(m.group(2))))
    return rows


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_kernel_spills_registers_or_uses_scratch():
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    assert len(srcs) >= 8
    with tempfile.TemporaryDirectory() as tmp, cf.ThreadPoolExecutor(4) as ex:
        rows = [r for rs in ex.map(lambda s: _usage(s, tmp), srcs) for r in rs]
    kernels = {(f, n) for f, n, _, _ in rows}
    assert len(kernels) >= 60, f"only {len(kernels)} kernels seen: the remarks were not parsed"
    # SGPR "spills" are v_writelane moves into spare VGPR lanes, not memory: tolerated only in the two fp32 fallback kernels of
    # the adapter (weights held in scalar registers by design: hg_adapter.hip), which no batch-256 path runs
    sgpr_ok = ("adapter_kv_kernel", "adapter_decoder_kernel", "qkv_attn_kernel", "vae_fused_kernelILi0E", "mlp_pair_kernelILi2ELb1E")
    # (mlp_pair_kernel<2, gamma>, hg_mlp_pair.hip - the text tower's instance: two GEMM bodies in one kernel, three scalars are parked in
    # VGPR lanes around its tail; what must not happen inside the bodies is pinned by test_mlp_pair_kernel_codegen below)
    # (qkv_attn_kernel, hg_qkv_attn.hip, runs its K loop on 156 accumulator + 48 fragment registers and its attention phases beside
    # 78 registers of parked fp16 results: it used to park a handful of per-item values in scratch across the K loop; values the
    # allocator would keep live across the loop - a hoisted lane id of __shfl_xor, a hoisted `wave < 2`, a constant pair, the zero high
    # half of a 64-bit store offset - are now made where they are used.  No VGPR spill, no scratch: nothing is tolerated here.)
    # vae_fused_kernel<0> (hg_vae_fused.hip: the instance that holds the Encoder passes, options vae_fused = 2 only) keeps 256 accumulator
    # + 128 operand registers through its pass loops; at the joins between its three pass epilogues the allocator parks one accumulator
    # block in scratch for the duration of an epilogue (three per 128-row item of ~0.3 ms; 30 dwords).  Its pass LOOPS must be free
    # of scratch and vmcnt(0): test_vae_fused_pass_loops_are_scratch_free.  The Generator-only and MLP-block instances (<2>, <3>: what the
    # default dispatch launches) spill nothing and fall under the rule for every other kernel.
    few_ok = {"vae_fused_kernelILi0E": 40}
    def tolerated(r):
        f, n, k, v = r
        if k == "SGPRs Spill" and any(x in n for x in sgpr_ok):
            return True
        lim = next((m for x, m in few_ok.items() if x in n), None)
        return lim is not None and ((k == "VGPRs Spill" and v <= lim) or (k.startswith("ScratchSize") and v <= 4 * lim))
    bad = [r for r in rows if r[3] != 0 and not tolerated(r)]
    assert not bad, "kernels that spill or use scratch:\n" + "\n".join(f"  {f}: {n}: {k} = {v}" for f, n, k, v in bad)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_fused_kernel_k_loop_is_scratch_free():
    """The sequence-tile K loop (hg_seq_kloop_run.inc inside both instances of the fused in_proj + attention kernel:
    qkv_attn_kernel for 3 m K-tiles, qkv_attn_kernel_k1 for 3 m + 1) keeps every DMA in flight behind counted s_waitcnt vmcnt(6):
    between the first and the last of them the assembly must hold no scratch access, no vmcnt(0), and the kernel all 78 MFMAs of
    each of its K-tile bodies x two phases (six bodies = 468; seven = 546 with the extra tail tile of the 3 m + 1 schedule)."""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "qa.s")
        r = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-S", "--cuda-device-only",
                            os.path.join(CSRC, "hg_qkv_attn.hip"), "-o", out], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = open(out).read().split("\n")
    starts = [i for i, l in enumerate(lines) if "qkv_attn_kernel" in l and l.startswith("_ZN") and ":" in l]
    assert len(starts) == 2, "expected the two instances of the fused kernel"
    for start, mfmas in zip(starts, (468, 546)):
        end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or ".Lfunc_end" in lines[i])
        body = lines[start:end]
        w6 = [i for i, l in enumerate(body) if "s_waitcnt vmcnt(6)" in l]
        assert len(w6) >= 12, "the counted waits of the K loop were not found"
        loop = body[w6[0]:w6[-1] + 1]
        assert not [l for l in loop if "scratch_" in l], "scratch access inside the K loop"
        assert not [l for l in loop if "s_waitcnt vmcnt(0)" in l], "vmcnt(0) inside the K loop"
        assert sum("v_mfma_f32_16x16x32_f16" in l for l in body) == mfmas


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_vae_fused_pass_loops_are_scratch_free():
    """The pass loops of the three vae_fused_kernel instances (two iterations of 64 MFMAs per trip) keep six ring stages of LDS-DMA in flight behind
    counted s_waitcnt vmcnt(20): a scratch reload inside them waits for vmcnt(0) and drains the ring once per iteration (measured:
    4 750 instead of 2 100 cycles per iteration).  Every innermost loop that holds MFMAs must hold exactly 128 of them, 8 barriers,
    no scratch access, no vmcnt(0), and no AGPR<->VGPR copies (the layer-1 accumulators are VGPR-form inline asm for that reason)."""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "vf.s")
        r = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-S", "--cuda-device-only",
                            os.path.join(CSRC, "hg_vae_fused.hip"), "-o", out], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = open(out).read().split("\n")
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    with_128 = [(a, b) for a, b in loops if sum("v_mfma" in x for x in lines[a:b + 1]) == 128]
    # (the item loop of a one-pass instance holds the same 128 MFMAs as its pass loop: innermost loops only)
    inner = [(a, b) for a, b in with_128 if not any((a2, b2) != (a, b) and a <= a2 and b2 <= b for a2, b2 in with_128)]
    assert len(inner) == 5, f"expected five pass loops (instance <0>: Encoder x 2 + Generator, <2>: Generator, <3>: MLP block), found {len(inner)}"
    for a, b in inner:
        body = lines[a:b + 1]
        assert sum("s_barrier" in x for x in body) == 8
        assert not [x for x in body if "scratch_" in x], "scratch access inside a pass loop"
        assert not [x for x in body if "vmcnt(0)" in x], "vmcnt(0) inside a pass loop"
        assert not [x for x in body if "v_accvgpr" in x], "accumulator copies inside a pass loop"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_mlp_pair_kernel_codegen():
    """hg_mlp_pair.hip runs the c_fc and the c_proj body in ONE kernel; four ways in which that silently costs 15-50 us per launch were
    found while building it (profiles/r06_mlp_pair.txt) - each is visible in the assembly and held off here, per instance:
      * no flat_ instruction (pointers that lost their address space: flat accesses count in lgkmcnt, every fetch segment behind an
        epilogue then waits for the tile's stores);
      * the QuickGELU epilogue block keeps its interleaved schedule (<= 100 s_nop; 242 when scalar-register spilling makes the
        scheduler minimise pressure in every region - c_proj's arguments are therefore read from the kernel-argument segment behind the
        c_fc body, and get their address space back through a cast of their bits);
      * every MFMA operand tuple starts at a multiple of 4 registers (threadIdx.x kept alive in v0 across the bodies shifts them to 2 mod 4);
      * no lane moves of spilled scalars inside any block that issues MFMAs, no scratch anywhere."""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "pair.s")
        r = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-S", "--cuda-device-only",
                            os.path.join(CSRC, "hg_mlp_pair.hip"), "-o", out], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        txt = open(out).read()
    names = re.findall(r"^(_ZN2hg15mlp_pair_kernel\S+):", txt, re.M)
    assert len(names) == 5, names      # stream fp32 / hi-lo / hi-lo -> fp32, and the text tower's two with gamma in the activation copy
    for name in names:
        i = txt.index(name + ":")
        body = txt[i:txt.index("s_endpgm", i)]
        assert "flat_" not in body, name
        assert "scratch_" not in body, name
        blocks, cur = [], []
        for line in body.split("\n"):
            if re.match(r"^\.LBB\S+:", line):
                blocks.append(cur)
                cur = []
            else:
                ins = re.sub(r";.*", "", line).strip()
                if ins and not ins.startswith("."):
                    cur.append(ins)
        blocks.append(cur)
        n_mfma = sum(1 for b in blocks for x in b if x.startswith("v_mfma"))
        assert n_mfma == 640, (name, n_mfma)      # c_fc: 6 K-tile kinds x 64; c_proj: 8 x 32
        epilogues = [b for b in blocks if sum(1 for x in b if x.startswith("v_exp_f32")) >= 64]
        assert len(epilogues) == 1 and sum(1 for x in epilogues[0] if x.startswith("s_nop")) <= 100, name
        for b in blocks:
            if any(x.startswith("v_mfma") for x in b):
                assert not any(x.startswith(("v_readlane", "v_writelane")) for x in b), name
        for m in re.finditer(r"v_mfma\S+ v\[(\d+):\d+\], v\[(\d+):\d+\], v\[(\d+):\d+\]", body):
            assert int(m.group(1)) % 4 == 0 and int(m.group(2)) % 4 == 0, (name, m.group(0))
