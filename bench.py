#!/usr/bin/env python3
"""Headline benchmark: union-crop embeddings/sec, CLIP ViT-B/16, batch 256 per GPU (BASELINE.json
configs[1]); weak scaling over N GPUs (one process per GPU, RCCL all-gather of the [256,512] embeddings per step
on a side stream = BASELINE.json configs[4] at N = 8).

    python bench.py [--gpus N --steps K --warmup W]          # N > 1 without WORLD_SIZE: starts the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of ``encode_image`` over one batch of 256 synthetic N(0,1) crops already resident in HBM.
Rank 0 prints ONE JSON line.  Weights are the seeded synthetic ViT-B/16 of hoigen_amd.synth (no checkpoint is
reachable offline).  `value` is measured with every row of every block computed (option last_block_row0 = 0).

Extra objects on the line:
  roofline         the kernel with the largest share of the step (found by timing every GEMM / attention launch of
                   two untimed steps), timed live with hipEvent pairs on its launch stream during the first timed
                   steps: achieved = mean algorithmic FLOPs per launch / mean duration, vs the dense 16-bit MFMA peak
                   2516.6 TFLOP/s.  `e2e_frac` = whole-step algorithmic FLOPs / step time: THE figure to hold against
                   the 40 % target.
  kernels          per launch shape: launches per step, mean ms, FLOPs, fraction of MFMA peak, algorithmic HBM bytes
                   and fraction of 8 TB/s (from the two untimed, fully instrumented steps).
  step_ms          median / p10 / p90 of the individual timed steps (events on the compute stream).
  class_rows_only  the library's default (last block on the class-token rows only), same protocol.
  power            (N = 1) package power / shader clock while the headline step runs (rocm-smi beside an untimed burst).
  variant_c        (N = 1) the detector's call `visual(x, prior)` with instance adapters on the same crops.
  config3/config4  (N = 1) encode_text over the 600 HICO prompts and the CoOp-VAE on 100 000 rows, each with its own
                   CPU baseline sample.
  cpu_baseline     the CPU oracle (a port of the reference's CPU path, pinned to the reference's own outputs) timed on
                   this box's host cores on a bounded sample (rank 0, N = 1).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

MFMA_PEAK_TFLOPS = 2516.6          # 256 CU x 4096 FLOP/clk/CU x 2.4 GHz (dense fp16/bf16), BASELINE.md §2
HBM_PEAK_GBS = 8000.0
FLOPS_PER_CROP = 35.127e9          # BASELINE.md §2 (every row of every block)
# The library's default runs the LAST block's attention, out-proj and MLP on the class-token row only (its other 196
# rows never reach the embedding); per skipped row: the Q and out projections, c_fc + c_proj, its attention row.
FLOPS_DEAD_ROWS = 196 * (2 * 2 * 768 * 768 + 2 * 2 * 768 * 3072 + 4 * 197 * 768)
BATCH = 256
KIND_NAMES = {0: "gemm bias->f16", 1: "gemm bias+QuickGELU->f16", 2: "gemm bias+ReLU->f16", 3: "gemm bias+residual",
              4: "gemm bias->f32", 5: "gemm patch-embed", 6: "gemm bias+ReLU->f32", 7: "gemm scale+residual",
              8: "gemm_ring<LN-fold bias->f16>", 9: "gemm_ring<LN-fold bias+QuickGELU->f16>",
              10: "gemm_ring2<residual + x16 + row stats>", 11: "gemm adapter down_proj", 12: "gemm_duo<adapter up_proj>",
              13: "gemm_ring<VAE mean|log_var + reparameterise>", 14: "gemm_duo<adapter up_proj, fp16 copy only>",
              100: "attention_kernel", 101: "qkv_attn_kernel<in_proj + attention, q k v in LDS>",
              102: "vae_fused_kernel<Encoder/Generator as one kernel: M rows, N passes, K hidden>",
              103: "mlp_pair_kernel<c_fc + QuickGELU -> c_proj + residual as one launch: M rows, N hidden, K width>"}


# ----------------------------------------------------------------------------------------------------------------
def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH, help="crops per GPU per step (metric is quoted at 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the config3 / config4 objects")
    ap.add_argument("--no-class-rows", action="store_true",
                    help="skip the second timed region (the library's default last block); for kernel traces of the headline alone")
    ap.add_argument("--power", action="store_true",
                    help="with --no-extra-configs: still sample package power / clock beside the headline step (A/B runs)")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (one per GPU, like the
    reference's mp.spawn, main_tip_finetune.py:1205-1208).  Nothing in this process has touched the GPU runtime, and
    it never replaces itself: it relays rank 0's JSON line and the children's exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--batch", str(args.batch)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.no_extra_configs:
        cmd.append("--no-extra-configs")
    if args.no_class_rows:
        cmd.append("--no-class-rows")
    if args.power:
        cmd.append("--power")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in proc.stdout:
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


def host_cores() -> int:
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


# ---- CPU baselines (the oracle is the checker / baseline only; never on the timed path) ---------------------------
def cpu_baseline(budget_s: float = 16.0):
    import torch
    from hoigen_amd import synth
    from oracle import clip_oracle as co

    torch.set_num_threads(min(host_cores(), int(os.environ.get("HG_CPU_THREADS", "64"))))
    sd = co.reference_weight_rounding(synth.clip_state_dict(synth.VIT_B16, 0))
    with torch.no_grad():
        x = torch.from_numpy(synth.crops(8, 224, seed=1234))
        t0 = time.perf_counter()
        co.encode_image(sd, x)                         # warm-up, also sizes the sample
        per8 = time.perf_counter() - t0
        bs = 32 if per8 * 4 < budget_s else 8
        x = torch.from_numpy(synth.crops(bs, 224, seed=4321))
        n, t_tot = 0, 0.0
        while t_tot < budget_s / 2 and n < 3 * bs:
            t0 = time.perf_counter()
            co.encode_image(sd, x)
            t_tot += time.perf_counter() - t0
            n += bs
    return {"value": round(n / t_tot, 3), "unit": "crops/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} crops in batches of {bs} (fp32, oracle/clip_oracle.py encode_image, "
                      f"{n * FLOPS_PER_CROP / t_tot / 1e9:.0f} GFLOP/s)"}


def text_flops(L: int, width: int = 512, layers: int = 12, last_block_one_row: bool = False) -> float:
    """Algorithmic FLOPs of the text tower for one prompt of L tokens (5.960 GFLOP at L = 77)."""
    blk = 2 * L * width * 3 * width + 4 * L * L * width + 2 * L * width * width + 4 * L * width * 4 * width
    if not last_block_one_row:
        return layers * blk + 2 * width * width
    last = 2 * L * width * 2 * width + 2 * width * width + 4 * L * width + 2 * width * width + 4 * width * 4 * width
    return (layers - 1) * blk + last + 2 * width * width


def timed(fn, iters: int, warm: int = 2) -> float:
    """Median milliseconds of fn() over `iters` runs (events on the current stream)."""
    import torch
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def power_sample(step, sync, seconds: float = 2.5):
    """Package power and shader clock while the headline step runs (its own untimed burst after the timed regions): the step
    sits at the package power cap (profiles/r03_power.txt), which is what the sustained clock - and with it every MFMA
    fraction on this line - hangs on.  rocm-smi is polled from a thread; None when it is missing or prints nothing usable."""
    import re
    import shutil
    import subprocess
    import threading
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    # never under a profiler: its preloaded library would initialise the GPU inside every child process
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ):
        return None
    child_env = dict(os.environ)
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                txt = subprocess.run([exe, "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True,
                                     timeout=10, env=child_env).stdout
            except Exception:
                return
            w = re.search(r"GPU\[0\].*?Current Socket Graphics Package Power \(W\):\s*([0-9.]+)", txt)
            c = re.search(r"GPU\[0\].*?sclk clock level:.*?\((\d+)Mhz\)", txt)
            cap = re.search(r"GPU\[0\].*?Max Graphics Package Power \(W\):\s*([0-9.]+)", txt)
            if w and c:
                samples.append((float(w.group(1)), int(c.group(1)), float(cap.group(1)) if cap else None))

    th = threading.Thread(target=poll, daemon=True)
    t0 = time.perf_counter()
    th.start()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            step()
        sync()
        n += 20
    burst_s = time.perf_counter() - t0
    stop.set()
    th.join(timeout=15)
    busy = [x for x in samples if x[0] > 0.5 * max(y[0] for y in samples)] if samples else []
    if not busy:
        return None
    med = lambda v: sorted(v)[len(v) // 2]
    return {"what": "rocm-smi polled beside an untimed burst of the headline step (samples above half the maximum power)",
            "steps": n, "samples": len(busy), "burst_ms_per_step": round(burst_s / n * 1e3, 4),
            "joules_per_step": round(med([x[0] for x in busy]) * burst_s / n, 3),
            "package_watts": {"median": med([x[0] for x in busy]), "min": min(x[0] for x in busy), "max": max(x[0] for x in busy)},
            "sclk_mhz": {"median": med([x[1] for x in busy]), "min": min(x[1] for x in busy), "max": max(x[1] for x in busy)},
            "cap_watts": busy[0][2], "nominal_sclk_mhz": 2400}


def variant_c(dev, crops, ms_a: float):
    """Variant C - `visual(x, prior)` with trained instance adapters, the detector's call (CLIP_models_adapter_prior2.py:489-506,
    upt_tip_cache_model_free_finetune_distill3.py:1615) - on the same crops with 14 prior tokens (4 padded) per crop."""
    import torch
    from hoigen_amd import synth
    from hoigen_amd.model import build_model
    sd = synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))
    sd.update(synth.to_torch(synth.adapter_state_dict(synth.VIT_B16, 21)))
    m = build_model(sd, use_adapter=True, adapter_pos="all").to(dev)
    B, N = crops.shape[0], 14
    g = torch.Generator(device=dev).manual_seed(4321)
    pri = torch.randn(B, N, 64, device=dev, generator=g)
    mask = torch.zeros(B, N, dtype=torch.bool, device=dev)
    mask[:, N - 4:] = True
    for _ in range(3):
        out = m.visual(crops, (pri, mask))
    torch.cuda.synchronize(dev)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        out = m.visual(crops, (pri, mask))
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / reps * 1e3
    assert torch.isfinite(out[0]).all() and torch.isfinite(out[1]).all()
    del m
    return {"workload": f"visual(x, prior): adapters in all 12 blocks, {N} prior tokens, batch {B}; global [B,512] + local [B,512,14,14]",
            "ms": round(ms, 4), "crops_per_s": round(B / ms * 1e3, 1), "over_variant_a_all_rows": round(ms / ms_a, 4),
            "gflop_per_crop": round(35.127 + 0.1541 + 12 * 0.0494, 3)}


def config3(model, dev, with_cpu: bool):
    """BASELINE config 3: encode_text over the 600 HICO HOI prompts (fixture G0 token ids), 77 tokens."""
    import numpy as np
    import torch
    g0 = json.load(open(os.path.join(HERE, "tests", "golden", "g0_tokens.json")))
    rows = g0["hoi600"]["ids"]
    ids = np.zeros((len(rows), 77), np.int64)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = r
    ids_d = torch.from_numpy(ids).to(dev)
    T = ids.shape[0]
    Lt = int(ids.argmax(-1).max()) + 1
    out = {"workload": f"encode_text, {T} HICO HOI prompts x 77 tokens (hico_text_label.py; BASELINE.json configs[2])"}
    for name, trunc in (("full_77_tokens", False), ("truncated", True)):
        model.truncate_text = trunc
        ms = timed(lambda: model.encode_text(ids_d), 10)
        L = Lt if trunc else 77
        nominal, executed = T * text_flops(77), T * text_flops(L, last_block_one_row=True)
        out[name] = {"ms": round(ms, 4), "prompts_per_s": round(T / ms * 1e3, 1), "tokens_run": L,
                     "nominal_tflops": round(nominal / ms / 1e9, 2),
                     "frac_nominal": round(nominal / ms / 1e9 / MFMA_PEAK_TFLOPS, 4),
                     "executed_tflops": round(executed / ms / 1e9, 2),
                     "frac_executed": round(executed / ms / 1e9 / MFMA_PEAK_TFLOPS, 4)}
    # option text_ln_fold: 1 (default) = LayerNorm folded, its weight in the activation copy; the other two settings beside it -
    # 0 separate LayerNorm kernels, 2 the weight folded into fp16(W * gamma) (the fastest; spends parity margin: DESIGN.md 4)
    g3p = os.path.join(HERE, "tests", "golden", "g3_vitb16_text.npz")
    ref = torch.from_numpy(np.load(g3p)["hoi600"]).to(dev).float() if os.path.exists(g3p) else None
    rel = lambda e: round(float(((e.float() - ref).norm() / ref.norm()).item()), 6) if ref is not None else None
    worst = lambda e: (round(float(((e.float() - ref).norm(dim=1) / ref.norm(dim=1)).max().item()), 6) if ref is not None else None)
    model.truncate_text = False
    e77 = model.encode_text(ids_d)
    out["full_77_tokens"]["rel_l2_vs_reference_fixture"] = rel(e77)
    out["full_77_tokens"]["worst_prompt_rel_l2"] = worst(e77)
    out["text_ln_fold_settings"] = {}
    prev_fold = model.get_option("text_ln_fold")      # (restored below: the caller's setting, HG_TEXT_LN_FOLD or the default)
    try:
        for mode, what in ((0, "separate LayerNorm kernels"), (2, "LayerNorm weight folded into the GEMM weights")):
            model.set_option("text_ln_fold", mode)
            ms = timed(lambda: model.encode_text(ids_d), 10)
            out["text_ln_fold_settings"][str(mode)] = {"what": what, "ms": round(ms, 4), "prompts_per_s": round(T / ms * 1e3, 1),
                                                       "frac_nominal": round(T * text_flops(77) / ms / 1e9 / MFMA_PEAK_TFLOPS, 4),
                                                       "rel_l2_vs_reference_fixture": rel(model.encode_text(ids_d)),
                                                       "worst_prompt_rel_l2": worst(model.encode_text(ids_d))}
    finally:
        model.set_option("text_ln_fold", prev_fold)
    model.truncate_text = True
    if with_cpu:
        from hoigen_amd import synth
        from oracle import clip_oracle as co
        sd = co.reference_weight_rounding(synth.clip_state_dict(synth.VIT_B16, 0))
        sample = torch.from_numpy(ids[:64])
        with torch.no_grad():
            t0 = time.perf_counter()
            co.encode_text(sd, sample)
            dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(64 / dt, 2), "unit": "prompts/s", "cores": torch.get_num_threads(),
                               "kind": "port", "sample": "first 64 prompts x 77 tokens, oracle/clip_oracle.py encode_text"}
    return out


def generation(model, dev):
    """the generation loop of main_tip_finetune.py:759-824 - 100 iterations x three HICO branches (hoi / human / object,
    600 targets each): z -> Generator -> PromptLearner -> TextEncoder -> L2 -> mlp_net = 180 000 generated features - through
    hoigen_amd.generation.FeatureSampler on seeded synthetic branch weights; and the text tower's share of one step on its own."""
    import torch
    from hoigen_amd import vae
    from hoigen_amd.generation import hico_sampler
    g0 = json.load(open(os.path.join(HERE, "tests", "golden", "g0_tokens.json")))
    sampler = hico_sampler(model, g0["_classnames"])
    iters, bi = 100, sampler._auto_batch(100)
    gen = torch.Generator(device=dev).manual_seed(5)
    sampler.sample(iterations=2 * bi, generator=gen, batch_iters=bi)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    feat, tgt = sampler.sample(iterations=iters, generator=gen, batch_iters=bi)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    assert feat.shape == (iters * 1800, 512) and bool(torch.isfinite(feat).all())
    # the text tower on the prompts of one step (bi iterations x 1 800 prompts), timed alone
    prompts, toks = [], []
    lt = sampler._tokens_run()
    for name, br in sampler.branches.items():
        t = br.target.to(dev).repeat(bi)
        prompts.append(br.prompt_learner(br.generator(torch.randn(len(t), 512, device=dev, generator=gen)), t, tokens=lt))
        toks.append(br.prompt_learner.tokenized_prompts[t])
    prompts, toks = torch.cat(prompts, dim=0), torch.cat(toks, dim=0)
    Lt = int(toks.argmax(-1).max()) + 1
    ms_text = timed(lambda: sampler.text_encoder(prompts, toks), 5)
    executed = prompts.shape[0] * text_flops(Lt, last_block_one_row=True)
    nominal = prompts.shape[0] * text_flops(77)
    # the same loop with the LayerNorm weight folded into the GEMM weights (option text_ln_fold = 2: faster, spends parity margin, see config3)
    prev_fold = model.get_option("text_ln_fold")
    model.set_option("text_ln_fold", 2)
    try:
        sampler.sample(iterations=bi, generator=gen, batch_iters=bi)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        feat2, _ = sampler.sample(iterations=iters, generator=gen, batch_iters=bi)
        torch.cuda.synchronize(dev)
        dt2 = time.perf_counter() - t0
        ms_text2 = timed(lambda: sampler.text_encoder(prompts, toks), 5)
        assert bool(torch.isfinite(feat2).all())
    finally:
        model.set_option("text_ln_fold", prev_fold)
    fold = {"option": "text_ln_fold = 2 (not the default)", "ms_per_iteration": round(dt2 / iters * 1e3, 4),
            "features_per_s": round(feat2.shape[0] / dt2, 0), "text_tower_ms": round(ms_text2, 4),
            "text_tower_frac_executed": round(executed / ms_text2 / 1e9 / MFMA_PEAK_TFLOPS, 4)}
    return {"with_text_ln_fold_2": fold, "workload": "generation loop of main_tip_finetune.py:759-824: 100 iterations x (hoi, human, object) x 600 targets -> 180 000 "
                        "features [z -> Generator -> PromptLearner -> TextEncoder -> L2 -> mlp_net], seeded synthetic branch weights, "
                        f"{bi} iterations per pass through the kernels",
            "iterations": iters, "features": int(feat.shape[0]), "total_ms": round(dt * 1e3, 2),
            "ms_per_iteration": round(dt / iters * 1e3, 4), "features_per_s": round(feat.shape[0] / dt, 0),
            "text_tower": {"prompts": int(prompts.shape[0]), "tokens_run": Lt, "ms": round(ms_text, 4),
                           "executed_tflops": round(executed / ms_text / 1e9, 2),
                           "frac_executed": round(executed / ms_text / 1e9 / MFMA_PEAK_TFLOPS, 4),
                           "nominal_tflops_at_77_tokens": round(nominal / ms_text / 1e9, 2)}}


def config4(dev, with_cpu: bool):
    """BASELINE config 4: CoOp-VAE (Encoder -> reparameterise -> Generator) on 100 000 rows x 512."""
    import torch
    from hoigen_amd import _lib, synth, vae
    R = 100_000
    se, sg = synth.encoder_state_dict(2), synth.generator_state_dict(3)
    E, G = vae.Encoder().to(dev), vae.Generator().to(dev)
    E.load_state_dict(synth.to_torch(se))
    G.load_state_dict(synth.to_torch(sg))
    V = vae.VAE(E, G)
    gen = torch.Generator(device=dev).manual_seed(44)
    x = vae.l2_normalize(torch.randn(R, 512, device=dev, generator=gen))
    eps = torch.randn(R, 512, device=dev, generator=gen)
    z = torch.randn(R, 512, device=dev, generator=gen)
    ms_full = timed(lambda: V(x, eps), 10)
    ms_gen = timed(lambda: G(z), 10)
    # the three dispatches of option vae_fused (default 1: Encoder on the GEMM path, Generator of the whole rounds of items as ONE kernel)
    modes = {}
    prev_vf = vae.get_option("vae_fused", dev)
    try:
        for m_, nm in ((0, "gemm_path"), (2, "one_kernel_every_row")):
            vae.set_option("vae_fused", m_, dev)
            modes[nm] = {"ms": round(timed(lambda: V(x, eps), 10), 4), "generator_only_ms": round(timed(lambda: G(z), 10), 4)}
    finally:
        vae.set_option("vae_fused", prev_vf if prev_vf is not None else 1, dev)
    _, recs = _lib.profile(V._slot.get(dev)[1], _lib.HG_PROF_ALL, 64, lambda: V(x, eps))
    out = {"workload": f"CoOp-VAE Encoder->reparameterise->Generator, {R} rows x 512 (BASELINE.json configs[3]); "
                       "inputs and the four outputs (mean, log_var, z, bias) fp32 in HBM",
           "ms": round(ms_full, 4), "rows_per_s": round(R / ms_full * 1e3, 0),
           "tflops": round(R * 14.680e6 / ms_full / 1e9, 2),
           "frac": round(R * 14.680e6 / ms_full / 1e9 / MFMA_PEAK_TFLOPS, 4),
           "generator_only": {"ms": round(ms_gen, 4), "tflops": round(R * 8.389e6 / ms_gen / 1e9, 2),
                              "frac": round(R * 8.389e6 / ms_gen / 1e9 / MFMA_PEAK_TFLOPS, 4)},
           "vae_fused_option": {"default_1": {"ms": round(ms_full, 4), "generator_only_ms": round(ms_gen, 4)}, **modes},
           "launches_per_call": len(recs),
           "kernels": [{"kind": KIND_NAMES.get(k, str(k)), "M": m, "N": n, "K": kk, "ms": round(t, 4)}
                       for k, m, n, kk, t in recs]}
    if with_cpu:
        from oracle import clip_oracle as co, vae_oracle as vo
        n = 8192
        xs, es = x[:n].cpu(), eps[:n].cpu()
        with torch.no_grad():
            t0 = time.perf_counter()
            vo.vae_forward(co.as_tensors(se), co.as_tensors(sg), xs, es)
            dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(n / dt, 1), "unit": "rows/s", "cores": torch.get_num_threads(),
                               "kind": "port", "sample": f"{n} rows, oracle/vae_oracle.py vae_forward (fp32)"}
    return out


# ---- per-kernel accounting ---------------------------------------------------------------------------------------
RLN_STREAM_BYTES = 10.0      # algorithmic stream bytes per element of a LayerNorm-emitting residual GEMM (set in run())


def kernel_row(kind, M, N, K):
    """(name, algorithmic FLOPs, algorithmic HBM bytes) of one launch."""
    if kind == 100:                       # attention: M sequences of N tokens, K heads of 64
        rows, D = M * N, K * 64
        return f"attention (L={N}, {K} heads)", 4.0 * M * K * N * N * 64, rows * 3 * D * 2 + rows * D * 2
    if kind == 101:                       # in_proj + attention in one kernel (q, k, v stay in LDS): x16 in, W once, att out
        rows, D = M * N, K * 64
        return (f"in_proj + attention fused (L={N}, {K} heads)", 2.0 * rows * 3 * D * D + 4.0 * M * K * N * N * 64,
                rows * D * 2 + 3 * D * D * 2 + rows * D * 2)
    if kind == 103:                       # the MLP as one launch: x16 in, both weights once, fc written and read once, the stream in and out
        return (f"c_fc + QuickGELU -> c_proj + residual, one launch (hidden {N}, width {K})", 2 * 2.0 * M * N * K,
                M * K * 2 + 2 * N * K * 2 + 2 * M * N * 2 + M * K * RLN_STREAM_BYTES)
    fl = 2.0 * M * N * K
    w = N * K * 2
    if kind in (0, 8):
        return f"in_proj / QKV (N={N}, K={K})", fl, M * K * 2 + w + M * N * 2
    if kind in (1, 9):
        return f"c_fc + QuickGELU (N={N}, K={K})", fl, M * K * 2 + w + M * N * 2
    if kind in (3, 10):
        # fp32 stream: read + write 4 B each (+ the 2-byte centred copy for the next LayerNorm-folded GEMM); stream held as
        # centre + hi + lo (option stream_hilo): hi IS the copy - 3 B in, 3 B out with lo as bf8 (RLN_STREAM_BYTES: mean over the step's launches)
        stream = RLN_STREAM_BYTES if kind == 10 else 8.0
        nm = "out_proj" if K == N else "c_proj"
        return f"{nm} + residual (N={N}, K={K})", fl, M * K * 2 + w + M * N * stream
    if kind == 5:
        return f"patch embedding (K={K})", fl, M * K * 2 + w + M * N * 4
    return f"{KIND_NAMES.get(kind, kind)} (M={M}, N={N}, K={K})", fl, M * K * 2 + w + M * N * 4


def aggregate(recs, steps):
    rows = {}
    for kind, M, N, K, ms in recs:
        name, fl, by = kernel_row(kind, M, N, K)
        r = rows.setdefault((kind, M, N, K), {"name": name, "kernel": KIND_NAMES.get(kind, str(kind)), "kind": kind,
                                              "n": 0, "ms": 0.0, "flops": fl, "hbm_bytes": by})
        r["n"] += 1
        r["ms"] += ms
    out = []
    for r in rows.values():
        avg = r["ms"] / r["n"]
        out.append({"name": r["name"], "kernel": r["kernel"], "kind": r["kind"], "launches_per_step": r["n"] / steps,
                    "avg_ms": round(avg, 4), "ms_per_step": round(r["ms"] / steps, 4),
                    "gflop": round(r["flops"] / 1e9, 2), "tflops": round(r["flops"] / avg / 1e9, 1),
                    "frac": round(r["flops"] / avg / 1e9 / MFMA_PEAK_TFLOPS, 4),
                    "hbm_bytes": int(r["hbm_bytes"]), "hbm_gbs": round(r["hbm_bytes"] / avg / 1e6, 0),
                    "hbm_frac": round(r["hbm_bytes"] / avg / 1e6 / HBM_PEAK_GBS, 4)})
    out.sort(key=lambda r: -r["ms_per_step"])
    return out


# ----------------------------------------------------------------------------------------------------------------
def run(args):
    import torch
    torch.set_grad_enabled(False)      # inference only (the façade refuses gradient callers)
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # HG_BENCH_FORCE_COMM=1 (tests): one rank, but the RCCL process group, the side-stream event chain and the in-place
    # all-gather run exactly as with N ranks - the only way to execute that leg on a 1-GPU box
    force_comm = world == 1 and os.environ.get("HG_BENCH_FORCE_COMM", "0") == "1" and "RANK" in os.environ
    if world > 1 or force_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL

    from hoigen_amd import _lib, synth
    from hoigen_amd.distributed import ShardedEncoder
    from hoigen_amd.model import build_model

    model = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    crops = torch.randn(args.batch, 3, 224, 224, device=dev, generator=gen)     # resident in HBM
    sharded = (ShardedEncoder(model.visual.encode_into, args.batch, 512, dev, force_comm=force_comm)
               if (world > 1 or force_comm) else None)

    def step():
        if sharded is not None:
            return sharded.step(crops)[0]                # encode into this rank's slice + all-gather on the side stream
        return model.visual(crops)                       # [B,512] in model.dtype (fp16, like the reference on GPU)

    def fence():
        if sharded is not None:
            sharded.finish()
        torch.cuda.synchronize(dev)
        if world > 1 or force_comm:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed_region(profile_kind=None, profile_steps=0, launches_per_step=0):
        """Exactly args.steps steps between two fences -> (seconds, per-step ms list, live kernel records, last out)."""
        h = model.visual._ctx.handle
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        recs = []
        fence()
        if profile_kind is not None:
            _lib.lib().hg_profile_begin(h, profile_kind, profile_steps * launches_per_step)
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(args.steps):
            out = step()
            ev[i + 1].record()
        fence()
        dt = time.perf_counter() - t0
        if profile_kind is not None:
            import ctypes as C
            buf = (_lib.hg_prof_rec * max(1, profile_steps * launches_per_step))()
            n = C.c_int32()
            _lib.lib().hg_profile_end(h, buf, profile_steps * launches_per_step, C.byref(n))
            recs = [(r.kind, r.M, r.N, r.K, float(r.ms)) for r in buf[: n.value]]
        per_step = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
        return dt, per_step, recs, out

    prev_row0 = model.visual.get_option("last_block_row0")
    model.visual.set_option("last_block_row0", 0)       # headline: every row of every block (hg_set_option on this context)
    for _ in range(args.warmup):
        step()
    global RLN_STREAM_BYTES
    hilo = ctypes.c_int32(0)
    _lib.lib().hg_get_option(model.visual._ctx.handle, b"stream_hilo", ctypes.byref(hilo))
    if hilo.value and args.batch * 197 >= 512:
        # 23 such launches per all-rows step, lo = b bytes (bf8: 1, the default build; fp16: 2): 21 read and write hi + lo (4 + 2b), the first
        # reads fp32 and writes hi + lo (6 + b), the last reads hi + lo and writes fp32 + copy (8 + b)
        lo_bits = ctypes.c_int32(16)
        _lib.lib().hg_get_option(model.visual._ctx.handle, b"stream_lo_bits", ctypes.byref(lo_bits))
        b = lo_bits.value / 8.0
        RLN_STREAM_BYTES = (21 * (4 + 2 * b) + (6 + b) + (8 + b)) / 23
    # two more untimed steps with every GEMM / attention launch bracketed by events: the per-kernel table, and which
    # kernel the live `roofline` measurement of the timed region follows
    PROF_STEPS = 4
    _, all_recs = _lib.profile(model.visual._ctx.handle, _lib.HG_PROF_ALL, PROF_STEPS * 128,
                               lambda: [step() for _ in range(PROF_STEPS)])
    kernels = aggregate(all_recs, PROF_STEPS)
    by_kind = {}
    for k in kernels:
        by_kind[k["kind"]] = by_kind.get(k["kind"], 0.0) + k["ms_per_step"]
    dom_kind = max(by_kind, key=by_kind.get)
    dom_launches = int(round(sum(k["launches_per_step"] for k in kernels if k["kind"] == dom_kind)))

    # `roofline`: the dominant kernel's launches of those untimed steps (hipEvent pairs on the stream it is launched on, this process,
    # right before the timed region) - the timed region itself carries no instrumentation
    live = [r for r in all_recs if r[0] == dom_kind]
    dt, per_step, _, out = timed_region()
    assert torch.isfinite(out).all()
    out = out.clone()                                    # (the gather buffers are reused by the next steps)

    # the library's default path (last block on the class-token rows only), same protocol
    if args.no_class_rows:
        dt2, rel = dt, 0.0
    else:
        model.visual.set_option("last_block_row0", prev_row0 if prev_row0 is not None else 1)
        for _ in range(args.warmup):
            step()
        dt2, per_step2, _, out2 = timed_region()
        assert torch.isfinite(out2).all()
        rel = float(((out2.float() - out.float()).norm() / out.float().norm()).item())

    # parity beside the speed: the reference's own fixture (4 seeded crops, tests/golden/g2_vitb16_image.npz), whole matrix and worst row
    parity = None
    g2p = os.path.join(HERE, "tests", "golden", "g2_vitb16_image.npz")
    if rank == 0 and os.path.exists(g2p):
        import numpy as np
        ref = torch.from_numpy(np.load(g2p)["encode_image"]).to(dev).float()
        gold = torch.from_numpy(synth.crops(4, 224, seed=1234)).to(dev)
        def _rel(o):
            d = (o.float() - ref)
            return {"rel_l2": float(f"{(d.norm() / ref.norm()).item():.3e}"),
                    "worst_row_rel_l2": float(f"{(d.norm(dim=1) / ref.norm(dim=1)).max().item():.3e}")}
        parity = {"what": "encode_image of the reference's 4 fixture crops vs the reference's CPU fp32 output (tolerance 1e-3)"}
        cur = model.visual.get_option("last_block_row0")
        for name, mode in (("all_rows", 0), ("class_rows_only", 1)):
            model.visual.set_option("last_block_row0", mode)
            parity[name] = _rel(model.visual.forward_trace(gold)[0])
        model.visual.set_option("last_block_row0", cur)

    t = torch.tensor([dt, dt2], device=dev, dtype=torch.float64)
    if world > 1 or force_comm:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, dt2 = float(t[0].item()), float(t[1].item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * args.batch * args.steps / dt
        flops_l = sum(kernel_row(*r[:4])[1] for r in live) / max(1, len(live))
        bytes_l = sum(kernel_row(*r[:4])[2] for r in live) / max(1, len(live))
        ms_l = sum(r[4] for r in live) / max(1, len(live))
        ach = flops_l / ms_l / 1e9 if ms_l > 0 else 0.0
        traffic = step_traffic = traffic_source = None
        tpath = os.path.join(HERE, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(f"kind{dom_kind}_hbm_bytes_per_launch")
                # HBM-side bytes of the whole step (FETCH_SIZE x 2 + WRITE_SIZE over every kernel of one all-rows step), from the
                # same PMC passes (tools/gpu_energy_ab.sh); the default configuration's entry
                st = tj.get("step_total_MB", {})
                key = next((k for k in st if "(default)" in k and "stagger" in k), next((k for k in st if "(default)" in k), None))
                step_traffic = int(st[key] * 1e6) if key else None
                # these two are CONSTANTS read from the committed file (separate rocprofv3 --pmc passes), not counters of this run
                traffic_source = (f"profiles/traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of round {tj.get('round', '?')}"
                                  f" on the library at commit {tj.get('commit', 'unrecorded')}; file constants, NOT counters of this run")
            except Exception:
                traffic = step_traffic = traffic_source = None
        ps = sorted(per_step)
        pct = lambda q: round(ps[min(len(ps) - 1, int(q * len(ps)))], 4)
        e2e = value / world * FLOPS_PER_CROP / 1e12
        shapes = sorted({(r[1], r[2], r[3]) for r in live})
        line = {
            "metric": "union-crop embeddings/sec ViT-B/16 bs=256",
            "value": round(value, 2), "unit": "crops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "CLIP ViT-B/16 union-region encode (encode_image), 224x224 crops, "
                                   f"batch {args.batch} per GPU, synthetic N(0,1) crops + seeded synthetic weights "
                                   "(BASELINE.json configs[1]" + ("; configs[4] at 8 GPUs: 2048 crops per step" if world > 1 else "") + ")",
                       "last_block": "all rows (option last_block_row0 = 0)",
                       "flops_per_crop_executed": round(FLOPS_PER_CROP / 1e9, 3),
                       "batch_per_gpu": args.batch, "global_batch": world * args.batch,
                       "parallelism": f"dp{world}" + (f" + in-place all_gather[{world}x{args.batch}x512 f32] per step on a side stream" if (world > 1 or force_comm) else "")},
            "step_ms": {"median": pct(0.5), "p10": pct(0.1), "p90": pct(0.9), "how": "hipEvents on the compute stream around every timed step"},
            "roofline": {"bound": "mfma",
                         "kernel": f"{KIND_NAMES.get(dom_kind, dom_kind)}: the kernel with the largest share of the step "
                                   f"({by_kind[dom_kind]:.3f} of {sum(by_kind.values()):.3f} ms of GEMM+attention time), "
                                   f"{dom_launches} launches per step, shapes (M,N,K) {shapes}",
                         "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "step_traffic": step_traffic,
                         "traffic_source": traffic_source,
                         "avg_kernel_ms": round(ms_l, 4), "avg_gflop_per_launch": round(flops_l / 1e9, 2),
                         "launches_timed": len(live),
                         "how": f"hipEvent pairs around every launch of this kernel in {PROF_STEPS} untimed steps run right before the "
                                "timed region (same process, same stream); the timed region carries no events but one per step",
                         "algorithmic_hbm_bytes_per_launch": int(bytes_l),
                         "hbm_frac_algorithmic": round(bytes_l / ms_l / 1e6 / HBM_PEAK_GBS, 4) if ms_l > 0 else None,
                         "e2e_tflops": round(e2e, 2), "e2e_frac": round(e2e / MFMA_PEAK_TFLOPS, 4),
                         "target_e2e_frac": 0.40},
            "kernels": [{k: v for k, v in r.items() if k != "kind"} for r in kernels],
        }
        if parity is not None:
            line["parity"] = parity
        v2 = world * args.batch * args.steps / dt2
        f2 = FLOPS_PER_CROP - FLOPS_DEAD_ROWS
        if not args.no_class_rows:
            line["class_rows_only"] = {
                "what": "library default: last block = K/V for all rows, then attention / out-proj / MLP for the class-token "
                        "row only (rows that cannot reach the embedding are not computed)",
                "value": round(v2, 2), "unit": "crops/s", "ms_per_step": round(dt2 / args.steps * 1e3, 4),
                "flops_per_crop_executed": round(f2 / 1e9, 3),
                "e2e_frac": round(v2 / world * f2 / 1e12 / MFMA_PEAK_TFLOPS, 4),
                "rel_l2_vs_all_rows": float(f"{rel:.3e}")}
        if world == 1:
            if (not args.no_extra_configs or args.power) and not force_comm:
                model.visual.set_option("last_block_row0", 0)
                try:
                    pw = power_sample(step, lambda: torch.cuda.synchronize(dev))
                finally:
                    model.visual.set_option("last_block_row0", 1)
                if pw is not None:
                    pw["mfma_peak_at_sustained_clock_tflops"] = round(MFMA_PEAK_TFLOPS * pw["sclk_mhz"]["median"] / 2400.0, 1)
                    pw["e2e_frac_of_that"] = round(line["roofline"]["e2e_tflops"] / pw["mfma_peak_at_sustained_clock_tflops"], 4)
                    line["power"] = pw
            with_cpu = not args.no_cpu_baseline
            if with_cpu:
                line["cpu_baseline"] = cpu_baseline()
            if not args.no_extra_configs:
                line["variant_c"] = variant_c(dev, crops, ms_per_step)      # (its own model; variant C always runs every row)
                line["config3"] = config3(model, dev, with_cpu)
                line["config4"] = config4(dev, with_cpu)
                line["generation"] = generation(model, dev)
        print(json.dumps(line), flush=True)
    if world > 1 or force_comm:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None:
        if args.gpus > 1:                     # decided from argv / env alone, before anything touches the GPU runtime
            sys.exit(self_launch(args))
    elif int(world_env) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world_env}; start it with "
                         f"--nproc-per-node {args.gpus} (or without a launcher: bench.py starts the ranks itself)")
    run(args)


if __name__ == "__main__":
    main()
