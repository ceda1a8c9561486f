#!/usr/bin/env python3
"""Headline benchmark: union-crop embeddings/sec, CLIP ViT-B/16, batch 256 per GPU (BASELINE.json
configs[1]); weak scaling over N GPUs with one RCCL all-gather of the [256,512] embeddings per step.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of ``encode_image`` (hg_encode_image through the façade) over one batch of 256
synthetic N(0,1) crops already resident in HBM.  Rank 0 prints ONE JSON line.  Weights are the seeded
synthetic ViT-B/16 of hoigen_amd.synth (no checkpoint is reachable offline).

Extra objects on the line:
  roofline      dominant kernel (the c_fc GEMM: M=50432, N=3072, K=768) timed live with hipEvent pairs on
                its own stream during the timed region; achieved = 2*M*N*K / mean duration vs the dense
                16-bit MFMA peak 2516.6 TFLOP/s.  `e2e_frac` = whole-step algorithmic FLOPs / step time.
  cpu_baseline  the CPU oracle (oracle/clip_oracle.py, a port of the reference's CPU path pinned to the
                reference's own outputs) timed on this box's host cores on a bounded sample (rank 0, N=1).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

MFMA_PEAK_TFLOPS = 2516.6          # 256 CU x 4096 FLOP/clk/CU x 2.4 GHz (dense fp16/bf16), BASELINE.md §2
FLOPS_PER_CROP = 35.127e9          # BASELINE.md §2 (every row of every block)
# The library's default runs the LAST block's attention, out-proj and MLP on the class-token row only (its other 196
# rows never reach the embedding).  The headline `value` is measured with HG_LAST_BLOCK_ROW0=0, i.e. with every row
# of every block computed like the reference does; the default path is timed right after it and reported in the
# extra object `class_rows_only` with the FLOPs it really executes (per skipped row: the Q and out projections,
# c_fc + c_proj, and its attention row over 197 keys).
FLOPS_DEAD_ROWS = 196 * (2 * 2 * 768 * 768 + 2 * 2 * 768 * 3072 + 4 * 197 * 768)
BATCH = 256
GEMM_CLASS_FC = 1                  # EPI_BIAS_QGELU_F16: the c_fc GEMM (M=B*197, N=3072, K=768)


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


def cpu_baseline(budget_s: float = 20.0):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="crops per GPU per step (metric is quoted at 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL

    from hoigen_amd import _lib, synth
    from hoigen_amd.model import build_model

    model = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    crops = torch.randn(args.batch, 3, 224, 224, device=dev, generator=gen)     # resident in HBM
    gathered = torch.empty(world * args.batch, 512, device=dev, dtype=torch.float32) if world > 1 else None

    def step():
        emb = model.visual(crops)                        # [B,512] in model.dtype (fp16, like the reference on GPU)
        if world > 1:
            dist.all_gather_into_tensor(gathered, emb.float())
            return gathered
        return emb

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    os.environ["HG_LAST_BLOCK_ROW0"] = "0"       # headline: every row of every block (read per call by the library)
    for _ in range(args.warmup):
        step()
    h = model.visual._ctx.handle
    lib = _lib.lib()
    n_layers = model.visual.transformer.layers
    # hipEvent pairs around every c_fc launch of the first (up to) 4 timed steps: an event record costs ~5 us on the
    # stream, so timing all launches would inflate ms_per_step by ~1 %
    lib.hg_profile_begin(h, GEMM_CLASS_FC, min(args.steps, 4) * n_layers)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    avg_ms, launches, flops = C.c_double(), C.c_int32(), C.c_double()
    mnk = (C.c_int32 * 3)()
    lib.hg_profile_end(h, C.byref(avg_ms), C.byref(launches), C.byref(flops), mnk)
    assert torch.isfinite(out).all()
    out = out.clone()                                    # (`gathered` is reused by the next steps)

    # the library's default path (last block on the class-token rows only), same protocol
    os.environ["HG_LAST_BLOCK_ROW0"] = "1"
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out2 = step()
    fence()
    dt2 = time.perf_counter() - t0
    assert torch.isfinite(out2).all()
    rel = float(((out2.float() - out.float()).norm() / out.float().norm()).item())

    t = torch.tensor([dt, dt2], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, dt2 = float(t[0].item()), float(t[1].item())

    if rank == 0:
        flops_per_crop = FLOPS_PER_CROP
        ms_per_step = dt / args.steps * 1e3
        value = world * args.batch * args.steps / dt
        ach = flops.value / (avg_ms.value * 1e-3) / 1e12 if avg_ms.value > 0 else 0.0
        traffic = None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("c_fc_gemm_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "union-crop embeddings/sec ViT-B/16 bs=256",
            "value": round(value, 2), "unit": "crops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "CLIP ViT-B/16 union-region encode (encode_image), 224x224 crops, "
                                   f"batch {args.batch} per GPU, synthetic N(0,1) crops + seeded synthetic weights "
                                   "(BASELINE.json configs[1])",
                       "last_block": "all rows (HG_LAST_BLOCK_ROW0=0)",
                       "flops_per_crop_executed": round(flops_per_crop / 1e9, 3),
                       "batch_per_gpu": args.batch, "global_batch": world * args.batch,
                       "parallelism": f"dp{world}" + (" + all_gather[256x512 f32]/step" if world > 1 else "")},
            "roofline": {"bound": "mfma", "kernel": f"gemm_ring (c_fc: M={mnk[0]} N={mnk[1]} K={mnk[2]}, LayerNorm-folded bias+QuickGELU->f16)",
                         "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "avg_kernel_ms": round(avg_ms.value, 4), "launches_timed": launches.value,
                         "e2e_tflops": round(value / world * flops_per_crop / 1e12, 2),
                         "e2e_frac": round(value / world * flops_per_crop / 1e12 / MFMA_PEAK_TFLOPS, 4)},
        }
        v2 = world * args.batch * args.steps / dt2
        f2 = FLOPS_PER_CROP - FLOPS_DEAD_ROWS
        line["class_rows_only"] = {
            "what": "library default: last block = K/V for all rows, then attention / out-proj / MLP for the class-token "
                    "row only (rows that cannot reach the embedding are not computed)",
            "value": round(v2, 2), "unit": "crops/s", "ms_per_step": round(dt2 / args.steps * 1e3, 4),
            "flops_per_crop_executed": round(f2 / 1e9, 3),
            "e2e_frac": round(v2 / world * f2 / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "rel_l2_vs_all_rows": float(f"{rel:.3e}")}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
