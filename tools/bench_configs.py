#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs 3 and 4 (bench.py carries the headline config 2):
   config 3  encode_text over the 600 HICO prompts (77 tokens), with and without causal truncation
   config 4  CoOp-VAE: Encoder -> reparameterise -> Generator on 100k rows (and Generator-only sampling)
Prints one JSON line per measurement.  Needs a HIP device."""
import json
import os
import sys
import time

import torch
torch.set_grad_enabled(False)   # inference only

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoigen_amd import clip, synth, vae  # noqa: E402
from hoigen_amd.model import build_model  # noqa: E402

PEAK = 2516.6


def timeit(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    dev = torch.device("cuda:0")
    g0 = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g0_tokens.json")))
    model = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(dev)
    ids = clip.tokenize(g0["hoi600"]["text"]).to(dev)
    for trunc in (False, True):
        model.truncate_text = trunc
        dt = timeit(lambda: model.encode_text(ids))
        L = 77 if not trunc else int(ids.argmax(-1).max()) + 1
        nominal = 600 * 5.960e9
        print(json.dumps({"config": 3, "what": f"encode_text 600 HICO prompts, L_exec={L}", "ms": round(dt * 1e3, 3),
                          "prompts_per_s": round(600 / dt, 1), "nominal_tflops_at_L77": round(nominal / dt / 1e12, 1),
                          "frac_of_mfma_peak_nominal": round(nominal / dt / 1e12 / PEAK, 4)}))
    E, G = vae.Encoder().to(dev), vae.Generator().to(dev)
    E.load_state_dict(synth.to_torch(synth.encoder_state_dict(2)))
    G.load_state_dict(synth.to_torch(synth.generator_state_dict(3)))
    fused = vae.VAE(E, G)
    R = 100_000
    x = torch.nn.functional.normalize(torch.randn(R, 512, device=dev), dim=-1)
    eps = torch.randn(R, 512, device=dev)
    dt = timeit(lambda: fused(x, eps), reps=5)
    print(json.dumps({"config": 4, "what": "VAE enc+reparam+gen, 100k rows", "ms": round(dt * 1e3, 3),
                      "rows_per_s": round(R / dt, 0), "tflops": round(R * 14.68e6 / dt / 1e12, 1),
                      "frac_of_mfma_peak": round(R * 14.68e6 / dt / 1e12 / PEAK, 4)}))
    z = torch.randn(R, 512, device=dev)
    dt = timeit(lambda: G(z), reps=5)
    print(json.dumps({"config": 4, "what": "Generator only (sampling), 100k rows", "ms": round(dt * 1e3, 3),
                      "rows_per_s": round(R / dt, 0), "tflops": round(R * 8.389e6 / dt / 1e12, 1)}))


if __name__ == "__main__":
    main()
