#!/usr/bin/env python3
"""Copy the summaries of tools/gpu_r6_collect.sh (gpurun_out/r06/) into profiles/ as r06_<tag>_* and rewrite profiles/traffic.json from
its FETCH_SIZE / WRITE_SIZE passes.  usage: r6_profiles.py <tag> (e.g. v1); the library's commit is read from git here."""
import collections, csv, glob, json, os, re, shutil, subprocess, sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, "gpurun_out", "r06")
tag = sys.argv[1] if len(sys.argv) > 1 else "v1"
P = os.path.join(R, "profiles")
for src, dst in (("headline_allrows_kernel_stats.csv", f"r06_{tag}_headline_allrows_kernel_stats.csv"),
                 ("headline_allrows_bench_line.json", f"r06_{tag}_headline_allrows_bench_line.json"),
                 ("bench_kernel_stats.csv", f"r06_{tag}_bench_kernel_stats.csv"), ("pmc_summary.txt", f"r06_{tag}_pmc_summary.txt")):
    if os.path.exists(os.path.join(O, src)):
        shutil.copy(os.path.join(O, src), os.path.join(P, dst))
line = [l for l in open(os.path.join(O, "bench_final.log")) if l.startswith("{")][-1]
open(os.path.join(P, f"r06_{tag}_bench_line.json"), "w").write(line)
pl = [l for l in open(os.path.join(O, "bench_power.log")) if l.startswith("{")]
power = json.loads(pl[-1]).get("power") if pl else None

def load(d, counter):
    fs = glob.glob(os.path.join(O, d) + "/**/*counter_collection.csv", recursive=True)
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):      # (gpurun merges: older passes may still lie beside the newest)
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(.*", "", re.sub(r"^void hg::", "", r["Kernel_Name"]))[:48]
        agg[k] += float(r["Counter_Value"]) * 1024
        cnt[k] += 1
    return agg, cnt

STEPS = 7      # bench.py --steps 2 --warmup 1 + its 4 profiled steps
f, fc = load("pmc_FETCH_SIZE", "FETCH_SIZE")
w, wc = load("pmc_WRITE_SIZE", "WRITE_SIZE")
per = {k: [round(2 * f.get(k, 0) / max(fc.get(k, 1), 1) / 1e6, 1), round(w.get(k, 0) / max(wc.get(k, 1), 1) / 1e6, 1), round(max(fc.get(k, 0), wc.get(k, 0)) / STEPS, 1)]
       for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0)))[:14]}
pair = next((k for k in per if k.startswith("mlp_pair_kernel")), None)
commit = subprocess.run(["git", "log", "-1", "--format=%h %s"], cwd=R, capture_output=True, text=True).stdout.strip()
M, D = 50432, 768
alg = M * D * 2 + 2 * 4 * D * D * 2 + 2 * M * 4 * D * 2 + 2 * M * D * 3      # x16 in, both weights, fc out + in, hi + lo (fp16 + bf8) in and out
tj = {
    "kernel": "mlp_pair_kernel<2> (c_fc + QuickGELU -> c_proj + residual of a block as one launch, M = 50432, hidden 3072, width 768, stream held as "
              "centre + hi (fp16) + lo (bf8)): the kernel with the largest share of the all-rows step, 11 launches per step",
    "round": 6,
    "commit": commit + " (library of the PMC passes: tools/gpu_r6_collect.sh; profiles/r06_%s_pmc_summary.txt)" % tag,
    "fetch_bytes_per_launch_corrected": int(per[pair][0] * 1e6) if pair else None,
    "write_bytes_per_launch": int(per[pair][1] * 1e6) if pair else None,
    "kind103_hbm_bytes_per_launch": int((per[pair][0] + per[pair][1]) * 1e6) if pair else None,
    "algorithmic_bytes_per_launch_mean": alg,
    "ratio": round((per[pair][0] + per[pair][1]) * 1e6 / alg, 2) if pair else None,
    "per_kernel_per_launch_MB (fetch x2 corrected / write / launches per step)": per,
    "step_total_MB": {"round 6 (default): mlp_pair on, stream hi / lo (bf8), fused in_proj + attention, stagger per XCD":
                      round((sum(2 * v for v in f.values()) + sum(w.values())) / STEPS / 1e6, 0)},
    "power_beside_the_headline": power,
    "how": "FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs "
           "--no-class-rows` (7 all-rows steps per pass); FETCH_SIZE x 2 (gfx950 counts 128-byte requests as 64 B: MI355X_MICROARCH.md, HBM section); "
           "both counters in KiB.  Earlier rounds' passes: profiles/r05_v6_pmc_summary.txt (29.01 GB per step, gemm_ring2<10, hl> 626 MB per launch "
           "against 431 algorithmic), profiles/r04_pmc_summary.txt.",
}
json.dump(tj, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print(json.dumps({k: tj[k] for k in ("kind103_hbm_bytes_per_launch", "algorithmic_bytes_per_launch_mean", "ratio", "step_total_MB")}, indent=1))
