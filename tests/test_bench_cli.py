"""bench.py's launch logic, checked without a GPU: `--gpus N` with no launcher must start N child ranks itself
(decided before anything touches the GPU runtime) and relay their exit code; a WORLD_SIZE that disagrees with
--gpus is an error, not silently accepted (ADVICE r1 bench.py:95)."""
import os
import subprocess
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and "--gpus 4" in r.stderr


def test_gpus_n_without_launcher_starts_child_ranks():
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without GPUs: here the children would run the real benchmark")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    # the children (one per requested GPU) were started through torch.distributed.run and each refused to run
    # without a HIP device; the parent relayed the failure instead of swallowing it
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a HIP device") >= 1
    assert "needs torch.distributed.run" not in r.stderr


import json  # noqa: E402

import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_line_schema_on_gpu():
    """One short run of the real benchmark: the JSON line carries the contract's fields, the roofline of the dominant
    kernel, the per-kernel table and the two extra configurations."""
    r = _run(["--steps", "3", "--warmup", "1", "--no-cpu-baseline"], timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "kernels", "step_ms", "config3", "config4"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "crops/s" and d["dtype"] == "f16" and d["vs_baseline"] is None
    assert abs(d["value"] - 256 * 1e3 / d["ms_per_step"]) < 1e-3 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and 0 < rf["e2e_frac"] < 1 and rf["launches_timed"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # the dominant kernel is the one with the largest time share in the table, and the table accounts for the step
    by_kernel = {}
    for k in d["kernels"]:
        by_kernel[k["kernel"]] = by_kernel.get(k["kernel"], 0.0) + k["ms_per_step"]
    top = max(by_kernel, key=by_kernel.get)
    assert rf["kernel"].startswith(top)
    # (the table is timed launch by launch on its own steps: with 3 timed steps on a cold box it can exceed the step by a few per cent)
    assert 0.8 * d["ms_per_step"] < sum(by_kernel.values()) < 1.12 * d["ms_per_step"]
    assert d["config4"]["ms"] > 0 and d["config3"]["truncated"]["ms"] < d["config3"]["full_77_tokens"]["ms"]
    assert 1.0 < d["variant_c"]["over_variant_a_all_rows"] < 1.3 and d["variant_c"]["crops_per_s"] > 0


@pytest.mark.gpu
def test_bench_under_torchrun_one_rank_runs_the_rccl_leg():
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` with HG_BENCH_FORCE_COMM=1 runs
    init_process_group("nccl", device_id=...), the barriers, the side-stream event chain and the in-place
    all_gather_into_tensor of ShardedEncoder on the one GPU there is (launch model: main_tip_finetune.py:1205-1208,
    328-332), so that the first real multi-GPU run does not die on plumbing."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"HG_BENCH_FORCE_COMM": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "MASTER_ADDR": "127.0.0.1"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--no-extra-configs", "--no-class-rows"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and "all_gather" in d["config"]["parallelism"] and d["value"] > 0
