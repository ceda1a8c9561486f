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
