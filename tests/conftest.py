import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _inference_mode_like_the_reference():
    """Every inference call site of the reference runs under torch.no_grad() with frozen, eval() modules
    (main_tip_finetune.py:470-529, main_coop_vae.py:316-318,436); the façade refuses gradient callers
    (tests/test_abi_and_facade.py::test_gradient_callers_fail_loudly switches grad mode back on)."""
    import torch

    with torch.no_grad():
        yield
