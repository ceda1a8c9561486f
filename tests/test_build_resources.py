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
    sgpr_ok = ("adapter_kv_kernel", "adapter_decoder_kernel")
    bad = [r for r in rows if r[3] != 0 and not (r[2] == "SGPRs Spill" and any(k in r[1] for k in sgpr_ok))]
    assert not bad, "kernels that spill or use scratch:\n" + "\n".join(f"  {f}: {n}: {k} = {v}" for f, n, k, v in bad)
