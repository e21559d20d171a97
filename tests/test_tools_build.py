"""The micro-benchmarks under tools/ that DESIGN.md quotes hardware facts from (texture-addresser rate, atomic adds per
address, vmcnt and stores issued with EXEC = 0) are single-file HIP programs run by hand on the GPU box; here they are
only cross-compiled for gfx950, so that they cannot rot unnoticed."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.parametrize("src", sorted(glob.glob(os.path.join(ROOT, "tools", "*.hip"))), ids=os.path.basename)
def test_tool_compiles_for_gfx950(src, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc here")
    out = str(tmp_path / (os.path.basename(src) + ".o"))
    run = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-result", "-c", "-o", out, src],
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    assert os.path.getsize(out) > 0
