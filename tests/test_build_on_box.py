"""The library is built from source ON the GPU box (hipcc --offload-arch=gfx950, the Makefile's flags) into a scratch
path and that build — not the .so that travelled with the tree — runs __graft_entry__.smoke() in a child process:
the sources compile where the kernels run, and what they compile to passes the oracle check."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_library_rebuilt_on_the_gpu_box_passes_smoke(tmp_path):
    out = str(tmp_path / "libsgtd_accel_rebuilt.so")
    # the shipped build's own recipe (sgtd_amd/csrc/Makefile: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off ...), another output path
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "sgtd_amd", "csrc"), "-s", "-B", "OUT=" + out])
    assert os.path.getsize(out) > 500000
    env = dict(os.environ, SGTD_ACCEL_LIB=out)
    code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as g; from sgtd_amd import _lib; "
            "assert _lib.LIB_PATH == %r; g.smoke()" % (ROOT, out))
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "smoke ok" in run.stdout


@pytest.mark.gpu
def test_planner_with_room_for_two_groups_per_round(tmp_path):
    """plan_passes_kernel stages as many groups per round as its LDS room takes (probe_kernels.hip.h, SGTD_PLAN_QUADS:
    448 quarters, all 16 groups of a round on the synthetic maps).  Built with room for two groups (one with a tail
    segment), every wave needs many rounds and the groups left over wait: the parity tests of the batch path, the tail
    segment and the per-query record passes must not notice."""
    out = str(tmp_path / "libsgtd_accel_small_room.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "sgtd_amd", "csrc"), "-s", "-B", "OUT=" + out, "EXTRA=-DSGTD_PLAN_QUADS=112"])
    env = dict(os.environ, SGTD_ACCEL_LIB=out)
    run = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                          "-k", "select_parity or tail or append or twins"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-2000:]
    assert " passed" in run.stdout
