"""Builds tests/cpp/test_manager.cpp (the C++ host-side mirror of STDescManager
above the C ABI) with g++ and runs it on the GPU box; the compile itself is
checked on CPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_manager")


def _build():
    from oracle import oracle
    from sgtd_amd import _lib
    _lib.build_library()
    oracle.build_library()
    src = os.path.join(ROOT, "tests", "cpp", "test_manager.cpp")
    cmd = ["g++", "-std=c++17", "-O2", "-include", "algorithm", src, "-o", EXE,
           "-L" + os.path.join(ROOT, "sgtd_amd"), "-lsgtd_accel",
           "-L" + os.path.join(ROOT, "oracle"), "-lsgtd_oracle",
           "-Wl,-rpath," + os.path.join(ROOT, "sgtd_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-pthread"]
    subprocess.check_call(cmd)


SHIM_EXE = os.path.join(ROOT, "tests", "cpp", "test_shim")


def _build_shim():
    from sgtd_amd import _lib
    _lib.build_library()
    src = os.path.join(ROOT, "tests", "cpp", "test_shim.cpp")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), src, "-o", SHIM_EXE,
           "-L" + os.path.join(ROOT, "sgtd_amd"), "-lsgtd_accel", "-Wl,-rpath," + os.path.join(ROOT, "sgtd_amd"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-pthread"]
    subprocess.check_call(cmd)


def test_reference_typed_shim_compiles_and_links():
    """adapter/STDesc_shim.hpp instantiated with declarations shaped like the reference's
    (Eigen-style vectors, PCL-style cloud pointer, the reference's struct names and method
    signatures); without a GPU the program stops at SGTD_ERR_NO_DEVICE"""
    _build_shim()
    out = subprocess.run([SHIM_EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim" in out.stdout


@pytest.mark.gpu
def test_reference_typed_shim_runs_the_callers_loops():
    _build_shim()
    out = subprocess.run([SHIM_EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim ok" in out.stdout


def test_cpp_host_mirror_compiles():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_host_mirror_matches_oracle():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cpp host mirror ok" in out.stdout
    # the same program with the manager's table sharded over two "devices" (sgtd_create_multi, both on GPU 0)
    out2 = subprocess.run([EXE, "multi"], capture_output=True, text=True, timeout=300)
    assert out2.returncode == 0, out2.stdout + out2.stderr
    assert out2.stdout == out.stdout
