"""The host C++ of the product under sanitizers, on the CPU build (no GPU sanitizer exists on the
pool): the adapter's templates (adapter/STDesc_shim.hpp through include/sgtd/STDescManager.hpp) against a host-only
stand-in for the C ABI, the graph-JSON scanner, the binary graph cache and the saved table's header under mutation
fuzzing, and the differential test against the reference's JSON library — AddressSanitizer + UndefinedBehaviorSanitizer
(every report fatal), and ThreadSanitizer for the adapter's fill team.  Sources: tests/cpp/sanitize/."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "cpp", "sanitize")
ASAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")


def _build(tmp_path, name, sources, flags):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g"] + flags + sources + ["-I" + os.path.join(ROOT, "include"), "-pthread", "-o", exe])
    return exe


def _run(exe, args, token, timeout=600):
    out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=timeout, env=ENV)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert token in out.stdout and "ERROR: " not in out.stderr and "WARNING: ThreadSanitizer" not in out.stderr, out.stdout[-2000:] + out.stderr[-4000:]


def test_adapter_under_asan_and_ubsan(tmp_path):
    exe = _build(tmp_path, "shim_asan", [os.path.join(SAN, "shim_driver.cpp"), os.path.join(SAN, "stub_abi.cpp")], ASAN)
    _run(exe, ["24"], "shim under the sanitizers: ok")
    # the opt-in policy that builds only the best candidate's loop_std_pair
    out = subprocess.run([exe, "12"], capture_output=True, text=True, timeout=600, env=dict(ENV, SGTD_SHIM_FILL="best"))
    assert out.returncode == 0 and "shim under the sanitizers: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_adapter_fill_team_under_tsan(tmp_path):
    exe = _build(tmp_path, "shim_tsan", [os.path.join(SAN, "shim_driver.cpp"), os.path.join(SAN, "stub_abi.cpp")], ["-fsanitize=thread"])
    _run(exe, ["12"], "shim under the sanitizers: ok")


def test_file_parsers_fuzzed_under_asan_and_ubsan(tmp_path):
    """JSON scanner (both duplicate-key policies, surrogate rules), sgtd_graphs_load_cache (every truncation + bit
    flips), the saved table's header: OK or an error, never a report"""
    exe = _build(tmp_path, "fuzz_files", [os.path.join(SAN, "fuzz_files.cpp"), os.path.join(SAN, "ingest_host.cpp")], ASAN)
    _run(exe, ["20000", "3000", str(tmp_path / "fuzz")], "file fuzzing: ok")


NLOHMANN = "/opt/conda/include/json.hpp"


@pytest.mark.skipif(not os.path.exists(NLOHMANN), reason="nlohmann/json.hpp (the reference's JSON library) is not in this image")
def test_nlohmann_differential_under_asan_and_ubsan(tmp_path):
    """tests/cpp/test_ingest_nlohmann.cpp linked against the host-only ingest (no libsgtd_accel, no HIP) with the sanitizers on"""
    exe = _build(tmp_path, "nlohmann_asan", [os.path.join(ROOT, "tests", "cpp", "test_ingest_nlohmann.cpp"), os.path.join(SAN, "ingest_host.cpp")], ASAN)
    _run(exe, ["400", str(tmp_path / "docs")], "ingest equals nlohmann::json")
