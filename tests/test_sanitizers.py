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


HIPCC = "/opt/rocm/bin/hipcc"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(CLANG)), reason="needs the ROCm compilers (host-only compile of the engine)")
def test_engine_host_code_under_asan_and_ubsan(tmp_path):
    """sgtd_accel.hip + multi_impl.hip.h — the engine's ~3 200 lines of host orchestration — compiled for the host only with the
    sanitizers on and linked against tests/cpp/sanitize/hip_stub.cpp (device memory = zeroed host memory, kernels = a hook that
    leaves behind what a scenario needs): tables frame by frame with tail segments and their merge, batches whose sweep / whose
    reservations / whose candidate pairs outgrow the work buffers (re-runs, growth), the verification on a batch of 4e7 stand-in
    pairs, sgtd_search_frame's four ways out (one wait, too little room, a gather enqueued again, the fall-back after an
    overflow), views over an owner whose table changes under a pending batch, save / load / append and truncated table files,
    two "devices" behind one handle — and at the end every device buffer freed.  (First run: sgtd_destroy kept the entry-id
    map's four buffers, 8 bytes per map frame plus up to 8 per entry, of every handle it destroyed.)"""
    san = ASAN + ["-fno-omit-frame-pointer"]
    hip = [HIPCC, "-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-host-only", "-Wno-unused-function", "-Wno-unused-result"] + san
    objs = {n: str(tmp_path / (n + ".o")) for n in ("accel", "driver", "stub", "fatbin")}
    subprocess.check_call(hip + ["-c", os.path.join(ROOT, "sgtd_amd", "csrc", "sgtd_accel.hip"), "-o", objs["accel"]], stderr=subprocess.DEVNULL)
    subprocess.check_call(hip + ["-c", "-x", "hip", os.path.join(SAN, "engine_driver.cpp"), "-o", objs["driver"]], stderr=subprocess.DEVNULL)
    subprocess.check_call([CLANG, "-O1", "-g", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + san +
                          ["-c", os.path.join(SAN, "hip_stub.cpp"), "-o", objs["stub"]], stderr=subprocess.DEVNULL)
    # the host objects refer to their (absent) device code objects by a symbol whose name carries a hash of the translation unit
    undefined = subprocess.run(["nm", "-u", objs["accel"], objs["driver"]], capture_output=True, text=True, check=True).stdout
    names = sorted({w for line in undefined.splitlines() for w in line.split() if w.startswith("__hip_fatbin_")})
    assert names
    src = tmp_path / "fatbin.cpp"
    src.write_text("".join('extern "C" const char %s[16] = {0};\n' % n for n in names))
    subprocess.check_call([CLANG, "-c", str(src), "-o", objs["fatbin"]])
    exe = str(tmp_path / "engine_asan")
    subprocess.check_call([CLANG] + san + [objs["accel"], objs["driver"], objs["stub"], objs["fatbin"], "-pthread", "-o", exe])
    _run(exe, ["30", str(tmp_path)], "engine host code under the sanitizers: ok")
