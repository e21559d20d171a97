"""Graph-JSON ingest: the native loader against a plain Python reading
of the same files (json -> double -> f32 / int casts, the two steps nlohmann::json's
get<float>/get<int> perform, Semantic_Graph.hpp:131-164).  Host code only: runs without a GPU."""
import json
import os

import numpy as np
import pytest

from sgtd_amd import _lib, ingest, synth


def python_reading(path):
    with open(path) as f:
        j = json.load(f)
    return (np.array([[np.float32(v) for v in c[:3]] for c in j["centers"]], np.float32).reshape(-1, 3),
            np.array([np.uint32(int(v) & 0xFFFFFFFF) for v in j["nodes"]], np.uint32),
            np.array([np.float32(v) for v in j["poses"]][:12], np.float32))


def write_frames(tmp_path, n_frames=7, n_kp=40):
    m = synth.make_map(n_frames, n_kp, stream=5)
    paths = []
    for f in range(n_frames):
        pose = np.zeros(12, np.float32)
        pose[[0, 5, 10]] = 1
        pose[3], pose[7], pose[11] = m.pose[f, 0], m.pose[f, 1], 0.25 * f
        p = tmp_path / ("%06d.json" % f)
        ingest.write_graph_json(p, m.xyz[f], m.label[f], pose)
        paths.append(p)
    return m, paths


def test_loader_matches_python_reading(tmp_path):
    m, paths = write_frames(tmp_path)
    b = ingest.load_graphs(paths, threads=3)
    assert b.n_frames == len(paths)
    for f, p in enumerate(paths):
        xyz, lab, pose = python_reading(p)
        lo, hi = b.kp_off[f], b.kp_off[f + 1]
        assert np.array_equal(b.xyz[lo:hi], xyz) and np.array_equal(b.label[lo:hi], lab)
        assert np.array_equal(b.poses[f], pose)
        # the generator's f32 values survive the text round trip
        assert np.array_equal(b.xyz[lo:hi], m.xyz[f]) and np.array_equal(b.label[lo:hi], m.label[f])
    assert np.array_equal(b.position()[:, 2], 0.25 * np.arange(len(paths), dtype=np.float32))
    # frame order = order of the path list (the map order of the reference is the caller's, quirk 13)
    rev = ingest.load_graphs(paths[::-1], threads=2)
    assert np.array_equal(rev.xyz[:rev.kp_off[1]], b.xyz[b.kp_off[-2]:])


def test_loader_free_form_documents(tmp_path):
    doc = ('\n {"weights" : [0.5, 1e-3], "note": {"a": [1, {"b": null}], "s": "x\\"y]}"}, \t"poses":[1,0,0,1.5e1, 0,1,0,-2.25,0,0,1,3],\n'
           ' "centers":[[1, 2.5, -3e0],[0.1,0.2,0.30000000000000004, 9.0]], "nodes":[7, 11.0],"flag":true, "z":false}')
    p = tmp_path / "free.json"
    p.write_text(doc)
    b = ingest.load_graphs([p])
    xyz, lab, pose = python_reading(p)
    assert np.array_equal(b.xyz, xyz) and np.array_equal(b.label, lab) and np.array_equal(b.poses[0], pose)
    assert b.xyz[1, 2] == np.float32(0.30000000000000004) and b.poses[0, 3] == 15.0 and list(b.label) == [7, 11]


def test_loader_errors_name_the_file(tmp_path):
    _, paths = write_frames(tmp_path, 2, 12)
    missing = tmp_path / "nope.json"
    with pytest.raises(_lib.SgtdError) as e:
        ingest.load_graphs([paths[0], missing])
    assert "Error opening file" in str(e.value) and "nope.json" in str(e.value)     # Semantic_Graph.hpp:173
    bad = tmp_path / "bad.json"
    bad.write_text('{"nodes":[1,2], "centers":[[0,0,0]], "poses":[0,0,0,0,0,0,0,0,0,0,0,0]}')
    with pytest.raises(_lib.SgtdError) as e:
        ingest.load_graphs([bad])
    assert "bad.json" in str(e.value) and "differ" in str(e.value)
    bad.write_text('{"nodes":[1], "centers":[[0,0,0]]')
    with pytest.raises(_lib.SgtdError):
        ingest.load_graphs([bad])
    bad.write_text('{"nodes":[1], "centers":[[0,0]], "poses":[]}')
    with pytest.raises(_lib.SgtdError):
        ingest.load_graphs([bad])
    bad.write_text('{"centers":[[0,0,0]], "poses":[]}')
    with pytest.raises(_lib.SgtdError) as e:
        ingest.load_graphs([bad])
    assert "nodes" in str(e.value)


def test_cache_round_trip_and_empty(tmp_path):
    _, paths = write_frames(tmp_path, 5, 17)
    c = tmp_path / "graphs.bin"
    b = ingest.cache_graphs(paths, c, threads=2)
    r = ingest.load_cache(c)
    for name in ("xyz", "label", "kp_off", "poses"):
        assert np.array_equal(getattr(b, name), getattr(r, name))
    (tmp_path / "junk.bin").write_bytes(b"not a cache")
    with pytest.raises(_lib.SgtdError):
        ingest.load_cache(tmp_path / "junk.bin")
    e = ingest.load_graphs([])
    assert e.n_frames == 0 and e.xyz.shape == (0, 3) and list(e.kp_off) == [0]


@pytest.mark.gpu
def test_map_from_json_equals_map_from_arrays(tmp_path):
    from sgtd_amd.manager import STDescManager
    m, paths = write_frames(tmp_path, 10, 60)
    b = ingest.load_graphs(paths)
    q = synth.make_queries(m, 3, stream=5)
    out = []
    for xyz, lab, off in ((m.xyz, m.label, None), (b.xyz, b.label, b.kp_off)):
        mgr = STDescManager()
        mgr.add_frames(xyz, lab, off)
        res = mgr.query_frames(q.xyz, q.label)
        out.append((res.n_cand.copy(), res.cand_frame.copy(), res.cand_votes.copy(), res.pair_off.copy()))
        mgr.close()
    for a, c in zip(out[0], out[1]):
        assert np.array_equal(a, c)


NLOHMANN = "/opt/conda/include/json.hpp"


@pytest.mark.skipif(not os.path.exists(NLOHMANN), reason="nlohmann/json.hpp (the reference's JSON library) is not in this image")
def test_ingest_equals_the_references_json_library_bit_for_bit():
    """tests/cpp/test_ingest_nlohmann.cpp: 1500 fuzzed documents parsed with nlohmann::json exactly as
    Semantic_Graph.hpp:122-184 does (operator>>, get<std::vector<int>>, item[k].get<float>(),
    get<std::vector<float>>) and with sgtd_graphs_load — xyz, labels, poses and frame offsets equal bit
    for bit: exponents, -0 (an integer token: +0.0f), integers written as 11.0, integers beyond 2^24 /
    2^53 / 2^64, denormals, nested unknown keys, keys spelled with \\u escapes, keys that occur twice"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from sgtd_amd import _lib
    _lib.build_library()
    exe = os.path.join(root, "tests", "cpp", "test_ingest_nlohmann")
    lib_dir = os.path.join(root, "sgtd_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(root, "tests", "cpp", "test_ingest_nlohmann.cpp"), "-I" + os.path.join(root, "include"),
                           "-o", exe, "-L" + lib_dir, "-lsgtd_accel", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
                           "-lamdhip64", "-pthread"])
    out = subprocess.run([exe, "1500"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ingest equals nlohmann::json" in out.stdout


def test_number_scanner_equals_strtod_and_the_integer_conversions(tmp_path):
    """tests/cpp/test_ingest_numbers.cpp: the scanner's numbers on 3 M random tokens — float32 values dumped with 17
    digits (what the producer writes), coordinates of 9-17 digits, random digit strings with exponents, any finite
    double, integers around 2^63 / 2^64 — against strtod / strtoull / strtoll bit for bit (the scanner reaches most
    doubles through one x87 extended-precision operation instead of strtod: graph_ingest.hip.h, fast_double)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_ingest_numbers")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-x", "c++", os.path.join(root, "tests", "cpp", "test_ingest_numbers.cpp"), "-o", exe, "-pthread"])
    out = subprocess.run([exe, "3000000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "number scanner equals the C library" in out.stdout
