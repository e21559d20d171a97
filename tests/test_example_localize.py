"""examples/localize.cpp — the node's main loop on the C ABI (graph-JSON directories in,
localization statistics out).  Compiled on CPU; on the GPU box its statistics must equal the
Python harness (sgtd_amd/evaluate.py) on the same files."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "localize")


def _build():
    from sgtd_amd import _lib
    _lib.build_library()
    cmd = ["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "localize.cpp"),
           "-o", EXE, "-L" + os.path.join(ROOT, "sgtd_amd"), "-lsgtd_accel",
           "-Wl,-rpath," + os.path.join(ROOT, "sgtd_amd"), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-pthread"]
    subprocess.check_call(cmd)


def test_example_compiles():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_example_matches_the_python_harness(tmp_path):
    from sgtd_amd import evaluate as ev, ingest, synth
    from sgtd_amd.manager import STDescManager
    _build()
    smap = synth.make_map(60, 150, stream=19)
    q = synth.make_queries(smap, 14, stream=19)
    (tmp_path / "map").mkdir()
    (tmp_path / "query").mkdir()
    for f in range(60):
        ingest.write_graph_json(tmp_path / "map" / ("%06d.json" % f), smap.xyz[f], smap.label[f], ev.pose_row(*smap.pose[f]))
    for i in range(14):
        ingest.write_graph_json(tmp_path / "query" / ("%06d.json" % i), q.xyz[i], q.label[i], ev.pose_row(*q.pose[i]))
    out = subprocess.run([EXE, str(tmp_path / "map"), str(tmp_path / "query"), "5"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"map frames (\d+), queries (\d+): loops (\d+), success\(5m,10deg\) (\d+) .*candidate<10m (\d+), top-1 hit (\d+)", out.stdout)
    assert m, out.stdout
    got = [int(x) for x in m.groups()]

    mgr = STDescManager()
    mgr.add_frames(smap.xyz, smap.label)
    # the example reads f32 pose rows from the files: give the harness the same matrices
    map_pose = np.stack([ev.matrix_from_row(ev.pose_row(*p)) for p in smap.pose])
    q_pose = np.stack([ev.matrix_from_row(ev.pose_row(*p)) for p in q.pose])
    met = ev.evaluate_batch(mgr, map_pose, q.xyz, q.label, q_pose)
    mgr.close()
    assert got == [60, 14, met.detected, met.score_num, met.test_10, int(met.STD_num[0])], (got, met.summary())
    assert met.score_num >= 12
    # the same run with the table sharded over "two devices" behind the one handle (both shards on GPU 0 here)
    out2 = subprocess.run([EXE, str(tmp_path / "map"), str(tmp_path / "query"), "5"], capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, SGTD_DEVICES="0,0", LOCALIZE_PER_FRAME="14"))
    assert "14/14 agree with the batched run" in out2.stdout, out2.stdout
    assert out2.returncode == 0, out2.stdout + out2.stderr
    assert out2.stdout.splitlines()[:2] == out.stdout.splitlines()[:2] and "2 device(s)" in out2.stdout
