"""The diagnostic tools that put numbers into profiles/ stay alive: tools/full_batch_parity.py at a small size (the
committed runs compare whole benchmark batches: profiles/r05_full_batch_parity*.json)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"FULL_PARITY_VERIFY": "1"}, {"FULL_PARITY_SKEW": "1"}], ids=["uniform+verify", "skewed"])
def test_full_batch_parity_tool(tmp_path, env):
    out = str(tmp_path / "p.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "full_batch_parity.py"), "300", "12", out],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, SGTD_SYNTH_CACHE="", **env))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.load(open(out))
    assert d["identical_ordered_match_lists"] == d["queries_in_the_batch"] == 12 and d["match_list_pairs_compared"] > 1000
    assert d["P_visits_gpu_counter"] == d["P_visits_oracle"] and d["M_matches_gpu_counter"] == d["M_matches_oracle"]
    if "FULL_PARITY_VERIFY" in env:
        assert d["candidates_verified"] == d["identical_score_pose_and_inlier_set"] > 100 and d["identical_search_loop_choice"] == 12


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"SGTD_COPY_IN_BLOCK": "0", "SGTD_FRAME_DIRECT": "0", "SGTD_SMALL_ORDER": "0"}], ids=["product", "general forms"])
def test_stress_session_in_a_process_of_its_own(env):
    """tools/stress_parity.py for half a minute (randomised maps, batches, views, sgtd_search_frame with ordinary and with
    page-locked arrays, the exchange kernels against the oracle) — once as shipped, once with the switches that are read
    at a process's first call set to the general forms: descriptors copied in field by field, the frame's results through
    the handle's own block, a one-frame batch ordered by the general kernels"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_parity.py"), "30", "4242"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert p.returncode == 0 and "stress ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
