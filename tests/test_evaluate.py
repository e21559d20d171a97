"""Evaluation harness: the node's metrics
(semantic_graph_localization.cpp:605-745, utility.hpp:109-123) on hand-made and synthetic data."""
import numpy as np
import pytest

from sgtd_amd import evaluate as ev


def test_compute_adj_rpe_known_answers():
    a = ev.pose_matrix(10.0, -4.0, 0.3)
    assert ev.compute_adj_rpe(a, a) == (0.0, 0.0) or max(ev.compute_adj_rpe(a, a)) < 1e-3
    b = ev.pose_matrix(13.0, 0.0, 0.3)                    # 3-4-5 translation, same heading
    t, r = ev.compute_adj_rpe(a, b)
    assert abs(t - 5.0) < 1e-4 and r < 1e-2
    c = ev.pose_matrix(10.0, -4.0, 0.3 + np.deg2rad(20.0))
    t, r = ev.compute_adj_rpe(a, c)
    assert t < 1e-4 and abs(r - 20.0) < 1e-2
    assert np.allclose(ev.matrix_from_row(ev.pose_row(1, 2, 0.5, 3)), ev.pose_matrix(1, 2, 0.5, 3))


def test_account_follows_the_node_loop():
    map_pose = np.stack([ev.pose_matrix(2.0 * f, 0.0, 0.0) for f in range(40)])
    gt = ev.pose_matrix(20.5, 0.2, 0.1)                  # true place = frame 10
    ident_rot, zero_t = np.eye(3), np.zeros(3)
    m = ev.LoopMetrics(5)
    # "no loop" and frame 0 are skipped (search_result.first > 0, quirk 10)
    assert ev.account(m, gt, map_pose, -1, None, None, (), ()) is None
    assert ev.account(m, gt, map_pose, 0, ident_rot, zero_t, [0], [9.0]) is None
    assert (m.total_num, m.detected, m.score_num) == (2, 0, 0)
    # candidates (frame, fitness): sorted by int fitness desc -> frames 30 (far), 11 (near), 10
    rel = np.linalg.inv(map_pose[11]) @ gt                # exact loop transform for frame 11
    t_err, r_err = ev.account(m, gt, map_pose, 11, rel[:3, :3], rel[:3, 3], [10, 30, 11], [5.2, 40.0, 17.9])
    assert t_err < 1e-3 and r_err < 0.1
    assert m.detected == 1 and m.score_num == 1 and m.test_10 == 1 and m.STD_num.tolist() == [0, 1, 0, 0, 0]
    # a wrong pose estimate: detected, candidate within 10 m at rank 0, but no success
    ev.account(m, gt, map_pose, 30, ident_rot, zero_t, [10], [8.0])
    assert m.score_num == 1 and m.STD_num.tolist() == [1, 1, 0, 0, 0]
    s = m.summary()
    assert s["queries"] == 4 and s["success_rate_5m_10deg"] == 0.25 and s["loops_detected"] == 2


@pytest.mark.gpu
def test_synthetic_localization_end_to_end():
    from sgtd_amd import synth
    from sgtd_amd.manager import STDescManager
    smap = synth.make_map(120, 200, stream=9)
    q = synth.make_queries(smap, 24, stream=9)
    mgr = STDescManager()
    mgr.add_frames(smap.xyz, smap.label)
    map_pose = np.stack([ev.pose_matrix(*p) for p in smap.pose])
    q_pose = np.stack([ev.pose_matrix(*p) for p in q.pose])
    m = ev.evaluate_batch(mgr, map_pose, q.xyz, q.label, q_pose)
    s = m.summary()
    assert s["queries"] == 24
    skipped = int(np.sum(q.gt_frame == 0))                # frame 0 can never be reported (quirk 10)
    assert s["loops_detected"] >= 24 - skipped - 1
    assert s["success_rate_5m_10deg"] >= (24 - skipped - 1) / 24.0
    assert s["mean_t_error_m"] < 0.6 and s["mean_r_error_deg"] < 2.0   # one-triangle Kabsch at 5 cm noise
    mgr.close()
