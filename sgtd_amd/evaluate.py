"""Evaluation harness of the reference's query loop:
`semantic_graph_localization.cpp:567-646,716-745` without ROS / GICP — SearchLoop per
query on the device (candidate selection + candidate_verify), then the node's metrics:

* `compute_adj_rpe` (`include/utility.hpp:109-123`): translation / rotation error of a pose
  estimate against the ground truth, in f32 4x4 matrices like the node;
* the top-k hit histogram `STD_num` (`:619-645`): rank of the first candidate — candidates
  sorted by `match_fitness` descending (`compareLOOP_RESULT`, `:12-14`) — whose map pose is
  within 10 m of the query's true pose;
* success rate (`:735-745`): `T_error < 5 && R_error < 10` for
  `Global_SG[match].pose * loop_transform` (GICP disabled => `transformation` = identity,
  BASE2OUSTER = identity for synthetic data).

Quirks kept: a query whose SearchLoop result is frame 0 or "no loop" is skipped
(`search_result.first > 0`, `:606-620`, quirk 10); `match_fitness` is an int (quirk 15).
`std::sort` is unstable; here ties keep candidate order (documented difference without
observable effect on the metrics' definition)."""
import numpy as np


def pose_matrix(x, y, yaw, z=0.0):
    """sensor -> world as the 4x4 f32 the node builds from the 12-float pose row (:723-733)"""
    c, s = np.cos(yaw), np.sin(yaw)
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    m[0, 3], m[1, 3], m[2, 3] = x, y, z
    return m


def pose_row(x, y, yaw, z=0.0):
    """the 12 floats a graph file carries (row-major 3x4)"""
    return pose_matrix(x, y, yaw, z)[:3, :].reshape(12).copy()


def matrix_from_row(row12):
    m = np.eye(4, dtype=np.float32)
    m[:3, :] = np.asarray(row12, np.float32).reshape(3, 4)
    return m


def compute_adj_rpe(gt, lo):
    """utility.hpp:109-123: delta = lo^-1 * gt; t_e = |translation|, r_e in degrees"""
    delta = (np.linalg.inv(lo.astype(np.float32)) @ gt.astype(np.float32)).astype(np.float32)
    t_e = float(np.linalg.norm(delta[:3, 3]))
    c = min(max((float(np.trace(delta[:3, :3])) - 1.0) / 2.0, -1.0), 1.0)
    r_e = abs(np.arccos(c)) / np.pi * 180.0
    return t_e, r_e


class LoopMetrics:
    """running counters of the node's query loop"""

    def __init__(self, candidate_num=50):
        self.total_num = 0          # queries processed (:580)
        self.detected = 0           # SearchLoop returned a frame > 0
        self.score_num = 0          # T_error < 5 && R_error < 10 (:735)
        self.test_10 = 0            # a candidate within 10 m exists (:640)
        self.STD_num = np.zeros(candidate_num, np.int64)
        self.t_errors, self.r_errors = [], []

    def summary(self):
        n = max(self.total_num, 1)
        return {"queries": self.total_num, "loops_detected": self.detected,
                "success_rate_5m_10deg": self.score_num / n, "candidate_within_10m_rate": self.test_10 / n,
                "top1_hit_rate": float(self.STD_num[0]) / n,
                "mean_t_error_m": float(np.mean(self.t_errors)) if self.t_errors else None,
                "mean_r_error_deg": float(np.mean(self.r_errors)) if self.r_errors else None,
                "STD_num": self.STD_num.tolist()}


def account(metrics, gt_pose4, map_pose4, search_frame, loop_rot, loop_t, cand_frames, cand_fitness):
    """one iteration of the node's loop body after SearchLoop (:605-745).
    cand_frames / cand_fitness: match_result_list in candidate order (fitness = verify_score)."""
    metrics.total_num += 1
    if not search_frame > 0:                               # :606-620
        return None
    metrics.detected += 1
    order = np.argsort(-np.asarray(cand_fitness).astype(np.int64), kind="stable")   # compareLOOP_RESULT on the int member
    for rank, k in enumerate(order):                       # :621-645
        t_e1, _ = compute_adj_rpe(gt_pose4, map_pose4[int(cand_frames[k])])
        if t_e1 < 10:
            metrics.test_10 += 1
            metrics.STD_num[rank] += 1
            break
    new_trans = np.eye(4, dtype=np.float32)                # :716-720
    new_trans[:3, :3] = np.asarray(loop_rot, np.float64).astype(np.float32)
    new_trans[:3, 3] = np.asarray(loop_t, np.float64).astype(np.float32)
    mat = (map_pose4[int(search_frame)] @ new_trans).astype(np.float32)   # transform_j1 * new_trans * I (:733)
    t_err, r_err = compute_adj_rpe(gt_pose4, mat)
    if t_err < 5 and r_err < 10:                           # :735
        metrics.score_num += 1
        metrics.t_errors.append(t_err)
        metrics.r_errors.append(r_err)
    return t_err, r_err


def evaluate_batch(mgr, map_pose4, query_xyz, query_label, query_pose4, metrics=None, kp_off=None):
    """SearchLoop for a batch of query frames on the device + the node's accounting.
    map_pose4[f] = 4x4 pose of map frame f; query_pose4[q] = ground truth of query q."""
    if metrics is None:
        metrics = LoopMetrics(mgr.config_setting_["candidate_num"])
    res = mgr.query_frames(query_xyz, query_label, kp_off)
    mgr.verify()
    bc, bf, bs = mgr.search_loop()
    for q in range(len(bf)):
        n_c = int(res.n_cand[q])
        if bf[q] > 0:
            score, rot, t = mgr.result_verify(q)
            k = int(bc[q])
            account(metrics, query_pose4[q], map_pose4, int(bf[q]), rot[k], t[k], res.cand_frame[q, :n_c], score[:n_c])
        else:
            account(metrics, query_pose4[q], map_pose4, int(bf[q]), None, None, (), ())
    return metrics
