#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (run in the build container).

The reference ships no golden vectors for this path and cannot
be built here, so these fixtures are outputs of the repo's own oracle — they
pin the oracle against regressions and give the GPU path a frozen target that
does not depend on re-running the oracle.  Inputs come from the seeded
synthetic generator (sgtd_amd/synth.py); frames with exact k-NN distance ties
are rejected because FLANN's tie order is unpinned.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.oracle import OracleManager  # noqa: E402
from sgtd_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (oracle config, n_frames, n_keypoints, n_queries, stream)
    "shipped_n32_f8": (dict(), 8, 32, 2, 101),
    "k6_res05_n40_f8": (dict(descriptor_near_num=6, std_side_resolution=0.5, descriptor_min_len=1.0,
                             descriptor_max_len=30.0, rough_dis_threshold=0.05, candidate_num=4), 8, 40, 2, 102),
}

DESC_FIELDS = ("side", "angle", "center", "vertex", "label", "frame", "node_id")


def make(name, cfg, n_frames, n_kp, n_q, stream):
    m = synth.make_map(n_frames, n_kp, stream=stream)
    qs = synth.make_queries(m, n_q, stream=stream)
    k = cfg.get("descriptor_near_num", 10)
    for f in range(n_frames):
        assert not synth.has_knn_ties(m.xyz[f], k), "tie in map frame %d: change the stream" % f
    for q in range(n_q):
        assert not synth.has_knn_ties(qs.xyz[q], k), "tie in query %d: change the stream" % q
    o = OracleManager(**cfg)
    out = dict(cfg_keys=np.array(sorted(cfg.keys())), cfg_vals=np.array([float(cfg[k]) for k in sorted(cfg.keys())]),
               map_xyz=m.xyz, map_label=m.label, q_xyz=qs.xyz, q_label=qs.label)
    counts = []
    for f in range(n_frames):
        d = o.build(m.xyz[f], m.label[f])
        counts.append(d.n)
        if f < 2:   # full descriptors of the first two frames
            for fld in DESC_FIELDS:
                out["map%d_%s" % (f, fld)] = getattr(d, fld)
        o.add_last()
    out["map_desc_count"] = np.array(counts, np.int64)
    keys, off, ids = o.table_dump()
    out["table_keys"], out["table_off"], out["table_ids"] = keys, off, ids
    for q in range(n_q):
        d = o.build(qs.xyz[q], qs.label[q])
        for fld in ("side", "label", "node_id"):
            out["q%d_%s" % (q, fld)] = getattr(d, fld)
        r = o.select()
        rm = o.rough_matches()
        c = o.counters()
        out["q%d_counters" % q] = np.array([c["D"], c["P"], c["M"]], np.int64)
        for kk in ("cand_frame", "cand_votes", "cand_off", "q_idx", "db_entry"):
            out["q%d_%s" % (q, kk)] = r[kk]
        for kk in ("q_idx", "cell", "db_entry", "frame", "dis"):
            out["q%d_rough_%s" % (q, kk)] = rm[kk]
        out["q%d_votes" % q] = o.votes()[:n_frames + 1]
        # next stage (candidate_verify, STDesc.cpp:462-547): score, pose (rot row-major + t), inliers
        sc, ps, inl, ioff = [], [], [], [0]
        for k in range(len(r["cand_frame"])):
            s_, t_, rot_, idx_ = o.verify(k, int(r["cand_off"][k + 1] - r["cand_off"][k]))
            sc.append(s_)
            ps.append(np.concatenate([rot_.reshape(9), t_]) if s_ >= 0 else np.zeros(12))
            inl.append(idx_)
            ioff.append(ioff[-1] + len(idx_))
        out["q%d_verify_score" % q] = np.array(sc, np.float64)
        out["q%d_verify_pose" % q] = np.array(ps, np.float64).reshape(-1, 12)
        out["q%d_verify_inliers" % q] = np.concatenate(inl).astype(np.int32) if inl else np.zeros(0, np.int32)
        out["q%d_verify_inlier_off" % q] = np.array(ioff, np.int64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, "%.1f KB" % (os.path.getsize(path) / 1024.0), "D per frame", counts)


if __name__ == "__main__":
    for name, (cfg, nf, nk, nq, stream) in CASES.items():
        make(name, cfg, nf, nk, nq, stream)
