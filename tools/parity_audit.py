#!/usr/bin/env python3
"""Parity-risk audit: a NUMBER on "parity unpinned".

The CPU restatement (oracle/) had to infer three pieces of third-party arithmetic the reference's tree does not
contain: Eigen's Vector3d::norm() association, whether the real binary contracts multiplies and
adds into FMAs, FLANN's order among exactly tied neighbours.  This tool runs the reference's loops (STDesc.cpp:183-308,
351-400) on the benchmark workloads with every such piece evaluated in its plausible alternatives side by side and counts
the DECISIONS that come out differently — gate tests, distance tests, length limits, sort comparisons, dedup keys, cells —
plus +-1 ulp on dis_threshold, and reports how close the closest call is.  CPU only (uses the oracle: test infrastructure).

  python tools/parity_audit.py [--out profiles/r05_parity_audit.json] [--quick]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def audit_uniform(name, n_frames, n_kp, n_queries, build_frames, stream, label_lo=3, label_hi=11, shard=None, threads=8):
    from oracle.oracle import OracleManager, OrcAudit
    from sgtd_amd import synth
    t0 = time.time()
    smap = synth.make_map(n_frames, n_kp, stream=stream, label_lo=label_lo, label_hi=label_hi)
    qs = synth.make_queries(smap, n_queries, stream=stream)
    lo, hi = (0, n_frames) if shard is None else shard
    o = OracleManager(num_threads=threads, max_frame_n=max(20000, n_frames + 1))
    o.set_current_frame_id(lo)
    for f0 in range(lo, hi, 500):       # (orc_add_frames keeps a chunk's descriptors twice while it inserts them)
        o.add_frames(smap.xyz[f0:min(hi, f0 + 500)], smap.label[f0:min(hi, f0 + 500)])
    o.set_current_frame_id(n_frames)          # the query frames' id (one beyond the newest map frame)
    acc = OrcAudit()
    for f in range(lo, min(hi, lo + build_frames)):      # BuildSingleScanSTD of map frames
        o.audit_build(smap.xyz[f], smap.label[f], acc)
    for q in range(n_queries):                           # ... and of every query frame, then its candidate_selector loop
        o.audit_build(qs.xyz[q], qs.label[q], acc)
        o.audit_select(acc)
    d = acc.as_dict()
    d.update(workload=name, map_frames=n_frames, table_frames=[lo, hi], keypoints=n_kp, queries=n_queries,
             frames_built=min(hi - lo, build_frames) + n_queries, seconds=round(time.time() - t0, 1))
    return d


def audit_skewed(name, n_frames, n_queries, build_frames, threads=8):
    from oracle.oracle import OracleManager, OrcAudit
    from sgtd_amd import synth
    t0 = time.time()
    smap, world = synth.make_skewed_map(n_frames, stream=31)
    qs = synth.make_skewed_queries(world, n_queries, stream=3100)
    o = OracleManager(num_threads=threads, max_frame_n=max(20000, n_frames + 1))
    acc = OrcAudit()
    for f in range(n_frames):
        x, l = smap.frame(f)
        if f < build_frames:
            o.audit_build(x, l, acc)
        else:
            o.build(x, l, export=False)
        o.add_last()
    for q in range(n_queries):
        x, l = qs.frame(q)
        o.audit_build(x, l, acc)
        o.audit_select(acc)
    d = acc.as_dict()
    d.update(workload=name, map_frames=n_frames, queries=n_queries, frames_built=build_frames + n_queries, seconds=round(time.time() - t0, 1))
    return d


def other_svd_hypotheses(cov, qc, ec):
    """triangle_solver (STDesc.cpp:549-571) with ANOTHER SVD of the same covariance matrices: LAPACK's (numpy.linalg.svd,
    gesdd) instead of the restatement's one-sided Jacobi.  Returns [n, 12] (R row-major, t) and sigma_2 / sigma_1."""
    n = cov.shape[0]
    out = np.zeros((n, 12))
    ratio = np.ones(n)
    for i in range(n):
        U, S, Vt = np.linalg.svd(cov[i])
        V = Vt.T
        R = V @ U.T
        if np.linalg.det(R) < 0:
            R = V @ np.diag([1.0, 1.0, -1.0]) @ U.T
        out[i, :9] = R.reshape(9)
        out[i, 9:] = -(R @ qc[i]) + ec[i]
        ratio[i] = S[1] / S[0] if S[0] > 0 else 0.0
    return out, ratio


def audit_verify(name, n_frames, n_kp, n_queries, stream, threads=8, skewed=False, label_lo=3, label_hi=11):
    """candidate_verify of every candidate of n_queries query frames with both SVDs' hypotheses (orc_audit_verify)"""
    from oracle.oracle import OracleManager, OrcVerifyAudit
    from sgtd_amd import synth
    t0 = time.time()
    o = OracleManager(num_threads=threads, max_frame_n=max(20000, n_frames + 1))
    if skewed:
        smap, world = synth.make_skewed_map(n_frames, stream=31)
        qs = synth.make_skewed_queries(world, n_queries, stream=3100)
        for f in range(n_frames):
            x, l = smap.frame(f)
            o.build(x, l, export=False)
            o.add_last()
        frames = [qs.frame(q) for q in range(n_queries)]
    else:
        smap = synth.make_map(n_frames, n_kp, stream=stream, label_lo=label_lo, label_hi=label_hi)
        qs = synth.make_queries(smap, n_queries, stream=stream)
        for f0 in range(0, n_frames, 500):
            o.add_frames(smap.xyz[f0:min(n_frames, f0 + 500)], smap.label[f0:min(n_frames, f0 + 500)])
        frames = [(qs.xyz[q], qs.label[q]) for q in range(n_queries)]
    o.set_current_frame_id(n_frames)
    acc = OrcVerifyAudit()
    flattest = 1.0
    for x, l in frames:
        o.build(x, l, export=False)
        sel = o.select()
        for c in range(len(sel["cand_frame"])):
            cov, qc, ec = o.verify_hyp_inputs(c)
            if cov.shape[0] == 0:
                continue
            rt, ratio = other_svd_hypotheses(cov, qc, ec)
            flattest = min(flattest, float(ratio.min()))
            o.audit_verify(c, rt, acc)
    d = acc.as_dict()
    d.update(workload=name, map_frames=n_frames, queries=n_queries, flattest_triangle_sigma2_over_sigma1=flattest,
             seconds=round(time.time() - t0, 1))
    return d


def main_verify(args):
    T = args.threads
    if args.quick:
        plan = [("cfg2-like (quick)", lambda n: audit_verify(n, 120, 200, 4, 2, threads=T))]
    else:
        plan = [("cfg2: 1k-frame map", lambda n: audit_verify(n, 1000, 200, 48, 2, threads=T)),
                ("cfg3: 4 541-frame map (KITTI-00 length)", lambda n: audit_verify(n, 4541, 200, 32, 3, threads=T)),
                ("north star: 10k-frame map", lambda n: audit_verify(n, 10000, 200, 48, 1, threads=T)),
                ("cfg5 labels: 13 wild classes, 5k-frame map", lambda n: audit_verify(n, 5000, 200, 32, 5, threads=T, label_lo=0, label_hi=12)),
                ("skewed workload, 2 500 frames", lambda n: audit_verify(n, 2500, 0, 24, 0, threads=T, skewed=True))]
    runs = []
    for name, fn in plan:
        runs.append(fn(name))
        print("done:", name, runs[-1]["seconds"], "s", flush=True)
    tot = {k: sum(r[k] for r in runs) for k in ("candidates", "hypotheses", "pair_tests", "vertex_tests", "vertex_flips", "pair_flips",
                                                "vote_list_diffs", "best_index_diffs", "score_diffs", "inlier_set_diffs", "near_calls")}
    tot["min_margin"] = min(r["min_margin"] for r in runs)
    for k in ("max_norm_diff", "max_rot_diff", "max_t_diff"):
        tot[k] = max(r[k] for r in runs)
    tot["flattest_triangle_sigma2_over_sigma1"] = min(r["flattest_triangle_sigma2_over_sigma1"] for r in runs)
    out = {"what": "candidate_verify (STDesc.cpp:462-547) of every candidate with the hypotheses of two SVDs side by side: the restatement's "
                   "one-sided Jacobi (what the GPU path computes, bit for bit) and LAPACK's (numpy.linalg.svd) — a stand-in for the "
                   "Eigen::JacobiSVD of the real binary, which is not in this image; decisions = vertex tests ||R v + t - w|| < 3 (:488-505), "
                   "vote counts, the first maximum (:507-514), the score (:539), the kept pairs (:516-539)",
           "totals": tot, "runs": runs}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(tot, indent=1))


def tie_rate(n_frames, n_kp, sigma):
    """how often the generator's tie filter fires: frames of raw draws (no regeneration) with an exact f32 tie among the
    squared distances of some point's K + 1 nearest"""
    from scipy.spatial import cKDTree
    from sgtd_amd import synth
    smap = synth.make_map(n_frames, n_kp, stream=1)
    rng = np.random.Generator(np.random.PCG64(12345))
    tree = cKDTree(smap.landmarks[:, :2])
    raw, _ = synth._observe(smap.landmarks, smap.landmark_label, tree, smap.pose, n_kp, sigma, rng, tie_k=0)
    bad = synth.frames_with_knn_ties(raw, 10)
    return {"frames": n_frames, "keypoints": n_kp, "frames_with_an_exact_knn_distance_tie_before_the_filter": int(bad.size)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_parity_audit.json"))
    ap.add_argument("--quick", action="store_true", help="small sizes (the CPU test)")
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--verify", action="store_true", help="audit candidate_verify against another SVD instead (default --out: ..._verify.json)")
    args = ap.parse_args()
    if args.verify:
        if args.out.endswith("r05_parity_audit.json"):
            args.out = args.out.replace("r05_parity_audit.json", "r05_parity_audit_verify.json")
        return main_verify(args)
    T = args.threads
    if args.quick:
        plan = [("cfg2-like (quick)", lambda n: audit_uniform(n, 120, 200, 6, 40, 2, threads=T)),
                ("skewed (quick)", lambda n: audit_skewed(n, 80, 4, 20, threads=T))]
        tie_args = (300, 200, 0.02)
    else:
        plan = [
            ("cfg2: 1k-frame map, 200 keypoints", lambda n: audit_uniform(n, 1000, 200, 256, 1000, 2, threads=T)),
            ("cfg3: 4 541-frame map (KITTI-00 length)", lambda n: audit_uniform(n, 4541, 200, 128, 1000, 3, threads=T)),
            ("north star: 10k-frame map", lambda n: audit_uniform(n, 10000, 200, 64, 1000, 1, threads=T)),
            ("cfg5 labels: 13 wild classes, 5k-frame map", lambda n: audit_uniform(n, 5000, 200, 64, 500, 5, label_lo=0, label_hi=12, threads=T)),
            ("cfg4: one rank's shard (frames 0..12 499) of the 100k-frame map", lambda n: audit_uniform(n, 100000, 200, 32, 500, 4, shard=(0, 12500), threads=T)),
            ("skewed workload: Zipf labels, 50-400 keypoints, clusters; 2 500 frames", lambda n: audit_skewed(n, 2500, 48, 500, threads=T)),
        ]
        tie_args = (10000, 200, 0.02)
    # every run's result is written as soon as it exists (a 12 500-frame table of 416-byte entries is 30 GB: a run that
    # dies must not take the others with it); a second start resumes
    part = args.out + ".partial"
    done = json.load(open(part)) if os.path.exists(part) else {}
    for name, fn in plan:
        if name in done:
            continue
        done[name] = fn(name)
        with open(part, "w") as f:
            json.dump(done, f)
        print("done:", name, done[name]["seconds"], "s", flush=True)
    runs = [done[name] for name, _ in plan]
    ties = tie_rate(*tie_args)
    tot = {k: 0 for k in ("gate_tests", "visits", "near_calls", "knn_points", "knn_tied_points", "knn_fma_order_diffs", "triplets")}
    vec = {k: [0, 0, 0] for k in ("gate_flips", "gate_flip_visits", "match_flips", "side_value_diffs", "build_flips")}
    thr = [0, 0]
    mins = {k: float("inf") for k in ("min_margin", "min_margin_ulps", "min_gate_margin", "min_len_margin", "min_cell_margin")}
    for r in runs:
        for k in tot:
            tot[k] += r[k]
        for k in vec:
            vec[k] = [a + b for a, b in zip(vec[k], r[k])]
        thr = [a + b for a, b in zip(thr, r["thr_ulp_flips"])]
        for k in mins:
            mins[k] = min(mins[k], r[k])
    out = {"what": "decisions of the reference's loops that differ under alternative third-party arithmetic (oracle/sgtd_oracle.h: orc_audit); "
                   "variant 0 = right association, 1 = left association FMA-contracted, 2 = right association FMA-contracted",
           "totals": dict(tot, **vec, thr_ulp_flips_plus_minus=thr, **mins), "knn_tie_filter": ties, "runs": runs}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    if os.path.exists(part):
        os.remove(part)
    print(json.dumps(out["totals"], indent=1))


if __name__ == "__main__":
    main()
