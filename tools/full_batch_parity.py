#!/usr/bin/env python3
"""Every query frame of ONE full benchmark batch against the CPU restatement (diagnostic; uses the oracle: test
infrastructure): the north-star map (F frames, 200 keypoints), Q query frames through sgtd_query_frames in one call, then
per query the oracle's candidate_selector — candidate frames, votes, and every candidate's ordered match list
(query descriptor index, table entry id).  bench.py's cpu_baseline leg does this for 200 queries per run; this is the
whole batch.      python tools/full_batch_parity.py [F] [Q] [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from oracle.oracle import OracleManager
    from sgtd_amd import synth
    from sgtd_amd.manager import STDescManager
    from sgtd_amd.synth import effective_cpus
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    Q = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    stream = int(os.environ.get("FULL_PARITY_STREAM", "1000"))       # bench.py's first rotating batch
    skew = os.environ.get("FULL_PARITY_SKEW") == "1"       # bench.py's workload_skew leg: Zipf labels, 50-400 keypoints, clusters
    dev = torch.device("cuda", 0)
    g = STDescManager(device_id=0, max_frame_n=max(20000, F + 1))
    o = OracleManager(num_threads=effective_cpus(), max_frame_n=max(20000, F + 1))
    if skew:
        smap, world = synth.make_skewed_map(F, stream=31)
        qs = synth.make_skewed_queries(world, Q, stream=3100)
        g.add_frames(smap.xyz, smap.label, kp_off=smap.kp_off)
        g.finalize()
        res = g.query_frames(qs.xyz, qs.label, kp_off=qs.kp_off)
        t0 = time.time()
        for f in range(F):
            x, l = smap.frame(f)
            o.build(x, l, export=False)
            o.add_last()
        frame_of = qs.frame
    else:
        smap = synth.make_map(F, 200, stream=1)
        qs = synth.make_queries(smap, Q, stream=stream)
        g.add_frames(torch.from_numpy(smap.xyz).to(dev), torch.from_numpy(smap.label.astype(np.int64)).to(dev).to(torch.int32))
        g.finalize()
        res = g.query_frames(qs.xyz, qs.label)
        t0 = time.time()
        for f0 in range(0, F, 500):
            o.add_frames(smap.xyz[f0:min(F, f0 + 500)], smap.label[f0:min(F, f0 + 500)])

        def frame_of(q):
            return qs.xyz[q], qs.label[q]
    t_map = time.time() - t0
    st = g.stats()
    verify = os.environ.get("FULL_PARITY_VERIFY") == "1"   # also candidate_verify + SearchLoop's choice of every query (STDesc.cpp:462-571, :105-146)
    v_cands = v_same = v_accepted = v_inliers = choice_same = 0
    if verify:
        g.verify()
        bc, bf, bs = g.search_loop()
    same_c = same_l = 0
    pairs = P = M = 0
    bad = []
    t0 = time.time()
    for q in range(Q):
        o.build(*frame_of(q), export=False)
        r = o.select()
        c = o.counters()
        P += c["P"]; M += c["M"]
        nc = int(res.n_cand[q])
        ok_c = nc == len(r["cand_frame"]) and np.array_equal(res.cand_frame[q, :nc], r["cand_frame"]) and np.array_equal(res.cand_votes[q, :nc], r["cand_votes"])
        ok_l = False
        if ok_c:
            qi, de = g.result_pairs(q, res)
            ok_l = np.array_equal(qi, r["q_idx"]) and np.array_equal(de, r["db_entry"])
            pairs += len(qi)
        same_c += int(ok_c); same_l += int(ok_l)
        if verify and ok_c and ok_l:
            score, rot, t = g.result_verify(q)
            best_s, best_k = 0.0, -1
            for k in range(nc):
                n_pairs = int(res.pair_off[q, k + 1] - res.pair_off[q, k])
                o_score, o_t, o_rot, o_idx = o.verify(k, n_pairs)
                ok = score[k] == o_score
                if ok and o_score >= 0:
                    ok = np.array_equal(t[k], o_t) and np.array_equal(rot[k], o_rot) and np.array_equal(g.result_inliers(q, k, n_pairs), o_idx)
                    v_accepted += 1
                    v_inliers += len(o_idx)
                v_cands += 1
                v_same += int(bool(ok))
                if o_score > best_s:
                    best_s, best_k = o_score, k
            thr = g.icp_threshold_
            want = (best_k, int(r["cand_frame"][best_k]), best_s) if best_s > thr else (-1, -1, 0.0)
            choice_same += int((int(bc[q]), int(bf[q]), float(bs[q])) == want)
            ok_l = ok_l and v_same == v_cands
        if not (ok_c and ok_l) and len(bad) < 10:
            bad.append(q)
        if (q + 1) % (32 if skew else 256) == 0:
            print("%d / %d compared, %d identical, %.0f s" % (q + 1, Q, same_l, time.time() - t0), flush=True)
    out = {"workload": "skewed (Zipf labels, 50-400 keypoints, clusters)" if skew else "uniform, 200 keypoints per frame",
           "map_frames": F, "queries_in_the_batch": Q, "query_stream": 3100 if skew else stream,
           "identical_candidates_and_votes": same_c, "identical_ordered_match_lists": same_l, "first_differing_queries": bad,
           "match_list_pairs_compared": int(pairs), "P_visits_oracle": int(P), "M_matches_oracle": int(M),
           "P_visits_gpu_counter": int(st["last_P"]), "M_matches_gpu_counter": int(st["last_M"]),
           "oracle_threads": effective_cpus(), "oracle_map_build_s": round(t_map, 1), "oracle_seconds": round(time.time() - t0, 1),
           "select_form": int(st["select_form"])}
    if verify:
        out.update(candidates_verified=v_cands, identical_score_pose_and_inlier_set=v_same, candidates_accepted=v_accepted,
                   inlier_pairs_compared=int(v_inliers), identical_search_loop_choice=choice_same)
    print(json.dumps(out))
    if out_path:
        with open(out_path, "w") as fh:
            json.dump(out, fh, indent=1)
    g.close()
    sys.exit(0 if same_l == Q and int(st["last_P"]) == P and int(st["last_M"]) == M and v_same == v_cands and (not verify or choice_same == Q) else 1)


if __name__ == "__main__":
    main()
