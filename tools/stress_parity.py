"""Randomised differential stress of the GPU path against the oracle (not part of the pytest
suite: run on the GPU box, `python tools/stress_parity.py [seconds] [seed] [log.jsonl]`).  Every round draws
a configuration (K, resolution, thresholds, labels), a small map, an insertion pattern
(frames in one batch / frame by frame / appended after queries = tail segment / caller-stamped
frame ids out of order), optionally a multi-device handle, and compares candidates, votes,
match lists, counters and the ordered rough list with the oracle."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import oracle  # noqa: E402
from sgtd_amd import manager, synth  # noqa: E402


FORMS = {}


def one_round(rng, rnd):
    k = int(rng.integers(3, 13))
    cfg = dict(descriptor_near_num=k, std_side_resolution=float(rng.choice([0.25, 0.5, 1.0, 2.0])),
               descriptor_min_len=float(rng.choice([0.0, 0.5, 2.0])), descriptor_max_len=float(rng.choice([15.0, 30.0, 50.0])),
               rough_dis_threshold=float(rng.choice([0.01, 0.03, 0.06, 0.12])), candidate_num=int(rng.integers(1, 30)))
    big = rnd % 25 == 24                      # every 25th round: a map large enough for long lists, pairs and slab refills
    n_kp = max(k, int(rng.integers(150, 220) if big else rng.integers(12, 120)))
    n_frames = int(rng.integers(120, 400) if big else rng.integers(3, 40))
    labels = [(3, 11), (0, 12), (5, 6), (0, 40)][int(rng.integers(0, 4))]
    if big:   # keep the ORACLE's cost bounded: its serial tail copies 832 B per match (tens of millions of
              # matches per query with two labels, 2 m cells and a 0.12 threshold take it hours)
        cfg["rough_dis_threshold"] = min(cfg["rough_dis_threshold"], 0.03)
        cfg["std_side_resolution"] = min(cfg["std_side_resolution"], 1.0)
        if labels == (5, 6):
            labels = (3, 11)
        if k > 10:
            k = 10
            cfg["descriptor_near_num"] = 10
    stream = int(rng.integers(100, 100000))
    pattern = ["batch", "per_frame", "tail", "stamped"][int(rng.integers(0, 4))]
    multi = pattern in ("batch", "per_frame", "tail") and rng.random() < 0.3
    sigma = float(rng.choice([0.0, 0.02, 0.3]))          # 0.0: re-observed frames are exact copies (many twins)
    m = synth.make_map(n_frames, n_kp, stream=stream, label_lo=labels[0], label_hi=labels[1], sigma=max(sigma, 1e-4))
    q = synth.make_queries(m, 3, stream=stream)
    # the passes over the match records: the five-kernel form, or one workgroup per query (select_kernels.hip.h) —
    # production picks the latter only for batches with a query per CU, the hook forces it on these small ones
    form = "2" if rng.random() < 0.6 else "1"
    __import__("os").environ["SGTD_SELECT_MODE"] = form
    FORMS[form] = FORMS.get(form, 0) + 1
    g = manager.STDescManager(devices=[0, 0, 0] if multi else None, **cfg)
    o = oracle.OracleManager(**cfg)
    desc = "round %d: K=%d res=%g rough=%g cand=%d n_kp=%d F=%d labels=%s %s%s sigma=%g" % (
        rnd, k, cfg["std_side_resolution"], cfg["rough_dis_threshold"], cfg["candidate_num"], n_kp, n_frames, labels, pattern,
        " multi" if multi else "", sigma)
    if __import__("os").environ.get("STRESS_VERBOSE") == "1":
        print("  config:", desc, "stream", stream, flush=True)

    def add(lo, hi, how):
        if how == "batch":
            g.add_frames(m.xyz[lo:hi], m.label[lo:hi])
            for f in range(lo, hi):
                o.build(m.xyz[f], m.label[f], export=False); o.add_last()
        else:
            for f in range(lo, hi):
                d = g.BuildSingleScanSTD(m.xyz[f], m.label[f])
                od = o.build(m.xyz[f], m.label[f])
                if how == "stamped":            # caller-stamped ids, out of insertion order
                    fid = int(rng.integers(0, 50))
                    d.frame[:] = fid; od.frame[:] = fid
                    o.add(od)
                else:
                    o.add_last()
                g.AddSTDescs(d)

    def check():
        # round 5's paths, drawn per check: the batch through a VIEW of the table (sgtd_attach_table: own work buffers and
        # stream, the owner's table), and one query frame through sgtd_search_frame against the calls it stands for
        if not multi and rng.random() < 0.35:
            view = manager.STDescManager(**cfg)
            view.attach_table(g)
            rv = view.query_frames(q.xyz, q.label)
            r0 = g.query_frames(q.xyz, q.label)
            assert np.array_equal(rv.n_cand, r0.n_cand) and np.array_equal(rv.cand_frame, r0.cand_frame) and np.array_equal(rv.cand_votes, r0.cand_votes), desc + " view"
            for i in range(3):
                a, b = view.result_pairs(i, rv), g.result_pairs(i, r0)
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), desc + " view lists"
            view.close()
            FORMS["view"] = FORMS.get("view", 0) + 1
        if not multi and rng.random() < 0.35:
            d0 = g.BuildSingleScanSTD(q.xyz[1], q.label[1])
            if d0.n > 0:
                cands = g.candidate_selector(d0)
                r1 = g.results()
                g.verify()
                sc, rot, tt = g.result_verify(0)
                cap = int(r1.pair_off[0, -1])
                off, qi1, ent1 = g.result_inlier_entries(0, cap)
                fs = g.search_frame(d0, capacity=max(cap, 1), page_locked=bool(rng.random() < 0.5))      # (page-locked arrays: written in place by the device)
                assert fs["status"] == 0 and fs["n_cand"] == int(r1.n_cand[0]), desc + " frame"
                assert np.array_equal(fs["cand_frame"], r1.cand_frame[0]) and np.array_equal(fs["cand_votes"], r1.cand_votes[0]) and np.array_equal(fs["pair_off"], r1.pair_off[0]), desc + " frame tables"
                assert np.array_equal(fs["score"], sc) and np.array_equal(fs["rot"], rot) and np.array_equal(fs["t"], tt), desc + " frame verify"
                assert np.array_equal(fs["inlier_off"], off) and np.array_equal(fs["inlier_q_idx"], qi1) and np.array_equal(fs["entries"].side, ent1.side) and np.array_equal(fs["entries"].vertex, ent1.vertex), desc + " frame inliers"
                # candidate_selector alone in the one call: every pair of every list with its entry, against the calls it stands for
                if rng.random() < 0.5:
                    qa, da = g.result_pairs(0, r1)
                    lo = g.search_frame(d0, capacity=max(cap, 1), page_locked=bool(rng.random() < 0.5), lists_only=True)
                    assert lo["status"] == 0 and lo["n_inliers"] == cap and np.array_equal(lo["inlier_off"], r1.pair_off[0]) and np.array_equal(lo["inlier_q_idx"], qa), desc + " frame lists"
                    ea = g.fetch_entries(da) if len(da) else None
                    assert ea is None or (np.array_equal(lo["entries"].side, ea.side) and np.array_equal(lo["entries"].node_id, ea.node_id) and np.array_equal(lo["entries"].frame, ea.frame)), desc + " frame lists entries"
                # ... and the verification itself against the oracle's (same one-sided Jacobi SVD restated on the CPU)
                o.build(q.xyz[1], q.label[1], export=False)
                ow = o.select()
                for kc in range(min(len(ow["cand_frame"]), 3)):
                    o_score, o_t, o_rot, o_idx = o.verify(kc, int(ow["cand_off"][kc + 1] - ow["cand_off"][kc]))
                    assert sc[kc] == o_score and (o_score < 0 or (np.array_equal(tt[kc], o_t) and np.array_equal(rot[kc], o_rot))), desc + " verify vs oracle"
                FORMS["frame"] = FORMS.get("frame", 0) + 1
        r = g.query_frames(q.xyz, q.label)
        for i in range(3):
            o.build(q.xyz[i], q.label[i], export=False)
            want = o.select()
            nc = int(r.n_cand[i])
            assert np.array_equal(r.cand_frame[i, :nc], want["cand_frame"]), desc
            assert np.array_equal(r.cand_votes[i, :nc], want["cand_votes"]), desc
            qi, de = g.result_pairs(i, r)
            assert np.array_equal(qi, want["q_idx"]), desc
            if multi:
                got = g.fetch_entries(de[:200]); ref = o.fetch_entries(want["db_entry"][:200])
                assert np.array_equal(got.side, ref.side) and np.array_equal(got.frame, ref.frame), desc
            else:
                assert np.array_equal(de, want["db_entry"]), desc
        if not multi:
            c = o.counters(); st = g.stats()
            one = g.query_frames(q.xyz[2:3], q.label[2:3])
            st = g.stats()
            assert st["last_P"] == c["P"] and st["last_M"] == c["M"], desc
            gr, orr = g.result_rough(0), o.rough_matches()
            for key in ("q_idx", "cell", "db_entry", "frame", "dis"):
                assert np.array_equal(gr[key], orr[key]), desc + " rough " + key

    def exchange_check():
        """the multi-GPU step's device side in one process: 2-4 frame-range shards, packed export behind the vote pass, the
        merge kernel, the winners' lists (deferred, masked) — merged candidates, votes and the winners' ordered lists
        against the oracle's single table"""
        import torch
        from sgtd_amd.dist import shard_range
        W = int(rng.integers(2, 5))
        cn = cfg["candidate_num"]
        dev = torch.device("cuda", 0)
        ints = 2 * 3 * cn + 4
        shards, packed, base = [], [], [0]
        for r_ in range(W):
            lo, hi = shard_range(n_frames, W, r_)
            sm = manager.STDescManager(first_frame_id=lo, **cfg)
            if hi > lo:
                sm.add_frames(m.xyz[lo:hi], m.label[lo:hi])
            pk = torch.zeros(ints, dtype=torch.int32, device=dev)
            sm.set_candidate_export(pk)
            sm.set_deferred_lists(True)
            sm.query_frames(q.xyz, q.label, fetch=False)
            shards.append(sm); packed.append(pk)
            base.append(base[-1] + sm.stats()["n_entries"])
        torch.cuda.synchronize()
        gathered = torch.cat(packed).contiguous()
        outs = []
        for r_, sm in enumerate(shards):
            of = torch.empty((3, cn), dtype=torch.int32, device=dev)
            ov = torch.empty_like(of); osrc = torch.empty_like(of)
            on = torch.empty(3, dtype=torch.int32, device=dev)
            keep = torch.empty(3, dtype=torch.int64, device=dev)
            flags = torch.zeros(4, dtype=torch.int32, device=dev)
            sm.merge_candidates_dev(0, gathered, W, r_, 3, of, ov, on, osrc, keep, flags)
            torch.cuda.synchronize()
            if int(flags[0]) != 0:          # a shard's first batch outgrew a work buffer: re-run, export again, merge again
                for s2 in shards:
                    s2.sync()
                torch.cuda.synchronize()
                gathered = torch.cat(packed).contiguous()
                sm.merge_candidates_dev(0, gathered, W, r_, 3, of, ov, on, osrc, keep, flags)
                torch.cuda.synchronize()
                assert int(flags[0]) == 0, desc + " exchange flags"
            sm.finish_lists(keep)
            outs.append((of, ov, on, osrc, keep))
        torch.cuda.synchronize()
        local = [sm.results() for sm in shards]
        for i in range(3):
            o.build(q.xyz[i], q.label[i], export=False)
            want = o.select()
            nc = len(want["cand_frame"])
            of, ov, on, osrc, _ = outs[0]
            assert int(on[i]) == nc and np.array_equal(of[i, :nc].cpu().numpy(), want["cand_frame"]) and np.array_equal(ov[i, :nc].cpu().numpy(), want["cand_votes"]), desc + " exchange merge"
            lists = [shards[r_].result_pairs(i, local[r_]) for r_ in range(W)]
            src = osrc[i].cpu().numpy()
            for kc in range(nc):
                r_, sl = src[kc] >> 8, src[kc] & 255
                lo_, hi_ = local[r_].pair_off[i, sl], local[r_].pair_off[i, sl + 1]
                wl, wh = want["cand_off"][kc], want["cand_off"][kc + 1]
                assert hi_ - lo_ == wh - wl, desc + " exchange list length"
                assert np.array_equal(lists[r_][0][lo_:hi_], want["q_idx"][wl:wh]) and np.array_equal(lists[r_][1][lo_:hi_] + base[r_], want["db_entry"][wl:wh]), desc + " exchange lists"
        for sm in shards:
            sm.close()
        FORMS["exchange"] = FORMS.get("exchange", 0) + 1

    if pattern == "tail":
        cut = max(1, n_frames // 2)
        add(0, cut, "batch"); check()
        add(cut, n_frames, "per_frame" if rng.random() < 0.5 else "batch"); check()
    else:
        add(0, n_frames, pattern)
        check()
        if pattern == "batch" and not multi and rng.random() < 0.4:
            exchange_check()
    g.close()
    return desc


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    oracle.build_library()
    rng = np.random.default_rng(seed)
    t0, rnd = time.time(), 0
    verbose = __import__("os").environ.get("STRESS_VERBOSE") == "1"
    while time.time() - t0 < budget:
        if verbose:
            print("start round %d at %.1f s" % (rnd, time.time() - t0), flush=True)
        d = one_round(rng, rnd)
        rnd += 1
        if rnd % 10 == 0 or verbose:
            print(d, flush=True)
    print("stress ok: %d rounds in %.0f s (seed %d)" % (rnd, time.time() - t0, seed))
    if len(sys.argv) > 3:     # one line per session, appended to a JSON-lines log (profiles/r03_stress.jsonl)
        import json
        import subprocess
        root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
        head = __import__("os").environ.get("STRESS_HEAD", "")      # (the GPU box's copy of the tree has no .git)
        try:
            head = head or subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        except Exception:
            pass
        with open(sys.argv[3], "a") as fh:
            fh.write(json.dumps({"rounds_without_a_difference": rnd, "rounds_per_query_workgroups": FORMS.get("2", 0), "rounds_block_passes": FORMS.get("1", 0), "checks_through_a_view": FORMS.get("view", 0), "checks_of_search_frame_and_verify": FORMS.get("frame", 0), "checks_of_the_exchange_kernels": FORMS.get("exchange", 0),
                                 "seconds": round(time.time() - t0, 1), "seed": seed, "git_head": head,
                                 "compared": "candidates, votes, ordered match lists, P/M counters, ordered rough list (q, cell, entry, frame, dis) against oracle/sgtd_oracle.cpp",
                                 "last_round": d}) + "\n")


if __name__ == "__main__":
    main()
