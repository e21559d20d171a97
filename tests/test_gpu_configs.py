"""GPU tests at the sizes BASELINE.json's configs name.

cfg2  F = 1 000, N = 200, 1 000 queries: sampled oracle comparison at exactly configs[1]'s size;
cfg3  F = 4 541 (the KITTI-00 length), self-localization: every map frame re-observed
      (semantic_graph_localization.cpp:567) — properties on all 4 541 queries, the full
      oracle comparison (candidates, votes, ordered match lists) on a sample;
north-star point  F = 10 000, 1 GPU: sampled oracle comparison + identical top-1, and the
      same map as 8 frame-range shards merged with the reference's rule (cfg4's mechanism
      at a tenth of its size, all shards on this one GPU);
cfg5  two sessions of one world (independent noise draws), 13 "wild" label classes
      (get_json_wild.cpp:10-12), 2 x 2 500 frames: sampled oracle comparison + both
      sessions retrieved.
(The oracle maps are inserted with orc_add_frames: builds on all host threads, a few seconds each.)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from oracle import oracle
    from sgtd_amd import manager, synth
    oracle.build_library()
    return oracle, manager, synth


def _oracle_map(oracle, smaps, threads=0):
    from sgtd_amd.synth import effective_cpus
    n = sum(m.xyz.shape[0] for m in smaps)
    # (the reference's thread rule, nproc - 4, on the CPUs this container may really use)
    o = oracle.OracleManager(num_threads=threads or max(1, effective_cpus() - 4), max_frame_n=max(20000, n + 1))
    for m in smaps:
        o.add_frames(m.xyz, m.label)     # = build + add_last per frame (tests/test_oracle_kat.py), builds on all host threads
    return o


def _same_as_oracle(g, o, res, q, xyz, label):
    o.build(xyz, label, export=False)
    r = o.select()
    nc = int(res.n_cand[q])
    assert np.array_equal(res.cand_frame[q, :nc], r["cand_frame"]), "candidate frames differ (query %d)" % q
    assert np.array_equal(res.cand_votes[q, :nc], r["cand_votes"]), "votes differ (query %d)" % q
    qi, de = g.result_pairs(q, res)
    assert np.array_equal(qi, r["q_idx"]) and np.array_equal(de, r["db_entry"]), "match lists differ (query %d)" % q
    assert np.array_equal(res.pair_off[q, :nc + 1], r["cand_off"])
    return r


def test_cfg1_one_scan_per_call_against_a_100_frame_map_graph_json_in(mods, tmp_path):
    """BASELINE configs[0] at its own size: a 100-frame map and 200 scans, every one a graph-JSON file in the producer's format
    (get_json.cpp:332-341; synthetic stand-in: KITTI-00 is not in the tree) read by the ingest, one scan per call with host
    pointers in (the reference's call pattern, semantic_graph_localization.cpp:590-603) and the same scans as one batch —
    candidates, votes, ordered match lists, the verification's score / pose and SearchLoop's choice of all 200 against the oracle"""
    oracle, manager, synth = mods
    from sgtd_amd import evaluate as ev, ingest
    F, N, NQ = 100, 200, 200
    smap = synth.make_map(F, N, stream=0)
    qs = synth.make_queries(smap, NQ, stream=0)
    mp, qp = [], []
    for f in range(F):
        mp.append(str(tmp_path / ("map_%04d.json" % f)))
        ingest.write_graph_json(mp[-1], smap.xyz[f], smap.label[f], ev.pose_row(*smap.pose[f]))
    for q in range(NQ):
        qp.append(str(tmp_path / ("scan_%04d.json" % q)))
        ingest.write_graph_json(qp[-1], qs.xyz[q], qs.label[q], ev.pose_row(*qs.pose[q]))
    gm, gq = ingest.load_graphs(mp), ingest.load_graphs(qp)
    assert gm.n_frames == F and gq.n_frames == NQ and np.array_equal(gm.xyz.reshape(F, N, 3), smap.xyz) and np.array_equal(gq.xyz.reshape(NQ, N, 3), qs.xyz)
    g = manager.STDescManager()
    g.add_frames(gm.xyz, gm.label, kp_off=gm.kp_off)
    o = _oracle_map(oracle, [smap])
    qx, ql = gq.xyz.reshape(NQ, N, 3), gq.label.reshape(NQ, N)
    batch = g.query_frames(qx, ql)
    g.verify()
    b_choice = g.search_loop()
    b_pairs = [g.result_pairs(q, batch) for q in range(NQ)]
    b_ver = [g.result_verify(q) for q in range(NQ)]
    loops = 0
    for q in range(NQ):
        one = g.query_frames(qx[q:q + 1], ql[q:q + 1])          # one scan per call
        r = _same_as_oracle(g, o, one, 0, qx[q], ql[q])
        nc = int(one.n_cand[0])
        assert int(batch.n_cand[q]) == nc and np.array_equal(batch.cand_frame[q, :nc], r["cand_frame"]) and np.array_equal(batch.cand_votes[q, :nc], r["cand_votes"])
        assert np.array_equal(b_pairs[q][0], r["q_idx"]) and np.array_equal(b_pairs[q][1], r["db_entry"])
        g.verify()
        score, rot, t = g.result_verify(0)
        best_s, best_k = 0.0, -1
        for k in range(nc):
            o_score, o_t, o_rot, _ = o.verify(k, int(one.pair_off[0, k + 1] - one.pair_off[0, k]))
            assert score[k] == o_score, (q, k)
            if o_score >= 0:
                assert np.array_equal(t[k], o_t) and np.array_equal(rot[k], o_rot), (q, k)
            if o_score > best_s:
                best_s, best_k = o_score, k
        assert all(np.array_equal(a[:nc], b[:nc]) for a, b in zip(b_ver[q], (score, rot, t)))
        bc, bf, bs = g.search_loop()
        assert (int(bc[0]), int(bf[0]), float(bs[0])) == (int(b_choice[0][q]), int(b_choice[1][q]), float(b_choice[2][q]))
        if best_s > g.icp_threshold_:          # SearchLoop's choice (STDesc.cpp:105-146): the first candidate with the largest score
            assert int(bc[0]) == best_k and float(bs[0]) == best_s and int(bf[0]) == int(r["cand_frame"][best_k])
        else:
            assert int(bf[0]) == -1
        loops += int(bf[0] >= 0)
    assert loops >= NQ * 9 // 10
    g.close()


def _list_properties(g, res, q):
    """what every candidate_selector result satisfies whatever the size: votes descending with
    ties by ascending frame, at least 5 votes, list lengths == votes, q_idx ascending per list"""
    nc = int(res.n_cand[q])
    v, f = res.cand_votes[q, :nc], res.cand_frame[q, :nc]
    assert np.all(v >= 5)
    assert np.all((np.diff(v) < 0) | ((np.diff(v) == 0) & (np.diff(f) > 0)))
    assert np.array_equal(np.diff(res.pair_off[q, :nc + 1]), v)        # a candidate's list holds its votes
    return nc


def test_cfg2_1k_frame_map_200_keypoints_sampled_parity(mods):
    """BASELINE configs[1] at exactly its size: synthetic 200 keypoints/frame, 1 000-frame map,
    descriptor build + match of 1 000 query frames on one GPU; the oracle compared on a sample
    (candidates, votes, ordered match lists, list offsets, P/M counters), properties on the rest"""
    oracle, manager, synth = mods
    F, N, Q = 1000, 200, 1000
    m = synth.make_map(F, N, stream=2)
    qs = synth.make_queries(m, Q, stream=2)
    g = manager.STDescManager()
    g.add_frames(m.xyz, m.label)
    res = g.query_frames(qs.xyz, qs.label)
    assert np.all(res.n_cand > 0)
    d = np.linalg.norm(m.pose[np.clip(res.top1(), 0, F - 1), :2] - qs.pose[:, :2], axis=1)
    assert np.mean(d < 5.0) >= 0.99
    for q in range(0, Q, 41):
        _list_properties(g, res, q)
    o = _oracle_map(oracle, [m])
    P = M = 0
    checked = (0, 333, 500, 999)
    for q in checked:
        r = _same_as_oracle(g, o, res, q, qs.xyz[q], qs.label[q])
        assert res.cand_frame[q, 0] == r["cand_frame"][0]
        c = o.counters()
        P += c["P"]; M += c["M"]
    sub = g.query_frames(qs.xyz[list(checked)], qs.label[list(checked)])
    st = g.stats()
    assert st["last_P"] == P and st["last_M"] == M and np.array_equal(sub.cand_frame, res.cand_frame[list(checked)])
    g.close()


def test_cfg3_kitti_length_map_every_frame_reobserved(mods):
    oracle, manager, synth = mods
    F, N, B = 4541, 200, 1024
    m = synth.make_map(F, N, stream=3)
    qs = synth.make_queries(m, F, stream=3, frames=np.arange(F))
    g = manager.STDescManager()
    g.add_frames(m.xyz, m.label)
    assert g.current_frame_id_ == F
    o = None
    sample = {7, 1500, 3000, 4540}
    within = 0
    for b0 in range(0, F, B):
        b1 = min(F, b0 + B)
        res = g.query_frames(qs.xyz[b0:b1], qs.label[b0:b1])
        top1 = res.top1()
        assert np.all(res.n_cand > 0)
        d = np.linalg.norm(m.pose[np.clip(top1, 0, F - 1), :2] - qs.pose[b0:b1, :2], axis=1)
        within += int(np.sum(d < 5.0))
        for q in range(0, b1 - b0, 97):
            nc = _list_properties(g, res, q)
            qi, de = g.result_pairs(q, res)
            lo, hi = res.pair_off[q, 0], res.pair_off[q, 1]
            assert np.all(np.diff(qi[lo:hi]) >= 0)
            ent = g.fetch_entries(de[lo:hi][:128])
            assert np.all(ent.frame == res.cand_frame[q, 0]) and nc > 0
        for f in sorted(sample):
            if b0 <= f < b1:
                if o is None:
                    o = _oracle_map(oracle, [m])
                _same_as_oracle(g, o, res, f - b0, qs.xyz[f], qs.label[f])
    assert within >= 0.995 * F          # top-1 lands within the reference's 5 m success radius


def test_north_star_10k_frame_map_sampled_parity_and_8_shards(mods):
    import torch
    from sgtd_amd.dist import merge_candidates, shard_range
    oracle, manager, synth = mods
    F, N, Q, G = 10000, 200, 256, 8
    m = synth.make_map(F, N, stream=1)
    qs = synth.make_queries(m, Q, stream=1)
    g = manager.STDescManager()
    g.add_frames(m.xyz, m.label)
    res = g.query_frames(qs.xyz, qs.label)
    assert np.all(res.n_cand > 0)
    for q in range(0, Q, 16):
        _list_properties(g, res, q)
    o = _oracle_map(oracle, [m])
    P = M = 0
    checked = (0, 101, 255)
    for q in checked:
        r = _same_as_oracle(g, o, res, q, qs.xyz[q], qs.label[q])
        assert res.cand_frame[q, 0] == r["cand_frame"][0]            # identical top-1 (north star)
        c = o.counters()
        P += c["P"]; M += c["M"]
    # the device's own counters of visited entries / rough matches, against the oracle's
    sub = g.query_frames(qs.xyz[list(checked)], qs.label[list(checked)])
    st = g.stats()
    assert st["last_P"] == P and st["last_M"] == M and np.array_equal(sub.cand_frame, res.cand_frame[list(checked)])
    # the safe batch size the library reports: ~740 k matches per query here; records are named by granules of four (6.9e10
    # of them), what binds is the 32-bit index of a batch's candidate pairs (0.2 .. 0.5 of the matches) and the memory
    mb = g.max_batch(N)
    assert 4096 <= mb <= (0xFFFFFFF0 << 2) // 700000, mb
    fresh = manager.STDescManager()                 # before any batch: the estimate from the table statistics
    fresh.add_frames(m.xyz[:2000], m.label[:2000])
    assert fresh.max_batch(N) >= 1024
    fresh.close()
    # cfg4's mechanism: the same map as 8 frame-range shards, local top-50 each, merged
    cn = g.config_setting_["candidate_num"]
    fr, vo = [], []
    for r_ in range(G):
        lo, hi = shard_range(F, G, r_)
        s = manager.STDescManager(first_frame_id=lo)
        s.add_frames(m.xyz[lo:hi], m.label[lo:hi])
        rs = s.query_frames(qs.xyz, qs.label)
        f_ = np.full((Q, cn), -1, np.int32); v_ = np.zeros((Q, cn), np.int32)
        for q in range(Q):
            nc = int(rs.n_cand[q])
            f_[q, :nc] = rs.cand_frame[q, :nc]; v_[q, :nc] = rs.cand_votes[q, :nc]
        fr.append(torch.from_numpy(f_)); vo.append(torch.from_numpy(v_))
        s.close()
    mf, mv, mn = merge_candidates(torch.stack(fr), torch.stack(vo), cn)
    for q in range(Q):
        nc = int(res.n_cand[q])
        assert int(mn[q]) == nc
        assert np.array_equal(mf[q, :nc].numpy(), res.cand_frame[q, :nc]) and np.array_equal(mv[q, :nc].numpy(), res.cand_votes[q, :nc])


def test_cfg5_two_sessions_wild_labels_at_size(mods):
    oracle, manager, synth = mods
    F, N, Q = 2500, 200, 64
    s1 = synth.make_map(F, N, stream=5, label_lo=0, label_hi=12)
    s2 = synth.make_map(F, N, stream=5, label_lo=0, label_hi=12, sigma=0.04)   # same world, another noise draw
    assert np.array_equal(s1.pose, s2.pose) and not np.array_equal(s1.xyz, s2.xyz)
    g = manager.STDescManager()
    g.add_frames(s1.xyz, s1.label)
    g.add_frames(s2.xyz, s2.label)               # second session appended to the first (frames F..2F-1)
    assert g.current_frame_id_ == 2 * F
    qs = synth.make_queries(s1, Q, stream=5)
    res = g.query_frames(qs.xyz, qs.label)
    both = 0
    for q in range(Q):
        nc = _list_properties(g, res, q)
        cf = res.cand_frame[q, :nc]
        both += int(np.any(cf < F) and np.any(cf >= F))
        d = np.linalg.norm(s1.pose[cf[0] % F, :2] - qs.pose[q, :2])
        assert d < 5.0
    assert both >= 0.9 * Q                       # a place is retrieved from both sessions
    o = _oracle_map(oracle, [s1, s2])
    for q in (0, 31, 63):
        _same_as_oracle(g, o, res, q, qs.xyz[q], qs.label[q])


def test_cfg4_100k_frame_map_sharded_eight_ways_behind_one_handle(mods):
    """BASELINE configs[3] at its real size on what one GPU box offers: a 100 000-frame map
    (444 M table entries) as ONE table and as EIGHT frame-block shards behind one handle
    (sgtd_create_multi, all eight on this GPU): per-shard top-50 tables merged with the
    reference's rule must equal the single-table result — candidates, votes, list offsets,
    match lists — and the place must be found.  The reference itself cannot hold this map
    (MAX_FRAME_N = 20 000, STDesc.h:33); the oracle is not run at this size."""
    _, manager, synth = mods
    F, N, Q = 100000, 200, 48
    m = synth.make_map(F, N, stream=4)
    qs = synth.make_queries(m, Q, stream=4)
    single = manager.STDescManager(max_frame_n=F + 1)
    single.add_frames(m.xyz, m.label)
    sres = single.query_frames(qs.xyz, qs.label)
    st = single.stats()
    assert st["n_entries"] > 4e8 and st["overflowed"] in (0, 1)
    assert np.all(sres.n_cand > 0)
    d = np.linalg.norm(m.pose[sres.top1(), :2] - qs.pose[:, :2], axis=1)
    assert np.mean(d < 5.0) >= 0.95
    pairs = [single.result_pairs(q, sres) for q in range(0, Q, 12)]
    ents = [single.fetch_entries(p[1][:64]) for p in pairs]
    single.close()
    del single
    multi = manager.STDescManager(max_frame_n=F + 1, devices=[0] * 8)
    multi.add_frames(m.xyz, m.label)
    assert multi.stats()["n_entries"] == st["n_entries"]
    mres = multi.query_frames(qs.xyz, qs.label)
    np.testing.assert_array_equal(mres.n_cand, sres.n_cand)
    np.testing.assert_array_equal(mres.cand_frame, sres.cand_frame)
    np.testing.assert_array_equal(mres.cand_votes, sres.cand_votes)
    np.testing.assert_array_equal(mres.pair_off, sres.pair_off)
    for k, q in enumerate(range(0, Q, 12)):
        mq, me = multi.result_pairs(q, mres)
        np.testing.assert_array_equal(mq, pairs[k][0])
        got = multi.fetch_entries(me[:64])
        for f in ("side", "frame", "label", "node_id"):
            np.testing.assert_array_equal(getattr(got, f), getattr(ents[k], f))
    multi.close()


def test_cfg5_two_50k_frame_sessions_wild_labels_at_cfg4s_size(mods):
    """BASELINE configs[4] at configs[3]'s size: two sessions of 50 000 frames of ONE world (the second an independent
    noise draw of the first's observations), 13 "wild" label classes (get_json_wild.cpp:10-12), appended one behind
    the other in one handle (100 000 frames) and, the same frames, as EIGHT frame-block shards behind one handle on this
    GPU.  The shard merge must equal the single table (candidates, votes, offsets, match lists), a place must come
    back from BOTH sessions, the lists must have the properties every candidate_selector result has, and the product
    sweep's counters of visited entries and rough matches must equal the diagnostic sweep's (the reference's distance
    test verbatim in f64 on every visited entry) on a sample.  The oracle cannot hold this map."""
    _, manager, synth = mods
    F, N, Q = 50000, 200, 48
    s1 = synth.make_map(F, N, stream=5, label_lo=0, label_hi=12)
    rng = np.random.Generator(np.random.PCG64(20251121 + 55))
    x2 = (s1.xyz.astype(np.float64) + rng.normal(0.0, 0.03, size=s1.xyz.shape)).astype(np.float32)      # the same places seen again
    qs = synth.make_queries(s1, Q, stream=55)
    single = manager.STDescManager(max_frame_n=2 * F + 1)
    single.add_frames(s1.xyz, s1.label)
    single.add_frames(x2, s1.label)               # second session appended: frames F .. 2F-1
    assert single.current_frame_id_ == 2 * F
    sres = single.query_frames(qs.xyz, qs.label)
    st = single.stats()
    assert st["n_entries"] > 4e8
    both = 0
    for q in range(Q):
        nc = _list_properties(single, sres, q)
        cf = sres.cand_frame[q, :nc]
        assert nc > 0
        both += int(np.any(cf < F) and np.any(cf >= F))
        assert np.linalg.norm(s1.pose[cf[0] % F, :2] - qs.pose[q, :2]) < 5.0
    assert both >= 0.9 * Q                       # a place is retrieved from both sessions
    pairs = [single.result_pairs(q, sres) for q in range(0, Q, 12)]
    for k, q in enumerate(range(0, Q, 12)):       # a list's entries belong to its candidate's frame, q_idx ascends
        lo, hi = sres.pair_off[q, 0], sres.pair_off[q, 1]
        assert np.all(np.diff(pairs[k][0][lo:hi]) >= 0)
        assert np.all(single.fetch_entries(pairs[k][1][lo:hi][:128]).frame == sres.cand_frame[q, 0])
    # product sweep (f32 pre-test, sub-cell pruning, four descriptors per visit list) against the diagnostic sweep
    # (one descriptor at a time, the reference's f64 test on every entry of every gated cell) on two of the queries
    sub = single.query_frames(qs.xyz[:2], qs.label[:2])
    prod = single.stats()
    rough = single.result_rough(0)                # re-runs the two-query batch with the diagnostic sweep
    diag = single.stats()
    assert diag["last_P"] == prod["last_P"] and diag["last_M"] == prod["last_M"] and prod["last_M"] > 0
    assert len(rough["q_idx"]) > 0 and np.array_equal(sub.cand_frame, sres.cand_frame[:2])
    lo_v, v = single.result_votes(0)
    counts = np.bincount(rough["frame"].astype(np.int64) - lo_v, minlength=len(v))
    assert np.array_equal(counts[:len(v)], v)     # the diagnostic list's frames add up to the vote histogram
    single.close()
    del single
    multi = manager.STDescManager(max_frame_n=2 * F + 1, devices=[0] * 8)
    multi.add_frames(s1.xyz, s1.label)
    multi.add_frames(x2, s1.label)
    assert multi.stats()["n_entries"] == st["n_entries"]
    mres = multi.query_frames(qs.xyz, qs.label)
    np.testing.assert_array_equal(mres.n_cand, sres.n_cand)
    np.testing.assert_array_equal(mres.cand_frame, sres.cand_frame)
    np.testing.assert_array_equal(mres.cand_votes, sres.cand_votes)
    np.testing.assert_array_equal(mres.pair_off, sres.pair_off)
    for k, q in enumerate(range(0, Q, 12)):
        mq, me = multi.result_pairs(q, mres)
        np.testing.assert_array_equal(mq, pairs[k][0])
    multi.close()


def test_skewed_reference_shaped_workload_sampled_parity(mods):
    """Zipf-distributed labels over the 13 wild classes (get_json_wild.cpp:10-12), 50-400 keypoints per
    frame (ragged batches: CSR offsets, frames beyond the LDS-resident dedup take the global one), clustered landmarks;
    2 500-frame map.  Long, unevenly filled buckets: the first batch's work buffers are sized from bucket statistics that
    run far off here and are regrown by re-runs; the oracle is compared on a sample (candidates, votes, ordered lists),
    properties on the rest, and the P / M counters on the sample as a batch of its own."""
    oracle, manager, synth = mods
    F, Q = 2500, 48
    smap, world = synth.make_skewed_map(F, stream=31)
    qs = synth.make_skewed_queries(world, Q, stream=3100)
    n_kp = np.diff(smap.kp_off)
    assert n_kp.min() >= 50 and n_kp.max() <= 400 and n_kp.max() > 380 and np.bincount(smap.label).max() > 0.3 * len(smap.label)
    g = manager.STDescManager()
    g.add_frames(smap.xyz, smap.label, kp_off=smap.kp_off)
    res = g.query_frames(qs.xyz, qs.label, kp_off=qs.kp_off)
    st = g.stats()
    assert st["bucket_len_sq_over_E"] > 300          # (the uniform generator: about 50 at this map size)
    assert np.all(res.n_cand > 0)
    # (frames of 50 .. 400 keypoints: the most votes often go to a neighbouring frame that sees more of the place — the
    # candidate list must contain the place, candidate_verify picks from it)
    for q in range(Q):
        nc = _list_properties(g, res, q)
        near = np.linalg.norm(smap.pose[res.cand_frame[q, :nc], :2] - qs.pose[q, :2], axis=1) < 5.0
        assert near.any(), "no candidate of query %d lies within 5 m of its place" % q
    from sgtd_amd.synth import effective_cpus
    o = oracle.OracleManager(num_threads=max(1, effective_cpus() - 4), max_frame_n=20000)
    for f in range(F):
        x, l = smap.frame(f)
        o.build(x, l, export=False)
        o.add_last()
    assert o.counters()["E"] == st["n_entries"]
    checked = (0, 17, 30, 47)
    P = M = 0
    for q in checked:
        x, l = qs.frame(q)
        _same_as_oracle(g, o, res, q, x, l)
        c = o.counters()
        P += c["P"]; M += c["M"]
    sub = qs.take(list(checked))
    rs = g.query_frames(sub.xyz, sub.label, kp_off=sub.kp_off)
    st = g.stats()
    assert st["last_P"] == P and st["last_M"] == M and np.array_equal(rs.cand_frame, res.cand_frame[list(checked)])
    g.close()
