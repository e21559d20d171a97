"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): bit-exact candidate indices, vote counts, match
lists and table layout; descriptor distances within 1e-5 (asserted bit-equal).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DESC_FIELDS = ("side", "angle", "center", "vertex", "label", "frame", "node_id")


def assert_descs_equal(g, o):
    assert g.n == o.n
    for f in DESC_FIELDS:
        a, b = getattr(g, f), getattr(o, f)
        np.testing.assert_array_equal(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64),
                                      err_msg=f)


@pytest.fixture(scope="module")
def mods():
    from oracle import oracle
    from sgtd_amd import manager, synth
    oracle.build_library()
    return oracle, manager, synth


def _pair(mods, **kw):
    oracle, manager, _ = mods
    return manager.STDescManager(**kw), oracle.OracleManager(**kw)


def test_build_parity_shipped_config(mods):
    _, _, synth = mods
    g, o = _pair(mods)
    m = synth.make_map(6, 200, stream=11)
    for f in range(6):
        assert_descs_equal(g.BuildSingleScanSTD(m.xyz[f], m.label[f]), o.build(m.xyz[f], m.label[f]))


@pytest.mark.parametrize("n_kp", [3, 9, 10, 11, 37, 64, 257, 333])
def test_build_parity_ragged_sizes(mods, n_kp):
    _, _, synth = mods
    g, o = _pair(mods)
    m = synth.make_map(2, n_kp, stream=12 + n_kp)
    for f in range(2):
        assert_descs_equal(g.BuildSingleScanSTD(m.xyz[f], m.label[f]), o.build(m.xyz[f], m.label[f]))


@pytest.mark.parametrize("cfg", [
    dict(descriptor_near_num=6, std_side_resolution=0.5, descriptor_min_len=1.0, descriptor_max_len=30.0),
    dict(descriptor_near_num=16, std_side_resolution=2.0, descriptor_min_len=0.1, descriptor_max_len=80.0),
    dict(descriptor_near_num=3),
])
def test_build_parity_other_configs(mods, cfg):
    _, _, synth = mods
    g, o = _pair(mods, **cfg)
    m = synth.make_map(2, 120, stream=21)
    for f in range(2):
        assert_descs_equal(g.BuildSingleScanSTD(m.xyz[f], m.label[f]), o.build(m.xyz[f], m.label[f]))


def test_build_exact_ties_and_duplicates(mods):
    """grid points give many exactly equal k-NN distances and equal side lengths
    (the dedup then collapses congruent triangles): tie rule = lower index first"""
    g, o = _pair(mods)
    xs, ys = np.meshgrid(np.arange(8, dtype=np.float32) * 2.0, np.arange(8, dtype=np.float32) * 2.0)
    xyz = np.stack([xs.ravel(), ys.ravel(), np.zeros(64, np.float32)], axis=1)
    xyz[5] = xyz[4]  # a duplicate point
    lab = (np.arange(64) % 9 + 3).astype(np.uint32)
    assert_descs_equal(g.BuildSingleScanSTD(xyz, lab), o.build(xyz, lab))


def _fill_both(mods, g, o, m, via_frames=True):
    F = m.xyz.shape[0]
    if via_frames:
        g.add_frames(m.xyz, m.label)
    for f in range(F):
        d = o.build(m.xyz[f], m.label[f])
        o.add_last()
        if not via_frames:
            dg = g.BuildSingleScanSTD(m.xyz[f], m.label[f])
            assert_descs_equal(dg, d)
            g.AddSTDescs(dg)
    assert g.current_frame_id_ == o.current_frame_id == F


@pytest.mark.parametrize("via_frames", [True, False])
def test_table_parity(mods, via_frames):
    _, _, synth = mods
    g, o = _pair(mods)
    m = synth.make_map(12, 150, stream=31)
    _fill_both(mods, g, o, m, via_frames)
    gk, goff, gid = g.table_dump()
    ok, ooff, oid = o.table_dump()
    np.testing.assert_array_equal(gk, ok)
    np.testing.assert_array_equal(goff, ooff)
    np.testing.assert_array_equal(gid, oid)   # the dump lists every bucket in insertion order
    ids = np.arange(0, len(gid), 97, dtype=np.int64)
    assert_descs_equal(g.fetch_entries(ids), o.fetch_entries(ids))


def _check_query(g, o, res, q, oq_descs, check_rough=True):
    r = o.select()
    nc = int(res.n_cand[q])
    np.testing.assert_array_equal(res.cand_frame[q, :nc], r["cand_frame"])
    np.testing.assert_array_equal(res.cand_votes[q, :nc], r["cand_votes"])
    np.testing.assert_array_equal(res.pair_off[q, :nc + 1], r["cand_off"])
    qi, de = g.result_pairs(q, res)
    np.testing.assert_array_equal(qi, r["q_idx"])
    np.testing.assert_array_equal(de, r["db_entry"])
    assert_descs_equal(g.result_query_descs(q), oq_descs)
    lo, v = g.result_votes(q)
    ov = o.votes()
    np.testing.assert_array_equal(v.astype(np.float64), ov[lo:lo + len(v)])
    assert ov[:lo].sum() == 0 and ov[lo + len(v):].sum() == 0
    if check_rough:
        gr, orr = g.result_rough(q), o.rough_matches()
        for k in ("q_idx", "cell", "db_entry", "frame"):
            np.testing.assert_array_equal(gr[k], orr[k], err_msg=k)
        # north_star tolerance on descriptor distances is 1e-5; we require bit equality
        np.testing.assert_array_equal(gr["dis"], orr["dis"])
        assert np.max(np.abs(gr["dis"] - orr["dis"]), initial=0.0) <= 1e-5
    return r


def test_select_parity_batch(mods):
    """a batch of query frames: every result of candidate_selector against the oracle,
    and the device's own counters of visited entries / rough matches against the oracle's"""
    _, _, synth = mods
    g, o = _pair(mods)
    m = synth.make_map(40, 200, stream=41)
    _fill_both(mods, g, o, m)
    qs = synth.make_queries(m, 6, stream=41)
    res = g.query_frames(qs.xyz, qs.label)
    st = g.stats()
    P = M = 0
    for q in range(6):
        od = o.build(qs.xyz[q], qs.label[q])
        r = _check_query(g, o, res, q, od)
        c = o.counters()
        P += c["P"]; M += c["M"]
        assert res.n_cand[q] > 0 and res.cand_frame[q, 0] == r["cand_frame"][0]
    assert st["last_P"] == P and st["last_M"] == M


def test_select_parity_ragged_batch_and_small_frames(mods):
    _, _, synth = mods
    g, o = _pair(mods)
    m = synth.make_map(16, 90, stream=51)
    _fill_both(mods, g, o, m)
    sizes = [90, 5, 33, 90, 12]           # includes a frame below K (no descriptors)
    qs = synth.make_queries(m, len(sizes), stream=51)
    xyz = np.concatenate([qs.xyz[i, :n] for i, n in enumerate(sizes)])
    lab = np.concatenate([qs.label[i, :n] for i, n in enumerate(sizes)])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    res = g.query_frames(xyz, lab, kp_off=off)
    for q, n in enumerate(sizes):
        od = o.build(qs.xyz[q, :n], qs.label[q, :n])
        _check_query(g, o, res, q, od)


def test_candidate_selector_on_descriptors(mods):
    _, _, synth = mods
    g, o = _pair(mods, rough_dis_threshold=0.05, candidate_num=7)
    m = synth.make_map(30, 100, stream=61)
    _fill_both(mods, g, o, m, via_frames=False)
    qs = synth.make_queries(m, 2, stream=61)
    for q in range(2):
        dg = g.BuildSingleScanSTD(qs.xyz[q], qs.label[q])
        o.build(qs.xyz[q], qs.label[q])
        r = o.select()
        lists = g.candidate_selector(dg)
        assert [l.match_id_[1] for l in lists] == list(r["cand_frame"])
        assert [l.votes for l in lists] == list(r["cand_votes"])
        assert all(l.match_id_[0] == o.current_frame_id for l in lists)
        np.testing.assert_array_equal(np.concatenate([l.q_idx for l in lists] or [np.zeros(0)]), r["q_idx"])
        np.testing.assert_array_equal(np.concatenate([l.db_entry for l in lists] or [np.zeros(0)]), r["db_entry"])


def test_quirks_through_the_abi(mods):
    """double count below side 1, unsigned frame test, vote threshold and tie
    order (tests/test_oracle_kat.py derives the expected values by hand)"""
    oracle, manager, _ = mods
    g = manager.STDescManager()

    def one(side, frame, labels=(3, 4, 5)):
        d = manager.Descs(1)
        d.side[0] = side; d.label[0] = labels; d.frame[0] = frame
        return d
    g.AddSTDescs(one([0.45, 5.2, 5.3], 0))
    assert g.candidate_selector(one([0.55, 5.2, 5.3], 7)) == []
    rough = g.result_rough(0)
    np.testing.assert_array_equal(rough["cell"], [4, 13])      # same bucket scanned twice
    lo, v = g.result_votes(0)
    assert lo == 0 and v[0] == 2
    g.candidate_selector(one([0.45, 5.2, 5.3], 0))             # equal frame ids never match
    assert g.stats()["last_M"] == 0

    g2 = manager.STDescManager()
    sides = np.array([[5.1 + 0.01 * k, 6.2, 7.3] for k in range(5)])
    for fid, n in ((0, 4), (1, 5), (2, 5)):
        d = manager.Descs(n)
        d.side[:] = sides[:n]; d.label[:] = (3, 4, 5); d.frame[:] = fid
        g2.AddSTDescs(d)
    lists = g2.candidate_selector(one([5.12, 6.2, 7.3], 3))
    assert [l.match_id_[1] for l in lists] == [1, 2] and [l.votes for l in lists] == [5, 5]
    np.testing.assert_array_equal(np.concatenate([l.db_entry for l in lists]), np.arange(4, 14))


def test_frame_limit_is_an_error(mods):
    _, manager, _ = mods
    g = manager.STDescManager(max_frame_n=3)
    d = manager.Descs(1)
    d.side[0] = [5.1, 6.2, 7.3]; d.label[0] = (3, 4, 5); d.frame[0] = 3
    with pytest.raises(manager.SgtdError):
        g.AddSTDescs(d)


def test_incremental_add_after_query(mods):
    _, _, synth = mods
    g, o = _pair(mods)
    m = synth.make_map(20, 120, stream=71)
    qs = synth.make_queries(m, 1, stream=71)
    for lo, hi in ((0, 8), (8, 20)):
        g.add_frames(m.xyz[lo:hi], m.label[lo:hi])
        for f in range(lo, hi):
            o.build(m.xyz[f], m.label[f]); o.add_last()
        res = g.query_frames(qs.xyz, qs.label)
        od = o.build(qs.xyz[0], qs.label[0])
        _check_query(g, o, res, 0, od)


def test_roundtrip_properties_medium_map(mods):
    """size-independent properties at a size the oracle does not cover cheaply:
    every map frame re-observed exactly is its own top-1 with >= D votes, the
    per-candidate lists are sorted by (q_idx) and reference only entries of the
    candidate's frame"""
    _, manager, synth = mods
    g = manager.STDescManager()
    m = synth.make_map(300, 200, stream=81)
    g.add_frames(m.xyz, m.label)
    pick = np.arange(0, 300, 37)
    res = g.query_frames(m.xyz[pick], m.label[pick])
    for q, f in enumerate(pick):
        d = g.result_query_descs(q)
        assert res.cand_frame[q, 0] == f and res.cand_votes[q, 0] >= d.n
        qi, de = g.result_pairs(q, res)
        lo, hi = res.pair_off[q, 0], res.pair_off[q, 1]
        assert np.all(np.diff(qi[lo:hi]) >= 0)
        ent = g.fetch_entries(de[lo:hi][:256])
        assert np.all(ent.frame == f)
        assert np.all(np.diff(res.cand_votes[q, :res.n_cand[q]]) <= 0)


def test_query_against_empty_table(mods):
    _, manager, synth = mods
    g = manager.STDescManager()
    m = synth.make_map(2, 60, stream=91)
    res = g.query_frames(m.xyz, m.label)
    assert res.n_cand.tolist() == [0, 0]
    assert g.stats()["last_M"] == 0 and g.stats()["last_P"] == 0
    d = g.BuildSingleScanSTD(m.xyz[0], m.label[0])
    assert g.candidate_selector(d) == []
    empty = manager.Descs(0)
    assert g.candidate_selector(empty) == []


@pytest.mark.parametrize("n_kp", [420, 900])
def test_large_frames_use_the_global_dedup_path(mods, n_kp):
    """frames whose dedup tables do not fit LDS (N > ~380) take the global-memory variant"""
    _, _, synth = mods
    g, o = _pair(mods)
    m = synth.make_map(3, n_kp, stream=95)
    for f in range(3):
        assert_descs_equal(g.BuildSingleScanSTD(m.xyz[f], m.label[f]), o.build(m.xyz[f], m.label[f]))
    g.add_frames(m.xyz[:2], m.label[:2])
    for f in range(2):
        o.build(m.xyz[f], m.label[f]); o.add_last()
    res = g.query_frames(m.xyz[2:3], m.label[2:3])
    od = o.build(m.xyz[2], m.label[2])
    _check_query(g, o, res, 0, od)


def test_skewed_labels_and_max_candidates(mods):
    """one dominant class (few label codes, long buckets), candidate_num = 64, K = 16"""
    _, _, synth = mods
    cfg = dict(candidate_num=64, descriptor_near_num=16, rough_dis_threshold=0.02)
    g, o = _pair(mods, **cfg)
    m = synth.make_map(70, 40, stream=97, label_lo=5, label_hi=6)
    _fill_both(mods, g, o, m)
    qs = synth.make_queries(m, 2, stream=97)
    res = g.query_frames(qs.xyz, qs.label)
    for q in range(2):
        od = o.build(qs.xyz[q], qs.label[q])
        r = _check_query(g, o, res, q, od)
        assert len(r["cand_frame"]) == 64


def test_vote_counts_beyond_the_topk_histogram_range(mods):
    """more than 8191 votes for one frame (clipped histogram bin) and heavy ties at the
    threshold (general top-k path): 9000 identical table entries per frame"""
    oracle, manager, _ = mods
    g = manager.STDescManager(candidate_num=3)
    o = oracle.OracleManager(candidate_num=3)
    for fid, n in ((0, 9000), (1, 9000), (2, 8500), (3, 9000)):
        for mk in (manager.Descs, oracle.Descs):
            d = mk(n)
            d.side[:] = (5.1, 6.2, 7.3); d.label[:] = (3, 4, 5); d.frame[:] = fid
            (g if mk is manager.Descs else o).__getattribute__("AddSTDescs" if mk is manager.Descs else "add")(d)
    q = manager.Descs(1); q.side[0] = (5.1, 6.2, 7.3); q.label[0] = (3, 4, 5); q.frame[0] = 9
    oq = oracle.Descs(1); oq.side[0] = (5.1, 6.2, 7.3); oq.label[0] = (3, 4, 5); oq.frame[0] = 9
    lists = g.candidate_selector(q)
    r = o.select(oq)
    assert [l.match_id_[1] for l in lists] == list(r["cand_frame"]) == [0, 1, 3]
    assert [l.votes for l in lists] == list(r["cand_votes"]) == [9000, 9000, 9000]
    np.testing.assert_array_equal(np.concatenate([l.db_entry for l in lists]), r["db_entry"])


def test_many_tied_frames_take_the_general_topk_path(mods):
    """1500 frames with exactly 6 votes each: more ties at the threshold than the pool holds"""
    oracle, manager, _ = mods
    g = manager.STDescManager(candidate_num=50, max_frame_n=4000)
    o = oracle.OracleManager(candidate_num=50, max_frame_n=4000)
    n_f = 1500
    dg, do = manager.Descs(6 * n_f), oracle.Descs(6 * n_f)
    for d in (dg, do):
        d.side[:] = (5.1, 6.2, 7.3); d.label[:] = (3, 4, 5)
        d.frame[:] = np.repeat(np.arange(n_f), 6)
    g.AddSTDescs(dg); o.add(do)
    q = manager.Descs(1); q.side[0] = (5.1, 6.2, 7.3); q.label[0] = (3, 4, 5); q.frame[0] = 3999
    oq = oracle.Descs(1); oq.side[0] = (5.1, 6.2, 7.3); oq.label[0] = (3, 4, 5); oq.frame[0] = 3999
    lists = g.candidate_selector(q)
    r = o.select(oq)
    assert [l.match_id_[1] for l in lists] == list(r["cand_frame"]) == list(range(50))
    assert all(l.votes == 6 for l in lists)


# ---------------------------------------------------------------------------
# candidate_verify + triangle_solver on the device vs the oracle's
# restatement (STDesc.cpp:462-571).  Same arithmetic order, no FMA => exact equality.
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n_kp,n_frames", [(40, 12), (120, 30), (200, 40)])
def test_verify_parity(mods, n_kp, n_frames):
    _, _, synth = mods
    smap = synth.make_map(n_frames, n_kp, stream=77)
    queries = synth.make_queries(smap, 6, stream=77)
    mgr, orc = _pair(mods)
    for f in range(n_frames):
        orc.build(smap.xyz[f], smap.label[f], export=False)
        orc.add_last()
        mgr.AddSTDescs(mgr.BuildSingleScanSTD(smap.xyz[f], smap.label[f]))
    mgr.query_frames(queries.xyz, queries.label)
    res = mgr.results()
    mgr.verify()
    bc, bf, bs = mgr.search_loop()
    checked = 0
    for q in range(queries.xyz.shape[0]):
        orc.build(queries.xyz[q], queries.label[q], export=False)
        sel = orc.select()
        n_c = len(sel["cand_frame"])
        assert n_c == int(res.n_cand[q])
        score, rot, t = mgr.result_verify(q)
        best_s, best_k = 0.0, -1
        for k in range(n_c):
            n_pairs = int(res.pair_off[q, k + 1] - res.pair_off[q, k])
            o_score, o_t, o_rot, o_idx = orc.verify(k, n_pairs)
            assert score[k] == o_score, (q, k)
            if o_score >= 0:
                assert np.array_equal(t[k], o_t), (q, k)
                assert np.array_equal(rot[k], o_rot), (q, k)
                assert np.array_equal(mgr.result_inliers(q, k, n_pairs), o_idx), (q, k)
                checked += 1
            if o_score > best_s:
                best_s, best_k = o_score, k
        assert np.all(score[n_c:] == -1)
        # the fused call: every candidate's inlier pairs with their table entries, one device pass
        qi_all, de_all = mgr.result_pairs(q, res)
        off, iq, ent = mgr.result_inlier_entries(q, int(res.pair_off[q, n_c]))
        for k in range(n_c):
            n_pairs = int(res.pair_off[q, k + 1] - res.pair_off[q, k])
            o_score, _, _, o_idx = orc.verify(k, n_pairs)
            a, b = int(off[k]), int(off[k + 1])
            if o_score < 0:
                continue
            assert b - a == len(o_idx)
            at = int(res.pair_off[q, k]) + np.asarray(o_idx, np.int64)
            assert np.array_equal(iq[a:b], qi_all[at])
            want = mgr.fetch_entries(de_all[at])
            assert np.array_equal(ent.side[a:b], want.side) and np.array_equal(ent.frame[a:b], want.frame)
            assert np.array_equal(ent.vertex[a:b], want.vertex) and np.array_equal(ent.node_id[a:b], want.node_id)
        if best_s > mgr.icp_threshold_:
            assert bc[q] == best_k and bf[q] == res.cand_frame[q, best_k] and bs[q] == best_s
        else:
            assert bc[q] == -1 and bf[q] == -1 and bs[q] == 0
    assert checked > 0
    # a buffer that is too small: SGTD_ERR_CAPACITY with the offsets and the needed count still delivered
    import ctypes as C
    off = np.zeros(mgr.config_setting_["candidate_num"] + 1, np.int64)
    qi = np.zeros(1, np.int32)
    need = C.c_int64(0)
    st = mgr._L.sgtd_result_inlier_entries(mgr._h, 0, off.ctypes.data_as(C.c_void_p), qi.ctypes.data_as(C.c_void_p), None, 1, C.byref(need))
    assert st == -4 and need.value > 1 and off[int(res.n_cand[0])] == need.value
    assert mgr._L.sgtd_result_inlier_entries(mgr._h, 0, None, None, None, 0, C.byref(need)) == -1      # no offsets array
    p = C.c_void_p(0)
    assert mgr._L.sgtd_host_alloc(1 << 20, C.byref(p)) == 0 and p.value
    assert mgr._L.sgtd_host_free(p) == 0 and mgr._L.sgtd_host_free(None) == 0
    mgr.close()


def test_verify_f32_pretest_equals_the_all_f64_path_at_size(mods, monkeypatch):
    """vertex A is pre-tested in packed f32 with a conservative bound and decided in f64 when in
    doubt: scores, poses and inlier flags of 64 queries x 50 candidates on a 600-frame map must
    equal the run with the pre-test switched off (SGTD_VERIFY_EXACT=1)"""
    _, manager, synth = mods
    smap = synth.make_map(600, 200, stream=41)
    qs = synth.make_queries(smap, 64, stream=41)
    out = []
    for exact in ("0", "1"):
        monkeypatch.setenv("SGTD_VERIFY_EXACT", exact)
        g = manager.STDescManager()
        g.add_frames(smap.xyz, smap.label)
        res = g.query_frames(qs.xyz, qs.label)
        g.verify()
        rows = []
        for q in range(64):
            score, rot, t = g.result_verify(q)
            nc = int(res.n_cand[q])
            inl = [g.result_inliers(q, k, int(res.pair_off[q, k + 1] - res.pair_off[q, k])) for k in range(nc) if score[k] >= 0]
            rows.append((score.copy(), rot.copy(), t.copy(), inl))
        out.append(rows)
        g.close()
    n_inl = 0
    for a, b in zip(*out):
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_array_equal(a[2], b[2])
        assert len(a[3]) == len(b[3])
        for x, y in zip(a[3], b[3]):
            np.testing.assert_array_equal(x, y)
            n_inl += len(x)
    assert n_inl > 1000


# ---------------------------------------------------------------------------
# saved table == rebuilt table, and appending after a load
# ---------------------------------------------------------------------------
def test_saved_table_round_trip_and_append(mods, tmp_path):
    _, manager, synth = mods
    smap = synth.make_map(24, 80, stream=31)
    q = synth.make_queries(smap, 5, stream=31)

    def snapshot(mgr):
        res = mgr.query_frames(q.xyz, q.label)
        pairs = [mgr.result_pairs(i, res) for i in range(5)]
        return (res.n_cand.copy(), res.cand_frame.copy(), res.cand_votes.copy(), res.pair_off.copy(),
                [p[0] for p in pairs], [p[1] for p in pairs], mgr.current_frame_id_)

    def same(a, b):
        for x, y in zip(a[:4], b[:4]):
            assert np.array_equal(x, y)
        for x, y in zip(a[4] + a[5], b[4] + b[5]):
            assert np.array_equal(x, y)
        assert a[6] == b[6]

    full = manager.STDescManager()
    full.add_frames(smap.xyz, smap.label)
    want_full = snapshot(full)

    first = manager.STDescManager()
    first.add_frames(smap.xyz[:16], smap.label[:16])
    want_first = snapshot(first)
    path = tmp_path / "map16.tbl"
    first.save_table(path)

    again = manager.STDescManager()
    again.load_table(path)
    same(snapshot(again), want_first)                      # the saved table answers like the original
    assert np.array_equal(again.table_dump()[0], first.table_dump()[0])
    again.add_frames(smap.xyz[16:], smap.label[16:])       # new session appends to the old map
    same(snapshot(again), want_full)

    other = manager.STDescManager(std_side_resolution=0.5)
    with pytest.raises(manager.SgtdError):
        other.load_table(path)                             # sides are stored scaled
    with pytest.raises(manager.SgtdError):
        other.load_table(tmp_path / "missing.tbl")
    (tmp_path / "junk.tbl").write_bytes(b"x" * 100)
    with pytest.raises(manager.SgtdError):
        again.load_table(tmp_path / "junk.tbl")
    for m in (full, first, again, other):
        m.close()


# ---------------------------------------------------------------------------
# BASELINE cfg5 ingredients: labels from 13 classes {0..12} (get_json_wild.cpp:10-12), labels
# beyond 15 (low-4-bit wrap of Combinatorial_Binary_Encoding), two "sessions" of one world
# ---------------------------------------------------------------------------
def test_wild_labels_and_two_sessions(mods):
    _, _, synth = mods
    g, o = _pair(mods)
    s1 = synth.make_map(10, 90, stream=41, label_lo=0, label_hi=12)
    s2 = synth.make_map(10, 90, stream=41, label_lo=0, label_hi=12, sigma=0.04)   # same world, other noise draw
    s2.label[:, ::7] += 16                                   # 16 + l collides with l in the 12-bit code
    for sess in (s1, s2):
        for f in range(10):
            d = g.BuildSingleScanSTD(sess.xyz[f], sess.label[f])
            assert_descs_equal(d, o.build(sess.xyz[f], sess.label[f]))
            g.AddSTDescs(d)
            o.add_last()
    q = synth.make_queries(s1, 4, stream=41)
    res = g.query_frames(q.xyz, q.label)
    for i in range(4):
        o.build(q.xyz[i], q.label[i], export=False)
        r = o.select()
        nc = int(res.n_cand[i])
        assert np.array_equal(res.cand_frame[i, :nc], r["cand_frame"]) and np.array_equal(res.cand_votes[i, :nc], r["cand_votes"])
        qi, de = g.result_pairs(i, res)
        assert np.array_equal(qi, r["q_idx"]) and np.array_equal(de, r["db_entry"])
    assert res.n_cand.max() > 0


# ---------------------------------------------------------------------------
# SURVEY §8e on one GPU: G frame-range shards (first_frame_id), local top-50 each, merged with
# the reference's rule == the single-table candidate list; owners hold the match lists
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n_shards", [2, 3])
def test_table_shards_merge_to_the_single_table_result(mods, n_shards):
    import torch
    from sgtd_amd.dist import merge_candidates, shard_range
    _, manager, synth = mods
    smap = synth.make_map(30, 100, stream=51)
    q = synth.make_queries(smap, 6, stream=51)
    single = manager.STDescManager()
    single.add_frames(smap.xyz, smap.label)
    want = single.query_frames(q.xyz, q.label)
    cn = single.config_setting_["candidate_num"]
    frames, votes, shards = [], [], []
    for r in range(n_shards):
        lo, hi = shard_range(30, n_shards, r)
        m = manager.STDescManager(first_frame_id=lo)
        m.add_frames(smap.xyz[lo:hi], smap.label[lo:hi])
        res = m.query_frames(q.xyz, q.label)
        f = np.where(np.arange(cn)[None, :] < res.n_cand[:, None], res.cand_frame, -1)
        v = np.where(np.arange(cn)[None, :] < res.n_cand[:, None], res.cand_votes, 0)
        frames.append(torch.from_numpy(f.astype(np.int32)))
        votes.append(torch.from_numpy(v.astype(np.int32)))
        shards.append((m, res, lo, hi))
    mf, mv, n = merge_candidates(torch.stack(frames), torch.stack(votes), cn)
    for i in range(6):
        nc = int(want.n_cand[i])
        assert int(n[i]) == nc
        assert np.array_equal(mf[i, :nc].numpy(), want.cand_frame[i, :nc]) and np.array_equal(mv[i, :nc].numpy(), want.cand_votes[i, :nc])
        # the owner of the global top-1 frame holds its complete match list, in the same order
        top = int(want.cand_frame[i, 0])
        m, res, lo, hi = next(s for s in shards if s[2] <= top < s[3])
        k = int(np.where(res.cand_frame[i, :res.n_cand[i]] == top)[0][0])
        qi_s, de_s = m.result_pairs(i, res)
        qi_w, de_w = single.result_pairs(i, want)
        a, b = res.pair_off[i, k], res.pair_off[i, k + 1]
        assert np.array_equal(qi_s[a:b], qi_w[want.pair_off[i, 0]:want.pair_off[i, 1]])
        ent_s = m.fetch_entries(de_s[a:b])
        ent_w = single.fetch_entries(de_w[want.pair_off[i, 0]:want.pair_off[i, 1]])
        assert np.array_equal(ent_s.side, ent_w.side) and np.array_equal(ent_s.frame, ent_w.frame)
    single.close()
    for s in shards:
        s[0].close()


# ---------------------------------------------------------------------------
# Property test: random configurations and random small worlds — descriptors,
# candidates, votes and match lists identical to the CPU restatement
# ---------------------------------------------------------------------------
def test_random_configs_property(mods):
    from hypothesis import HealthCheck, given, settings, strategies as st
    _, _, synth = mods

    @settings(max_examples=12, deadline=None, derandomize=True,
              suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
    @given(k=st.integers(3, 12), res=st.sampled_from([0.25, 0.5, 1.0, 2.0]), min_len=st.sampled_from([0.0, 0.5, 2.0]),
           max_len=st.sampled_from([15.0, 30.0, 50.0]), rough=st.sampled_from([0.01, 0.03, 0.1]),
           cand=st.integers(1, 20), n_kp=st.integers(12, 70), n_frames=st.integers(2, 9),
           labels=st.sampled_from([(3, 11), (0, 12), (5, 6), (0, 40)]), stream=st.integers(100, 10_000))
    def run(k, res, min_len, max_len, rough, cand, n_kp, n_frames, labels, stream):
        if n_kp < k:
            n_kp = k
        cfg = dict(descriptor_near_num=k, std_side_resolution=res, descriptor_min_len=min_len, descriptor_max_len=max_len,
                   rough_dis_threshold=rough, candidate_num=cand)
        g, o = _pair(mods, **cfg)
        try:
            m = synth.make_map(n_frames, n_kp, stream=stream, label_lo=labels[0], label_hi=labels[1])
            q = synth.make_queries(m, 2, stream=stream)
            for f in range(n_frames):
                d = g.BuildSingleScanSTD(m.xyz[f], m.label[f])
                assert_descs_equal(d, o.build(m.xyz[f], m.label[f]))
                g.AddSTDescs(d)
                o.add_last()
            r = g.query_frames(q.xyz, q.label)
            for i in range(2):
                o.build(q.xyz[i], q.label[i], export=False)
                want = o.select()
                nc = int(r.n_cand[i])
                assert np.array_equal(r.cand_frame[i, :nc], want["cand_frame"])
                assert np.array_equal(r.cand_votes[i, :nc], want["cand_votes"])
                qi, de = g.result_pairs(i, r)
                assert np.array_equal(qi, want["q_idx"]) and np.array_equal(de, want["db_entry"])
        finally:
            g.close()

    run()


# SearchLoop over frame-range shards (every candidate verified by its owner, results merged)
# == SearchLoop on the single table
def test_sharded_search_loop_equals_single_table(mods):
    import torch
    from sgtd_amd.dist import ShardedMap, merge_candidates, merge_verified, search_loop_choice, shard_range
    _, manager, synth = mods
    smap = synth.make_map(30, 100, stream=61)
    q = synth.make_queries(smap, 6, stream=61)
    single = manager.STDescManager()
    single.add_frames(smap.xyz, smap.label)
    want = single.query_frames(q.xyz, q.label)
    single.verify()
    w_bc, w_bf, w_bs = single.search_loop()
    cn = single.config_setting_["candidate_num"]
    dev = torch.device("cuda", 0)
    sf, sv, ss, sp, shards = [], [], [], [], []
    for r in range(3):
        lo, hi = shard_range(30, 3, r)
        m = manager.STDescManager(first_frame_id=lo)
        m.add_frames(smap.xyz[lo:hi], smap.label[lo:hi])
        m.query_frames(q.xyz, q.label, fetch=False)
        f = torch.empty((6, cn), dtype=torch.int32, device=dev)
        v = torch.empty((6, cn), dtype=torch.int32, device=dev)
        m.export_candidates(f, v)
        m.verify()
        s = torch.empty((6, cn), dtype=torch.float64, device=dev)
        p = torch.empty((6, cn, 12), dtype=torch.float64, device=dev)
        m.export_verify(s, p)
        m.sync()
        sf.append(f); sv.append(v); ss.append(s); sp.append(p); shards.append(m)
    torch.cuda.synchronize()
    gf, gv, n = merge_candidates(torch.stack(sf), torch.stack(sv), cn)
    scores, poses = merge_verified(gf, torch.stack(sf), torch.stack(ss), torch.stack(sp))
    bc, bf, bs = search_loop_choice(gf, n, scores, single.icp_threshold_)
    for i in range(6):
        w_score, w_rot, w_t = single.result_verify(i)
        nc = int(want.n_cand[i])
        assert np.array_equal(gf[i, :nc].cpu().numpy(), want.cand_frame[i, :nc])
        assert np.array_equal(scores[i].cpu().numpy(), w_score)
        got_pose = poses[i].cpu().numpy()
        assert np.array_equal(got_pose[:, :9].reshape(cn, 3, 3), w_rot) and np.array_equal(got_pose[:, 9:], w_t)
    assert np.array_equal(bc.cpu().numpy(), w_bc) and np.array_equal(bf.cpu().numpy(), w_bf) and np.array_equal(bs.cpu().numpy(), w_bs)
    # the one-rank form of the collective path
    one = ShardedMap(30, 0, 1)
    one.add_shard_frames(smap.xyz, smap.label)
    out = one.search_loop(q.xyz, q.label)
    assert np.array_equal(out[6].cpu().numpy(), w_bf) and np.array_equal(out[7].cpu().numpy(), w_bs)
    for m in shards + [single, one.mgr]:
        m.close()


@pytest.mark.gpu
def test_pass_pool_and_group_rows_grow_on_overflow(mods, monkeypatch):
    """the planner's work buffers (pass records, GroupRows) start small and are regrown by a re-run of the
    batch, like the match-record buffer: the repaired batch equals the oracle's lists"""
    oracle, manager, synth = mods
    m = synth.make_map(50, 160, stream=77)
    qs = synth.make_queries(m, 6, stream=77)
    o = oracle.OracleManager()
    o.add_frames(m.xyz, m.label)
    for env in ({"SGTD_POOL_UNITS": "64"}, {"SGTD_GROUP_CAP": "3"}, {"SGTD_POOL_UNITS": "64", "SGTD_GROUP_CAP": "1", "SGTD_REC_CAP": "2048"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = manager.STDescManager()
        for k in env:
            monkeypatch.delenv(k)
        g.add_frames(m.xyz, m.label)
        res = g.query_frames(qs.xyz, qs.label)
        assert g.stats()["overflowed"] == 1, env
        for q in range(6):
            o.build(qs.xyz[q], qs.label[q], export=False)
            r = o.select()
            nc = int(res.n_cand[q])
            assert np.array_equal(res.cand_frame[q, :nc], r["cand_frame"]) and np.array_equal(res.cand_votes[q, :nc], r["cand_votes"]), env
            qi, de = g.result_pairs(q, res)
            assert np.array_equal(qi, r["q_idx"]) and np.array_equal(de, r["db_entry"]), env
        res2 = g.query_frames(qs.xyz, qs.label)        # the grown buffers serve the next batch without a re-run
        assert g.stats()["overflowed"] == 0 and np.array_equal(res2.cand_frame, res.cand_frame)
        g.close()


def test_overflowed_batch_is_resolved_before_the_sharded_export(mods, monkeypatch):
    """a batch that outgrows the match-record buffer has empty candidate tables
    until it is re-run; ShardedMap.query / search_loop must export the repaired tables"""
    import torch
    from sgtd_amd.dist import ShardedMap
    _, manager, synth = mods
    monkeypatch.setenv("SGTD_REC_CAP", "4096")       # every non-trivial batch overflows at first
    m = synth.make_map(60, 200, stream=91)
    qs = synth.make_queries(m, 8, stream=91)
    sm = ShardedMap(60, 0, 1)
    sm.add_shard_frames(m.xyz, m.label)
    dx = torch.from_numpy(qs.xyz).cuda().contiguous()
    dl = torch.from_numpy(qs.label.astype(np.int64)).cuda().to(torch.int32).contiguous()
    frames, votes, n_cand, scores, poses, bc, bf, bs = sm.search_loop(dx, dl)
    assert sm.mgr.stats()["overflowed"] == 1
    monkeypatch.delenv("SGTD_REC_CAP")
    g = manager.STDescManager()
    g.add_frames(m.xyz, m.label)
    res = g.query_frames(qs.xyz, qs.label)
    g.verify()
    gc, gf, gs = g.search_loop()
    assert res.n_cand.max() > 0
    for q in range(8):
        nc = int(res.n_cand[q])
        assert int(n_cand[q]) == nc
        assert np.array_equal(frames[q, :nc].cpu().numpy(), res.cand_frame[q, :nc])
        assert np.array_equal(votes[q, :nc].cpu().numpy(), res.cand_votes[q, :nc])
    assert np.array_equal(bf.cpu().numpy(), gf) and np.array_equal(bs.cpu().numpy(), gs)
    # a candidate-pair buffer that is too small re-runs only the write pass
    monkeypatch.setenv("SGTD_PAIR_CAP", "256")
    p = manager.STDescManager()
    p.add_frames(m.xyz, m.label)
    pres = p.query_frames(qs.xyz, qs.label)
    assert p.stats()["overflowed"] == 1
    for q in range(8):
        assert np.array_equal(pres.pair_off[q], res.pair_off[q])
        a, b = p.result_pairs(q, pres), g.result_pairs(q, res)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ---------------------------------------------------------------------------
# probe layout: sub-cells + overflow slice inside a bucket, f32 pre-test with exact fallback
# ---------------------------------------------------------------------------
def _random_descs(oracle_mod, manager_mod, rng, n, frame, small):
    """caller-made descriptors (AddSTDescs path) concentrated in a few cells: many entries of
    one frame share a bucket, some of them close enough to match the same query descriptor"""
    if small:
        side = np.stack([rng.uniform(3, 5, n), rng.uniform(4, 6, n), rng.uniform(5, 7, n)], 1)
    else:
        side = np.stack([rng.uniform(20, 22, n), rng.uniform(30, 32, n), rng.uniform(40, 42, n)], 1)
    lab = np.where(rng.random((n, 1)) < 0.5, np.int32([[3, 4, 5]]), np.int32([[3, 4, 9]]))
    out = []
    for mod in (manager_mod, oracle_mod):
        d = mod.Descs(n)
        d.side[:] = side
        d.label[:] = lab
        d.frame[:] = frame
        d.node_id[:, 0] = np.arange(n)
        out.append(d)
    return out


def test_entry_ids_grow_with_the_largest_frame(mods):
    """an entry id holds the rank of the entry inside its frame in 13 bits until a frame needs more:
    appending a frame of 9 000 descriptors to a finalized table re-ids (and rebuilds) the table;
    candidates, votes and ordered match lists stay the reference's before and after"""
    oracle, manager, _ = mods
    rng = np.random.default_rng(91)
    g, o = _pair(mods, rough_dis_threshold=0.04, candidate_num=20)

    def check():
        for small in (True, False):
            gq, oq = _random_descs(oracle, manager, rng, 200, 99, small)
            g.candidate_selector(gq)
            res = g.results()
            r = o.select(oq)
            nc = int(res.n_cand[0])
            np.testing.assert_array_equal(res.cand_frame[0, :nc], r["cand_frame"])
            np.testing.assert_array_equal(res.cand_votes[0, :nc], r["cand_votes"])
            qi, de = g.result_pairs(0, res)
            np.testing.assert_array_equal(qi, r["q_idx"])
            np.testing.assert_array_equal(de, r["db_entry"])
        return g.stats()

    for f in range(6):
        gd, od = _random_descs(oracle, manager, rng, 700, f, small=(f % 2 == 0))
        g.AddSTDescs(gd); o.add(od)
    check()
    gd, od = _random_descs(oracle, manager, rng, 9000, 6, small=True)      # > 8192 entries in one frame
    g.AddSTDescs(gd); o.add(od)
    st = check()
    assert st["tail_entries"] == 0                                          # a change of the id width cannot live in a tail
    gd, od = _random_descs(oracle, manager, rng, 300, 7, small=False)       # and the next small append can again
    g.AddSTDescs(gd); o.add(od)
    st = check()
    assert st["tail_entries"] == 300


def test_entry_id_envelope_is_refused_loudly(mods):
    """frame span x largest frame beyond a 32-bit entry id: SGTD_ERR_UNSUPPORTED with a message, not
    a wrong answer (600 001 frames need 20 bits, a 5 000-entry frame 13)"""
    oracle, manager, _ = mods
    from sgtd_amd._lib import SgtdError
    rng = np.random.default_rng(92)
    g = manager.STDescManager(max_frame_n=700000)
    gd, _ = _random_descs(oracle, manager, rng, 5000, 0, small=True)
    g.AddSTDescs(gd)
    gd, _ = _random_descs(oracle, manager, rng, 100, 600000, small=True)
    g.AddSTDescs(gd)
    gq, _ = _random_descs(oracle, manager, rng, 50, 650000, small=True)
    with pytest.raises(SgtdError) as ei:
        g.candidate_selector(gq)
    assert ei.value.status == -6 and "entry id" in str(ei.value)
    g.close()
    # the same two frames 1 000 ids apart fit
    g = manager.STDescManager(max_frame_n=700000)
    gd, _ = _random_descs(oracle, manager, rng, 5000, 0, small=True)
    g.AddSTDescs(gd)
    gd, _ = _random_descs(oracle, manager, rng, 100, 1000, small=True)
    g.AddSTDescs(gd)
    g.candidate_selector(gq)
    g.close()


@pytest.mark.parametrize("coarse_at", [None, "0", "6"])
@pytest.mark.parametrize("monotone", [True, False])
def test_bucket_slices_keep_the_reference_order_per_frame(mods, monotone, coarse_at, monkeypatch):
    # coarse_at: the plan of a visit list with more ranges than this falls back to unpruned cells
    # (62 in production, where it needs dozens of overflow slices): 0 = every list, 6 = a mix
    if coarse_at is not None:
        monkeypatch.setenv("SGTD_COARSE_AT", coarse_at)
    oracle, manager, _ = mods
    rng = np.random.default_rng(77 if monotone else 78)
    g, o = _pair(mods, rough_dis_threshold=0.04, candidate_num=20)
    ids = list(range(14))
    if not monotone:
        rng.shuffle(ids)                      # frame ids out of insertion order: the (key, frame) pre-sort path
    for k, f in enumerate(ids):
        n = 500 if k % 5 else 1600            # 1600: (bucket, frame) runs beyond SGTD_RUN_MAX go to the overflow slice untested
        gd, od = _random_descs(oracle, manager, rng, n, f, small=(k % 2 == 0))
        if not monotone and k % 3 == 0:       # one AddSTDescs call carrying two frame ids
            gd.frame[n // 2:] = ids[(k + 5) % len(ids)]
            od.frame[n // 2:] = ids[(k + 5) % len(ids)]
        g.AddSTDescs(gd)
        o.add(od)
    gk, goff, gid = g.table_dump()
    ok, ooff, oid = o.table_dump()
    np.testing.assert_array_equal(gk, ok)
    np.testing.assert_array_equal(goff, ooff)
    np.testing.assert_array_equal(gid, oid)
    for small in (True, False):
        gq, oq = _random_descs(oracle, manager, rng, 300, 99, small)
        lists = g.candidate_selector(gq)
        res = g.results()
        r = o.select(oq)
        nc = int(res.n_cand[0])
        assert nc > 0
        np.testing.assert_array_equal(res.cand_frame[0, :nc], r["cand_frame"])
        np.testing.assert_array_equal(res.cand_votes[0, :nc], r["cand_votes"])
        qi, de = g.result_pairs(0, res)
        np.testing.assert_array_equal(qi, r["q_idx"])
        np.testing.assert_array_equal(de, r["db_entry"])     # (i, cell, j) order inside every frame's list
        c = o.counters()
        st = g.stats()
        assert st["last_P"] == c["P"] and st["last_M"] == c["M"]
        assert 0 < st["last_P_swept"] <= st["last_P"]
        lo, v = g.result_votes(0)
        np.testing.assert_array_equal(v.astype(np.float64), o.votes()[lo:lo + len(v)])
        gr, orr = g.result_rough(0), o.rough_matches()       # diagnostic sweep: full (i, cell, j) order + distances
        for key in ("q_idx", "cell", "db_entry", "frame", "dis"):
            np.testing.assert_array_equal(gr[key], orr[key], err_msg=key)
        assert len(lists) == nc


def test_f32_pretest_decides_like_the_exact_test_at_the_threshold(mods):
    """entries placed within a few ulps / 1e-7 relative of the match threshold: the votes (product
    sweep: f32 test, provisional records, exact resolution) equal the oracle's f64 decisions"""
    oracle, manager, _ = mods
    g, o = _pair(mods)
    q = np.array([10.3, 20.7, 25.9])
    thr = np.sqrt((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) * 0.03
    rels = [0.0, 1e-16, -1e-16, 3e-16, -3e-16, 1e-13, -1e-13, 1e-9, -1e-9, 1e-7, -1e-7, 3e-6, -3e-6, 1e-5, -1e-5, 1e-3, -1e-3]
    dirs = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [0.6, 0.8, 0], [0.36, 0.48, 0.8], [-1, 0, 0], [0, -0.6, 0.8]])
    f = 0
    for dvec in dirs:
        for rel in rels:
            e = q + dvec * (thr * (1.0 + rel))
            if np.any((e + 0.5).astype(int) - (q + 0.5).astype(int) > 1):
                continue
            for mod, mgr in ((manager, g), (oracle, o)):
                d = mod.Descs(1)
                d.side[0] = e
                d.label[0] = (3, 4, 5)
                d.frame[0] = f
                mgr.AddSTDescs(d) if mgr is g else mgr.add(d)
            f += 1
    assert f > 80
    gq, oq = manager.Descs(1), oracle.Descs(1)
    for d in (gq, oq):
        d.side[0] = q
        d.label[0] = (3, 4, 5)
        d.frame[0] = 100000 % 20000 + 5000
    g.candidate_selector(gq)
    o.select(oq)
    lo, v = g.result_votes(0)
    ov = o.votes()
    np.testing.assert_array_equal(v.astype(np.float64), ov[lo:lo + len(v)])
    assert 0 < v.sum() < f                      # some inside, some outside
    st = g.stats()
    assert st["last_M"] == o.counters()["M"]


# ---------------------------------------------------------------------------
# one handle over several devices (sgtd_create_multi): frame blocks dealt round robin,
# host-side merge == the single table.  Three shards on this one GPU exercise all of it.
# ---------------------------------------------------------------------------
def _same_multi_vs_single(mg, sg, mres, sres, nq):
    cn = sg.config_setting_["candidate_num"]
    np.testing.assert_array_equal(mres.n_cand, sres.n_cand)
    np.testing.assert_array_equal(mres.cand_frame, sres.cand_frame)
    np.testing.assert_array_equal(mres.cand_votes, sres.cand_votes)
    np.testing.assert_array_equal(mres.pair_off, sres.pair_off)
    for q in range(nq):
        mq, me = mg.result_pairs(q, mres)
        sq, se = sg.result_pairs(q, sres)
        np.testing.assert_array_equal(mq, sq)
        pick = np.arange(0, len(se), max(1, len(se) // 400))
        assert_descs_equal(mg.fetch_entries(me[pick]), sg.fetch_entries(se[pick]))   # the same table entries, global frame ids
        lo1, v1 = mg.result_votes(q)
        lo2, v2 = sg.result_votes(q)
        assert lo1 == lo2
        n = min(len(v1), len(v2))
        np.testing.assert_array_equal(v1[:n], v2[:n])
        assert v1[n:].sum() == 0 and v2[n:].sum() == 0


def test_multi_device_handle_equals_single_table(mods):
    _, manager, synth = mods
    m = synth.make_map(200, 120, stream=71)          # 200 frames = 3+ blocks of 64 per shard round
    qs = synth.make_queries(m, 12, stream=71)
    sg = manager.STDescManager()
    mg = manager.STDescManager(devices=[0, 0, 0])
    assert mg.device_count == 3 and sg.device_count == 1
    sg.add_frames(m.xyz, m.label)
    mg.add_frames(m.xyz[:70], m.label[:70])          # uneven calls: block boundaries inside and between
    mg.add_frames(m.xyz[70:], m.label[70:])
    assert mg.current_frame_id_ == sg.current_frame_id_ == 200
    assert mg.stats()["n_entries"] == sg.stats()["n_entries"]
    sres = sg.query_frames(qs.xyz, qs.label)
    mres = mg.query_frames(qs.xyz, qs.label)
    assert sres.n_cand.max() > 0
    _same_multi_vs_single(mg, sg, mres, sres, 12)
    # verification on the owners + SearchLoop's choice on the merged list
    sg.verify(); mg.verify()
    for q in range(12):
        s1, r1, t1 = sg.result_verify(q)
        s2, r2, t2 = mg.result_verify(q)
        np.testing.assert_array_equal(s1, s2)
        np.testing.assert_array_equal(r1, r2)
        np.testing.assert_array_equal(t1, t2)
        for k in range(int(sres.n_cand[q])):
            if s1[k] >= 0:
                n = int(sres.pair_off[q, k + 1] - sres.pair_off[q, k])
                np.testing.assert_array_equal(sg.result_inliers(q, k, n), mg.result_inliers(q, k, n))
    for a, b in zip(sg.search_loop(), mg.search_loop()):
        np.testing.assert_array_equal(a, b)
    # the per-frame adapter path: BuildSingleScanSTD + AddSTDescs frame by frame, then candidate_selector
    sg2, mg2 = manager.STDescManager(), manager.STDescManager(devices=[0, 0])
    for f in range(140):
        d1 = sg2.BuildSingleScanSTD(m.xyz[f], m.label[f])
        d2 = mg2.BuildSingleScanSTD(m.xyz[f], m.label[f])
        assert_descs_equal(d2, d1)
        sg2.AddSTDescs(d1); mg2.AddSTDescs(d2)
    dq = sg2.BuildSingleScanSTD(qs.xyz[0], qs.label[0])
    l1, l2 = sg2.candidate_selector(dq), mg2.candidate_selector(mg2.BuildSingleScanSTD(qs.xyz[0], qs.label[0]))
    assert len(l1) == len(l2) > 0
    for a, b in zip(l1, l2):
        assert a.match_id_ == b.match_id_ and a.votes == b.votes
        np.testing.assert_array_equal(a.q_idx, b.q_idx)
        assert_descs_equal(mg2.fetch_entries(b.db_entry[:50]), sg2.fetch_entries(a.db_entry[:50]))
    for x in (sg, mg, sg2, mg2):
        x.close()


def test_multi_device_handle_on_two_gpus(mods):
    """the same handle with its shards on two different GPUs (skipped on a one-GPU box)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _, manager, synth = mods
    m = synth.make_map(300, 150, stream=72)
    qs = synth.make_queries(m, 8, stream=72)
    sg = manager.STDescManager()
    mg = manager.STDescManager(devices=[0, 1])
    sg.add_frames(m.xyz, m.label); mg.add_frames(m.xyz, m.label)
    _same_multi_vs_single(mg, sg, mg.query_frames(qs.xyz, qs.label), sg.query_frames(qs.xyz, qs.label), 8)


# ---------------------------------------------------------------------------
# incremental insert: appends after a query go to a tail segment
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("rate", ["1", "256"])
def test_match_lists_that_outgrow_their_room_move(mods, monkeypatch, rate):
    # a descriptor's match list is given room for the matches it is EXPECTED to have (SGTD_REC_RATE / 256 of
    # its visit list + 256 records; measured on the batch before in production) and moves to a fresh slab
    # when it outgrows what its slab has left.  1: nearly every slab ends with a move; 256: the worst case is
    # reserved, nothing ever moves.  Undecided records are queued by their index in the list, so they survive it.
    monkeypatch.setenv("SGTD_REC_RATE", rate)
    _, manager, synth = mods
    g, o = _pair(mods, rough_dis_threshold=0.04)      # (longer lists than the default threshold gives)
    m = synth.make_map(60, 150, stream=36)
    qs = synth.make_queries(m, 4, stream=36)
    g.add_frames(m.xyz, m.label)
    for f in range(60):
        o.build(m.xyz[f], m.label[f], export=False)
        o.add_last()
    for _ in range(2):
        res = g.query_frames(qs.xyz, qs.label)
        for q in range(4):
            _check_query(g, o, res, q, o.build(qs.xyz[q], qs.label[q]))


def test_key_sharded_ablation_selects_the_same_candidates(mods):
    # SURVEY §8e's ablation (sgtd_amd/dist.py KeyShardedMap): buckets dealt to the ranks by key; the sum of the
    # ranks' vote histograms (what the all-reduce delivers) gives the single table's candidates and votes
    import torch
    from sgtd_amd.dist import KeyShardedMap, topk_from_votes
    _, manager, synth = mods
    m = synth.make_map(40, 120, stream=37)
    qs = synth.make_queries(m, 5, stream=37)
    single = manager.STDescManager()
    single.add_frames(m.xyz, m.label)
    want = single.query_frames(qs.xyz, qs.label)
    shards = [KeyShardedMap(r, 3) for r in range(3)]
    for sh in shards:
        sh.add_frames(m.xyz, m.label)
    assert sum(sh.kept for sh in shards) == single.stats()["n_entries"] and min(sh.kept for sh in shards) > 0
    votes = sum(sh.local_votes(qs.xyz, qs.label) for sh in shards)
    f, v, n = topk_from_votes(votes, shards[0].cand_num)
    for q in range(5):
        nc = int(want.n_cand[q])
        assert int(n[q]) == nc and nc > 0
        np.testing.assert_array_equal(f[q, :nc].numpy(), want.cand_frame[q, :nc])
        np.testing.assert_array_equal(v[q, :nc].numpy(), want.cand_votes[q, :nc])
    single.close()
    for sh in shards:
        sh.mgr.close()


def test_eight_byte_compact_words_keep_parity(mods, monkeypatch):
    # the compact candidate lists between the two assembly passes use 4-byte words (slot, descriptor in
    # block, rank in frame) whenever an entry's rank among its frame's entries fits 19 bits, 8-byte words
    # (slot, q_idx, id) otherwise — frames of more than 524 288 descriptors; SGTD_WIDE_PAIRS forces those
    monkeypatch.setenv("SGTD_WIDE_PAIRS", "1")
    _, manager, synth = mods
    g, o = _pair(mods)
    m = synth.make_map(40, 150, stream=35)
    qs = synth.make_queries(m, 3, stream=35)
    g.add_frames(m.xyz, m.label)
    for f in range(40):
        o.build(m.xyz[f], m.label[f], export=False)
        o.add_last()
    res = g.query_frames(qs.xyz, qs.label)
    for q in range(3):
        _check_query(g, o, res, q, o.build(qs.xyz[q], qs.label[q]))


@pytest.mark.parametrize("plan", [None, ("0", "62"), ("0", "0"), ("6", "12")])
def test_appends_go_to_a_tail_segment_and_keep_parity(mods, monkeypatch, plan):
    # plan: (SGTD_COARSE_AT, SGTD_WHOLE_AT) — the planner's fallbacks for visit lists of more ranges than
    # the sweep has lanes: a cell's halves as one range, then (only with a tail: two segments' ranges per
    # cell) a whole bucket as one range; 62 / 62 in production
    if plan is not None:
        monkeypatch.setenv("SGTD_COARSE_AT", plan[0])
        monkeypatch.setenv("SGTD_WHOLE_AT", plan[1])
        monkeypatch.setenv("SGTD_TAIL_MAX", "1000000")   # (and no room reserved behind the main segment: the tail's build moves the layout)
    _, manager, synth = mods
    g, o = _pair(mods)
    m = synth.make_map(60, 150, stream=33)
    qs = synth.make_queries(m, 3, stream=33)

    def add(lo, hi):
        g.add_frames(m.xyz[lo:hi], m.label[lo:hi])
        for f in range(lo, hi):
            o.build(m.xyz[f], m.label[f], export=False)
            o.add_last()

    def check():
        res = g.query_frames(qs.xyz, qs.label)
        for q in range(3):
            od = o.build(qs.xyz[q], qs.label[q])
            _check_query(g, o, res, q, od)        # candidates, votes, lists, rough list + distances
        return g.stats()

    add(0, 40)
    st = check()
    assert st["tail_entries"] == 0
    main_entries = st["n_entries"]
    add(40, 50)                                   # appended to a finalized table: only these are sorted
    st = check()
    assert st["tail_entries"] == st["n_entries"] - main_entries > 0
    add(50, 60)                                   # the tail grows (rebuilt from the first appended entry)
    st = check()
    assert st["tail_entries"] == st["n_entries"] - main_entries
    for _ in range(4):                            # a tail nobody appends to any more is merged after a few batches
        st = check()
    assert st["tail_entries"] == 0
    gk, goff, gid = g.table_dump()                # the dump merges the segments: whole buckets, insertion order
    ok, ooff, oid = o.table_dump()
    np.testing.assert_array_equal(gk, ok)
    np.testing.assert_array_equal(goff, ooff)
    np.testing.assert_array_equal(gid, oid)
    assert g.stats()["tail_entries"] == 0
    check()
    # descriptors stamped with an OLD frame id cannot live in a tail (a frame's entries stay in one
    # segment): the append is merged
    d = g.BuildSingleScanSTD(qs.xyz[0], qs.label[0])
    d.frame[:] = 7
    od = o.build(qs.xyz[0], qs.label[0])
    od.frame[:] = 7
    g.AddSTDescs(d); o.add(od)
    st = check()
    assert st["tail_entries"] == 0
    # a tail that outgrows its bound is merged
    monkeypatch.setenv("SGTD_TAIL_MAX", "2000")
    g2 = manager.STDescManager()
    g2.add_frames(m.xyz[:40], m.label[:40]); g2.query_frames(qs.xyz, qs.label)
    g2.add_frames(m.xyz[40:41], m.label[40:41])   # one frame's descriptors (> 2000) exceed the bound
    r2 = g2.query_frames(qs.xyz, qs.label)
    assert g2.stats()["tail_entries"] == 0 and r2.n_cand.max() > 0
