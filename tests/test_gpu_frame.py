"""The one-frame-per-call path (the reference's call pattern, semantic_graph_localization.cpp:590-603): sgtd_build's
one-block transfer form and sgtd_search_frame — candidate_selector + candidate_verify + the inlier pairs with their
table entries in one call — against the calls they stand for, value for value, and against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from oracle import oracle
    from sgtd_amd import manager, synth
    oracle.build_library()
    return oracle, manager, synth


def _five_calls(g, d):
    """sgtd_query_descs + sgtd_result_candidates + sgtd_result_pairs + sgtd_verify + sgtd_result_verify + sgtd_result_inlier_entries"""
    cands = g.candidate_selector(d)
    res = g.results()
    g.verify()
    score, rot, t = g.result_verify(0)
    off, qi, ent = g.result_inlier_entries(0, int(res.pair_off[0, -1]))
    return cands, res, score, rot, t, off, qi, ent


def _same(fs, ref, cn):
    cands, res, score, rot, t, off, qi, ent = ref
    nc = int(res.n_cand[0])
    assert fs["n_cand"] == nc
    assert np.array_equal(fs["cand_frame"], res.cand_frame[0]) and np.array_equal(fs["cand_votes"], res.cand_votes[0])
    assert np.array_equal(fs["pair_off"], res.pair_off[0])
    assert np.array_equal(fs["score"], score) and np.array_equal(fs["rot"], rot) and np.array_equal(fs["t"], t)
    assert np.array_equal(fs["inlier_off"], off) and fs["n_inliers"] == len(qi)
    if fs["status"] == 0:
        assert np.array_equal(fs["inlier_q_idx"], qi)
        for name, _, _ in ent.FIELDS:
            assert np.array_equal(getattr(fs["entries"], name), getattr(ent, name)), name


def test_search_frame_equals_the_calls_it_stands_for(mods, monkeypatch):
    oracle, manager, synth = mods
    m = synth.make_map(160, 200, stream=131)
    qs = synth.make_queries(m, 6, stream=131)
    g = manager.STDescManager()
    g.add_frames(m.xyz, m.label)
    o = oracle.OracleManager()
    o.add_frames(m.xyz, m.label)
    cn = g.config_setting_["candidate_num"]
    for q in range(6):
        d = g.BuildSingleScanSTD(qs.xyz[q], qs.label[q])
        od = o.build(qs.xyz[q], qs.label[q])
        assert d.n == od.n and np.array_equal(d.side, od.side) and np.array_equal(d.vertex, od.vertex) and np.array_equal(d.node_id, od.node_id)
        ref = _five_calls(g, d)
        room = int(ref[1].pair_off[0, -1])           # (every pair an inlier: always enough)
        before = g.stats()["batches_total"]
        fs = g.search_frame(d, capacity=room)
        assert fs["status"] == 0 and fs["n_inliers"] > 0
        _same(fs, ref, cn)
        # the one-wait path leaves the handle's running totals and stage times as a waited batch does
        st = g.stats()
        assert st["batches_total"] == before + 1 and st["last_queries"] == 1 and st["ms_total"] >= 0
        # the caller's arrays page-locked (sgtd_host_alloc): the device writes the inlier pairs in place — the same pairs, with room
        # to spare, with exactly enough, and with one pair too little (then nothing but the pairs is missing)
        for cap in (room, fs["n_inliers"]):
            pl = g.search_frame(d, capacity=cap, page_locked=True)
            assert pl["status"] == 0
            _same(pl, ref, cn)
        # candidate_selector alone in the one call (SGTD_FRAME_LISTS_ONLY): every pair of every match list with its table entry, in the
        # reference's order — the four calls it stands for give the same; ordinary and page-locked arrays; too little room
        qi_all, de_all = g.result_pairs(0, ref[1])
        ent_all = g.fetch_entries(de_all)
        for pl_mem in (False, True):
            lo = g.search_frame(d, capacity=room, page_locked=pl_mem, lists_only=True)
            assert lo["status"] == 0 and lo["n_cand"] == int(ref[1].n_cand[0]) and lo["n_inliers"] == room
            assert np.array_equal(lo["cand_frame"], ref[1].cand_frame[0]) and np.array_equal(lo["cand_votes"], ref[1].cand_votes[0])
            assert np.array_equal(lo["pair_off"], ref[1].pair_off[0]) and np.array_equal(lo["inlier_off"], ref[1].pair_off[0])
            assert np.array_equal(lo["inlier_q_idx"], qi_all)
            for name, _, _ in ent_all.FIELDS:
                assert np.array_equal(getattr(lo["entries"], name), getattr(ent_all, name)), name
        short = g.search_frame(d, capacity=room - 1, page_locked=True, lists_only=True)
        assert short["status"] == -4 and short["n_inliers"] == room
        # (the handle is left as sgtd_query_descs leaves it: verification is a call of its own again)
        g.verify()
        sc2, _, _ = g.result_verify(0)
        assert np.array_equal(sc2, fs["score"])
        tight = g.search_frame(d, capacity=fs["n_inliers"] - 1, page_locked=True)
        assert tight["status"] == -4 and tight["n_inliers"] == fs["n_inliers"]
        _same(tight, ref, cn)
        # the handle is left as the five calls leave it: the lists of the same batch can still be read, and are the oracle's
        r = o.select()
        qi, de = g.result_pairs(0, g.results())
        assert np.array_equal(qi, r["q_idx"]) and np.array_equal(de, r["db_entry"])
        # too little room for the inlier pairs: everything else is valid, the pairs come from the second call
        small = g.search_frame(d, capacity=7)
        assert small["status"] == -4 and small["n_inliers"] == fs["n_inliers"]
        _same(small, ref, cn)
        off, qi2, ent2 = g.result_inlier_entries(0, small["n_inliers"])
        assert np.array_equal(qi2, fs["inlier_q_idx"]) and np.array_equal(ent2.side, fs["entries"].side)
    # an empty frame, a frame without a candidate
    none = g.search_frame(g.BuildSingleScanSTD(qs.xyz[0][:5], qs.label[0][:5]))
    assert none["status"] == 0 and none["n_cand"] == 0 and none["n_inliers"] == 0
    far = qs.xyz[1] * 3.7
    fr = g.search_frame(g.BuildSingleScanSTD(far, qs.label[1]))
    assert fr["status"] == 0 and fr["n_inliers"] == 0 and np.all(fr["score"][:fr["n_cand"]] <= 0)
    g.close()
    # a first frame that outgrows its work buffers: the call falls back to the re-run and gives the same
    monkeypatch.setenv("SGTD_REC_CAP", "4096")
    h = manager.STDescManager()
    monkeypatch.delenv("SGTD_REC_CAP")
    h.add_frames(m.xyz, m.label)
    d = h.BuildSingleScanSTD(qs.xyz[2], qs.label[2])
    fs = h.search_frame(d, capacity=1 << 20, page_locked=True)
    assert h.stats()["overflowed"] == 1 and fs["status"] == 0
    _same(fs, _five_calls(h, d), cn)
    h.close()
    # ... and the same for the lists alone
    monkeypatch.setenv("SGTD_REC_CAP", "4096")
    h = manager.STDescManager()
    monkeypatch.delenv("SGTD_REC_CAP")
    h.add_frames(m.xyz, m.label)
    d = h.BuildSingleScanSTD(qs.xyz[2], qs.label[2])
    lo = h.search_frame(d, capacity=1 << 20, lists_only=True)
    assert h.stats()["overflowed"] == 1 and lo["status"] == 0
    h.candidate_selector(d)
    r1 = h.results()
    qi_all, de_all = h.result_pairs(0, r1)
    assert lo["n_inliers"] == len(qi_all) and np.array_equal(lo["inlier_q_idx"], qi_all) and np.array_equal(lo["entries"].side, h.fetch_entries(de_all).side)
    h.close()


def test_build_one_block_form_and_the_general_form_agree(mods):
    """sgtd_build of one frame takes the one-transfer form up to 4 MB of descriptors and the general form beyond (more
    than ~850 keypoints: also the global-memory dedup); both against the oracle"""
    oracle, manager, synth = mods
    rng = np.random.default_rng(7)
    g = manager.STDescManager()
    o = oracle.OracleManager()
    for n in (9, 10, 57, 200, 400, 900):
        xyz = (rng.random((n, 3)) * np.array([60.0, 60.0, 3.0])).astype(np.float32)
        lab = rng.integers(3, 12, n).astype(np.uint32)
        if synth.has_knn_ties(xyz, 10):
            continue
        d = g.BuildSingleScanSTD(xyz, lab)
        od = o.build(xyz, lab)
        assert d.n == od.n, n
        for name in ("side", "angle", "center", "vertex", "label", "frame", "node_id"):
            assert np.array_equal(getattr(d, name), getattr(od, name)), (n, name)
    g.close()


def test_search_frame_when_the_inlier_pairs_outgrow_the_gather_room_twice(mods):
    """found by tools/stress_parity.py (seed 5001, round 1773): frames that are exact copies of each other give tens of
    thousands of inlier pairs; the first call grows the gather's room to 1.5 x its count — not a multiple of the launch's
    256 threads — and a later frame with more pairs than that room must not write past it"""
    oracle, manager, synth = mods
    cfg = dict(descriptor_near_num=10, std_side_resolution=2.0, rough_dis_threshold=0.03, candidate_num=22)
    m = synth.make_map(28, 113, stream=78507, sigma=1e-4)
    q = synth.make_queries(m, 3, stream=78507)
    g = manager.STDescManager(**cfg)
    g.add_frames(m.xyz[:14], m.label[:14])
    counts = []
    for stage in range(2):
        if stage == 1:
            g.add_frames(m.xyz[14:], m.label[14:])
        for i in range(3):
            d = g.BuildSingleScanSTD(q.xyz[i], q.label[i])
            ref = _five_calls(g, d)
            fs = g.search_frame(d, capacity=int(ref[1].pair_off[0, -1]))
            assert fs["status"] == 0
            _same(fs, ref, cfg["candidate_num"])
            counts.append(fs["n_inliers"])
    assert max(counts) > 16384 and counts[-1] != counts[0]
    g.close()


def test_one_frame_ordering_in_one_launch_equals_the_general_form(mods, monkeypatch):
    """a one-frame batch orders its descriptors by home key in ONE launch (small_order_kernel: counters cleared, keys, a radix
    sort in LDS, group ids, pass slots) where the general form takes twenty-seven; both against the oracle and against each
    other — candidates, votes, ordered match lists, the ordered rough list (the sweep's order shows there) and the counters"""
    oracle, manager, synth = mods
    m = synth.make_map(120, 200, stream=141)
    qs = synth.make_queries(m, 5, stream=141)
    o = oracle.OracleManager()
    o.add_frames(m.xyz, m.label)
    got = {}
    for form in ("1", "0"):
        monkeypatch.setenv("SGTD_SMALL_ORDER", form)
        g = manager.STDescManager()
        g.add_frames(m.xyz, m.label)
        rows = []
        for q in range(5):
            res = g.query_frames(qs.xyz[q:q + 1], qs.label[q:q + 1])
            st = g.stats()
            o.build(qs.xyz[q], qs.label[q], export=False)
            r = o.select()
            nc = int(res.n_cand[0])
            assert nc == len(r["cand_frame"]) and np.array_equal(res.cand_frame[0, :nc], r["cand_frame"]) and np.array_equal(res.cand_votes[0, :nc], r["cand_votes"])
            qi, de = g.result_pairs(0, res)
            assert np.array_equal(qi, r["q_idx"]) and np.array_equal(de, r["db_entry"])
            rows.append((res.cand_frame.copy(), res.cand_votes.copy(), qi, de, st["last_P"], st["last_M"], st["last_D"]))
        # a frame with very few descriptors and an empty one
        few = g.query_frames(qs.xyz[0:1, :12], qs.label[0:1, :12])
        rows.append((few.cand_frame.copy(), few.n_cand.copy()))
        got[form] = rows
        g.close()
    for a, b in zip(got["1"], got["0"]):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
