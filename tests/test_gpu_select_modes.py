"""GPU parity of the per-query passes over the match records (sgtd_amd/csrc/select_kernels.hip.h:
votes + top-k in one launch, the match lists in one launch, STDesc.cpp:404-453).

Production picks them for batches with at least one query frame per CU; SGTD_SELECT_MODE=2
forces them for every batch (1: the five-kernel form), so the small oracle-sized cases of
test_gpu_parity.py run through them here, and a batch large enough for the automatic choice is
compared query by query with the five-kernel form and, on a sample, with the oracle.
"""
import numpy as np
import pytest

import test_gpu_parity as P
from test_gpu_parity import mods  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def per_query_mode(request, monkeypatch):
    if "auto_mode" not in request.keywords:
        monkeypatch.setenv("SGTD_SELECT_MODE", "2")


# the oracle-sized cases of test_gpu_parity.py, through votes_topk_kernel / pairs_query_kernel
test_select_parity_batch = P.test_select_parity_batch
test_select_parity_ragged_batch_and_small_frames = P.test_select_parity_ragged_batch_and_small_frames
test_candidate_selector_on_descriptors = P.test_candidate_selector_on_descriptors
test_quirks_through_the_abi = P.test_quirks_through_the_abi
test_query_against_empty_table = P.test_query_against_empty_table
test_skewed_labels_and_max_candidates = P.test_skewed_labels_and_max_candidates
test_vote_counts_beyond_the_topk_histogram_range = P.test_vote_counts_beyond_the_topk_histogram_range
test_many_tied_frames_take_the_general_topk_path = P.test_many_tied_frames_take_the_general_topk_path
test_wild_labels_and_two_sessions = P.test_wild_labels_and_two_sessions
test_entry_ids_grow_with_the_largest_frame = P.test_entry_ids_grow_with_the_largest_frame
test_f32_pretest_decides_like_the_exact_test_at_the_threshold = P.test_f32_pretest_decides_like_the_exact_test_at_the_threshold


@pytest.mark.parametrize("monotone", [True, False])
def test_bucket_slices_keep_the_reference_order_per_frame(mods, monotone, monkeypatch):
    # (monotone = False: frame ids out of insertion order — the lists' entries go through IdMap::by_frame)
    P.test_bucket_slices_keep_the_reference_order_per_frame(mods, monotone, None, monkeypatch)


@pytest.mark.parametrize("rate", ["1"])
def test_match_lists_that_outgrow_their_room_move(mods, monkeypatch, rate):
    P.test_match_lists_that_outgrow_their_room_move(mods, monkeypatch, rate)


def test_appends_go_to_a_tail_segment_and_keep_parity(mods, monkeypatch):
    P.test_appends_go_to_a_tail_segment_and_keep_parity(mods, monkeypatch, None)


def test_verify_parity(mods):
    P.test_verify_parity(mods, 120, 30)


def test_pair_buffer_that_is_too_small_reruns_the_list_pass(mods, monkeypatch):
    """SGTD_PAIR_CAP: the first batch's candidate pairs do not fit — query_base_kernel raises the
    flag, sgtd_sync grows the buffer and runs the list pass again"""
    monkeypatch.setenv("SGTD_PAIR_CAP", "64")
    _, _, synth = mods
    g, o = P._pair(mods)
    m = synth.make_map(40, 200, stream=141)
    P._fill_both(mods, g, o, m)
    qs = synth.make_queries(m, 5, stream=141)
    res = g.query_frames(qs.xyz, qs.label)
    assert g.stats()["overflowed"] == 1 and g.stats()["rewrites_total"] >= 1 and g.stats()["select_form"] == 2
    for q in range(5):
        P._check_query(g, o, res, q, o.build(qs.xyz[q], qs.label[q]), check_rough=False)


def test_long_lists_span_many_tiles_and_super_blocks(mods):
    """frames of 700 keypoints: 15 000 descriptors per query = 30 super-blocks of 512 lists; a wide
    threshold gives lists of hundreds of records, so a wave's share of a tile starts in the
    middle of a list and a tile in the middle of a descriptor"""
    _, manager, synth = mods
    g, o = P._pair(mods, rough_dis_threshold=0.06)
    m = synth.make_map(24, 700, stream=142)
    g.add_frames(m.xyz, m.label)
    for f in range(24):
        o.build(m.xyz[f], m.label[f], export=False)
        o.add_last()
    qs = synth.make_queries(m, 3, stream=142)
    res = g.query_frames(qs.xyz, qs.label)
    for q in range(3):
        r = P._check_query(g, o, res, q, o.build(qs.xyz[q], qs.label[q]), check_rough=False)
        assert len(r["q_idx"]) > 3 * 8192      # more than three tiles of candidates alone


@pytest.mark.auto_mode
def test_batch_with_a_query_per_cu_takes_the_per_query_passes_and_equals_the_block_passes(mods, monkeypatch):
    """320 query frames (>= one per CU): the automatic choice is the per-query form; every query's
    candidates, votes, offsets and ordered match lists equal the five-kernel form's, a sample equals
    the oracle's, and the device's counters equal in both forms"""
    _, manager, synth = mods
    m = synth.make_map(150, 200, stream=143)
    qs = synth.make_queries(m, 320, stream=143)
    out = {}
    for mode in ("auto", "1"):
        if mode != "auto":
            monkeypatch.setenv("SGTD_SELECT_MODE", mode)
        g = manager.STDescManager()
        g.add_frames(m.xyz, m.label)
        res = g.query_frames(qs.xyz, qs.label)
        st = g.stats()
        out[mode] = (g, res, st)
    (ga, ra, sa), (gb, rb, sb) = out["auto"], out["1"]
    assert sa["select_form"] == 2 and sb["select_form"] == 0
    assert sa["last_M"] == sb["last_M"] and sa["last_P"] == sb["last_P"] and sa["last_cand_pairs"] == sb["last_cand_pairs"]
    np.testing.assert_array_equal(ra.n_cand, rb.n_cand)
    np.testing.assert_array_equal(ra.cand_frame, rb.cand_frame)
    np.testing.assert_array_equal(ra.cand_votes, rb.cand_votes)
    np.testing.assert_array_equal(ra.pair_off, rb.pair_off)
    for q in range(320):
        qa, ea = ga.result_pairs(q, ra)
        qb, eb = gb.result_pairs(q, rb)
        np.testing.assert_array_equal(qa, qb)
        np.testing.assert_array_equal(ea, eb)
        if q % 40 == 0:
            la, va = ga.result_votes(q)
            lb, vb = gb.result_votes(q)
            assert la == lb
            np.testing.assert_array_equal(va, vb)
    o = mods[0].OracleManager()
    for f in range(150):
        o.build(m.xyz[f], m.label[f], export=False)
        o.add_last()
    for q in (0, 77, 319):
        P._check_query(ga, o, ra, q, o.build(qs.xyz[q], qs.label[q]), check_rough=False)
    ga.close(); gb.close()


def test_frame_span_that_nearly_fills_lds(mods):
    """frame ids spread over 28 761 (caller-stamped): the query's vote histogram takes 115 KB of the 160 KB of LDS and
    votes_topk_kernel still runs it (140 KB with its top-k scratch), pairs_query_kernel's byte table 28 KB"""
    oracle, manager, synth = mods
    g = manager.STDescManager(max_frame_n=32000)
    o = oracle.OracleManager(max_frame_n=32000)
    m = synth.make_map(24, 150, stream=145)
    for f in range(24):
        d = g.BuildSingleScanSTD(m.xyz[f], m.label[f])
        od = o.build(m.xyz[f], m.label[f])
        fid = 11 + f * 1250
        d.frame[:] = fid; od.frame[:] = fid
        g.AddSTDescs(d); o.add(od)
    qs = synth.make_queries(m, 3, stream=145)
    res = g.query_frames(qs.xyz, qs.label)
    assert g.stats()["select_form"] == 2
    for q in range(3):
        o.build(qs.xyz[q], qs.label[q], export=False)
        r = o.select()
        nc = int(res.n_cand[q])
        assert nc > 0
        np.testing.assert_array_equal(res.cand_frame[q, :nc], r["cand_frame"])
        np.testing.assert_array_equal(res.cand_votes[q, :nc], r["cand_votes"])
        qi, de = g.result_pairs(q, res)
        np.testing.assert_array_equal(qi, r["q_idx"])
        np.testing.assert_array_equal(de, r["db_entry"])
        lo, v = g.result_votes(q)
        np.testing.assert_array_equal(v.astype(np.float64), o.votes()[lo:lo + len(v)])


@pytest.mark.parametrize("stride", [6521, 5200], ids=["span150k", "span120k"])
def test_frame_spans_beyond_lds_take_the_tiled_votes_and_the_candidates_hash(mods, stride):
    """frame ids spread over 150 000 (caller-stamped): the vote histogram of a query does not fit LDS — votes in frame
    tiles by votes_query_kernel, top-k by topk_kernel, the lists' offsets by cand_prefix_kernel — and neither does the
    frame -> slot byte table of pairs_query_kernel, which looks the candidates up in their 256-entry hash instead.
    span120k: 119 611 frames — the byte table and the tile image together are within the 150 KB the launch code allowed but,
    with the kernel's 12.6 KB of static LDS, beyond the 160 KB a workgroup has (the launch used to fail there)"""
    oracle, manager, synth = mods
    g = manager.STDescManager(max_frame_n=200000)
    o = oracle.OracleManager(max_frame_n=200000)
    m = synth.make_map(24, 150, stream=144)
    for f in range(24):
        d = g.BuildSingleScanSTD(m.xyz[f], m.label[f])
        od = o.build(m.xyz[f], m.label[f])
        fid = 11 + f * stride                    # ascending, up to 149 994 / 119 611
        d.frame[:] = fid; od.frame[:] = fid
        g.AddSTDescs(d); o.add(od)
    qs = synth.make_queries(m, 4, stream=144)
    res = g.query_frames(qs.xyz, qs.label)
    st = g.stats()
    assert st["select_form"] == 1                # the lists by one workgroup per query, votes and top-k by their own kernels
    for q in range(4):
        o.build(qs.xyz[q], qs.label[q], export=False)
        r = o.select()
        nc = int(res.n_cand[q])
        assert nc > 0
        np.testing.assert_array_equal(res.cand_frame[q, :nc], r["cand_frame"])
        np.testing.assert_array_equal(res.cand_votes[q, :nc], r["cand_votes"])
        np.testing.assert_array_equal(res.pair_off[q, :nc + 1], r["cand_off"])
        qi, de = g.result_pairs(q, res)
        np.testing.assert_array_equal(qi, r["q_idx"])
        np.testing.assert_array_equal(de, r["db_entry"])
        lo, v = g.result_votes(q)
        ov = o.votes()
        np.testing.assert_array_equal(v.astype(np.float64), ov[lo:lo + len(v)])
