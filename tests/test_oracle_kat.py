"""Known-answer tests that pin the CPU oracle.

The reference has no tests or golden vectors for this path, so every expected
value below is derived by hand from the reference source (file:line cited).
"""
import numpy as np
import pytest

from oracle.oracle import Descs, OracleManager, label_code


def tri_frame(extra=None):
    """3-4-5 triangle + 7 far-away helper points so that N >= K = 10.

    Helpers sit > 50 m (descriptor_max_len) from the triangle and from each
    other, so every triplet that touches one is rejected (STDesc.cpp:204-208)."""
    pts = [(0, 0, 0), (3, 0, 0), (0, 4, 0)]
    for k in range(7):
        pts.append((1000.0 + 100.0 * k, 2000.0 + 130.0 * k, 5.0 * k))
    lab = [3, 10, 11] + [5] * 7
    return np.array(pts, np.float32), np.array(lab, np.uint32)


def test_label_code():
    # STDesc.cpp:3-16 — three 4-bit fields, low 4 bits of each int
    assert label_code(3, 10, 11) == 939
    assert label_code(17, 5, -1) == 351
    assert label_code(0, 0, 0) == 0
    assert label_code(15, 15, 15) == 4095


def test_345_triangle():
    xyz, lab = tri_frame()
    om = OracleManager()
    d = om.build(xyz, lab)
    assert d.n == 1  # reached from all three vertices, first (i,m,n) wins (:249-251)
    np.testing.assert_array_equal(d.side[0], [3.0, 4.0, 5.0])
    # A = vertex shared by shortest & middle side, B shortest & longest, C middle & longest
    np.testing.assert_array_equal(d.vertex[0], [0, 0, 0, 3, 0, 0, 0, 4, 0])
    np.testing.assert_array_equal(d.label[0], [3, 10, 11])
    np.testing.assert_array_equal(d.angle[0], [0.8, 0.6, 0.0])  # :299-301
    np.testing.assert_array_equal(d.center[0], [1.0, (0.0 + 0.0 + 4.0) / 3, 0.0])
    np.testing.assert_array_equal(d.node_id[0], [0, 1, 2])  # i=0, ranks m=1,n=2
    assert d.frame[0] == 0


def test_scale_and_frame_stamp():
    xyz, lab = tri_frame()
    om = OracleManager(std_side_resolution=0.5)
    d = om.build(xyz, lab)
    np.testing.assert_array_equal(d.side[0], [6.0, 8.0, 10.0])  # scale = 1/res (:178)
    om.add_last()                       # current_frame_id_++ first (:151)
    assert om.current_frame_id == 1
    d2 = om.build(xyz, lab)
    assert d2.frame[0] == 1             # stamped before AddSTDescs increments (:305)


def test_isosceles_no_swap():
    # a == b: strict '>' leaves the enumeration order (STDesc.cpp:220-243)
    pts = [(0, 0, 0), (3, 0, 0), (0, 3, 0)]
    for k in range(7):
        pts.append((1000.0 + 100.0 * k, 2000.0 + 130.0 * k, 5.0 * k))
    xyz = np.array(pts, np.float32)
    lab = np.array([1, 2, 3] + [5] * 7, np.uint32)
    d = OracleManager().build(xyz, lab)
    assert d.n == 1
    # i=0: p1=(0,0,0); ties in k-NN distance (3 == 3) -> lower index first: p2=pt1, p3=pt2
    # a=|p1p2|=3, b=|p1p3|=3, c=|p3p2|=3*sqrt(2): no swap at all
    assert d.side[0][0] == 3.0 and d.side[0][1] == 3.0
    # l1=(1,2,0) a:(p1,p2); l2=(1,0,3) b:(p1,p3); l3=(0,2,3) c:(p2,p3)
    # A = shared(l1,l2)=p1, B = shared(l1,l3)=p2, C = shared(l2,l3)=p3
    np.testing.assert_array_equal(d.vertex[0], [0, 0, 0, 3, 0, 0, 0, 3, 0])
    np.testing.assert_array_equal(d.label[0], [1, 2, 3])


def test_length_filter():
    xyz, lab = tri_frame()
    assert OracleManager(descriptor_min_len=3.5).build(xyz, lab).n == 0   # a=3 < min
    assert OracleManager(descriptor_max_len=4.5).build(xyz, lab).n == 0   # c=5 > max


def _one_desc(side, labels=(3, 4, 5), frame=0):
    d = Descs(1)
    d.side[0] = side
    d.label[0] = labels
    d.frame[0] = frame
    return d


def test_insert_cell_rounding():
    # (int)(side + 0.5): 2.5 -> 3, 2.49 -> 2 (STDesc.cpp:155-157)
    om = OracleManager()
    om.add(_one_desc([2.5, 2.49, 7.0]))
    keys, off, ids = om.table_dump()
    np.testing.assert_array_equal(keys[0], [3, 2, 7, label_code(3, 4, 5)])
    np.testing.assert_array_equal(off, [0, 1])


def test_probe_double_count_below_one():
    # quirk 2: for a side < 1 the probe offsets -1 and 0 both truncate to cell 0
    # ((int)(-0.45) == (int)(0.55) == 0, STDesc.cpp:359-361), so an entry stored in
    # cell 0 (side < 0.5 -> (int)(side+0.5) == 0, :155) is scanned and counted twice.
    om = OracleManager()
    om.add(_one_desc([0.45, 5.2, 5.3], frame=0))         # stored with frame_id 0
    r = om.select(_one_desc([0.55, 5.2, 5.3], frame=7))
    rm = om.rough_matches()
    assert len(rm["q_idx"]) == 2                         # one entry, matched twice
    assert om.votes()[0] == 2.0
    # voxel_round index = (x+1)*9+(y+1)*3+(z+1): x=-1,y=0,z=0 -> 4 ; x=0,y=0,z=0 -> 13
    np.testing.assert_array_equal(rm["cell"], [4, 13])
    assert len(r["cand_frame"]) == 0                     # 2 votes < 5


def test_unsigned_frame_test():
    # (src.frame_id_ - db.frame_id_) > 0 on unsigned: only equal ids are skipped (:373)
    om = OracleManager()
    om.add(_one_desc([5.1, 5.2, 5.3], frame=3))
    om.select(_one_desc([5.1, 5.2, 5.3], frame=3))
    assert om.counters()["M"] == 0
    om.select(_one_desc([5.1, 5.2, 5.3], frame=2))      # smaller id still matches
    assert om.counters()["M"] == 1


def test_vote_threshold_and_tie_order():
    # frames 0..2 are added one per AddSTDescs call; entries carry their frame id
    om = OracleManager()
    sides = np.array([[5.1 + 0.01 * k, 6.2, 7.3] for k in range(5)])

    def frame(n_entries, fid):
        d = Descs(n_entries)
        d.side[:] = sides[:n_entries]
        d.label[:] = (3, 4, 5)
        d.frame[:] = fid
        return d
    om.add(frame(4, 0))   # frame 0: 4 entries
    om.add(frame(5, 1))   # frame 1: 5 entries
    om.add(frame(5, 2))   # frame 2: 5 entries
    q = _one_desc([5.12, 6.2, 7.3], frame=3)
    r = om.select(q)
    v = om.votes()
    np.testing.assert_array_equal(v[:3], [4, 5, 5])
    # 4 votes -> no candidate; 5 -> candidate; equal votes -> lower frame id first (:424-433)
    np.testing.assert_array_equal(r["cand_frame"], [1, 2])
    np.testing.assert_array_equal(r["cand_votes"], [5, 5])
    np.testing.assert_array_equal(r["cand_off"], [0, 5, 10])
    # match list order = insertion order inside the bucket
    np.testing.assert_array_equal(r["db_entry"], [4, 5, 6, 7, 8, 9, 10, 11, 12, 13])


def test_cell_gate():
    # ||side - (cell+0.5)|| < 1.5 prunes corner cells (:366-369).
    # side (5.5,5.5,5.5): centre cell distance 0, face 1, edge sqrt2, corner sqrt3 > 1.5
    om = OracleManager()
    d = Descs(2)
    d.side[0] = [4.6, 4.6, 4.6]   # inserts into cell (5,5,5) -> wait: (int)(4.6+.5)=5
    d.side[1] = [6.4, 6.4, 6.4]   # cell (6,6,6)
    d.label[:] = (3, 4, 5)
    d.frame[:] = 0
    om.add(d)
    q = _one_desc([5.5, 5.5, 5.5], frame=1)
    q2 = Descs(1); q2.side[0] = [5.5, 5.5, 5.5]; q2.label[0] = (3, 4, 5); q2.frame[0] = 1
    om2 = OracleManager(rough_dis_threshold=10.0)
    om2.add(d)
    om2.select(q2)
    rm = om2.rough_matches()
    # cell (5,5,5) is the centre cell (index 13) and is visited; (6,6,6) is a corner
    # (index 26) whose centre (6.5,6.5,6.5) is sqrt(3) away -> gated out
    np.testing.assert_array_equal(rm["cell"], [13])
    assert om2.counters()["P"] == 1


def test_dedup_first_wins_and_identity(oracle_mod):
    from sgtd_amd import synth
    m = synth.make_map(4, 64, stream=9)
    om = OracleManager()
    counts = []
    for f in range(4):
        d = om.build(m.xyz[f], m.label[f])
        counts.append(d.n)
        # dedup key unique within the frame (STDesc.cpp:244-251)
        key = np.floor((d.side * 1000).astype(np.float32)).astype(np.int64)
        assert len(np.unique(key, axis=0)) == d.n
        # node_id order is the enumeration order (i, m, n) and strictly increasing
        nid = d.node_id.astype(np.int64)
        lin = (nid[:, 0] * 100 + nid[:, 1]) * 100 + nid[:, 2]
        assert np.all(np.diff(lin) > 0)
        # sides sorted ascending
        assert np.all(d.side[:, 0] <= d.side[:, 1]) and np.all(d.side[:, 1] <= d.side[:, 2])
        om.add_last()
    # identity query: a frame identical to map frame 2 but with a distinct id
    d = om.build(m.xyz[2], m.label[2])
    assert np.all(d.frame == 4)
    r = om.select()
    assert r["cand_frame"][0] == 2
    assert r["cand_votes"][0] >= counts[2]


def test_few_points_yield_nothing():
    xyz = np.zeros((5, 3), np.float32)
    xyz[:, 0] = np.arange(5)
    assert OracleManager().build(xyz, np.ones(5, np.uint32)).n == 0


def test_batch_insert_equals_frame_by_frame():
    """orc_add_frames (builds on all host threads, inserts in order) leaves the table and the frame
    counter exactly as the reference's loop of BuildSingleScanSTD + AddSTDescs does"""
    from sgtd_amd import synth
    m = synth.make_map(40, 60, stream=23)
    a, b = OracleManager(), OracleManager()
    for f in range(40):
        a.build(m.xyz[f], m.label[f], export=False)
        a.add_last()
    b.add_frames(m.xyz[:25], m.label[:25])
    b.add_frames(m.xyz[25:], m.label[25:])
    assert a.current_frame_id == b.current_frame_id == 40
    ka, oa, ea = a.table_dump()
    kb, ob, eb = b.table_dump()
    assert np.array_equal(ka, kb) and np.array_equal(oa, ob) and np.array_equal(ea, eb)
    q = synth.make_queries(m, 2, stream=23)
    for i in range(2):
        ra = (a.build(q.xyz[i], q.label[i], export=False), a.select())[1]
        rb = (b.build(q.xyz[i], q.label[i], export=False), b.select())[1]
        for k in ra:
            assert np.array_equal(ra[k], rb[k]), k


def test_parity_audit_counts_no_flipped_decision_on_small_workloads(tmp_path):
    """tools/parity_audit.py --quick: the oracle's loops with the inferred third-party arithmetic in its alternatives side
    by side (right association, FMA contraction, +-1 ulp on the threshold) — no decision differs, and the audit's own
    counters are alive (gate tests, visits, triplets, sides that do differ in the last bit under the right association)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "audit.json")
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "parity_audit.py"), "--quick", "--out", out, "--threads", "2"],
                          stdout=subprocess.DEVNULL)
    t = json.load(open(out))["totals"]
    assert t["visits"] > 1e6 and t["gate_tests"] > 1e5 and t["triplets"] > 1e5 and t["knn_points"] > 1e3
    assert t["match_flips"] == [0, 0, 0] and t["gate_flips"] == [0, 0, 0] and t["build_flips"] == [0, 0, 0]
    assert t["thr_ulp_flips_plus_minus"] == [0, 0] and t["near_calls"] == 0 and t["knn_tied_points"] == 0
    assert t["side_value_diffs"][0] > 0 and t["side_value_diffs"][1] == 0        # squares of f32 differences are exact: contraction alone changes nothing
    assert t["min_margin_ulps"] > 1e3


def test_verify_audit_counts_what_it_should():
    """orc_audit_verify (tools/parity_audit.py --verify): candidate_verify with two solutions of every hypothesis side by
    side.  With the restatement's own hypotheses as the "other" ones nothing differs and its score is orc_verify's; with
    LAPACK's SVD no decision differs either (the hypotheses do, in their last bits); with a hypothesis pushed 5 cm every
    counter moves."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from parity_audit import other_svd_hypotheses
    from oracle.oracle import OrcVerifyAudit
    from sgtd_amd import synth
    m = synth.make_map(120, 200, stream=2)
    q = synth.make_queries(m, 1, stream=2)
    o = OracleManager(num_threads=2)
    o.add_frames(m.xyz, m.label)
    o.build(q.xyz[0], q.label[0], export=False)
    sel = o.select()
    assert len(sel["cand_frame"]) >= 3
    same, lapack, pushed = OrcVerifyAudit(), OrcVerifyAudit(), OrcVerifyAudit()
    for c in range(3):
        own = o.verify_hyp_solutions(c)
        cov, qc, ec = o.verify_hyp_inputs(c)
        assert own.shape[0] == cov.shape[0] > 0
        o.audit_verify(c, own, same)
        rt, ratio = other_svd_hypotheses(cov, qc, ec)
        assert np.abs(rt - own).max() < 1e-8 and (ratio > 0).all()
        o.audit_verify(c, rt, lapack)
        far = own.copy()
        far[:, 9:] += 0.05
        o.audit_verify(c, far, pushed)
    for a in (same, lapack):
        assert a.candidates == 3 and a.pair_tests > 1e4 and a.vertex_tests == 3 * a.pair_tests
        assert a.vertex_flips == a.pair_flips == a.vote_list_diffs == a.best_index_diffs == a.score_diffs == a.inlier_set_diffs == 0
    assert same.max_norm_diff == 0.0 and same.max_rot_diff == 0.0 and 0 < lapack.max_rot_diff < 1e-8
    assert pushed.vertex_flips > 0 and pushed.pair_flips > 0 and pushed.vote_list_diffs > 0 and pushed.max_t_diff > 0.049
