"""The device side of the multi-GPU step (sgtd_amd/csrc/exchange_kernels.hip.h behind the C ABI): the merge kernel
against the reference's rule (STDesc.cpp:423-433) restated with torch, the packed export that leaves the pipeline
before the match lists, lists and verification for the merge's winners only — three shards in ONE process here, so
that every step can be compared with a single table (the multi-process form: tests/test_dist_gpu.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from sgtd_amd import manager, synth
    return manager, synth


def _packed(frames, votes, flag=0, nq=None, cn=None):
    import torch
    w, q, c = frames.shape
    rows = []
    for t in range(w):
        rows.append(torch.cat([frames[t].reshape(-1), votes[t].reshape(-1),
                               torch.tensor([flag if t == w - 1 else 0, q if nq is None else nq, c if cn is None else cn, 7], dtype=torch.int32)]))
    return torch.cat(rows).to(torch.int32).cuda().contiguous()


@pytest.mark.parametrize("n_tables,cn", [(1, 50), (2, 50), (3, 6), (8, 50), (16, 64), (5, 17)])
def test_merge_kernel_is_the_reference_rule(mods, n_tables, cn):
    import torch
    from sgtd_amd.dist import merge_candidates
    manager, _ = mods
    g = manager.STDescManager(candidate_num=cn)
    rng = np.random.default_rng(100 * n_tables + cn)
    nq = 37
    frames = np.full((n_tables, nq, cn), -1, np.int32)
    votes = np.zeros((n_tables, nq, cn), np.int32)
    for t in range(n_tables):
        for q in range(nq):
            k = int(rng.integers(0, cn + 1))
            f = rng.choice(np.arange(t * 1000, (t + 1) * 1000), k, replace=False)        # disjoint frame ranges
            v = rng.integers(3, 12, k)                                                   # many ties, some below 5 votes
            order = np.lexsort((f, -v))                                                  # a shard's own list: votes desc, frame asc
            frames[t, q, :k], votes[t, q, :k] = f[order], v[order]
    frames[:, 5] = -1; votes[:, 5] = 0                                                   # a query without any candidate
    tf, tv = torch.from_numpy(frames), torch.from_numpy(votes)
    want_f, want_v, want_n = merge_candidates(tf, tv, cn)
    dev = torch.device("cuda", 0)
    gathered = _packed(tf, tv)
    for my in sorted({0, n_tables - 1, -1}):
        of = torch.empty((nq, cn), dtype=torch.int32, device=dev)
        ov = torch.empty_like(of); osrc = torch.empty_like(of)
        on = torch.empty(nq, dtype=torch.int32, device=dev)
        keep = torch.empty(nq, dtype=torch.int64, device=dev)
        flags = torch.full((4,), 9, dtype=torch.int32, device=dev)
        g.merge_candidates_dev(0, gathered, n_tables, my, nq, of, ov, on, osrc, keep, flags)
        torch.cuda.synchronize()
        assert int(flags[0]) == 0
        assert torch.equal(of.cpu(), want_f) and torch.equal(ov.cpu(), want_v) and torch.equal(on.cpu(), want_n)
        src = osrc.cpu().numpy()
        kp = keep.cpu().numpy().astype(np.uint64)
        for q in range(nq):
            n = int(want_n[q])
            assert (src[q, n:] == -1).all()
            mine = 0
            for k in range(n):
                t, s = src[q, k] >> 8, src[q, k] & 255
                assert frames[t, q, s] == int(want_f[q, k]) and votes[t, q, s] == int(want_v[q, k])
                if t == my:
                    mine |= 1 << int(s)
            assert int(kp[q]) == mine
    # a table from a batch that outgrew a work buffer, tables of another shape: reported, not merged silently
    flags = torch.zeros(4, dtype=torch.int32, device=dev)
    g.merge_candidates_dev(0, _packed(tf, tv, flag=1), n_tables, 0, nq, of, ov, on, osrc, keep, flags)
    torch.cuda.synchronize()
    assert int(flags[0]) == 1
    g.merge_candidates_dev(0, _packed(tf, tv, nq=nq + 1), n_tables, 0, nq, of, ov, on, osrc, keep, flags)
    torch.cuda.synchronize()
    assert int(flags[0]) & 2
    g.close()


@pytest.mark.parametrize("mode", ["1", "2"])
def test_three_shards_through_the_exchange_kernels_equal_a_single_table(mods, monkeypatch, mode):
    """export (before the lists) -> merge kernel -> lists of the winners only -> verification of the winners only ->
    results out of the owners' tables, against one table over all frames; SGTD_SELECT_MODE 1: the per-block passes
    (which always hold every local candidate's list), 2: the per-query passes (deferred, masked)"""
    import torch
    from sgtd_amd.dist import shard_range
    manager, synth = mods
    monkeypatch.setenv("SGTD_SELECT_MODE", mode)
    F, NQ, W = 90, 9, 3
    smap = synth.make_map(F, 140, stream=83)
    qs = synth.make_queries(smap, NQ, stream=83)
    single = manager.STDescManager()
    single.add_frames(smap.xyz, smap.label)
    want = single.query_frames(qs.xyz, qs.label)
    want_pairs = [single.result_pairs(q, want) for q in range(NQ)]
    single.verify()
    w_scores = [single.result_verify(q) for q in range(NQ)]
    cn = single.config_setting_["candidate_num"]
    dev = torch.device("cuda", 0)
    ints = 2 * NQ * cn + 4
    shards, packed, base = [], [], [0]
    for r in range(W):
        lo, hi = shard_range(F, W, r)
        m = manager.STDescManager(first_frame_id=lo)
        m.add_frames(smap.xyz[lo:hi], smap.label[lo:hi])
        p = torch.zeros(ints, dtype=torch.int32, device=dev)
        m.set_candidate_export(p)
        m.set_deferred_lists(True)
        m.query_frames(qs.xyz, qs.label, fetch=False)
        shards.append(m); packed.append(p)
        base.append(base[-1] + m.stats()["n_entries"])
    torch.cuda.synchronize()
    gathered = torch.cat(packed).contiguous()
    v_all = torch.empty((W, NQ * cn * 13), dtype=torch.float64, device=dev)
    outs = []
    for r, m in enumerate(shards):
        of = torch.empty((NQ, cn), dtype=torch.int32, device=dev)
        ov = torch.empty_like(of); osrc = torch.empty_like(of)
        on = torch.empty(NQ, dtype=torch.int32, device=dev)
        keep = torch.empty(NQ, dtype=torch.int64, device=dev)
        flags = torch.zeros(4, dtype=torch.int32, device=dev)
        m.merge_candidates_dev(0, gathered, W, r, NQ, of, ov, on, osrc, keep, flags)
        torch.cuda.synchronize()
        assert int(flags[0]) == 0
        m.finish_lists(keep)
        m.verify_masked(keep)
        m.export_verify(v_all[r, :NQ * cn], v_all[r, NQ * cn:])
        outs.append((of, ov, on, osrc, keep))
    torch.cuda.synchronize()
    of, ov, on, osrc, _ = outs[0]
    for o in outs[1:]:
        assert torch.equal(o[0], of) and torch.equal(o[1], ov) and torch.equal(o[3], osrc)     # the same merged list everywhere
    score = torch.empty((NQ, cn), dtype=torch.float64, device=dev)
    pose = torch.empty((NQ, cn, 12), dtype=torch.float64, device=dev)
    shards[0].gather_verified_dev(0, v_all.reshape(-1), W, osrc, NQ, score, pose)
    torch.cuda.synchronize()
    local = [m.results() for m in shards]
    for q in range(NQ):
        nc = int(want.n_cand[q])
        assert int(on[q]) == nc
        assert np.array_equal(of[q, :nc].cpu().numpy(), want.cand_frame[q, :nc]) and np.array_equal(ov[q, :nc].cpu().numpy(), want.cand_votes[q, :nc])
        w_score, w_rot, w_t = w_scores[q]
        assert np.array_equal(score[q].cpu().numpy(), w_score)
        got = pose[q].cpu().numpy()
        assert np.array_equal(got[:, :9].reshape(cn, 3, 3), w_rot) and np.array_equal(got[:, 9:], w_t)
        # the winners' match lists on their owners == the single table's lists (entry ids shifted by the shard's first entry)
        lists = [shards[r].result_pairs(q, local[r]) for r in range(W)]
        src = osrc[q].cpu().numpy()
        for k in range(nc):
            r, s = src[k] >> 8, src[k] & 255
            lo_, hi_ = local[r].pair_off[q, s], local[r].pair_off[q, s + 1]
            wl, wh = want.pair_off[q, k], want.pair_off[q, k + 1]
            assert hi_ - lo_ == wh - wl == want.cand_votes[q, k]
            assert np.array_equal(lists[r][0][lo_:hi_], want_pairs[q][0][wl:wh])
            assert np.array_equal(lists[r][1][lo_:hi_] + base[r], want_pairs[q][1][wl:wh])
        if mode == "2":       # ... and nothing else was written: the losers' lists are empty
            for r in range(W):
                kept = int(outs[r][4][q].item())
                for s in range(int(local[r].n_cand[q])):
                    if not (kept >> s) & 1:
                        assert local[r].pair_off[q, s + 1] == local[r].pair_off[q, s]
    # the lists of ALL local candidates once more (the records are intact), then the same batch without deferral
    m = shards[1]
    m.finish_lists(None)
    full = m.results()
    full_pairs = [m.result_pairs(q, full) for q in range(NQ)]
    m.set_deferred_lists(False)
    again = m.query_frames(qs.xyz, qs.label)
    assert np.array_equal(full.pair_off, again.pair_off) and full.pair_off[:, cn].sum() > 0
    for q in range(NQ):
        a = m.result_pairs(q, again)
        assert np.array_equal(a[0], full_pairs[q][0]) and np.array_equal(a[1], full_pairs[q][1])
    for m in shards + [single]:
        m.close()


def test_attached_view_queries_the_owners_table(mods):
    """sgtd_attach_table: a second handle borrows the finalized table (here with a tail segment) and gives the owner's
    results for the same batch — candidates, votes, ordered lists, verification — from its own work buffers and stream,
    while the owner works on another batch; a changed table is noticed; a view cannot add; an owner outlives its views"""
    import torch
    from sgtd_amd._lib import SgtdError
    manager, synth = mods
    smap = synth.make_map(70, 150, stream=97)
    qa, qb = synth.make_queries(smap, 7, stream=97), synth.make_queries(smap, 7, stream=98)
    own = manager.STDescManager()
    own.add_frames(smap.xyz[:60], smap.label[:60])
    own.finalize()
    own.add_frames(smap.xyz[60:], smap.label[60:])        # ... into a tail segment
    own.finalize()
    assert own.stats()["tail_entries"] > 0
    view = manager.STDescManager()
    s2 = torch.cuda.Stream()
    view.set_stream(s2.cuda_stream)
    view.attach_table(own)
    assert view.stats()["n_entries"] == own.stats()["n_entries"] and view.current_frame_id_ == own.current_frame_id_
    # two batches in flight: A on the view, B on the owner
    view.query_frames(qa.xyz, qa.label, fetch=False)
    own.query_frames(qb.xyz, qb.label, fetch=False)
    rv, ro_b = view.results(), own.results()
    pv = [view.result_pairs(q, rv) for q in range(7)]
    view.verify()
    sv = [view.result_verify(q) for q in range(7)]
    ro = own.query_frames(qa.xyz, qa.label)
    own.verify()
    assert rv.n_cand.max() > 0 and not np.array_equal(ro_b.cand_frame, ro.cand_frame)
    assert np.array_equal(rv.n_cand, ro.n_cand) and np.array_equal(rv.cand_frame, ro.cand_frame) and np.array_equal(rv.cand_votes, ro.cand_votes)
    assert np.array_equal(rv.pair_off, ro.pair_off)
    for q in range(7):
        a = own.result_pairs(q, ro)
        assert np.array_equal(a[0], pv[q][0]) and np.array_equal(a[1], pv[q][1])
        so = own.result_verify(q)
        assert all(np.array_equal(x, y) for x, y in zip(so, sv[q]))
    with pytest.raises(SgtdError):
        view.add_frames(smap.xyz[:2], smap.label[:2])
    with pytest.raises(SgtdError):
        own.close()                                        # a view is still attached
    own.add_frames(smap.xyz[:3], smap.label[:3])           # the table changes ...
    with pytest.raises(SgtdError):
        view.query_frames(qa.xyz, qa.label)                # ... and the view notices
    view.attach_table(own)
    r2 = view.query_frames(qa.xyz, qa.label)
    r3 = own.query_frames(qa.xyz, qa.label)
    assert np.array_equal(r2.cand_frame, r3.cand_frame) and np.array_equal(r2.cand_votes, r3.cand_votes)
    # a view's PENDING batch after the owner's table changed (the owner reallocates and frees what the view borrowed): the
    # verification, SearchLoop's choice and the gathers of entries refuse instead of reading freed memory
    view.query_frames(qa.xyz, qa.label, fetch=False)
    own.add_frames(smap.xyz[3:6], smap.label[3:6])
    own.finalize()
    for call in (view.verify, lambda: view.fetch_entries(np.arange(4, dtype=np.int64))):
        with pytest.raises(SgtdError):
            call()
    view.sync()                                            # (the refused batch is dropped: nothing is pending any more)
    view.attach_table(own)
    view.query_frames(qa.xyz, qa.label, fetch=False)
    own.add_frames(smap.xyz[6:8], smap.label[6:8])
    with pytest.raises(SgtdError):
        view.sync()                                        # the wait for a pending batch is where a re-run would touch the table
    view.attach_table(own)
    r4, r5 = view.query_frames(qb.xyz, qb.label), own.query_frames(qb.xyz, qb.label)
    view.verify(); own.verify()
    assert np.array_equal(r4.cand_frame, r5.cand_frame) and all(np.array_equal(x, y) for x, y in zip(view.result_verify(0), own.result_verify(0)))
    view.close()
    own.close()
