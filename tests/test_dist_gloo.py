"""N > 1 path on CPU: world_size-2 gloo run of the frame-range sharded map.

Each rank holds a table shard for its frame range (here backed by the CPU
oracle standing in for the HIP engine — tests may use the oracle as the
checker/stand-in, the product never does), computes its local top-k, and
`sgtd_amd.dist.gather_and_merge` must reproduce the single-table candidate
list bit for bit on every rank.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N_FRAMES, N_KP, N_Q, CAND = 24, 60, 3, 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_tables(rank, world):
    """(frames, votes) int32 [N_Q, CAND] of this rank's shard, and the single-table truth"""
    from oracle.oracle import OracleManager
    from sgtd_amd import synth
    from sgtd_amd.dist import shard_range
    m = synth.make_map(N_FRAMES, N_KP, stream=201)
    qs = synth.make_queries(m, N_Q, stream=201)
    lo, hi = shard_range(N_FRAMES, world, rank)
    shard = OracleManager(candidate_num=CAND)
    shard.set_current_frame_id(lo)          # sgtd_config.first_frame_id on the HIP engine
    full = OracleManager(candidate_num=CAND)
    for f in range(N_FRAMES):
        full.build(m.xyz[f], m.label[f], export=False)
        full.add_last()
        if lo <= f < hi:
            shard.build(m.xyz[f], m.label[f], export=False)
            shard.add_last()
    assert shard.current_frame_id == hi
    lf = np.full((N_Q, CAND), -1, np.int32)
    lv = np.zeros((N_Q, CAND), np.int32)
    ls = np.full((N_Q, CAND), -1.0)              # candidate_verify of the local candidates
    lp = np.zeros((N_Q, CAND, 12))
    truth = []

    def verified(orc, sel):
        sc, ps = [], []
        for k in range(len(sel["cand_frame"])):
            s_, t_, rot_, _ = orc.verify(k, int(sel["cand_off"][k + 1] - sel["cand_off"][k]))
            sc.append(s_)
            ps.append(np.concatenate([rot_.reshape(9), t_]) if s_ >= 0 else np.zeros(12))
        return np.array(sc), np.array(ps).reshape(-1, 12)

    for q in range(N_Q):
        shard.build(qs.xyz[q], qs.label[q], export=False)
        r = shard.select()
        n = len(r["cand_frame"])
        lf[q, :n], lv[q, :n] = r["cand_frame"], r["cand_votes"]
        ls[q, :n], lp[q, :n] = verified(shard, r)
        assert np.all((r["cand_frame"] >= lo) & (r["cand_frame"] < hi))
        full.build(qs.xyz[q], qs.label[q], export=False)
        t = full.select()
        truth.append((t["cand_frame"], t["cand_votes"]) + verified(full, t))
    return lf, lv, truth, ls, lp


def _worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sgtd_amd.dist import gather_and_merge
        lf, lv, truth, ls, lp = _local_tables(rank, world)
        mf, mv, n = gather_and_merge(torch.from_numpy(lf), torch.from_numpy(lv), CAND)
        # SearchLoop over the shards: every candidate is verified by its owner, one all_gather
        from sgtd_amd.dist import gather_verified, merge_verified, search_loop_choice
        sf, ss, sp = gather_verified(torch.from_numpy(lf), torch.from_numpy(ls), torch.from_numpy(lp))
        scores, poses = merge_verified(mf, sf, ss, sp)
        bc, bf, bs = search_loop_choice(mf, n, scores, 0.4)
        for q in range(N_Q):
            k = int(n[q])
            tf, tv, tscore, tpose = truth[q]
            assert scores[q, :k].tolist() == tscore.tolist() and (scores[q, k:] == -1).all()
            assert np.array_equal(poses[q, :k].numpy(), tpose)
            best = max(tscore.tolist() + [0.0])
            if best > 0.4:
                kk = tscore.tolist().index(best)
                assert (int(bc[q]), int(bf[q]), float(bs[q])) == (kk, int(tf[kk]), best)
            else:
                assert (int(bc[q]), int(bf[q]), float(bs[q])) == (-1, -1, 0.0)
            assert k == len(tf), (k, len(tf))
            assert mf[q, :k].tolist() == tf.tolist(), (mf[q].tolist(), tf.tolist())
            assert mv[q, :k].tolist() == tv.tolist()
            assert (mf[q, k:] == -1).all() and (mv[q, k:] == 0).all()
        # every rank holds the same merged list
        probe = torch.stack([mf, mv]).contiguous()
        ref = probe.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, probe)
        # query-sharded mode: every rank serves a slice of the batch against the full table
        from sgtd_amd.dist import gather_query_slices, query_slice
        full_f = np.full((N_Q, CAND), -1, np.int32)
        full_v = np.zeros((N_Q, CAND), np.int32)
        for q in range(N_Q):
            tf, tv = truth[q][:2]
            full_f[q, :len(tf)], full_v[q, :len(tv)] = tf, tv
        lo, hi = query_slice(N_Q, world, rank)
        gf, gv = gather_query_slices(torch.from_numpy(full_f[lo:hi].copy()), torch.from_numpy(full_v[lo:hi].copy()), N_Q)
        assert gf.shape == (N_Q, CAND)
        assert np.array_equal(gf.numpy(), full_f)
        valid = full_f >= 0
        assert np.array_equal(gv.numpy()[valid], full_v[valid])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_selection_matches_single_table_gloo():
    world = 2
    # a failing assertion in any rank makes mp.spawn raise
    mp.spawn(_worker, args=(world, _free_port()), nprocs=world, join=True)


def test_single_process_merge_matches_oracle_without_process_group():
    from sgtd_amd.dist import gather_and_merge
    lf, lv, truth = _local_tables(0, 1)[:3]
    mf, mv, n = gather_and_merge(torch.from_numpy(lf), torch.from_numpy(lv), CAND)
    for q in range(N_Q):
        k = int(n[q])
        assert mf[q, :k].tolist() == truth[q][0].tolist() and mv[q, :k].tolist() == truth[q][1].tolist()


def test_topk_from_votes_is_the_reference_rule():
    """the key-sharded ablation picks its candidates from the all-reduced histogram: arg-max rounds of
    STDesc.cpp:423-433 (most votes, ties to the lowest frame id, at least 5 votes)"""
    import numpy as np
    import torch
    from sgtd_amd.dist import topk_from_votes, key_owner
    rng = np.random.default_rng(5)
    votes = rng.integers(0, 9, size=(7, 300)).astype(np.int32)       # many ties
    votes[3, :] = 0                                                    # a query without a candidate
    votes[4, :20] = 4                                                  # fewer than candidate_num frames reach 5 votes
    votes[4, 20:] = 0
    votes[4, 7] = 11
    f, v, n = topk_from_votes(torch.from_numpy(votes), 50)
    for q in range(7):
        work = votes[q].astype(np.int64).copy()
        want_f, want_v = [], []
        for _ in range(50):                                            # the reference's rounds
            best = int(np.argmax(work))                                # first maximum = lowest frame id
            if work[best] < 5:
                break
            want_f.append(best); want_v.append(int(work[best]))
            work[best] = 0
        assert int(n[q]) == len(want_f)
        assert f[q, :len(want_f)].tolist() == want_f and v[q, :len(want_v)].tolist() == want_v
        assert (f[q, len(want_f):] == -1).all() and (v[q, len(want_f):] == 0).all()
    own = key_owner(rng.integers(0, 4096, 1000), rng.integers(0, 200, 1000), rng.integers(0, 200, 1000), rng.integers(0, 200, 1000), 8)
    assert own.min() >= 0 and own.max() < 8 and len(np.unique(own)) == 8


def _grid_worker(rank, world, port, r_t):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sgtd_amd.dist import grid_groups, make_grid_groups, merge_candidates
        tables, cols = grid_groups(world, r_t)
        tg, cg = make_grid_groups(world, r_t, rank)
        t, g = rank % r_t, rank // r_t
        # the table group's exchange: every rank contributes its shard's local table, the merge is identical inside the group
        lf = torch.full((2, CAND), -1, dtype=torch.int32)
        lv = torch.zeros((2, CAND), dtype=torch.int32)
        lf[:, 0] = 100 * t + g                      # one candidate per shard: frame ids disjoint inside a group
        lv[:, 0] = 5 + t
        packed = torch.stack([lf, lv]).contiguous()
        n_t = len(tables[g])
        out = torch.empty((n_t * 2, 2, CAND), dtype=torch.int32)
        if n_t > 1:
            dist.all_gather_into_tensor(out, packed, group=tg)
        else:
            out.copy_(packed)
        out = out.view(n_t, 2, 2, CAND)
        mf, mv, n = merge_candidates(out[:, 0], out[:, 1], CAND)
        want = sorted([(5 + tt, 100 * tt + g) for tt in range(r_t)], key=lambda kv: (-kv[0], kv[1]))
        assert n.tolist() == [r_t, r_t] and mf[0, :r_t].tolist() == [f for _, f in want] and mv[0, :r_t].tolist() == [v for v, _ in want]
        # the column group's gather of the groups' result tables: group-major on every rank
        res = torch.full((1, 3), g, dtype=torch.int32)
        n_c = len(cols[t])
        allr = torch.empty((n_c, 3), dtype=torch.int32)
        if n_c > 1:
            dist.all_gather_into_tensor(allr, res, group=cg)
        else:
            allr.copy_(res)
        assert allr[:, 0].tolist() == list(range(world // r_t))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("r_t", [1, 2, 4])
def test_grid_groups_exchange_inside_table_groups_and_across_query_groups(r_t):
    """the R_t x R_q grid of sgtd_amd/dist.py on four gloo ranks: the process groups every rank creates, the all-gather
    inside a table group (+ the reference's merge rule) and the gather of the groups' results across the column group"""
    from sgtd_amd.dist import grid_groups
    assert grid_groups(8, 4) == ([[0, 1, 2, 3], [4, 5, 6, 7]], [[0, 4], [1, 5], [2, 6], [3, 7]])
    assert grid_groups(4, 1) == ([[0], [1], [2], [3]], [[0, 1, 2, 3]])
    mp.spawn(_grid_worker, args=(4, _free_port(), r_t), nprocs=4, join=True)
