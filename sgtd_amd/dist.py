"""Multi-GPU forms of candidate_selector, one process per GPU, RCCL collectives.

The N ranks of a job form a grid of R_t table shards x R_q query groups (R_t * R_q = N, `Map2D`):
* inside a query group the map's hash table is sharded by frame range over R_t ranks:
  rank t owns map frames [lo_t, hi_t) and holds a complete table for them, so every vote of a frame
  is counted on exactly one rank and the final vote of each of its frames is local.  Per batch each
  rank probes its shard with the group's queries and takes its local top-`candidate_num` (votes
  desc, frame id asc, >= 5 votes — STDesc.cpp:423-433); ONE all_gather of the packed tables inside
  the group and an identical merge on every rank (a kernel: sgtd_merge_candidates_dev) reproduce the
  single-table candidate list bit for bit: the global top-k of disjoint frame sets is the top-k of
  the union of the local top-k lists.  The match lists of the winners stay on their owner.
* the R_q query groups each serve their own queries against their own copy of the (sharded) table;
  one all_gather across the groups hands every rank the result tables of the whole step.
R_t = N is the pure table-sharded form BASELINE.json's north_star names (`ShardedMap`), R_t = 1 the
replicated map with sharded queries (`ReplicatedMap`); `plan_2d` picks the smallest R_t whose shard
fits one GPU's envelope, because everything a rank does per QUERY (descriptor build, home-cell sort,
GroupRows, the plan) is repeated on all R_t ranks of a group and only the sweep shrinks with the shard.

The exchange is off the critical path: the packed local tables leave the pipeline right behind
votes_topk_kernel (sgtd_set_candidate_export), a side stream all-gathers and merges them while the
main stream writes the match lists (`lists="all"`), or the main stream waits for the merge and
writes the lists of the winners only (`lists="winners"`).  Either way sgtd_verify_masked verifies
only the candidates that survived the merge.
"""

import os

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_frames, world, rank):
    """contiguous frame range [lo, hi) of `rank`"""
    return rank * n_frames // world, (rank + 1) * n_frames // world


def owner_of(frame, n_frames, world):
    """rank whose shard holds `frame` (inverse of shard_range)"""
    r = min(world - 1, (int(frame) * world) // max(n_frames, 1))
    while frame < shard_range(n_frames, world, r)[0]:
        r -= 1
    while frame >= shard_range(n_frames, world, r)[1]:
        r += 1
    return r


def merge_candidates(frames, votes, cand_num, min_votes=5):
    """frames, votes: int32 tensors [W, Q, cn] of per-rank candidate tables (unused
    slots: frame -1 / votes 0).  Returns (frames [Q, cand_num], votes [Q, cand_num],
    n_cand [Q]) in the reference's order: votes descending, ties -> lowest frame id."""
    w, q, cn = frames.shape
    f = frames.permute(1, 0, 2).reshape(q, w * cn).to(torch.int64)
    v = votes.permute(1, 0, 2).reshape(q, w * cn).to(torch.int64)
    valid = (f >= 0) & (v >= min_votes)
    key = torch.where(valid, (v << 32) | (0xFFFFFFFF - f), torch.full_like(v, -1))
    key, _ = torch.sort(key, dim=1, descending=True)
    key = key[:, :cand_num]
    ok = key >= 0
    out_v = torch.where(ok, key >> 32, torch.zeros_like(key)).to(torch.int32)
    out_f = torch.where(ok, 0xFFFFFFFF - (key & 0xFFFFFFFF), torch.full_like(key, -1)).to(torch.int32)
    return out_f, out_v, ok.sum(dim=1).to(torch.int32)


def gather_and_merge(local_frames, local_votes, cand_num, group=None):
    """all_gather of the local candidate tables [Q, cn] (RCCL on GPUs, gloo on CPU)
    followed by the merge; every rank returns the same global candidate list"""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return merge_candidates(local_frames[None], local_votes[None], cand_num)
    # one collective: frames and votes travel together
    packed = torch.stack([local_frames, local_votes]).contiguous()
    out = torch.empty((world * 2,) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, packed, group=group)   # concatenation along dim 0 (RCCL and gloo)
    out = out.view((world, 2) + tuple(packed.shape[1:]))
    return merge_candidates(out[:, 0], out[:, 1], cand_num)


def merge_verified(global_frames, shard_frames, shard_scores, shard_poses):
    """Table-sharded verification: every candidate frame is verified by the rank that owns it
    (its match list lives there), the results travel in one all_gather.  global_frames [Q, cn]
    = merged candidate list (-1 unused); shard_frames [W, Q, cn] / shard_scores [W, Q, cn] /
    shard_poses [W, Q, cn, 12] = every rank's local candidate frames and their verify results.
    Returns (scores [Q, cn], poses [Q, cn, 12]) aligned with the global list (-1 / zeros where
    unused) — what candidate_verify gives on a single table, because a frame's match list and
    the descriptors in it are the same on its owner."""
    w, q, cn = shard_frames.shape
    sf = shard_frames.permute(1, 0, 2).reshape(q, w * cn)
    ss = shard_scores.permute(1, 0, 2).reshape(q, w * cn)
    sp = shard_poses.permute(1, 0, 2, 3).reshape(q, w * cn, 12)
    eq = (sf[:, None, :] == global_frames[:, :, None]) & (global_frames[:, :, None] >= 0)   # [Q, cn, W*cn]
    found = eq.any(dim=2)
    idx = eq.to(torch.int8).argmax(dim=2)                                                   # frames are unique across shards
    scores = torch.where(found, torch.gather(ss, 1, idx), torch.full_like(idx, -1, dtype=ss.dtype))
    poses = torch.gather(sp, 1, idx[:, :, None].expand(q, idx.shape[1], 12)) * found[:, :, None].to(sp.dtype)
    return scores, poses


def gather_verified(local_frames, score, pose, group=None):
    """one all_gather of every rank's (candidate frames [Q, cn] int32, verify_score [Q, cn] f64,
    pose [Q, cn, 12] f64) -> stacked [W, ...] tensors, identical on every rank.  Frames travel
    as f64 next to the scores (ids below 2^53 are exact)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_frames[None], score[None], pose[None]
    nq, cn = local_frames.shape
    packed = torch.cat([local_frames.to(torch.float64)[:, :, None], score[:, :, None], pose], dim=2).contiguous()
    out = torch.empty((world * nq, cn, 14), dtype=torch.float64, device=packed.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    out = out.view(world, nq, cn, 14)
    return out[..., 0].to(torch.int32), out[..., 1].contiguous(), out[..., 2:].contiguous()


def search_loop_choice(global_frames, n_cand, scores, icp_threshold):
    """SearchLoop's choice (STDesc.cpp:105-146) on device tensors: per query the first
    candidate with the strictly largest verify_score, accepted above icp_threshold.
    Returns (best_cand, best_frame, best_score) with -1 / -1 / 0 for "no loop"."""
    q, cn = scores.shape
    live = torch.arange(cn, device=scores.device)[None, :] < n_cand[:, None].to(torch.int64)
    s = torch.where(live, scores, torch.full_like(scores, -1.0))
    best, arg = s.max(dim=1)                         # torch.max returns the first maximum
    first = (s == best[:, None]).to(torch.int8).argmax(dim=1)
    ok = (best > 0) & (best > icp_threshold)
    best_frame = torch.gather(global_frames.to(torch.int64), 1, first[:, None])[:, 0]
    neg = torch.full_like(first, -1)
    return (torch.where(ok, first, neg).to(torch.int32), torch.where(ok, best_frame, neg).to(torch.int32),
            torch.where(ok, best, torch.zeros_like(best)))



def plan_2d(world, n_frames, queries_per_group, keypoints=200, span_limit=32000, record_limit=3.2e9, r_t=None):
    """(R_t, R_q) for `world` ranks: the smallest R_t (a divisor of world) whose shard of an n_frames map fits one GPU's
    envelope — the per-query vote histogram of votes_topk_kernel in LDS (span_limit frames), the batch's match records
    under the 32-bit record index (about 0.37 rough matches per keypoint, map frame and query on the synthetic maps:
    0.74 M per 200-keypoint query at 10 000 frames), the table in HBM (36 descriptors per keypoint, 155 B each, a
    quarter of 288 GB)."""
    if r_t is not None:
        assert world % r_t == 0
        return int(r_t), world // int(r_t)
    for t in [d for d in range(1, world + 1) if world % d == 0]:
        f = -(-n_frames // t)
        if f > span_limit and t < world:
            continue
        if 0.37 * keypoints * f * queries_per_group > record_limit and t < world:
            continue
        if 155.0 * 36 * keypoints * f > 288e9 / 4 and t < world:
            continue
        return t, world // t
    return world, 1


def grid_groups(world, r_t):
    """the rank lists of an R_t x R_q grid (rank = g * R_t + t): the R_q table groups (ranks that share query group g and
    hold the shards of one table copy) and the R_t column groups (ranks that hold the same shard t, one per query group)"""
    assert r_t >= 1 and world % r_t == 0
    r_q = world // r_t
    return [list(range(g * r_t, (g + 1) * r_t)) for g in range(r_q)], [list(range(t, world, r_t)) for t in range(r_t)]


def make_grid_groups(world, r_t, rank):
    """the process groups of `rank` in the grid, created by EVERY rank in the same order (torch.distributed's rule);
    None stands for the default group (a grid side as long as the world) or for a side of one (no collective)"""
    tables, cols = grid_groups(world, r_t)
    table_group = col_group = None
    for ranks in tables:
        grp = dist.new_group(ranks) if 1 < len(ranks) < world else None
        if rank in ranks:
            table_group = grp
    for ranks in cols:
        grp = dist.new_group(ranks) if 1 < len(ranks) < world else None
        if rank in ranks:
            col_group = grp
    return table_group, col_group


class Map2D:
    """one rank of an R_t x R_q grid: rank = g * R_t + t serves query group g with table shard t"""

    def __init__(self, n_frames_total, rank, world, r_t=None, device_id=0, lists="all", attach_to=None, **cfg):
        """attach_to: another Map2D of the same grid on this rank — this one borrows its table (sgtd_attach_table) and keeps
        work buffers, streams and exchange buffers of its own: steps issued alternately on the two overlap on the device"""
        from . import _lib
        from .manager import STDescManager
        assert lists in ("all", "winners")
        self.n_frames, self.rank, self.world = n_frames_total, rank, world
        self.r_t = world if r_t is None else int(r_t)
        assert self.r_t >= 1 and world % self.r_t == 0
        self.r_q = world // self.r_t
        self.t, self.g = rank % self.r_t, rank // self.r_t
        self.lo, self.hi = shard_range(n_frames_total, self.r_t, self.t)
        self.lists = lists
        cfg.setdefault("max_frame_n", max(20000, n_frames_total + 1))
        self.mgr = STDescManager(first_frame_id=self.lo, device_id=device_id, **cfg)
        self.cand_num = self.mgr.config_setting_["candidate_num"]
        self._L = _lib.lib()
        self.dev = torch.device("cuda", device_id)
        # every rank creates every group, in the same order (torch.distributed's rule)
        self.table_group, self.col_group = None, None
        if dist.is_initialized() and world > 1:
            assert dist.get_world_size() == world and dist.get_rank() == rank
            self.table_group, self.col_group = make_grid_groups(world, self.r_t, rank)
        if attach_to is None:
            self.main = torch.cuda.current_stream(self.dev)
        else:
            assert (attach_to.r_t, attach_to.rank, attach_to.world) == (self.r_t, rank, world)
            self.main = torch.cuda.Stream(self.dev)
        self.mgr.set_stream(self.main.cuda_stream)
        if attach_to is not None:
            self.mgr.attach_table(attach_to.mgr)
        self.side = torch.cuda.Stream(self.dev)
        self._nq_buf = -1
        self.merged = None
        self.exchanges = 0
        # defer = True: for transports whose all-gather of device tensors BLOCKS THE HOST (gloo copies through host memory,
        # so the call returns only once the packed table exists — a whole sweep after the step was enqueued — and the
        # host cannot feed the device meanwhile).  The all-gather is then issued asynchronously from a copy of the packed
        # table and its merge is enqueued at the start of the NEXT exchange (or by flush()): the merged tables lag one
        # step.  Measurement aid for one-GPU boxes (bench.py `exchange_exposed_ms_host_not_blocked`); RCCL collectives are
        # stream-ordered and never need it.
        self.defer = False
        self._pending = None
        # force_collective: issue the collectives even along a grid side of ONE rank (the all-gather of a single table is a
        # copy).  A one-GPU box can put the RCCL calls, their stream ordering against the engine's export events and the
        # merge behind them on hardware this way (tests/test_dist_gpu.py); never set in production.
        self.force_collective = os.environ.get("SGTD_FORCE_COLLECTIVE") == "1"
        # (a grid side of ONE rank has no process group of its own — None means WORLD to torch.distributed — so the switch only
        # makes sense in a job of one rank)
        assert not self.force_collective or world == 1, "SGTD_FORCE_COLLECTIVE=1 is for one-rank jobs"
        if lists == "winners":
            self.mgr.set_deferred_lists(True)

    # ---- map construction ---------------------------------------------------------------
    def add_shard_frames(self, xyz, label, kp_off=None):
        """xyz/label of THIS rank's frames [lo, hi) in frame order"""
        self.mgr.add_frames(xyz, label, kp_off)
        self.mgr.finalize()

    # ---- buffers --------------------------------------------------------------------------
    def _buffers(self, nq):
        if nq == self._nq_buf:
            return
        cn, rt, dev = self.cand_num, self.r_t, self.dev
        ints = int(self._L.sgtd_candidate_export_ints(nq, cn))
        self.packed = torch.empty(ints, dtype=torch.int32, device=dev)
        self.gathered = torch.empty(rt * ints, dtype=torch.int32, device=dev)
        self.staged = torch.empty(ints, dtype=torch.int32, device=dev)
        self.m_frame = torch.empty((nq, cn), dtype=torch.int32, device=dev)
        self.m_votes = torch.empty((nq, cn), dtype=torch.int32, device=dev)
        self.m_src = torch.empty((nq, cn), dtype=torch.int32, device=dev)
        self.m_n = torch.empty(nq, dtype=torch.int32, device=dev)
        self.m_keep = torch.empty(nq, dtype=torch.int64, device=dev)
        self.m_flags = torch.zeros(4, dtype=torch.int32, device=dev)
        self.v_local = torch.empty(nq * cn * 13, dtype=torch.float64, device=dev)        # [score | pose] of the local candidates
        self.v_gathered = torch.empty(rt * nq * cn * 13, dtype=torch.float64, device=dev)
        self.v_score = torch.empty((nq, cn), dtype=torch.float64, device=dev)
        self.v_pose = torch.empty((nq, cn, 12), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        self.mgr.set_candidate_export(self.packed)
        self._nq_buf = nq

    def _table_collective(self):
        return (self.r_t > 1 or self.force_collective) and dist.is_initialized()

    # ---- the step ---------------------------------------------------------------------------
    def flush(self):
        """finish a deferred exchange (defer = True): wait for its all-gather, enqueue its merge"""
        if self._pending is None:
            return
        work, nq = self._pending
        self._pending = None
        with torch.cuda.stream(self.side):      # wait() orders the CURRENT stream behind the collective: the merge runs on the side stream
            work.wait()
        self.mgr.merge_candidates_dev(self.side.cuda_stream, self.gathered, self.r_t, self.t, nq, self.m_frame, self.m_votes, self.m_n,
                                      self.m_src, self.m_keep, self.m_flags)

    def _exchange(self):
        """side stream: wait for the packed local table, all-gather it inside the table group, merge"""
        nq = self.mgr._nq
        if self.defer and self._table_collective() and self.lists == "all":
            self.flush()
            self.mgr.export_wait(self.side.cuda_stream)
            with torch.cuda.stream(self.side):
                self.staged.copy_(self.packed, non_blocking=True)      # the engine's buffer is free again at once
            self.mgr.export_release(self.side.cuda_stream)
            with torch.cuda.stream(self.side):
                work = dist.all_gather_into_tensor(self.gathered, self.staged, group=self.table_group, async_op=True)
            self._pending = (work, nq)
            self.exchanges += 1
            return
        self.mgr.export_wait(self.side.cuda_stream)
        with torch.cuda.stream(self.side):
            if self._table_collective():
                dist.all_gather_into_tensor(self.gathered, self.packed, group=self.table_group)
                src = self.gathered
            else:
                src = self.packed
        self.mgr.merge_candidates_dev(self.side.cuda_stream, src, self.r_t, self.t, nq, self.m_frame, self.m_votes, self.m_n,
                                      self.m_src, self.m_keep, self.m_flags)
        self.mgr.export_release(self.side.cuda_stream)
        self.exchanges += 1

    def query_async(self, xyz, label, kp_off=None):
        """enqueue one step for this rank's query group: the shard's sweep and the passes over its records on the main
        stream, the exchange (all_gather + merge kernel) on the side stream.  Nothing waits on the host; the merged tables
        (self.m_frame / m_votes / m_n / m_src / m_keep) are valid once both streams are, and self.m_flags[0] != 0 says
        that some rank's batch outgrew a work buffer (query() repairs that; a caller of query_async checks it)."""
        nq = self.mgr.frames_in(xyz, kp_off)
        self._buffers(nq)
        self.mgr.query_frames(xyz, label, kp_off, fetch=False)
        self._exchange()
        if self.lists == "winners":
            self.main.wait_stream(self.side)                 # (device-side wait)
            self.mgr.finish_lists(self.m_keep)

    def query(self, xyz, label, kp_off=None):
        """all ranks of a table group pass the same query batch; returns the group's merged candidate list
        (frames, votes, n_cand) as device tensors, identical on every rank of the group"""
        self.query_async(xyz, label, kp_off)
        self.flush()
        for _ in range(4):
            self.side.synchronize()
            if int(self.m_flags[0].item()) == 0:
                break
            if int(self.m_flags[0].item()) & 2:
                raise RuntimeError("sgtd_amd.dist: the ranks of a table group passed batches of different shapes")
            # some rank's batch outgrew a work buffer (every rank of the group sees the same flag): sgtd_sync re-runs it
            # where that happened — the re-run exports again — and the group exchanges once more
            self.mgr.sync()
            self._exchange()
            if self.lists == "winners":
                self.main.wait_stream(self.side)
                self.mgr.finish_lists(self.m_keep)
        else:
            raise RuntimeError("sgtd_amd.dist: a batch kept outgrowing its work buffers")
        self.main.wait_stream(self.side)
        # the merged tables are ordered against self.main — for a map with a stream of its own (attach_to) that is not the
        # caller's: whatever the caller enqueues next on ITS stream comes behind them
        self._caller_waits()
        return self.m_frame, self.m_votes, self.m_n

    def _caller_waits(self):
        cur = torch.cuda.current_stream(self.dev)
        if cur.cuda_stream != self.main.cuda_stream:
            cur.wait_stream(self.main)

    def gather_groups(self):
        """the merged tables of ALL query groups (group-major): (frames, votes) [R_q * nq, cn] on every rank"""
        if (self.r_q == 1 and not self.force_collective) or not dist.is_initialized():
            return self.m_frame, self.m_votes
        with torch.cuda.stream(self.main):       # torch ops and the collective behind the engine's work on self.main
            packed = torch.stack([self.m_frame, self.m_votes]).contiguous()
            out = torch.empty((self.r_q * 2,) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
            dist.all_gather_into_tensor(out, packed, group=self.col_group)
            out = out.view((self.r_q, 2) + tuple(packed.shape[1:]))
            res = out[:, 0].reshape(-1, self.cand_num), out[:, 1].reshape(-1, self.cand_num)
        self._caller_waits()
        return res

    def search_loop(self, xyz, label, kp_off=None, icp_threshold=None, group=None):
        """SearchLoop over the sharded map: query + merge, candidate_verify of this rank's WINNERS only
        (sgtd_verify_masked), one all_gather of (scores, poses) inside the table group, the merged candidates'
        results out of their owners' tables (sgtd_gather_verified_dev), SearchLoop's choice on the merged list.
        Returns (frames, votes, n_cand, scores, poses, best_cand, best_frame, best_score), identical on every rank."""
        frames, votes, n_cand = self.query(xyz, label, kp_off)
        nq, cn = frames.shape
        # everything below on self.main: the verification, its export copies, the all-gather of the results, the gather kernel and
        # the choice are one chain on the engine's stream (a map with a stream of its own — attach_to — would otherwise start the
        # all-gather on the caller's stream before the verification had written v_local)
        with torch.cuda.stream(self.main):
            self.mgr.verify_masked(self.m_keep)
            self.mgr.export_verify(self.v_local[:nq * cn], self.v_local[nq * cn:])
            if self._table_collective():
                dist.all_gather_into_tensor(self.v_gathered, self.v_local, group=self.table_group)
                src = self.v_gathered
            else:
                src = self.v_local
            self.mgr.gather_verified_dev(self.main.cuda_stream, src, self.r_t, self.m_src, nq, self.v_score, self.v_pose)
            thr = self.mgr.icp_threshold_ if icp_threshold is None else icp_threshold
            bc, bf, bs = search_loop_choice(frames, n_cand, self.v_score, thr)
        self._caller_waits()
        return frames, votes, n_cand, self.v_score, self.v_pose, bc, bf, bs


class ShardedMap(Map2D):
    """the pure table-sharded form (R_t = world): one rank's shard of the map + the collective query"""

    def __init__(self, n_frames_total, rank, world, device_id=0, **cfg):
        super().__init__(n_frames_total, rank, world, r_t=world, device_id=device_id, **cfg)


def query_slice(n_queries, world, rank):
    """contiguous slice [lo, hi) of a query batch that `rank` serves when the MAP is
    replicated and the QUERIES are sharded"""
    return rank * n_queries // world, (rank + 1) * n_queries // world


def gather_query_slices(local_frames, local_votes, n_queries, group=None):
    """query-sharded mode: every rank computed the candidate tables [q_hi-q_lo, cn] of its
    query slice against a full replica of the map; one all_gather gives every rank the
    tables of the whole batch.  Slices may differ by one row: they travel padded."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_frames, local_votes
    rows = (n_queries + world - 1) // world
    cn = local_frames.shape[1]
    packed = torch.full((2, rows, cn), -1, dtype=torch.int32, device=local_frames.device)
    packed[0, :local_frames.shape[0]] = local_frames
    packed[1, :local_votes.shape[0]] = local_votes
    out = torch.empty((world * 2, rows, cn), dtype=torch.int32, device=packed.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    out = out.view(world, 2, rows, cn)
    frames, votes = [], []
    for r in range(world):
        lo, hi = query_slice(n_queries, world, r)
        frames.append(out[r, 0, :hi - lo])
        votes.append(out[r, 1, :hi - lo])
    return torch.cat(frames), torch.cat(votes)


class ReplicatedMap(Map2D):
    """R_t = 1: every rank holds the whole table and serves its own query group (no collective on the data path but
    the gather of the results)"""

    def __init__(self, n_frames_total, rank, world, device_id=0, **cfg):
        super().__init__(n_frames_total, rank, world, r_t=1, device_id=device_id, **cfg)

    def add_frames(self, xyz, label, kp_off=None):
        self.add_shard_frames(xyz, label, kp_off)

    def query_group(self, xyz_group, label_group):
        """xyz_group/label_group: THIS rank's queries; returns (frames, votes) [world * nq, cn] on every rank"""
        self.query_async(xyz_group, label_group)
        self.main.wait_stream(self.side)
        with torch.cuda.stream(self.main):
            return self.gather_groups()


# ---------------------------------------------------------------------------
# ablation: the table sharded by bucket KEY instead of by frame range
# ---------------------------------------------------------------------------
def key_owner(code, x, y, z, world):
    """rank that owns the bucket (label code, cell): a multiplicative hash of the table key"""
    k = (np.asarray(code, np.uint64) << np.uint64(48)) | (np.asarray(x, np.uint64) << np.uint64(32)) | \
        (np.asarray(y, np.uint64) << np.uint64(16)) | np.asarray(z, np.uint64)
    return ((k * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)) % np.uint64(world)


def topk_from_votes(votes, cand_num, min_votes=5):
    """the reference's candidate rule (STDesc.cpp:423-433) on a full vote histogram [nq, F]:
    arg-max rounds = votes descending, ties to the lowest frame id, a frame needs >= 5 votes.
    Returns frames, votes int32 [nq, cand_num] (-1 / 0 in unused slots) and n_cand [nq]."""
    nq, F = votes.shape
    v = votes.to(torch.int64)
    frame = torch.arange(F, device=votes.device, dtype=torch.int64)[None, :].expand(nq, F)
    key = v * (F + 1) + (F - frame)                       # votes first, then the LOWER frame id
    k = min(cand_num, F)
    top = torch.topk(key, k, dim=1).indices
    tv = torch.gather(v, 1, top)
    ok = tv >= min_votes
    frames = torch.full((nq, cand_num), -1, dtype=torch.int32, device=votes.device)
    out_v = torch.zeros((nq, cand_num), dtype=torch.int32, device=votes.device)
    frames[:, :k] = torch.where(ok, top, torch.full_like(top, -1)).to(torch.int32)
    out_v[:, :k] = torch.where(ok, tv, torch.zeros_like(tv)).to(torch.int32)
    return frames, out_v, ok.sum(dim=1).to(torch.int32)


class KeyShardedMap:
    """ABLATION, not the product path: every rank holds the buckets whose key hashes to it — a slice of
    EVERY frame's descriptors — so a frame's votes are the sum of the ranks' votes and the exchange is an
    all-reduce of the whole Q x F histogram (4 B x Q x F per rank: 82 MB for 2048 queries on a 10 000-frame
    map) instead of the all-gather of the top-candidate_num tables of the frame-range form (400 B per
    query and rank).  Candidates and votes only: a candidate's match list is spread over all ranks.
    Host-side and slow by construction (descriptors are built once per frame and filtered on the host);
    it exists to show that both shardings select the same candidates and what the second one moves."""

    def __init__(self, rank, world, device_id=0, **cfg):
        from .manager import STDescManager
        from . import _lib
        self.rank, self.world = rank, world
        self.mgr = STDescManager(device_id=device_id, **cfg)
        self.cand_num = self.mgr.config_setting_["candidate_num"]
        self._L = _lib.lib()
        self.n_frames = 0
        self.kept = 0

    def add_frames(self, xyz, label):
        """ALL frames of the map, in order, on every rank; each keeps the descriptors of its buckets"""
        for f in range(len(xyz)):
            d = self.mgr.BuildSingleScanSTD(xyz[f], label[f])
            code = np.array([self._L.sgtd_label_code(int(a), int(b), int(c)) for a, b, c in d.label], np.uint64)
            cell = (d.side + 0.5).astype(np.int64)                       # position of STDesc.cpp:153-160
            mine = np.nonzero(key_owner(code, cell[:, 0], cell[:, 1], cell[:, 2], self.world) == self.rank)[0]
            self.mgr.AddSTDescs(d.take(mine))                            # (an empty slice still advances the frame counter)
            self.kept += len(mine)
            self.n_frames += 1
        self.mgr.finalize()

    def local_votes(self, xyz, label):
        """this rank's share of the vote histogram of a query batch, int32 [nq, n_frames + 1]"""
        res = self.mgr.query_frames(xyz, label)
        nq = len(res.n_cand)
        votes = torch.zeros((nq, self.n_frames + 1), dtype=torch.int32)
        for q in range(nq):
            lo, v = self.mgr.result_votes(q)
            votes[q, lo:lo + len(v)] = torch.from_numpy(v.astype(np.int32))
        return votes

    def query(self, xyz, label, group=None):
        """all ranks pass the same query batch; returns (frames, votes, n_cand) identical on every rank
        and the bytes this rank put into the all-reduce"""
        votes = self.local_votes(xyz, label)
        if self.world > 1:
            dist.all_reduce(votes, op=dist.ReduceOp.SUM, group=group)
        f, v, n = topk_from_votes(votes, self.cand_num)
        return f, v, n, votes.numel() * 4
