"""Multi-GPU forms of candidate_selector, one process per GPU, RCCL collectives.

Two modes (bench.py --shard auto|table|query):
* table-sharded (below): the map's hash table is sharded by frame range — for maps
  that do not fit, or should not be replicated on, one GPU;
* query-sharded (`ReplicatedMap`): the table is replicated and every rank serves a
  slice of each query batch — the throughput mode for maps that fit one GPU.

Sharding: rank r owns map frames [lo_r, hi_r) and holds a
complete table for them, so every vote of a frame is counted on exactly one
rank and the final vote of each of its frames is local.  Per query batch each
rank probes its shard with all queries, takes its local top-`candidate_num`
(votes desc, frame id asc, >= 5 votes — STDesc.cpp:423-433), then ONE
all_gather of the (frame, votes) tables (candidate_num * 8 B per query and
rank) and an identical merge on every rank reproduce the single-table
candidate list bit for bit: the global top-k of disjoint frame sets is the
top-k of the union of the local top-k lists.  The match lists of the winners
stay on their owner rank (`owner_of`).
"""
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


class ShardedMap:
    """one rank's shard of the map + the collective query (one process per GPU)"""

    def __init__(self, n_frames_total, rank, world, device_id=0, **cfg):
        from .manager import STDescManager
        self.n_frames, self.rank, self.world = n_frames_total, rank, world
        self.lo, self.hi = shard_range(n_frames_total, world, rank)
        cfg.setdefault("max_frame_n", max(20000, n_frames_total + 1))
        self.mgr = STDescManager(first_frame_id=self.lo, device_id=device_id, **cfg)
        self.cand_num = self.mgr.config_setting_["candidate_num"]
        self._bufs = None

    def add_shard_frames(self, xyz, label, kp_off=None):
        """xyz/label of THIS rank's frames [lo, hi) in frame order"""
        self.mgr.add_frames(xyz, label, kp_off)
        self.mgr.finalize()

    def query(self, xyz, label, kp_off=None):
        """all ranks pass the same query batch; returns the global candidate list
        (frames, votes, n_cand) as device tensors, identical on every rank"""
        self.mgr.query_frames(xyz, label, kp_off, fetch=False)
        nq = self.mgr._nq
        dev = torch.device("cuda", self.mgr.config_setting_["device_id"])
        if self._bufs is None or self._bufs[0].shape[0] != nq:
            self._bufs = (torch.empty((nq, self.cand_num), dtype=torch.int32, device=dev),
                          torch.empty((nq, self.cand_num), dtype=torch.int32, device=dev))
        self.mgr.export_candidates(*self._bufs)
        return gather_and_merge(self._bufs[0], self._bufs[1], self.cand_num)

    def search_loop(self, xyz, label, kp_off=None, icp_threshold=None, group=None):
        """SearchLoop over the sharded map: query + merge, candidate_verify of the local
        candidates on every rank, one all_gather of (frames, scores, poses), SearchLoop's
        choice on the merged list.  Returns (frames, votes, n_cand, scores, poses, best_cand,
        best_frame, best_score), identical on every rank."""
        frames, votes, n_cand = self.query(xyz, label, kp_off)
        self.mgr.verify()
        nq, cn = frames.shape
        dev = frames.device
        score = torch.empty((nq, cn), dtype=torch.float64, device=dev)
        pose = torch.empty((nq, cn, 12), dtype=torch.float64, device=dev)
        self.mgr.export_verify(score, pose)
        sf, ss, sp = gather_verified(self._bufs[0], score, pose, group)
        scores, poses = merge_verified(frames, sf, ss, sp)
        thr = self.mgr.icp_threshold_ if icp_threshold is None else icp_threshold
        bc, bf, bs = search_loop_choice(frames, n_cand, scores, thr)
        return frames, votes, n_cand, scores, poses, bc, bf, bs


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


class ReplicatedMap:
    """multi-GPU mode for maps that fit one GPU: every rank holds the whole table and
    serves its slice of each query batch (no collective on the data path but the
    gather of the results)"""

    def __init__(self, n_frames_total, rank, world, device_id=0, **cfg):
        from .manager import STDescManager
        self.rank, self.world = rank, world
        cfg.setdefault("max_frame_n", max(20000, n_frames_total + 1))
        self.mgr = STDescManager(device_id=device_id, **cfg)
        self.cand_num = self.mgr.config_setting_["candidate_num"]
        self._bufs = None

    def add_frames(self, xyz, label, kp_off=None):
        self.mgr.add_frames(xyz, label, kp_off)
        self.mgr.finalize()

    def query(self, xyz_slice, label_slice, n_queries_total):
        """xyz_slice/label_slice: THIS rank's slice of the batch (uniform frames [n, N, 3]);
        returns (frames, votes) [n_queries_total, cn] on every rank"""
        self.mgr.query_frames(xyz_slice, label_slice, fetch=False)
        nq = self.mgr._nq
        dev = torch.device("cuda", self.mgr.config_setting_["device_id"])
        if self._bufs is None or self._bufs[0].shape[0] != nq:
            self._bufs = (torch.empty((nq, self.cand_num), dtype=torch.int32, device=dev),
                          torch.empty((nq, self.cand_num), dtype=torch.int32, device=dev))
        self.mgr.export_candidates(*self._bufs)
        return gather_query_slices(self._bufs[0], self._bufs[1], n_queries_total)
