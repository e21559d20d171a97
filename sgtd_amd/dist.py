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
