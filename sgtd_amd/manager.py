"""Host-side mirror of the reference's operator API for the hot path.

`STDescManager` keeps the reference class's method names and argument meaning
(src/sgtd/include/desc/STDesc.h:342-440):

    BuildSingleScanSTD(cloud)  -> descriptors          STDesc.cpp:174-315
    AddSTDescs(descriptors)                            STDesc.cpp:149-172
    candidate_selector(descs)  -> [STDMatchList]       STDesc.cpp:318-460

plus the batched, device-resident forms the GPU wants (`add_frames`,
`query_frames`).  Everything computes in libsgtd_accel.so (HIP, gfx950); this
file only marshals numpy / torch buffers through the C ABI.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import Config, DescSoa, SgtdError, Stats

DEFAULTS = dict(descriptor_near_num=10, candidate_num=50, max_frame_n=20000, device_id=0,
                descriptor_min_len=0.5, descriptor_max_len=50.0, std_side_resolution=1.0,
                rough_dis_threshold=0.03, first_frame_id=0)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Descs:
    """descriptor structure-of-arrays (layout of sgtd_desc_soa), numpy backed"""

    FIELDS = (("side", np.float64, 3), ("angle", np.float64, 3), ("center", np.float64, 3),
              ("vertex", np.float32, 9), ("label", np.int32, 3), ("frame", np.uint32, 1),
              ("node_id", np.int32, 3))

    def __init__(self, n):
        self.n = int(n)
        for name, dt, w in self.FIELDS:
            setattr(self, name, np.zeros((self.n, w) if w > 1 else (self.n,), dtype=dt))

    def soa(self):
        s = DescSoa()
        for name, _, _ in self.FIELDS:
            setattr(s, name, _p(getattr(self, name)))
        return s

    def head(self, n):
        out = Descs(0)
        out.n = int(n)
        for name, _, _ in self.FIELDS:
            setattr(out, name, np.ascontiguousarray(getattr(self, name)[:n]))
        return out

    def take(self, idx):
        out = Descs(0)
        out.n = len(idx)
        for name, _, _ in self.FIELDS:
            setattr(out, name, np.ascontiguousarray(getattr(self, name)[idx]))
        return out


class STDMatchList:
    """STDMatchList (STDesc.h:120-124): match_id_ = (query frame id, map frame id);
    match_list_ as (query descriptor index, table entry index) pairs in order"""

    def __init__(self, query_frame, map_frame, votes, q_idx, db_entry):
        self.match_id_ = (int(query_frame), int(map_frame))
        self.votes = int(votes)
        self.q_idx = q_idx
        self.db_entry = db_entry

    def __len__(self):
        return len(self.q_idx)


class BatchResult:
    """candidate_selector output of a batch of query frames (host copies)"""

    def __init__(self, n_cand, cand_frame, cand_votes, pair_off, query_frame_id):
        self.n_cand = n_cand
        self.cand_frame = cand_frame
        self.cand_votes = cand_votes
        self.pair_off = pair_off
        self.query_frame_id = query_frame_id

    def top1(self):
        """map frame with the most votes per query (-1 if no candidate)"""
        return np.where(self.n_cand > 0, self.cand_frame[:, 0], -1)


class STDescManager:
    def __init__(self, **kw):
        """devices=[ids]: one handle over several GPUs of this process (sgtd_create_multi: the
        table sharded by frame blocks, host-side merge of the per-device candidate tables)"""
        cfg = dict(DEFAULTS)
        self.icp_threshold_ = float(kw.pop("icp_threshold", 0.4))   # SG_localization.yaml:89
        devices = kw.pop("devices", None)
        cfg.update(kw)
        self.config_setting_ = cfg
        self._L = _lib.lib()
        c = Config(**cfg)
        h = C.c_void_p()
        self._h = None
        if devices is None:
            self._check(self._L.sgtd_create(C.byref(c), C.byref(h)))
        else:
            ids = (C.c_int * len(devices))(*[int(d) for d in devices])
            self._check(self._L.sgtd_create_multi(C.byref(c), ids, len(devices), C.byref(h)))
        self._h = h
        self._keep = None

    @property
    def device_count(self):
        return self._L.sgtd_device_count(self._h)

    def close(self):
        if self._h is not None:
            st = self._L.sgtd_destroy(self._h)
            if st != 0:       # (an owner whose table other managers still borrow: close them first)
                raise SgtdError(st, self._L.sgtd_last_error(self._h).decode())
            self._h = None
            self._owner = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, status):
        if status != 0:
            detail = ""
            if self._h is not None:
                detail = self._L.sgtd_last_error(self._h).decode()
            raise SgtdError(status, detail)

    # ---- plumbing -------------------------------------------------------
    def set_stream(self, stream_ptr):
        self._check(self._L.sgtd_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_timing(self, on):
        self._check(self._L.sgtd_set_timing(self._h, int(bool(on))))

    @property
    def current_frame_id_(self):
        v = C.c_uint32(0)
        self._check(self._L.sgtd_current_frame_id(self._h, C.byref(v)))
        return v.value

    def stats(self):
        s = Stats()
        self._check(self._L.sgtd_get_stats(self._h, C.byref(s)))
        return {f[0]: getattr(s, f[0]) for f in Stats._fields_}

    def sync(self):
        self._check(self._L.sgtd_sync(self._h))

    def max_batch(self, n_keypoints):
        """largest safe number of query frames (of n_keypoints keypoints each) per query_frames call"""
        n = C.c_int64(0)
        self._check(self._L.sgtd_max_batch(self._h, int(n_keypoints), C.byref(n)))
        return n.value

    # ---- BuildSingleScanSTD ----------------------------------------------
    def BuildSingleScanSTD(self, xyz, label):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        label = np.ascontiguousarray(label, dtype=np.uint32)
        n = xyz.shape[0]
        cap = self._L.sgtd_max_descs(self._h, n)
        d = Descs(cap)
        s = d.soa()
        n_out = C.c_int64(0)
        self._check(self._L.sgtd_build(self._h, _p(xyz), _p(label), n, C.byref(s), cap, C.byref(n_out)))
        return d.head(n_out.value)

    # ---- AddSTDescs -------------------------------------------------------
    def AddSTDescs(self, d):
        s = d.soa()
        self._check(self._L.sgtd_add(self._h, C.byref(s), d.n))

    def add_frames(self, xyz, label, kp_off=None):
        """BuildSingleScanSTD + AddSTDescs for a whole run of frames on the device
        (the caller's map loop, semantic_graph_localization.cpp:419-458).
        xyz (F,N,3)/(total,3) numpy or torch.cuda tensor; kp_off None => uniform N."""
        xp, lp, off, nf, dev = self._frames_args(xyz, label, kp_off)
        self._check(self._L.sgtd_add_frames(self._h, xp, lp, _p(off), nf, dev))

    def finalize(self):
        self._check(self._L.sgtd_finalize(self._h))

    def attach_table(self, owner):
        """borrow the finalized table of `owner` (another manager on the same device): this manager then queries the
        same map with its own work buffers and stream — two batches in flight (include/sgtd_accel.h)"""
        self._check(self._L.sgtd_attach_table(self._h, owner._h))
        self._owner = owner         # (the owner must outlive its views)

    def _frames_args(self, xyz, label, kp_off):
        is_torch = hasattr(xyz, "data_ptr")
        if is_torch:
            assert xyz.is_cuda and label.is_cuda and xyz.is_contiguous() and label.is_contiguous()
            shape = tuple(xyz.shape)
            xp, lp, dev = C.c_void_p(xyz.data_ptr()), C.c_void_p(label.data_ptr()), 1
            self._keep = (xyz, label)
        else:
            xyz = np.ascontiguousarray(xyz, dtype=np.float32)
            label = np.ascontiguousarray(label, dtype=np.uint32)
            shape = xyz.shape
            xp, lp, dev = _p(xyz), _p(label), 0
            self._keep = (xyz, label)
        if kp_off is None:
            assert len(shape) == 3
            nf, n = shape[0], shape[1]
            off = np.arange(nf + 1, dtype=np.int64) * n
        else:
            off = np.ascontiguousarray(kp_off, dtype=np.int64)
            nf = len(off) - 1
        return xp, lp, off, nf, dev

    # ---- candidate_selector -------------------------------------------------
    def query_frames(self, xyz, label, kp_off=None, fetch=True):
        """fused BuildSingleScanSTD + candidate_selector for a batch of query frames
        (semantic_graph_localization.cpp:592,601 -> STDesc.cpp:98).  Asynchronous when
        fetch=False (results via .results())."""
        xp, lp, off, nq, dev = self._frames_args(xyz, label, kp_off)
        self._nq = nq
        self._check(self._L.sgtd_query_frames(self._h, xp, lp, _p(off), nq, dev))
        return self.results() if fetch else None

    def results(self):
        nq, cn = self._nq, self.config_setting_["candidate_num"]
        n_cand = np.zeros(nq, np.int32)
        cf = np.zeros((nq, cn), np.int32)
        cv = np.zeros((nq, cn), np.int32)
        po = np.zeros((nq, cn + 1), np.int64)
        self._check(self._L.sgtd_result_candidates(self._h, _p(n_cand), _p(cf), _p(cv), _p(po)))
        return BatchResult(n_cand, cf, cv, po, self.current_frame_id_)

    def export_candidates(self, d_frame, d_votes):
        """async D2D copy of the (n_queries, candidate_num) int32 candidate tables into
        torch.cuda tensors (no host synchronisation)"""
        self._check(self._L.sgtd_export_candidates_dev(self._h, C.c_void_p(d_frame.data_ptr()),
                                                       C.c_void_p(d_votes.data_ptr())))

    # ---- the multi-GPU step (include/sgtd_accel.h, sgtd_amd/dist.py) -----------------------------------------
    @staticmethod
    def frames_in(xyz, kp_off=None):
        """query frames a batch call carries"""
        return int(xyz.shape[0]) if kp_off is None else len(kp_off) - 1

    def set_candidate_export(self, d_packed):
        """every batch writes its packed local candidate tables into this torch.cuda int32 tensor as soon as they are
        final (None: off)"""
        if d_packed is None:
            self._check(self._L.sgtd_set_candidate_export(self._h, None, 0))
        else:
            self._check(self._L.sgtd_set_candidate_export(self._h, C.c_void_p(d_packed.data_ptr()), d_packed.numel()))
        self._export_keep = d_packed

    def export_wait(self, side_stream_ptr):
        self._check(self._L.sgtd_export_wait(self._h, C.c_void_p(side_stream_ptr)))

    def export_release(self, side_stream_ptr):
        self._check(self._L.sgtd_export_release(self._h, C.c_void_p(side_stream_ptr)))

    def merge_candidates_dev(self, stream_ptr, gathered, n_tables, my_table, nq, frame, votes, n_cand, src, keep, flags):
        """the merge of STDesc.cpp:423-433 over n_tables packed tables as one kernel on `stream_ptr` (torch.cuda tensors)"""
        self._check(self._L.sgtd_merge_candidates_dev(self._h, C.c_void_p(stream_ptr), C.c_void_p(gathered.data_ptr()), n_tables, my_table, nq,
                                                      C.c_void_p(frame.data_ptr()), C.c_void_p(votes.data_ptr()), C.c_void_p(n_cand.data_ptr()),
                                                      C.c_void_p(src.data_ptr()), C.c_void_p(keep.data_ptr()), C.c_void_p(flags.data_ptr())))

    def gather_verified_dev(self, stream_ptr, gathered, n_tables, src, nq, score, pose):
        self._check(self._L.sgtd_gather_verified_dev(self._h, C.c_void_p(stream_ptr), C.c_void_p(gathered.data_ptr()), n_tables,
                                                     C.c_void_p(src.data_ptr()), nq, C.c_void_p(score.data_ptr()), C.c_void_p(pose.data_ptr())))

    def set_deferred_lists(self, on):
        self._check(self._L.sgtd_set_deferred_lists(self._h, int(bool(on))))

    def finish_lists(self, keep=None):
        """write the match lists of the last batch — of the candidates in the u64-per-query device mask only (None: all)"""
        self._check(self._L.sgtd_finish_lists(self._h, None if keep is None else C.c_void_p(keep.data_ptr())))

    def verify_masked(self, keep):
        self._check(self._L.sgtd_verify_masked(self._h, None if keep is None else C.c_void_p(keep.data_ptr())))

    def result_pairs(self, q, res):
        cn = self.config_setting_["candidate_num"]
        n = int(res.pair_off[q, cn])
        qi = np.zeros(n, np.int32)
        de = np.zeros(n, np.int64)
        got = C.c_int64(0)
        self._check(self._L.sgtd_result_pairs(self._h, q, _p(qi), _p(de), n, C.byref(got)))
        return qi, de

    def result_query_descs(self, q):
        n = C.c_int64(0)
        self._check(self._L.sgtd_result_query_desc_count(self._h, q, C.byref(n)))
        d = Descs(n.value)
        s = d.soa()
        got = C.c_int64(0)
        self._check(self._L.sgtd_result_query_descs(self._h, q, C.byref(s), n.value, C.byref(got)))
        return d

    def result_votes(self, q):
        lo = C.c_uint32(0)
        n = C.c_int64(0)
        self._check(self._L.sgtd_result_votes(self._h, q, None, 0, C.byref(lo), C.byref(n)))
        v = np.zeros(n.value, np.uint32)
        self._check(self._L.sgtd_result_votes(self._h, q, _p(v), n.value, C.byref(lo), C.byref(n)))
        return lo.value, v

    def result_rough(self, q, with_dis=True):
        n = C.c_int64(0)
        st = self._L.sgtd_result_rough(self._h, q, None, None, None, None, None, 0, C.byref(n))
        if st not in (0, -4):
            self._check(st)
        m = n.value
        out = dict(q_idx=np.zeros(m, np.int32), cell=np.zeros(m, np.int32),
                   db_entry=np.zeros(m, np.int64), frame=np.zeros(m, np.uint32),
                   dis=np.zeros(m, np.float64) if with_dis else None)
        self._check(self._L.sgtd_result_rough(self._h, q, _p(out["q_idx"]), _p(out["cell"]),
                                              _p(out["db_entry"]), _p(out["frame"]),
                                              _p(out["dis"]), m, C.byref(n)))
        return out

    def candidate_selector(self, stds_vec):
        """one query frame given as descriptors -> list of STDMatchList"""
        s = stds_vec.soa()
        self._nq = 1
        self._check(self._L.sgtd_query_descs(self._h, C.byref(s), stds_vec.n))
        res = self.results()
        qi, de = self.result_pairs(0, res)
        out = []
        for k in range(int(res.n_cand[0])):
            lo, hi = res.pair_off[0, k], res.pair_off[0, k + 1]
            out.append(STDMatchList(self.current_frame_id_, res.cand_frame[0, k], res.cand_votes[0, k],
                                    qi[lo:hi], de[lo:hi]))
        return out

    # ---- table access -------------------------------------------------------
    # ---- geometric verification (STDesc.cpp:462-571) and SearchLoop's choice (:84-147)
    def verify(self):
        """candidate_verify for every (query, candidate) of the last batch, on the device"""
        self._check(self._L.sgtd_verify(self._h))

    def export_verify(self, d_score, d_pose):
        """async D2D copy of verify_score [n_queries, candidate_num] f64 and pose
        [n_queries, candidate_num, 12] f64 into torch.cuda tensors (no host synchronisation)"""
        self._check(self._L.sgtd_export_verify_dev(self._h, C.c_void_p(d_score.data_ptr()), C.c_void_p(d_pose.data_ptr())))

    def result_verify(self, q):
        """-> (score[candidate_num], rot[candidate_num,3,3], t[candidate_num,3])"""
        cn = self.config_setting_["candidate_num"]
        score = np.zeros(cn, np.float64)
        pose = np.zeros((cn, 12), np.float64)
        self._check(self._L.sgtd_result_verify(self._h, q, _p(score), _p(pose)))
        return score, pose[:, :9].reshape(cn, 3, 3).copy(), pose[:, 9:].copy()

    def result_inliers(self, q, cand, n_pairs):
        """sucess_match_vec of one candidate as positions into its match_list_"""
        idx = np.zeros(max(int(n_pairs), 1), np.int32)
        n = C.c_int64(0)
        self._check(self._L.sgtd_result_inliers(self._h, q, cand, _p(idx), len(idx), C.byref(n)))
        return idx[:n.value].copy()

    def result_inlier_entries(self, q, capacity):
        """sucess_match_vec of EVERY candidate of query q with the table side already fetched: ->
        (cand_off[candidate_num + 1], q_idx[n], Descs of the n table entries); capacity = the sum of the
        candidates' list lengths (result pair_off's last entry) always suffices"""
        cn = self.config_setting_["candidate_num"]
        off = np.zeros(cn + 1, np.int64)
        qi = np.zeros(max(int(capacity), 1), np.int32)
        d = Descs(max(int(capacity), 1))
        s = d.soa()
        n = C.c_int64(0)
        self._check(self._L.sgtd_result_inlier_entries(self._h, q, _p(off), _p(qi), C.byref(s), int(capacity), C.byref(n)))
        return off, qi[:n.value].copy(), d.head(n.value)

    def search_loop(self, icp_threshold=None):
        """SearchLoop's result for every query of the last batch (after verify()):
        (best_cand, best_frame, best_score) arrays; frame -1 / score 0 = no loop (:144)"""
        if icp_threshold is None:
            icp_threshold = self.icp_threshold_
        nq = self._nq
        bc = np.zeros(nq, np.int32)
        bf = np.zeros(nq, np.int32)
        bs = np.zeros(nq, np.float64)
        self._check(self._L.sgtd_search_loop(self._h, float(icp_threshold), _p(bc), _p(bf), _p(bs)))
        return bc, bf, bs

    def search_frame(self, stds_vec, capacity=16384, page_locked=False, lists_only=False):
        """sgtd_search_frame: candidate_selector + candidate_verify + the inlier pairs of every candidate with their table
        entries for ONE query frame given as descriptors, in one call -> dict(n_cand, cand_frame, cand_votes, pair_off,
        score, rot, t, inlier_off, inlier_q_idx, entries (Descs), n_inliers, status).  page_locked: the arrays the inlier
        pairs arrive in come from sgtd_host_alloc, as adapter/STDesc_shim.hpp keeps them — the device then writes them in
        place and the call has one wait (ordinary arrays are filled from the handle's own page-locked block).
        lists_only (SGTD_FRAME_LISTS_ONLY): candidate_selector alone — no verification, inlier_off = pair_off and the pairs
        handed back are all pairs of every candidate's match list"""
        from ._lib import FrameSearch
        cn = self.config_setting_["candidate_num"]
        cap = max(int(capacity), 1)
        out = dict(cand_frame=np.zeros(cn, np.int32), cand_votes=np.zeros(cn, np.int32), pair_off=np.zeros(cn + 1, np.int64),
                   score=np.zeros(cn, np.float64), pose=np.zeros((cn, 12), np.float64), inlier_off=np.zeros(cn + 1, np.int64))
        blocks = []
        if page_locked:
            def room(dt, w):
                p = C.c_void_p()
                self._check(self._L.sgtd_host_alloc(cap * w * np.dtype(dt).itemsize, C.byref(p)))
                blocks.append(p)
                a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), shape=(cap * w * np.dtype(dt).itemsize,)).view(dt)
                return a.reshape((cap, w)) if w > 1 else a
            ent = Descs(0)
            ent.n = cap
            for name, dt, w in Descs.FIELDS:
                setattr(ent, name, room(dt, w))
            out["inlier_q_idx"] = room(np.int32, 1)
        else:
            ent = Descs(cap)
            out["inlier_q_idx"] = np.zeros(cap, np.int32)
        try:
            fs = FrameSearch()
            for k in ("cand_frame", "cand_votes", "pair_off", "score", "pose", "inlier_off", "inlier_q_idx"):
                setattr(fs, k, out[k].ctypes.data)
            fs.entries = ent.soa()
            fs.capacity = int(capacity)
            fs.flags = 1 if lists_only else 0
            s = stds_vec.soa()
            self._nq = 1
            st = self._L.sgtd_search_frame(self._h, C.byref(s), stds_vec.n, C.byref(fs))
            if st not in (0, -4):
                self._check(st)
            n = int(fs.n_inliers)
            out.update(status=st, n_cand=int(fs.n_cand), n_inliers=n, rot=out["pose"][:, :9].reshape(cn, 3, 3).copy(), t=out["pose"][:, 9:].copy(),
                       inlier_q_idx=out["inlier_q_idx"][:min(n, capacity)].copy(), entries=ent.head(min(n, capacity)))
            if page_locked:     # (head() of a contiguous slice is a view: the results leave the block before it is given back)
                for name, _, _ in Descs.FIELDS:
                    setattr(out["entries"], name, getattr(out["entries"], name).copy())
        finally:
            ent = None
            for p in blocks:
                self._L.sgtd_host_free(p)
        return out

    def SearchLoop(self, stds_vec, icp_threshold=None):
        """mirror of STDescManager::SearchLoop (STDesc.cpp:84-147) for one query given as
        descriptors -> (loop_result (frame, score), (t, rot), success pair positions,
        match_result_list [(frame, score, (t, rot), positions)])"""
        if stds_vec.n == 0:                       # "No STDescs!" (:89-93)
            return (-1, 0.0), None, np.zeros(0, np.int32), []
        cands = self.candidate_selector(stds_vec)
        self.verify()
        score, rot, t = self.result_verify(0)
        mrl = []
        for k, c in enumerate(cands):
            inl = self.result_inliers(0, k, len(c.q_idx)) if score[k] >= 0 else np.zeros(0, np.int32)
            mrl.append((int(c.match_id_[1]), float(score[k]), (t[k], rot[k]), inl))
        bc, bf, bs = self.search_loop(icp_threshold)
        if bf[0] < 0:
            return (-1, 0.0), None, np.zeros(0, np.int32), mrl
        k = int(bc[0])
        return (int(bf[0]), float(bs[0])), (t[k], rot[k]), mrl[k][3], mrl

    # ---- persistent table
    def save_table(self, path):
        self._check(self._L.sgtd_save_table(self._h, os.fspath(path).encode()))

    def load_table(self, path):
        """replace this manager's table with a saved one; AddSTDescs / add_frames keep appending"""
        self._check(self._L.sgtd_load_table(self._h, os.fspath(path).encode()))

    def fetch_entries(self, db_entry):
        db_entry = np.ascontiguousarray(db_entry, dtype=np.int64)
        d = Descs(len(db_entry))
        s = d.soa()
        self._check(self._L.sgtd_fetch_entries(self._h, _p(db_entry), len(db_entry), C.byref(s)))
        return d

    def table_dump(self):
        self.finalize()
        # the dump shows whole buckets: a table with a tail segment is merged first (the sizing
        # call below does it and reports the capacity it needs)
        self._L.sgtd_table_dump(self._h, None, None, None, 0, 0)
        st = self.stats()
        u, e = st["n_buckets"], st["n_entries"]
        keys = np.zeros((u, 4), np.int64)
        off = np.zeros(u + 1, np.int64)
        ids = np.zeros(e, np.int64)
        self._check(self._L.sgtd_table_dump(self._h, _p(keys), _p(off), _p(ids), u, e))
        return keys, off, ids
