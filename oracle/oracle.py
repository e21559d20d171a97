"""ctypes binding of the CPU oracle (oracle/libsgtd_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (sgtd_amd) never imports it.
PARITY UNPINNED — see oracle/sgtd_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsgtd_oracle.so")


class OrcConfig(C.Structure):
    _fields_ = [
        ("descriptor_near_num", C.c_int32),
        ("candidate_num", C.c_int32),
        ("max_frame_n", C.c_int32),
        ("num_threads", C.c_int32),
        ("descriptor_min_len", C.c_double),
        ("descriptor_max_len", C.c_double),
        ("std_side_resolution", C.c_double),
        ("rough_dis_threshold", C.c_double),
    ]


class OrcSoa(C.Structure):
    _fields_ = [
        ("side", C.c_void_p),
        ("angle", C.c_void_p),
        ("center", C.c_void_p),
        ("vertex", C.c_void_p),
        ("label", C.c_void_p),
        ("frame", C.c_void_p),
        ("node_id", C.c_void_p),
    ]


class OrcCounters(C.Structure):
    _fields_ = [
        ("D", C.c_int64), ("P", C.c_int64), ("M", C.c_int64),
        ("E", C.c_int64), ("U", C.c_int64),
        ("probe_ms", C.c_double), ("select_ms", C.c_double),
        ("build_ms", C.c_double),
    ]


class OrcAudit(C.Structure):
    """orc_audit of sgtd_oracle.h (parity-risk audit, tools/parity_audit.py)"""
    _fields_ = [("gate_tests", C.c_int64), ("gate_flips", C.c_int64 * 3), ("gate_flip_visits", C.c_int64 * 3),
                ("visits", C.c_int64), ("match_flips", C.c_int64 * 3), ("thr_ulp_flips", C.c_int64 * 2), ("near_calls", C.c_int64),
                ("min_margin", C.c_double), ("min_margin_ulps", C.c_double), ("min_gate_margin", C.c_double),
                ("knn_points", C.c_int64), ("knn_tied_points", C.c_int64), ("knn_fma_order_diffs", C.c_int64), ("triplets", C.c_int64),
                ("side_value_diffs", C.c_int64 * 3), ("build_flips", C.c_int64 * 3), ("min_len_margin", C.c_double), ("min_cell_margin", C.c_double)]

    def __init__(self):
        super().__init__()
        self.min_margin = self.min_margin_ulps = self.min_gate_margin = self.min_len_margin = self.min_cell_margin = float("inf")

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out


class OrcVerifyAudit(C.Structure):
    """orc_verify_audit of sgtd_oracle.h (tools/parity_audit.py --verify)"""
    _fields_ = [("candidates", C.c_int64), ("hypotheses", C.c_int64), ("pair_tests", C.c_int64), ("vertex_tests", C.c_int64),
                ("vertex_flips", C.c_int64), ("pair_flips", C.c_int64), ("vote_list_diffs", C.c_int64),
                ("best_index_diffs", C.c_int64), ("score_diffs", C.c_int64), ("inlier_set_diffs", C.c_int64), ("near_calls", C.c_int64),
                ("min_margin", C.c_double), ("max_norm_diff", C.c_double), ("max_rot_diff", C.c_double), ("max_t_diff", C.c_double)]

    def __init__(self):
        super().__init__()
        self.min_margin = float("inf")

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


def build_library(force=False):
    """compile oracle/libsgtd_oracle.so with the committed Makefile"""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) <
            max(os.path.getmtime(os.path.join(_HERE, f))
                for f in ("sgtd_oracle.cpp", "sgtd_oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


# an idle OpenMP team must sleep, not spin: the GPU tests and the bench alternate between the oracle
# and HIP calls / torch CPU kernels on hosts whose CPU quota is far below their thread count
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

_lib = None


def lib():
    global _lib
    if _lib is None:
        build_library()
        L = C.CDLL(_LIB_PATH)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        L.orc_create.restype = vp
        L.orc_create.argtypes = [C.POINTER(OrcConfig)]
        L.orc_destroy.argtypes = [vp]
        L.orc_current_frame_id.restype = C.c_uint32
        L.orc_current_frame_id.argtypes = [vp]
        L.orc_set_current_frame_id.argtypes = [vp, C.c_uint32]
        L.orc_set_num_threads.argtypes = [vp, C.c_int]
        L.orc_label_code.restype = C.c_int
        L.orc_label_code.argtypes = [C.c_int] * 3
        for f in ("orc_cell_key_eq", "orc_milli_key_eq"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [vp, vp]
        for f in ("orc_cell_key_hash", "orc_milli_key_hash"):
            getattr(L, f).restype = i64
            getattr(L, f).argtypes = [vp]
        L.orc_build.restype = i64
        L.orc_build.argtypes = [vp, vp, vp, C.c_int]
        L.orc_last_export.argtypes = [vp, C.POINTER(OrcSoa)]
        L.orc_add_last.argtypes = [vp]
        L.orc_add_frames.argtypes = [vp, vp, vp, C.c_int, C.c_int]
        L.orc_add.argtypes = [vp, C.POINTER(OrcSoa), i64]
        L.orc_select.restype = C.c_int
        L.orc_select.argtypes = [vp, C.c_int, C.POINTER(OrcSoa), i64, vp, vp, vp, vp]
        L.orc_cand_match_total.restype = i64
        L.orc_cand_match_total.argtypes = [vp]
        L.orc_cand_matches.argtypes = [vp, vp, vp]
        L.orc_rough_total.restype = i64
        L.orc_rough_total.argtypes = [vp]
        L.orc_rough_matches.argtypes = [vp] + [vp] * 6
        L.orc_votes.argtypes = [vp, vp]
        L.orc_fetch_entries.argtypes = [vp, vp, i64, C.POINTER(OrcSoa)]
        L.orc_table_dump.argtypes = [vp, vp, vp, vp]
        L.orc_get_counters.argtypes = [vp, C.POINTER(OrcCounters)]
        L.orc_verify.restype = C.c_double
        L.orc_verify.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Descs:
    """descriptor SoA held in numpy arrays (layout of orc_desc_soa)"""

    FIELDS = (("side", np.float64, 3), ("angle", np.float64, 3),
              ("center", np.float64, 3), ("vertex", np.float64, 9),
              ("label", np.int32, 3), ("frame", np.uint32, 1),
              ("node_id", np.int32, 3))

    def __init__(self, n):
        self.n = int(n)
        for name, dt, w in self.FIELDS:
            shape = (self.n, w) if w > 1 else (self.n,)
            setattr(self, name, np.zeros(shape, dtype=dt))

    def soa(self):
        s = OrcSoa()
        for name, _, _ in self.FIELDS:
            setattr(s, name, _p(getattr(self, name)))
        return s

    def take(self, idx):
        out = Descs(len(idx))
        for name, _, _ in self.FIELDS:
            setattr(out, name, np.ascontiguousarray(getattr(self, name)[idx]))
        return out


DEFAULTS = dict(descriptor_near_num=10, candidate_num=50, max_frame_n=20000,
                num_threads=1, descriptor_min_len=0.5, descriptor_max_len=50.0,
                std_side_resolution=1.0, rough_dis_threshold=0.03)


class OracleManager:
    """Mirror of STDescManager's hot-path methods on the CPU oracle.

    Shipped-YAML defaults: src/sgtd/config/SG_localization.yaml:74-89.
    """

    def __init__(self, **kw):
        cfg = dict(DEFAULTS)
        cfg.update(kw)
        self.cfg = cfg
        c = OrcConfig(**cfg)
        self._h = lib().orc_create(C.byref(c))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    @property
    def current_frame_id(self):
        return lib().orc_current_frame_id(self._h)

    def set_current_frame_id(self, fid):
        lib().orc_set_current_frame_id(self._h, int(fid))

    def set_num_threads(self, n):
        lib().orc_set_num_threads(self._h, int(n))
        self.cfg["num_threads"] = int(n)

    def build(self, xyz, label, export=True):
        """BuildSingleScanSTD; returns Descs (or the count if export=False)"""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        label = np.ascontiguousarray(label, dtype=np.uint32)
        n = lib().orc_build(self._h, _p(xyz), _p(label), xyz.shape[0])
        if not export:
            return n
        d = Descs(n)
        s = d.soa()
        lib().orc_last_export(self._h, C.byref(s))
        return d

    def add_last(self):
        lib().orc_add_last(self._h)

    def audit_build(self, xyz, label, acc):
        """BuildSingleScanSTD of one frame with every inferred piece of arithmetic in its alternatives (accumulates into
        the OrcAudit `acc`); the frame becomes the last built one"""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        label = np.ascontiguousarray(label, dtype=np.uint32)
        L = lib()
        L.orc_audit_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(OrcAudit)]
        L.orc_audit_build.restype = None
        L.orc_audit_build(self._h, _p(xyz), _p(label), xyz.shape[0], C.byref(acc))

    def audit_select(self, acc):
        """candidate_selector's loop for the last built frame, audited (accumulates into `acc`)"""
        L = lib()
        L.orc_audit_select.argtypes = [C.c_void_p, C.POINTER(OrcAudit)]
        L.orc_audit_select.restype = None
        L.orc_audit_select(self._h, C.byref(acc))

    def verify_hyp_inputs(self, cand):
        """per hypothesis of candidate `cand` of the last select: covariance matrix (:558), query centre, table centre"""
        L = lib()
        L.orc_verify_hyp_inputs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_verify_hyp_inputs.restype = C.c_int
        n = L.orc_verify_hyp_inputs(self._h, cand, None, None, None)
        cov, qc, ec = np.zeros((n, 3, 3)), np.zeros((n, 3)), np.zeros((n, 3))
        if n:
            L.orc_verify_hyp_inputs(self._h, cand, _p(cov), _p(qc), _p(ec))
        return cov, qc, ec

    def verify_hyp_solutions(self, cand):
        """the restatement's own hypotheses of candidate `cand`: [use_size, 12] (R row-major, t)"""
        L = lib()
        L.orc_verify_hyp_inputs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_verify_hyp_inputs.restype = C.c_int
        n = L.orc_verify_hyp_inputs(self._h, cand, None, None, None)
        rt = np.zeros((n, 12))
        L.orc_verify_hyp_solutions.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_verify_hyp_solutions.restype = None
        if n:
            L.orc_verify_hyp_solutions(self._h, cand, _p(rt))
        return rt

    def audit_verify(self, cand, other_rt, acc):
        """candidate_verify of candidate `cand` with this restatement's hypotheses AND `other_rt` ([n_hyp, 12]: R row-major, t)
        side by side (accumulates into the OrcVerifyAudit `acc`)"""
        other_rt = np.ascontiguousarray(other_rt, dtype=np.float64).reshape(-1, 12)
        L = lib()
        L.orc_audit_verify.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(OrcVerifyAudit)]
        L.orc_audit_verify.restype = None
        L.orc_audit_verify(self._h, cand, _p(other_rt), other_rt.shape[0], C.byref(acc))

    def add_frames(self, xyz, label):
        """xyz (F, N, 3), label (F, N): what F build + add_last pairs give (builds on all host threads)"""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        label = np.ascontiguousarray(label, dtype=np.uint32)
        lib().orc_add_frames(self._h, _p(xyz), _p(label), xyz.shape[0], xyz.shape[1])

    def add(self, d):
        s = d.soa()
        lib().orc_add(self._h, C.byref(s), d.n)

    def select(self, q=None):
        """candidate_selector; q=None uses the last built descriptors.

        Returns dict(cand_frame, cand_votes, cand_off, q_idx, db_entry)."""
        k = self.cfg["candidate_num"]
        cf = np.zeros(k, np.int32)
        cv = np.zeros(k, np.int32)
        co = np.zeros(k + 1, np.int64)
        nc = C.c_int32(0)
        if q is None:
            lib().orc_select(self._h, 1, None, 0, _p(cf), _p(cv), _p(co), C.addressof(nc))
        else:
            s = q.soa()
            lib().orc_select(self._h, 0, C.byref(s), q.n, _p(cf), _p(cv), _p(co),
                             C.addressof(nc))
        nc = nc.value
        tot = lib().orc_cand_match_total(self._h)
        qi = np.zeros(tot, np.int32)
        de = np.zeros(tot, np.int64)
        lib().orc_cand_matches(self._h, _p(qi), _p(de))
        return dict(cand_frame=cf[:nc].copy(), cand_votes=cv[:nc].copy(),
                    cand_off=co[:nc + 1].copy(), q_idx=qi, db_entry=de)

    def rough_matches(self):
        n = lib().orc_rough_total(self._h)
        out = dict(q_idx=np.zeros(n, np.int32), cell=np.zeros(n, np.int32),
                   j=np.zeros(n, np.int32), db_entry=np.zeros(n, np.int64),
                   frame=np.zeros(n, np.uint32), dis=np.zeros(n, np.float64))
        lib().orc_rough_matches(self._h, _p(out["q_idx"]), _p(out["cell"]), _p(out["j"]),
                                _p(out["db_entry"]), _p(out["frame"]), _p(out["dis"]))
        return out

    def votes(self):
        v = np.zeros(self.cfg["max_frame_n"], np.float64)
        lib().orc_votes(self._h, _p(v))
        return v

    def fetch_entries(self, db_entry):
        db_entry = np.ascontiguousarray(db_entry, dtype=np.int64)
        d = Descs(len(db_entry))
        s = d.soa()
        lib().orc_fetch_entries(self._h, _p(db_entry), len(db_entry), C.byref(s))
        return d

    def counters(self):
        c = OrcCounters()
        lib().orc_get_counters(self._h, C.byref(c))
        return {f[0]: getattr(c, f[0]) for f in OrcCounters._fields_}

    def table_dump(self):
        c = self.counters()
        keys = np.zeros((c["U"], 4), np.int64)
        off = np.zeros(c["U"] + 1, np.int64)
        ids = np.zeros(c["E"], np.int64)
        lib().orc_table_dump(self._h, _p(keys), _p(off), _p(ids))
        return keys, off, ids

    def verify(self, cand, n_pairs):
        t = np.zeros(3)
        rot = np.zeros(9)
        idx = np.zeros(max(n_pairs, 1), np.int32)
        ns = C.c_int32(0)
        score = lib().orc_verify(self._h, cand, _p(t), _p(rot), _p(idx), C.addressof(ns))
        return score, t, rot.reshape(3, 3), idx[:ns.value].copy()


def label_code(a, b, c):
    return lib().orc_label_code(a, b, c)


REF_PIN_PATH = os.path.join(_HERE, "_ref", "libsgtd_ref_pin.so")


def ref_pin():
    """oracle/_ref/libsgtd_ref_pin.so — the pieces of the reference itself that compile with
    standard headers (oracle/ref_pin.cpp, built by `make -C oracle` where /root/reference
    exists; the built library travels to the GPU box).  None if it was never built."""
    if not os.path.exists(REF_PIN_PATH):
        return None
    L = C.CDLL(REF_PIN_PATH)
    L.ref_label_code.restype = C.c_int
    L.ref_label_code.argtypes = [C.c_int] * 3
    for f in ("ref_hash_p", "ref_max_n", "ref_max_frame_n"):
        getattr(L, f).restype = C.c_int64
    for f in ("ref_voxel_eq", "ref_loc_eq"):
        getattr(L, f).restype = C.c_int
        getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
    for f in ("ref_voxel_hash", "ref_loc_hash"):
        getattr(L, f).restype = C.c_int64
        getattr(L, f).argtypes = [C.c_void_p]
    return L
