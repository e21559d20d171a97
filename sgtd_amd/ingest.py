"""Graph-JSON ingest: the reference's on-disk graph files
(`include/Semantic_Graph.hpp:122-184`, producer `src/get_json.cpp:332-341`) -> the keypoint
batches `STDescManager.add_frames` / `query_frames` take.  Parsing runs in the native library
on host threads; this module only wraps the C ABI and writes files in the same format for
the synthetic generator."""
import ctypes as C
import json
import os

import numpy as np

from . import _lib


class GraphBatch:
    """xyz f32 [K,3], label u32 [K], kp_off i64 [F+1], poses f32 [F,12] (copies)"""

    def __init__(self, xyz, label, kp_off, poses):
        self.xyz, self.label, self.kp_off, self.poses = xyz, label, kp_off, poses

    @property
    def n_frames(self):
        return len(self.kp_off) - 1

    def position(self):
        """this_poses of the node: (poses[3], poses[7], poses[11])
        (semantic_graph_localization.cpp:447)"""
        return self.poses[:, [3, 7, 11]]


def _take(L, h):
    try:
        nf, nk = C.c_int(0), C.c_int64(0)
        px, pl, po, pp = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        st = L.sgtd_graphs_view(h, C.byref(nf), C.byref(nk), C.byref(px), C.byref(pl), C.byref(po), C.byref(pp))
        if st != 0:
            raise _lib.SgtdError(st, "sgtd_graphs_view")
        nf, nk = nf.value, nk.value

        def arr(ptr, ctype, n, dtype, shape):
            if n == 0:
                return np.zeros(shape, dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).astype(dtype, copy=True).reshape(shape)

        return GraphBatch(arr(px, C.c_float, nk * 3, np.float32, (nk, 3)), arr(pl, C.c_uint32, nk, np.uint32, (nk,)),
                          arr(po, C.c_int64, nf + 1, np.int64, (nf + 1,)), arr(pp, C.c_float, nf * 12, np.float32, (nf, 12)))
    finally:
        L.sgtd_graphs_free(h)


def load_graphs(paths, threads=None):
    """parse graph JSON files (frame order = order of `paths`); raises SgtdError(IO) naming
    the file that could not be opened or parsed"""
    L = _lib.lib()
    paths = [os.fspath(p) for p in paths]
    arr = (C.c_char_p * max(len(paths), 1))(*[p.encode() for p in paths])
    h = C.c_void_p()
    st = L.sgtd_graphs_load(arr, len(paths), int(threads or min(32, os.cpu_count() or 1)), C.byref(h))
    if st != 0:
        msg = L.sgtd_graphs_error(h).decode() if h else ""
        if h:
            L.sgtd_graphs_free(h)
        raise _lib.SgtdError(st, msg)
    return _take(L, h)


def cache_graphs(paths, cache_path, threads=None):
    """parse `paths` once and write the binary cache; returns the batch"""
    L = _lib.lib()
    paths = [os.fspath(p) for p in paths]
    arr = (C.c_char_p * max(len(paths), 1))(*[p.encode() for p in paths])
    h = C.c_void_p()
    st = L.sgtd_graphs_load(arr, len(paths), int(threads or min(32, os.cpu_count() or 1)), C.byref(h))
    if st != 0:
        msg = L.sgtd_graphs_error(h).decode() if h else ""
        if h:
            L.sgtd_graphs_free(h)
        raise _lib.SgtdError(st, msg)
    st = L.sgtd_graphs_save_cache(h, os.fspath(cache_path).encode())
    if st != 0:
        L.sgtd_graphs_free(h)
        raise _lib.SgtdError(st, "cannot write %s" % cache_path)
    return _take(L, h)


def load_cache(cache_path):
    L = _lib.lib()
    h = C.c_void_p()
    st = L.sgtd_graphs_load_cache(os.fspath(cache_path).encode(), C.byref(h))
    if st != 0:
        msg = L.sgtd_graphs_error(h).decode() if h else ""
        if h:
            L.sgtd_graphs_free(h)
        raise _lib.SgtdError(st, msg)
    return _take(L, h)


def write_graph_json(path, xyz, label, pose12, extra=True):
    """one frame in the producer's format (Graph::toJSON, Semantic_Graph.hpp:79-110);
    floats are written with repr precision so that f32 values round-trip"""
    xyz = np.asarray(xyz, np.float32)
    doc = {"nodes": [int(v) for v in np.asarray(label)]}
    if extra:
        doc["edges"] = [[0.0, 1.0]] if len(xyz) > 1 else []
        doc["weights"] = [0.5] if len(xyz) > 1 else []
    doc["centers"] = [[float(a), float(b), float(c)] for a, b, c in xyz]
    doc["poses"] = [float(v) for v in np.asarray(pose12, np.float32)]
    if extra:
        doc["volumes"] = []
        doc["densitys"] = []
    with open(path, "w") as f:
        json.dump(doc, f)


CACHE_MAGIC = b"SGTDGB01"


def write_cache(cache_path, xyz, label, pose12):
    """the binary graph-batch cache of sgtd_graphs_save_cache (graph_ingest.hip.h) straight from
    arrays: xyz (F, N, 3) f32, label (F, N), pose12 (F, 12) f32 — what sgtd_graphs_load_cache and
    examples/localize read without parsing a JSON file per frame"""
    xyz = np.ascontiguousarray(xyz, np.float32)
    f, n = xyz.shape[0], xyz.shape[1]
    with open(cache_path, "wb") as fh:
        fh.write(CACHE_MAGIC)
        fh.write(np.array([f, f * n], np.int64).tobytes())
        fh.write((np.arange(f + 1, dtype=np.int64) * n).tobytes())
        fh.write(np.ascontiguousarray(pose12, np.float32).reshape(f, 12).tobytes())
        fh.write(np.ascontiguousarray(label).astype(np.uint32).reshape(f * n).tobytes())
        fh.write(xyz.reshape(f * n * 3).tobytes())
