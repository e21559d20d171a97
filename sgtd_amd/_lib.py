"""ctypes binding of libsgtd_accel.so (the C ABI in include/sgtd_accel.h).

There is no fallback: if the shared library is missing, or no gfx950 device is
usable, this module raises — the product path never computes on the CPU.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SGTD_ACCEL_LIB points at another build of the same ABI (kernel variants under test)
LIB_PATH = os.environ.get("SGTD_ACCEL_LIB") or os.path.join(_HERE, "libsgtd_accel.so")
CSRC = os.path.join(_HERE, "csrc")

SGTD_OK = 0
ERRORS = {-8: "IO", -1: "INVALID", -2: "NO_DEVICE", -3: "HIP", -4: "CAPACITY",
          -5: "FRAME_LIMIT", -6: "UNSUPPORTED", -7: "STATE"}

# every symbol include/sgtd_accel.h declares
SYMBOLS = [
    "sgtd_default_config", "sgtd_create", "sgtd_create_multi", "sgtd_device_count", "sgtd_device_handle", "sgtd_destroy", "sgtd_strerror", "sgtd_last_error", "sgtd_label_code", "sgtd_table_key", "sgtd_dedup_key",
    "sgtd_set_stream", "sgtd_set_timing", "sgtd_current_frame_id", "sgtd_max_descs", "sgtd_build",
    "sgtd_add", "sgtd_add_frames", "sgtd_finalize", "sgtd_query_frames", "sgtd_query_descs", "sgtd_max_batch",
    "sgtd_result_candidates", "sgtd_export_candidates_dev", "sgtd_result_query_desc_count", "sgtd_result_pairs",
    "sgtd_result_query_descs", "sgtd_result_votes", "sgtd_result_rough", "sgtd_fetch_entries", "sgtd_host_alloc", "sgtd_host_free",
    "sgtd_table_dump", "sgtd_sync", "sgtd_get_stats",
    "sgtd_verify", "sgtd_export_verify_dev", "sgtd_result_verify", "sgtd_result_inliers", "sgtd_result_inlier_pairs", "sgtd_result_inlier_entries", "sgtd_search_loop",
    "sgtd_graphs_load", "sgtd_graphs_save_cache", "sgtd_graphs_load_cache", "sgtd_graphs_view",
    "sgtd_graphs_error", "sgtd_graphs_free", "sgtd_save_table", "sgtd_load_table",
    "sgtd_candidate_export_ints", "sgtd_set_candidate_export", "sgtd_export_wait", "sgtd_export_release", "sgtd_merge_candidates_dev",
    "sgtd_gather_verified_dev", "sgtd_set_deferred_lists", "sgtd_finish_lists", "sgtd_verify_masked", "sgtd_attach_table", "sgtd_search_frame",
]


class SgtdError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__("sgtd_accel: %s (%d) %s" % (ERRORS.get(status, "?"), status, detail))


class Config(C.Structure):
    _fields_ = [
        ("descriptor_near_num", C.c_int32),
        ("candidate_num", C.c_int32),
        ("max_frame_n", C.c_int32),
        ("device_id", C.c_int32),
        ("descriptor_min_len", C.c_double),
        ("descriptor_max_len", C.c_double),
        ("std_side_resolution", C.c_double),
        ("rough_dis_threshold", C.c_double),
        ("first_frame_id", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class DescSoa(C.Structure):
    _fields_ = [("side", C.c_void_p), ("angle", C.c_void_p), ("center", C.c_void_p),
                ("vertex", C.c_void_p), ("label", C.c_void_p), ("frame", C.c_void_p),
                ("node_id", C.c_void_p)]


class FrameSearch(C.Structure):
    """sgtd_frame_search"""
    _fields_ = [("n_cand", C.c_int32), ("flags", C.c_int32), ("cand_frame", C.c_void_p), ("cand_votes", C.c_void_p),
                ("pair_off", C.c_void_p), ("score", C.c_void_p), ("pose", C.c_void_p), ("inlier_off", C.c_void_p),
                ("inlier_q_idx", C.c_void_p), ("entries", DescSoa), ("capacity", C.c_int64), ("n_inliers", C.c_int64)]


class Stats(C.Structure):
    _fields_ = [
        ("n_entries", C.c_int64), ("n_buckets", C.c_int64), ("n_frames", C.c_int64),
        ("last_queries", C.c_int64), ("last_D", C.c_int64), ("last_P", C.c_int64),
        ("last_M", C.c_int64), ("last_cand_pairs", C.c_int64), ("hbm_bytes_table", C.c_int64),
        ("ms_build", C.c_float), ("ms_sort", C.c_float), ("ms_probe", C.c_float),
        ("ms_votes", C.c_float), ("ms_topk", C.c_float),
        ("ms_count", C.c_float), ("ms_scan", C.c_float), ("ms_write", C.c_float),
        ("ms_total", C.c_float), ("overflowed", C.c_int32), ("select_form", C.c_int32),
        ("last_P_swept", C.c_int64), ("bucket_len_sq_over_E", C.c_double),
        ("tail_entries", C.c_int64), ("ms_finalize", C.c_float), ("reserved2", C.c_float),
        ("batches_total", C.c_int64), ("overflow_launches_total", C.c_int64), ("reruns_total", C.c_int64), ("rewrites_total", C.c_int64),
        ("list_moves_total", C.c_int64), ("last_list_moves", C.c_int64), ("device_allocs_total", C.c_int64),
    ]


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of the in-tree shared library"""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(os.path.dirname(_HERE), "include", "sgtd_accel.h")]
    stale = (not os.path.exists(LIB_PATH) or
             os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs))
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SgtdError(-2, "libsgtd_accel.so is not built (run __graft_entry__.build() or "
                            "make -C sgtd_amd/csrc); there is no CPU fallback")
    # One HIP runtime per process: PyTorch ships its own libamdhip64 (same SONAME as the system
    # one this library is linked against).  Loaded first, it also serves this library; loaded
    # second, it finds the GPU already opened by the other copy and reports "No HIP GPUs".  The
    # package uses torch for device buffers, streams and torch.distributed anyway (dist.py,
    # bench.py), so pin the order here.  C/C++ hosts link the system runtime and never see torch.
    if os.environ.get("SGTD_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.sgtd_default_config.argtypes = [C.POINTER(Config)]
    L.sgtd_default_config.restype = None
    L.sgtd_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.sgtd_destroy.argtypes = [vp]
    L.sgtd_create_multi.argtypes = [C.POINTER(Config), C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.sgtd_device_count.argtypes = [vp]
    L.sgtd_device_handle.argtypes = [vp, C.c_int]
    L.sgtd_device_handle.restype = vp
    L.sgtd_strerror.argtypes = [C.c_int]
    L.sgtd_strerror.restype = C.c_char_p
    L.sgtd_last_error.argtypes = [vp]
    L.sgtd_last_error.restype = C.c_char_p
    L.sgtd_label_code.argtypes = [C.c_int] * 3
    L.sgtd_label_code.restype = C.c_uint32
    L.sgtd_table_key.argtypes = [C.c_uint32] * 4
    L.sgtd_table_key.restype = C.c_uint64
    L.sgtd_dedup_key.argtypes = [C.c_uint64] * 3
    L.sgtd_dedup_key.restype = C.c_uint64
    L.sgtd_set_stream.argtypes = [vp, vp]
    L.sgtd_set_timing.argtypes = [vp, C.c_int]
    L.sgtd_current_frame_id.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.sgtd_max_descs.argtypes = [vp, C.c_int]
    L.sgtd_max_descs.restype = i64
    L.sgtd_build.argtypes = [vp, vp, vp, C.c_int, C.POINTER(DescSoa), i64, C.POINTER(i64)]
    L.sgtd_add.argtypes = [vp, C.POINTER(DescSoa), i64]
    L.sgtd_add_frames.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int]
    L.sgtd_finalize.argtypes = [vp]
    L.sgtd_query_frames.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int]
    L.sgtd_query_descs.argtypes = [vp, C.POINTER(DescSoa), i64]
    L.sgtd_result_candidates.argtypes = [vp, vp, vp, vp, vp]
    L.sgtd_export_candidates_dev.argtypes = [vp, vp, vp]
    L.sgtd_result_query_desc_count.argtypes = [vp, C.c_int, C.POINTER(i64)]
    L.sgtd_result_pairs.argtypes = [vp, C.c_int, vp, vp, i64, C.POINTER(i64)]
    L.sgtd_result_query_descs.argtypes = [vp, C.c_int, C.POINTER(DescSoa), i64, C.POINTER(i64)]
    L.sgtd_result_votes.argtypes = [vp, C.c_int, vp, i64, C.POINTER(C.c_uint32), C.POINTER(i64)]
    L.sgtd_result_rough.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, i64, C.POINTER(i64)]
    L.sgtd_fetch_entries.argtypes = [vp, vp, i64, C.POINTER(DescSoa)]
    L.sgtd_table_dump.argtypes = [vp, vp, vp, vp, i64, i64]
    L.sgtd_sync.argtypes = [vp]
    L.sgtd_max_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_int64)]
    L.sgtd_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.sgtd_verify.argtypes = [vp]
    L.sgtd_result_verify.argtypes = [vp, C.c_int, vp, vp]
    L.sgtd_export_verify_dev.argtypes = [vp, vp, vp]
    L.sgtd_result_inliers.argtypes = [vp, C.c_int, C.c_int, vp, i64, C.POINTER(i64)]
    L.sgtd_result_inlier_pairs.argtypes = [vp, C.c_int, vp, vp, vp, i64, C.POINTER(i64)]
    L.sgtd_result_inlier_entries.argtypes = [vp, C.c_int, vp, vp, C.POINTER(DescSoa), i64, C.POINTER(i64)]
    L.sgtd_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.sgtd_host_free.argtypes = [vp]
    L.sgtd_search_loop.argtypes = [vp, C.c_double, vp, vp, vp]
    L.sgtd_graphs_load.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(vp)]
    L.sgtd_graphs_save_cache.argtypes = [vp, C.c_char_p]
    L.sgtd_graphs_load_cache.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.sgtd_graphs_view.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(i64), C.POINTER(vp), C.POINTER(vp),
                                   C.POINTER(vp), C.POINTER(vp)]
    L.sgtd_graphs_error.argtypes = [vp]
    L.sgtd_graphs_error.restype = C.c_char_p
    L.sgtd_graphs_free.argtypes = [vp]
    L.sgtd_graphs_free.restype = None
    L.sgtd_candidate_export_ints.argtypes = [C.c_int, C.c_int]
    L.sgtd_candidate_export_ints.restype = i64
    L.sgtd_set_candidate_export.argtypes = [vp, vp, i64]
    L.sgtd_export_wait.argtypes = [vp, vp]
    L.sgtd_export_release.argtypes = [vp, vp]
    L.sgtd_merge_candidates_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    L.sgtd_gather_verified_dev.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, vp, vp]
    L.sgtd_set_deferred_lists.argtypes = [vp, C.c_int]
    L.sgtd_finish_lists.argtypes = [vp, vp]
    L.sgtd_verify_masked.argtypes = [vp, vp]
    L.sgtd_attach_table.argtypes = [vp, vp]
    L.sgtd_search_frame.argtypes = [vp, C.POINTER(DescSoa), i64, C.POINTER(FrameSearch)]
    L.sgtd_save_table.argtypes = [vp, C.c_char_p]
    L.sgtd_load_table.argtypes = [vp, C.c_char_p]
    for name in SYMBOLS:
        getattr(L, name)
        if getattr(L, name).restype is C.c_int:
            pass
    _lib = L
    return L
