"""The pin against the reference's own code (CPU, no GPU): oracle/_ref/libsgtd_ref_pin.so holds
Combinatorial_Binary_Encoding (STDesc.cpp:3-16), VOXEL_LOC / STDesc_LOC with their operator==
and std::hash (STDesc.h:126-154,217-250) and the constants (STDesc.h:31-33), cut out of the
reference by line range and compiled verbatim (oracle/Makefile, oracle/ref_pin.cpp).  Checked
here: the oracle's restatements AND the product's key helpers (the inline functions the kernels
use, include/sgtd_accel.h) agree with it — rows a7, a5, a8 and a12's constants of SURVEY.md §8a.
The rest of the reference's path cannot be compiled in this image (Eigen/PCL/ROS/Ceres)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as orc

REF = orc.ref_pin()
needs_ref = pytest.mark.skipif(REF is None, reason="oracle/_ref was not built (no /root/reference at build time)")


def _product():
    from sgtd_amd import _lib
    _lib.build_library()
    L = C.CDLL(_lib.LIB_PATH)     # the key helpers need no device
    L.sgtd_label_code.argtypes = [C.c_int] * 3
    L.sgtd_label_code.restype = C.c_uint32
    L.sgtd_table_key.argtypes = [C.c_uint32] * 4
    L.sgtd_table_key.restype = C.c_uint64
    L.sgtd_dedup_key.argtypes = [C.c_uint64] * 3
    L.sgtd_dedup_key.restype = C.c_uint64
    return L


def test_ref_pin_is_built_where_the_reference_is():
    if os.path.isdir("/root/reference/src/sgtd"):
        assert REF is not None, "run `make -C oracle` (or __graft_entry__.build())"


@needs_ref
def test_constants_are_the_references():
    assert (REF.ref_hash_p(), REF.ref_max_n(), REF.ref_max_frame_n()) == (116101, 10000000000, 20000)
    from sgtd_amd import _lib
    cfg = _lib.Config()
    _product().sgtd_default_config(C.byref(cfg))
    assert cfg.max_frame_n == REF.ref_max_frame_n()


@needs_ref
def test_label_code_all_inputs_and_wraps():
    P, O = _product(), orc.lib()
    # every (a, b, c) in 0..15, the known answers of SURVEY §8c, and wrap cases: negative labels,
    # labels beyond 15 (the wild label map), INT_MIN / INT_MAX
    cases = [(a, b, c) for a in range(16) for b in range(16) for c in range(16)]
    rng = np.random.default_rng(7)
    wild = [-2147483648, -17, -16, -1, 16, 17, 31, 255, 4096, 2147483647]
    cases += [(int(a), int(b), int(c)) for a, b, c in rng.integers(-40, 60, size=(2000, 3))]
    cases += [(a, b, c) for a in wild for b in wild for c in wild]
    for a, b, c in cases:
        r = REF.ref_label_code(a, b, c)
        assert O.orc_label_code(a, b, c) == r, (a, b, c)
        assert P.sgtd_label_code(a, b, c) == r, (a, b, c)
    assert REF.ref_label_code(3, 10, 11) == 939 and REF.ref_label_code(17, 5, -1) == 351


def _arr(v):
    return np.ascontiguousarray(v, np.int64)


@needs_ref
def test_table_key_equality_and_hash():
    """STDesc_LOC: equality on (x, y, z, a) only — b, c never matter (STDesc.h:235)"""
    P, O = _product(), orc.lib()
    rng = np.random.default_rng(11)
    n = 20000
    k1 = rng.integers(0, 6, size=(n, 6)).astype(np.int64)        # small range: many equal pairs
    k2 = rng.integers(0, 6, size=(n, 6)).astype(np.int64)
    k2[: n // 4, :4] = k1[: n // 4, :4]                           # equal in x, y, z, a; b, c differ
    big = rng.integers(0, 65536, size=(n, 6)).astype(np.int64)    # the envelope's cells
    big[:, 3] = rng.integers(0, 4096, size=n)
    n_eq = 0
    for p, q in list(zip(k1, k2)) + list(zip(big, np.roll(big, 1, axis=0))) + list(zip(big, big)):
        p, q = _arr(p), _arr(q)
        r = REF.ref_loc_eq(p.ctypes.data, q.ctypes.data)
        n_eq += r
        assert O.orc_cell_key_eq(p.ctypes.data, q.ctypes.data) == r
        same = P.sgtd_table_key(int(p[3]), int(p[0]), int(p[1]), int(p[2])) == P.sgtd_table_key(int(q[3]), int(q[0]), int(q[1]), int(q[2]))
        assert int(same) == r, (p, q)
        assert O.orc_cell_key_hash(p.ctypes.data) == REF.ref_loc_hash(p.ctypes.data)
    assert n_eq > n // 4 + n
    # the code field is Combinatorial_Binary_Encoding of the ordered labels (STDesc.cpp:161,365)
    for la, lb, lc in rng.integers(0, 16, size=(500, 3)):
        code = REF.ref_label_code(int(la), int(lb), int(lc))
        assert P.sgtd_table_key(P.sgtd_label_code(int(la), int(lb), int(lc)), 1, 2, 3) == P.sgtd_table_key(code, 1, 2, 3)


@needs_ref
def test_dedup_key_equality_and_hash():
    """VOXEL_LOC of the millimetre sides (STDesc.cpp:244-251)"""
    P, O = _product(), orc.lib()
    rng = np.random.default_rng(13)
    n = 20000
    a = rng.integers(0, 50001, size=(n, 3)).astype(np.int64)     # (int64)(float)(side * 1000), side <= 50 m
    b = a.copy()
    flip = rng.integers(0, 4, size=n)
    for i in range(n):
        if flip[i] < 3:
            b[i, flip[i]] += int(rng.integers(1, 3))
    edge = np.array([[0, 0, 0], [2097151, 2097151, 2097151], [2097151, 0, 0], [0, 2097151, 0], [0, 0, 2097151]], np.int64)
    pairs = list(zip(a, b)) + [(x, y) for x in edge for y in edge]
    for p, q in pairs:
        p, q = _arr(p), _arr(q)
        r = REF.ref_voxel_eq(p.ctypes.data, q.ctypes.data)
        assert O.orc_milli_key_eq(p.ctypes.data, q.ctypes.data) == r
        same = P.sgtd_dedup_key(int(p[0]), int(p[1]), int(p[2])) == P.sgtd_dedup_key(int(q[0]), int(q[1]), int(q[2]))
        assert int(same) == r, (p, q)
        assert O.orc_milli_key_hash(p.ctypes.data) == REF.ref_voxel_hash(p.ctypes.data)
