"""What a finer partition of the buckets would save the sweep — counted, not built.

The table's buckets (one reference cell + label code, STDesc.cpp:149-172) are cut into SGTD_YSLICES x SGTD_ZSLICES = 2 x 3
sub-cells plus an overflow slice (table_kernels.hip.h: sub_cell, slice_assign_kernel); plan_passes_kernel gives every pass of up
to four query descriptors, per gated cell and per half, ONE range from the first to the last third any of them reaches
(probe_kernels.hip.h: reached_slices, cell()), plus the overflow slice.  This script restates exactly that in numpy for
ny x nz sub-cells — the slices, the overflow rule, the passes, the ranges — on the benchmark's own table and a batch of its
queries, taken from the product on the GPU (descriptors, entries), and counts the entries a sweep would load (P_swept) and
the entry-descriptor tests it would make.  The 2 x 3 count must equal what the product's sweep reports for the same batch
(sgtd_stats.last_P_swept): that pins the model.  Everything else is arithmetic on the host.

    python tools/partition_model.py [frames=10000] [queries=256] [partitions=2x3,2x4,...] -> one JSON object (profiles/r06_partition_model*.json)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def f32_up(x):
    """the f32 not below x (write_query_rec's t_up)"""
    f = x.astype(np.float32)
    low = f.astype(np.float64) < x
    f[low] = np.nextafter(f[low], np.float32(np.inf))
    return f


def axis_slice(s, n):                       # table_kernels.hip.h: axis_slice
    y = s + 0.5
    cell = y.astype(np.int64)
    return np.clip((y * n).astype(np.int64) - cell * n, 0, n - 1)


def reached(dq, t_up, n):                   # probe_kernels.hip.h: reached_slices (f32 arithmetic)
    dq = dq.astype(np.float32)
    a = ((dq - t_up) + np.float32(0.5)) * np.float32(n) - np.float32(1e-4)
    b = ((dq + t_up) + np.float32(0.5)) * np.float32(n) + np.float32(1e-4)
    lo = np.where(~(a > 0), 0, np.where(a >= n, n, np.minimum(a, n).astype(np.int64)))
    hi = np.where(b < 0, -1, np.where(~(b < n), n - 1, np.maximum(b, -1).astype(np.int64)))
    return lo, hi


def pack_key(code, x, y, z):
    return (code.astype(np.int64) << 48) | (x.astype(np.int64) << 32) | (y.astype(np.int64) << 16) | z.astype(np.int64)


def label_code(lab):
    return ((lab[:, 0] & 15) << 8) | ((lab[:, 1] & 15) << 4) | (lab[:, 2] & 15)


def overflow_runs(bucket, frame, side, sub, rough, run_max=48):
    """slice_assign_kernel: a (bucket, frame) run two of whose members lie in different sub-cells and could both match one
    query descriptor goes to the overflow slice as a whole; runs beyond run_max untested"""
    E = len(bucket)
    order = np.lexsort((np.arange(E), frame, bucket))
    b, f = bucket[order], frame[order]
    head = np.ones(E, bool)
    head[1:] = (b[1:] != b[:-1]) | (f[1:] != f[:-1])
    run = np.cumsum(head) - 1
    length = np.bincount(run)
    first = np.flatnonzero(head)
    over_run = length > run_max
    fct = 2.0 * rough / (1.0 - rough) * (1.0 + 1e-9)
    norm = np.sqrt((side[:, 0] ** 2 + side[:, 1] ** 2) + side[:, 2] ** 2)
    for L in np.unique(length):
        if L < 2 or L > run_max:
            continue
        runs = np.flatnonzero(length == L)
        for c0 in range(0, len(runs), 200000 // int(L * L) + 1):
            rr = runs[c0:c0 + 200000 // int(L * L) + 1]
            idx = order[first[rr][:, None] + np.arange(L)[None, :]]              # (n, L) entries of the runs
            s, sb, nm = side[idx], sub[idx], norm[idx]
            d = s[:, :, None, :] - s[:, None, :, :]
            dist = np.sqrt((d[..., 0] ** 2 + d[..., 1] ** 2) + d[..., 2] ** 2)
            lim = fct * np.maximum(nm[:, :, None], nm[:, None, :]) + 1e-9
            close = ~(dist > lim) & (sb[:, :, None] != sb[:, None, :])
            over_run[rr] |= close.any(axis=(1, 2))
    over = np.zeros(E, bool)
    over[order] = over_run[run]
    return over


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    from sgtd_amd import manager, synth
    t0 = time.time()
    smap = synth.make_map(F, 200, stream=1)
    qs = synth.make_queries(smap, NQ, stream=1000)
    g = manager.STDescManager()
    rough = g.config_setting_["rough_dis_threshold"]
    g.add_frames(smap.xyz, smap.label)
    g.query_frames(qs.xyz, qs.label, fetch=False)
    g.sync()
    st = g.stats()
    qd = [g.result_query_descs(q) for q in range(NQ)]
    E = int(st["n_entries"]) if "n_entries" in st else None
    if E is None:
        gk, goff, gid = g.table_dump()
        E = len(gid)
    ent = g.fetch_entries(np.arange(E, dtype=np.int64))
    g.close()
    t1 = time.time()

    # ---- the table
    side, frame = ent.side, ent.frame.astype(np.int64)
    cell = (side + 0.5).astype(np.int64)                                             # STDesc.cpp:156-157
    key = pack_key(label_code(ent.label.astype(np.int64)), cell[:, 0], cell[:, 1], cell[:, 2])
    ukeys, bucket = np.unique(key, return_inverse=True)
    U = len(ukeys)

    # ---- the batch: descriptors in the sweep's order (home key: label code, home cell, position inside the cell: 2 x 2), passes of four
    qside = np.concatenate([d.side for d in qd])
    qlab = np.concatenate([d.label for d in qd]).astype(np.int64)
    D = len(qside)
    home = qside.astype(np.int64)
    frac = qside - home
    sub2 = np.clip((frac[:, 1] * 2).astype(np.int64), 0, 1) * 2 + np.clip((frac[:, 2] * 2).astype(np.int64), 0, 1)
    code = label_code(qlab)
    hkey = ((((code << 6 | home[:, 0]) << 6 | home[:, 1]) << 6) | home[:, 2]) << 2 | sub2
    order = np.argsort(hkey, kind="stable")
    grp = hkey[order] >> 2
    head = np.ones(D, bool)
    head[1:] = grp[1:] != grp[:-1]
    gfirst = np.flatnonzero(head)
    pos_in_group = np.arange(D) - gfirst[np.cumsum(head) - 1]
    pass_head = pos_in_group % 4 == 0
    pass_of = np.cumsum(pass_head) - 1
    n_pass = int(pass_of[-1]) + 1
    members = np.full((n_pass, 4), -1, np.int64)
    members[pass_of, pos_in_group % 4] = order
    K = (members >= 0).sum(1)
    cols = np.where(K == 3, 4, K)
    m0 = members[:, 0]

    thr = np.sqrt((qside[:, 0] ** 2 + qside[:, 1] ** 2) + qside[:, 2] ** 2) * rough
    t_up = f32_up(thr * (1.0 + 1e-9) + 1e-12)
    # gate: ||side - centre|| < 1.5 per probe cell (STDesc.cpp:366-369; common.hip.h: gate_mask)
    ex = [(qside - ((qside + (k - 1)).astype(np.int64) + 0.5)) ** 2 for k in range(3)]
    gate = np.zeros((D, 27), bool)
    for c in range(27):
        gate[:, c] = ((ex[c // 9][:, 0] + ex[(c // 3) % 3][:, 1]) + ex[c % 3][:, 2]) < 2.25

    out = {"what": "entries the sweep would load (P_swept) and entry-descriptor tests per batch under ny x nz sub-cells per bucket, counted on the host from the product's own table and query descriptors (tools/partition_model.py); 2 x 3 is what is built",
           "frames": F, "queries": NQ, "table_entries": E, "buckets": U, "descriptors": D, "passes": n_pass,
           "product_P_swept_of_this_batch": int(st["last_P_swept"]), "product_P_visited_of_this_batch": int(st["last_P"]), "partitions": {}}

    # the bucket of every (pass, cell): the home cell's neighbours (group_resolve_kernel)
    hx = home[m0]
    pcode = code[m0]
    bidx = np.full((n_pass, 27), -1, np.int64)
    for c in range(27):
        ix, iy, iz = c // 9 - 1, (c // 3) % 3 - 1, c % 3 - 1
        k = pack_key(pcode, np.maximum(hx[:, 0] + ix, 0), np.maximum(hx[:, 1] + iy, 0), np.maximum(hx[:, 2] + iz, 0))
        p = np.searchsorted(ukeys, k)
        p[p >= U] = U - 1
        bidx[:, c] = np.where(ukeys[p] == k, p, -1)
    blen = np.bincount(bucket, minlength=U)
    # the reference's visits: every entry of every gated cell's bucket, per descriptor
    visited = 0
    mem_valid = members >= 0
    memc = np.where(mem_valid, members, 0)
    for c in range(27):
        has = bidx[:, c] >= 0
        gl = gate[memc, c] & mem_valid                                            # (n_pass, 4)
        visited += int((gl.sum(1) * np.where(has, blen[np.maximum(bidx[:, c], 0)], 0)).sum())
    out["model_P_visited"] = visited

    parts = [tuple(int(v) for v in p.split("x")) for p in (sys.argv[3] if len(sys.argv) > 3 else "2x3,2x4,2x5,2x6,2x8,3x3,3x4,4x4,3x6,4x6").split(",")]
    for ny, nz in parts:
        ta = time.time()
        NS = ny * nz
        sub = axis_slice(side[:, 1], ny) * nz + axis_slice(side[:, 2], nz)
        over = overflow_runs(bucket, frame, side, sub, rough)
        sl = np.where(over, NS, sub)
        cnt = np.bincount(bucket * (NS + 1) + sl, minlength=U * (NS + 1)).reshape(U, NS + 1)
        cum = np.zeros((U, NS + 2), np.int64)
        cum[:, 1:] = np.cumsum(cnt, axis=1)                                       # cum[b, s] = entries of slices < s
        # reach per descriptor and offset o = -1, 0, +1 (the cell at (int)(q + o))
        ylo, yhi, zlo, zhi = [np.zeros((D, 3), np.int64) for _ in range(4)]
        for o in range(3):
            ylo[:, o], yhi[:, o] = reached(qside[:, 1] - (qside[:, 1] + (o - 1)).astype(np.int64), t_up, ny)
            zlo[:, o], zhi[:, o] = reached(qside[:, 2] - (qside[:, 2] + (o - 1)).astype(np.int64), t_up, nz)
        loads = np.zeros(n_pass, np.int64)
        ranges = np.zeros(n_pass, np.int64)
        for c in range(27):
            oy, oz = (c // 3) % 3, c % 3
            has = bidx[:, c] >= 0
            b = np.maximum(bidx[:, c], 0)
            gl = gate[memc, c] & mem_valid
            live = gl.any(1) & has
            zl, zh = zlo[memc, oz], zhi[memc, oz]
            yl, yh = ylo[memc, oy], yhi[memc, oy]
            for yy in range(ny):
                m = gl & (yl <= yy) & (yy <= yh) & (zl <= zh)
                first = np.where(m, zl, nz).min(1)
                last = np.where(m, zh, -1).max(1)
                on = live & (last >= first)
                ln = np.where(on, cum[b, yy * nz + np.minimum(last, nz - 1) + 1] - cum[b, yy * nz + np.minimum(first, nz - 1)], 0)
                loads += ln
                ranges += ln > 0
            ov = np.where(live, cnt[b, NS], 0)
            loads += ov
            ranges += ov > 0
        P = int(loads.sum())
        out["partitions"]["%dx%d" % (ny, nz)] = {
            "P_swept": P, "tests": int((loads * cols).sum()), "ranges": int(ranges.sum()), "entries_in_overflow_slices": int(over.sum()),
            "directory_row_words": 1 + NS + 1, "s": round(time.time() - ta, 1)}
    base = out["partitions"]["2x3"]
    for k, v in out["partitions"].items():
        v["P_swept_vs_2x3"] = round(v["P_swept"] / base["P_swept"], 4)
        v["tests_vs_2x3"] = round(v["tests"] / base["tests"], 4)
        # the sweep's measured cost model (DESIGN §3 "the pass width"): 0.69 ms per 1e9 entries loaded + 0.52 ms per 1e9 tests, scaled to the bench's 2048 queries
        v["model_sweep_ms_at_2048_queries"] = round((0.69e-9 * v["P_swept"] + 0.52e-9 * v["tests"]) * 2048 / NQ, 3)
    out["model_equals_product_for_2x3"] = base["P_swept"] == out["product_P_swept_of_this_batch"]
    out["P_visited_equals_product"] = visited == out["product_P_visited_of_this_batch"]
    out["seconds"] = {"product": round(t1 - t0, 1), "model": round(time.time() - t1, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
