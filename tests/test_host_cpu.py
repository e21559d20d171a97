"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol
the header declares, host-side merge logic, generator determinism."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_library_exports_every_declared_symbol():
    from sgtd_amd import _lib
    _lib.build_library()
    header = open(os.path.join(ROOT, "include", "sgtd_accel.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(sgtd_[a-z_0-9]+)\s*\(", header))
    assert len(declared) >= 25
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), name


def test_abi_non_compute_calls_without_gpu():
    """calls that need no device: defaults, error strings; create must fail loudly
    (no CPU fallback) when no gfx950 device is present"""
    import torch
    from sgtd_amd import _lib
    L = _lib.lib()
    cfg = _lib.Config()
    L.sgtd_default_config(ctypes.byref(cfg))
    assert (cfg.descriptor_near_num, cfg.candidate_num, cfg.max_frame_n) == (10, 50, 20000)
    assert (cfg.descriptor_min_len, cfg.descriptor_max_len) == (0.5, 50.0)
    assert (cfg.std_side_resolution, cfg.rough_dis_threshold) == (1.0, 0.03)
    assert L.sgtd_strerror(0) == b"ok" and b"gfx950" in L.sgtd_strerror(-2)
    if not torch.cuda.is_available():
        h = ctypes.c_void_p()
        assert L.sgtd_create(ctypes.byref(cfg), ctypes.byref(h)) == -2
        from sgtd_amd.manager import STDescManager, SgtdError
        with pytest.raises(SgtdError):
            STDescManager()
    bad = _lib.Config()
    L.sgtd_default_config(ctypes.byref(bad))
    bad.descriptor_near_num = 40
    h = ctypes.c_void_p()
    assert L.sgtd_create(ctypes.byref(bad), ctypes.byref(h)) == -6


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sgtd_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "oracle" not in open(os.path.join(dirpath, f)).read().lower(), os.path.join(dirpath, f)


def test_merge_candidates_rule():
    import torch
    from sgtd_amd.dist import merge_candidates
    # two ranks, one query, cand_num 4: votes desc, ties -> lowest frame id, >= 5 votes only
    f = torch.tensor([[[7, 2, -1, -1]], [[11, 12, 13, -1]]], dtype=torch.int32)
    v = torch.tensor([[[9, 6, 0, 0]], [[9, 6, 5, 0]]], dtype=torch.int32)
    mf, mv, n = merge_candidates(f, v, 4)
    assert mf.tolist() == [[7, 11, 2, 12]] and mv.tolist() == [[9, 9, 6, 6]] and n.tolist() == [4]
    mf, mv, n = merge_candidates(f[:1], v[:1], 4)
    assert mf.tolist() == [[7, 2, -1, -1]] and n.tolist() == [2]


def test_shard_ranges_cover_and_owner():
    from sgtd_amd.dist import owner_of, shard_range
    for n, w in ((10, 3), (1000, 8), (7, 8), (100000, 8)):
        ranges = [shard_range(n, w, r) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(ranges[i][1] == ranges[i + 1][0] for i in range(w - 1))
        for f in (0, n // 2, n - 1):
            lo, hi = ranges[owner_of(f, n, w)]
            assert lo <= f < hi


def test_generator_is_deterministic_and_tie_free_checker():
    from sgtd_amd import synth
    a = synth.make_map(5, 40, stream=77)
    b = synth.make_map(5, 40, stream=77)
    np.testing.assert_array_equal(a.xyz, b.xyz)
    np.testing.assert_array_equal(a.label, b.label)
    assert a.xyz.dtype == np.float32 and a.label.dtype == np.uint32
    assert a.label.min() >= 3 and a.label.max() <= 11
    qa = synth.make_queries(a, 3, stream=77)
    qb = synth.make_queries(b, 3, stream=77)
    np.testing.assert_array_equal(qa.xyz, qb.xyz)
    grid = np.zeros((16, 3), np.float32)
    grid[:, 0] = np.arange(16) % 4
    grid[:, 1] = np.arange(16) // 4
    assert synth.has_knn_ties(grid, 5)
    assert not synth.has_knn_ties(a.xyz[0], 10)


def test_merge_verified_and_search_loop_choice():
    import torch
    from sgtd_amd.dist import merge_verified, search_loop_choice
    # two ranks, two queries, cand_num 3.  Rank 0 owns frames < 10, rank 1 frames >= 10.
    sf = torch.tensor([[[3, 7, -1], [2, -1, -1]], [[12, 15, 11], [19, 14, -1]]], dtype=torch.int32)
    ss = torch.tensor([[[40.0, -1.0, -1.0], [8.0, -1.0, -1.0]], [[55.0, 40.0, 6.0], [-1.0, 8.0, -1.0]]], dtype=torch.float64)
    sp = torch.arange(2 * 2 * 3 * 12, dtype=torch.float64).reshape(2, 2, 3, 12)
    gf = torch.tensor([[12, 3, 15], [2, 14, -1]], dtype=torch.int32)      # merged lists (votes order)
    scores, poses = merge_verified(gf, sf, ss, sp)
    assert scores.tolist() == [[55.0, 40.0, 40.0], [8.0, 8.0, -1.0]]
    assert torch.equal(poses[0, 0], sp[1, 0, 0]) and torch.equal(poses[0, 1], sp[0, 0, 0]) and torch.equal(poses[1, 1], sp[1, 1, 1])
    assert torch.count_nonzero(poses[1, 2]) == 0
    n_cand = torch.tensor([3, 2], dtype=torch.int32)
    bc, bf, bs = search_loop_choice(gf, n_cand, scores, 0.4)
    assert bc.tolist() == [0, 0] and bf.tolist() == [12, 2] and bs.tolist() == [55.0, 8.0]   # ties -> first candidate
    bc, bf, bs = search_loop_choice(gf, n_cand, scores, 20.0)
    assert bc.tolist() == [0, -1] and bf.tolist() == [12, -1] and bs.tolist() == [55.0, 0.0]  # (-1, 0): no loop
    none = torch.full((1, 3), -1.0, dtype=torch.float64)
    bc, bf, bs = search_loop_choice(gf[:1], n_cand[:1], none, 0.4)
    assert bc.tolist() == [-1] and bs.tolist() == [0.0]


def test_bench_refuses_a_gpu_count_that_disagrees_with_the_launcher():
    """bench.py --gpus N under a launcher that started another number of ranks must fail loudly,
    before anything touches a GPU"""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=120, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=3" in (out.stderr + out.stdout)


def test_bench_finds_the_committed_traffic_row_of_its_default_workload():
    """bench.py's roofline object takes the HBM bytes per sweep launch and the VALU issue fraction
    from the newest profiles/r<NN>_traffic.json (PMC passes cannot run inside the bench): the row of the
    default workload must be there, with the fields the bench reads, and be self-consistent"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import sys
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    row = bench.load_traffic(args.frames, args.keypoints, args.queries, 1)
    assert row is not None, "no row for the default workload in profiles/r*_traffic.json"
    k = row["kernels"][row["kernel"]]        # the sweep's bytes: 2 x FETCH_SIZE + WRITE_SIZE, per launch
    assert row["bytes_per_launch"] == int(k["read_bytes"] + k["write_bytes"]) and k["launches_per_step"] == 1
    assert 0.0 < row["valu_issue_frac"] <= 1.0
    assert abs(row["valu_issue_frac"] - row["SQ_INSTS_VALU"] * 4.0 / (row["kernel_cycles"] * 1024.0)) < 1e-9
    assert os.path.exists(os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % row["profile_tag"]))


def test_committed_traffic_row_describes_the_kernels_the_bench_launches():
    """bench.py refuses a PMC row that was taken on other kernels than the ones the run launches: the row committed
    under profiles/ for the default workload must carry the kernels of the per-query form (sgtd_stats.select_form 2),
    every kernel name bench.py expects must exist in the sources, and a row of another form must be refused"""
    import importlib.util
    import json
    import pytest
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    src = "".join(open(os.path.join(ROOT, "sgtd_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "sgtd_amd", "csrc")) if f.endswith((".h", ".hip")))
    for form, names in bench.STEP_KERNELS.items():
        for n in names:
            assert ("void %s(" % n) in src or ("%s(" % n) in src, n
    row = bench.load_traffic(10000, 200, 2048, 1)
    assert row is not None and row["select_form"] == 2 and row.get("commit")
    bench.check_traffic_row(row, 2)                       # accepted
    with pytest.raises(SystemExit):
        bench.check_traffic_row(row, 0)                   # the five-kernel form was not what the row measured
    renamed = dict(row, select_form=None, kernels={k.replace("pairs_query_kernel", "pairs_kernel_v2"): v for k, v in row["kernels"].items()})
    with pytest.raises(SystemExit):
        bench.check_traffic_row(renamed, 2)               # a kernel of the step is missing from the row
    # every stage the row's kernels fall into is one of sgtd_stats' per-kernel times
    assert {bench.stage_of(k) for k in row["kernels"] if not k.startswith(("__amd", "at::"))} <= set(bench.KERNEL_KEYS)
    total = sum((v.get("read_bytes") or 0) + (v.get("write_bytes") or 0) for v in row["kernels"].values())
    assert abs(total - row["step"]["bytes"]) < 1e-6 * total


def test_step_summariser_cuts_steps_out_of_a_dispatch_sequence(tmp_path):
    """profiles/summarize_step.py: a step runs from the first build_frames_kernel after the previous step's last kernel to
    pairs_query_kernel / block_write_kernel; a kernel's launches inside a step are summed, the map's own build launches
    and a first-batch re-run in front do not count (the median over the last steps is taken)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("summarize_step", os.path.join(ROOT, "profiles", "summarize_step.py"))
    ss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ss)
    seq = ["build_frames_kernel<true, 10>"] * 3 + ["slice_assign_kernel"]                      # map construction: no step closes
    one = ["build_frames_kernel<true, 10>", "radix_scatter_kernel<unsigned int>", "radix_scatter_kernel<unsigned int>",
           "probe_sorted_kernel<false, false, false>", "votes_topk_kernel", "pairs_query_kernel<true>"]
    rows = [dict(name=n, v=1.0) for n in seq]
    for k in range(6):
        rows += [dict(name=n, v=float(10 * (k + 1) if n.startswith("probe") else 2.0)) for n in one]
    steps = ss.steps_of(rows, lambda r: r["v"])
    assert len(steps) == 6 and len(steps[0]["build_frames_kernel<true, 10>"]) == 4       # (the map's launches fall into the first, discarded, step)
    pk = ss.per_kernel(steps)
    assert pk["radix_scatter_kernel<unsigned int>"] == {"launches_per_step": 2, "per_step": 4.0}
    assert pk["probe_sorted_kernel<false, false, false>"]["per_step"] == 50.0            # median of the last four steps: 30, 40, 50, 60
    assert pk["build_frames_kernel<true, 10>"]["launches_per_step"] == 1


def test_plan_2d_prefers_query_groups_and_respects_the_envelope():
    """sgtd_amd/dist.py::plan_2d: the smallest number of table shards whose shard fits one GPU — everything a rank does
    per query is repeated on every rank of a table group, so the other factor of N goes to query groups"""
    from sgtd_amd.dist import plan_2d, shard_range
    assert plan_2d(8, 10000, 2048) == (1, 8)           # the north-star map fits one GPU: eight replicas' worth of query groups
    assert plan_2d(1, 10000, 2048) == (1, 1)
    assert plan_2d(8, 100000, 256) == (4, 2)           # cfg4: a 25 000-frame shard fits the LDS vote histogram, 50 000 do not
    assert plan_2d(2, 100000, 256) == (2, 1)           # nothing fits: as many shards as there are ranks
    assert plan_2d(8, 30000, 2048)[0] == 2             # the batch's match records under the 32-bit index
    assert plan_2d(8, 10000, 2048, r_t=8) == (8, 1) and plan_2d(4, 400, 24, r_t=2) == (2, 2)
    for world in (1, 2, 3, 8):
        cover = [shard_range(1001, world, r) for r in range(world)]
        assert cover[0][0] == 0 and cover[-1][1] == 1001 and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))


def test_bench_relays_exactly_one_result_line():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    noisy = '[Gloo] Rank 0 is connected to 1 peer ranks\n{"not": "a result"}\n{"metric": "m", "value": 1}\ntrailing chatter\n'
    assert b.last_result_line(noisy) == '{"metric": "m", "value": 1}'
    assert b.last_result_line("no json here\n") is None
    assert b.modelled_all_gather_ms(1 << 20, 1) == 0.0 and b.modelled_all_gather_ms(1 << 20, 8) > b.modelled_all_gather_ms(1 << 20, 2) > 0


def test_bench_line_repeats_its_evidence_as_scalar_keys():
    """the driver's record of a bench line keeps scalars (and the scalar members of `config` / `roofline`) only: what the nested
    objects say — delivered rate, one batch in flight, overflowed launches, parity count, verification time, skewed workload,
    the spread of the repeated timed region, for N > 1 the table-sharded and cfg4 figures, the sweep's useful fraction of the
    HBM roof — must also be there as scalar keys (bench.flatten_evidence)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    line = {"n_gpus": 8, "config": {"table_shards_R_t": 4, "query_groups_R_q": 2},
            "roofline": {"peak": 8000.0, "kernel_ms": {"ms_probe": 5.0},
                         "compulsory": {"probe_layout_once": 711000000, "match_records_once": 6090000000, "frac": 0.107}},
            "timed_region": {"launches_that_overflowed_a_work_buffer": 0, "region_ms_per_step_median": 11.0, "region_ms_per_step_min": 10.9,
                             "region_ms_per_step_max": 11.2, "regions_timed": 5},
            "delivered": {"frames_per_s": 184600.0}, "one_batch_in_flight": {"ms_per_step": 11.87},
            "parity": {"identical_candidates_votes_matchlists": 200}, "verify": {"ms_per_batch": 9.5},
            "workload_skew": {"frames_per_s": 18800.0}, "boundary": {"cpp_adapter_ms_per_frame": 6.5},
            "map_size_sweep": {"100_cfg1_json_in": {"identical_candidates_votes_matchlists": "200/200"}},
            "table_sharded": {"value": 60000.0}, "merged_list_equals_single_table": True, "cfg4": {"value": 150000.0},
            "scaling_parts": {"exchange_exposed_ms": 0.05}}
    out = bench.flatten_evidence(line)
    for k in bench.FLAT_CONFIG_KEYS + bench.FLAT_CONFIG_KEYS_MULTI:
        assert k in out["config"] and out["config"][k] is not None and not isinstance(out["config"][k], (dict, list)), k
    for k in bench.FLAT_ROOFLINE_KEYS:
        assert isinstance(out["roofline"][k], float), k
    assert abs(out["roofline"]["useful_frac"] - (0.711 + 6.09) / 5.0e-3 / 8000.0) < 1e-9 and out["config"]["cfg1_identical_of_200"] == 200
    # a line without the optional legs still gets the keys (null), and never fails
    bare = bench.flatten_evidence({"n_gpus": 1, "config": {}, "roofline": {"peak": 8000.0}})
    assert all(k in bare["config"] for k in bench.FLAT_CONFIG_KEYS) and bare["roofline"]["useful_frac"] is None
