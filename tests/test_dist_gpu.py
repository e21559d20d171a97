"""Real engine instances in several processes: bench.py --gpus N starts its own ranks; on a box with N GPUs they use
one GPU each and RCCL, on a one-GPU box all ranks share cuda:0 and the collectives run over gloo (the SGTD_BENCH_*
switches exist for exactly this).  Every form of the R_t x R_q grid (sgtd_amd/dist.py::Map2D) must give the lists of a
single table: replicas with sharded queries (R_t = 1), the pure table-sharded form (R_t = N: all_gather of the packed
local top-50 tables + the merge kernel on a side stream), a 2 x 2 grid on four ranks — with the match lists of every
local candidate written beside the exchange (lists "all") and with the winners' lists only (lists "winners")."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--frames", "400", "--queries", "24", "--steps", "2", "--warmup", "1", "--cpu-baseline", "off",
        "--sweep", "none", "--verify", "off", "--boundary", "off"]


def _bench(extra, env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS + extra, capture_output=True, text=True,
                         timeout=1200, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, "stdout must carry exactly one line, got %d: %r" % (len(lines), [l[:80] for l in lines])
    return json.loads(lines[0])


def _env(n):
    import torch
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if torch.cuda.device_count() < n:
        env.update(SGTD_BENCH_BACKEND="gloo", SGTD_BENCH_SHARE_GPU="1")
    return env


def _check_parts(sp, n, r_t):
    assert sp["table_shards_R_t"] == r_t and sp["table_shards_R_t"] * sp["query_groups_R_q"] == n
    for key in ("replicated_build_sort_plan", "sharded_sweep_and_record_passes", "exchange_alone_all_gather_merge_result_gather",
                "step_with_exchange", "step_without_exchange"):
        assert len(sp["per_rank_ms"][key]) == n and min(sp["per_rank_ms"][key]) > 0
    assert sp["predicted_speedup_over_one_gpu"] > 0 and sp["exchange_exposed_ms"] >= 0


@pytest.mark.gpu
def test_two_rank_bench_replicas_and_table_shards_agree():
    env = _env(2)
    two = _bench(["--gpus", "2"], env)
    # a 400-frame map fits one GPU: dist.plan_2d gives one table shard, two query groups — 24 query frames each per step
    cfg = two["config"]
    assert two["n_gpus"] == 2 and cfg["ranks_in_collective"] == 2 and cfg["mode"] == "query"
    assert cfg["table_shards_R_t"] == 1 and cfg["query_groups_R_q"] == 2
    assert two["scaling"] == "weak" and cfg["queries_per_step"] == 48 and cfg["queries_per_query_group"] == 24
    assert len(cfg["table_entries_per_rank"]) == 2 and cfg["table_entries_per_rank"][0] == cfg["table_entries_per_rank"][1]
    assert cfg["collective_backend"] == ("gloo" if "SGTD_BENCH_BACKEND" in env else "nccl")
    assert ("NOT RCCL" in cfg["sharding"]) == (cfg["collective_backend"] == "gloo")      # the line says what really ran
    _check_parts(two["scaling_parts"], 2, 1)
    tr = cfg["timed_region"]
    assert tr["batch_launches_in_timed_region"] >= 2 and tr["launches_that_overflowed_a_work_buffer"] == 0
    # beside it: north_star's form — the table sharded over both ranks, the same 24 queries on both, merged lists == a replica's
    ts = two["table_sharded"]
    assert ts["ranks_in_collective"] == 2 and ts["scaling"] == "strong" and ts["queries_per_step"] == 24
    assert ts["merged_list_equals_headline_list"] is True and two["merged_list_equals_single_table"] is True
    assert sum(ts["table_entries_per_rank"]) == cfg["table_entries_per_rank"][0]          # shards add up to the replica
    _check_parts(ts["scaling_parts"], 2, 2)
    assert two["fixed_total_batch"]["queries_per_step_total"] == 24
    one = _bench(["--gpus", "1", "--queries", "48"], dict(os.environ))
    assert one["n_gpus"] == 1 and one["recall"] == two["recall"]      # same 48 queries, same answers
    assert one["config"]["delivered"]["frames_per_s"] > 0 and one["config"]["timed_region"]["launches_that_overflowed_a_work_buffer"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("lists", ["all", "winners"])
def test_two_rank_table_sharded_headline(lists):
    env = _env(2)
    tb = _bench(["--gpus", "2", "--shard", "table", "--lists", lists], env)
    cfg = tb["config"]
    assert cfg["mode"] == "table" and cfg["table_shards_R_t"] == 2 and cfg["lists"] == lists
    assert tb["scaling"] == "strong" and cfg["queries_per_step"] == 24
    assert tb["merged_list_equals_single_table"] is True         # against a replica of the whole table
    _check_parts(tb["scaling_parts"], 2, 2)
    if cfg["collective_backend"] == "gloo" and lists == "all":
        # gloo's all-gather blocks the host; the same step with it issued asynchronously and merged a step late is measured too
        assert tb["scaling_parts"]["exchange_exposed_ms_host_not_blocked"] is not None
    assert tb["recall"]["top1_pose_within_5m"] > 0.9


@pytest.mark.gpu
def test_four_rank_two_by_two_grid():
    env = _env(4)
    g = _bench(["--gpus", "4", "--rt", "2", "--lists", "winners"], env)
    cfg = g["config"]
    assert cfg["mode"] == "2d" and cfg["table_shards_R_t"] == 2 and cfg["query_groups_R_q"] == 2 and cfg["ranks_in_collective"] == 4
    assert cfg["queries_per_step"] == 48 and g["scaling"] == "weak"
    _check_parts(g["scaling_parts"], 4, 2)
    # the four-shard form of the same map gives the lists of query group 0 (two shards): both are the single table's
    assert g["table_sharded"]["merged_list_equals_headline_list"] is True
    assert g["recall"]["top1_pose_within_5m"] > 0.9


def _search_loop(n, r_t, lists, extra_env=None, backend=None):
    import socket
    n_dev = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True).stdout.strip() or 0)
    backend = backend or ("nccl" if n_dev >= n else "gloo")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SGTD_TEST_BACKEND=backend, SGTD_TEST_RT=str(r_t), SGTD_TEST_LISTS=lists, **(extra_env or {}))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "tests", "_sharded_search_loop_worker.py")],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "sharded search_loop ok: %d ranks over %s" % (n, backend) in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("lists", ["all", "winners"])
def test_sharded_search_loop_over_the_collective_backend(lists):
    """Map2D.search_loop in two processes: over RCCL, one GPU per rank, whenever two devices are visible; on a
    one-GPU box both ranks share cuda:0 and the same collectives run over gloo.  Ranks are child processes of
    torch.distributed.run, started before anything here touches a GPU."""
    _search_loop(2, 2, lists)


@pytest.mark.gpu
@pytest.mark.parametrize("lists", ["all", "winners"])
def test_search_loop_on_a_map_attached_to_another(lists):
    """the same through a second Map2D of the rank that borrows the first one's table (attach_to): its engine runs on a stream
    of its own, and verification, export copies, the all-gather of the results and the choice must all be ordered against
    THAT stream (they once ran on the caller's current stream: the all-gather could start before the scores were written)"""
    _search_loop(2, 2, lists, {"SGTD_TEST_ATTACHED": "1"})


@pytest.mark.gpu
def test_search_loop_on_a_two_by_two_grid():
    _search_loop(4, 2, "winners")


@pytest.mark.gpu
@pytest.mark.parametrize("lists", ["all", "winners"])
def test_a_shard_whose_batch_outgrows_its_buffers_is_repaired_by_the_whole_group(lists):
    """every engine starts with a 4096-record buffer: each rank's first batch overflows, the flag travels in the packed
    tables, every rank of the group learns it from the same all-gather, re-runs and exchanges again (Map2D.query) —
    and the result is still the single table's"""
    _search_loop(2, 2, lists, {"SGTD_REC_CAP": "4096", "SGTD_TEST_EXPECT_REPAIR": "1"})


@pytest.mark.gpu
@pytest.mark.parametrize("lists", ["all", "winners"])
def test_rccl_calls_on_hardware_with_a_group_of_one(lists):
    """What a one-GPU box can put through RCCL itself: ONE rank, backend nccl, every collective of the step issued all the
    same (SGTD_FORCE_COLLECTIVE: the all-gather of a single table is a copy inside RCCL) — the communicator, the dtypes
    and shapes handed to all_gather_into_tensor, its stream ordering against the engine's export events on the side
    stream, the merge kernel and the verification gather behind it; the result must be the single table's."""
    _search_loop(1, 1, lists, {"SGTD_FORCE_COLLECTIVE": "1"}, backend="nccl")
