"""Two real engine instances in two processes: bench.py --gpus 2 starts its
own ranks; on a box with two GPUs they use one GPU each and RCCL, on a one-GPU box both ranks
share cuda:0 and the collectives run over gloo (the SGTD_BENCH_* switches exist for exactly
this).  Checks: the ranks really formed one group of 2, the table-sharded result (frame-range
shards, all_gather + merge) equals the replicated-map result query by query, and both equal the
single-process result."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--frames", "400", "--queries", "24", "--steps", "2", "--warmup", "1", "--cpu-baseline", "off",
        "--sweep", "", "--verify", "off", "--boundary", "off"]


def _bench(extra, env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS + extra, capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_two_rank_bench_shards_and_replicas_agree():
    import torch
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if torch.cuda.device_count() < 2:
        env.update(SGTD_BENCH_BACKEND="gloo", SGTD_BENCH_SHARE_GPU="1")
    two = _bench(["--gpus", "2"], env)
    # the headline for N > 1 is the form the north_star names: the map's table sharded by frame range,
    # the same Q queries on every rank, all_gather + merge of the per-shard top-50 (strong scaling)
    cfg = two["config"]
    assert two["n_gpus"] == 2 and cfg["ranks_in_collective"] == 2 and cfg["mode"] == "table"
    assert two["scaling"] == "strong" and cfg["queries_per_step"] == 24
    assert len(cfg["table_entries_per_rank"]) == 2
    assert cfg["collective_backend"] == ("gloo" if "SGTD_BENCH_BACKEND" in env else "nccl")
    assert ("NOT RCCL" in cfg["sharding"]) == (cfg["collective_backend"] == "gloo")      # the line says what really ran
    assert two["merged_list_equals_single_table"] is True
    # what a rank's step consists of and what that predicts (the replicated part bounds the strong-scaling curve)
    sp = two["scaling_parts"]
    for key in ("replicated_build_sort_plan", "sharded_sweep_and_record_passes", "exchange_all_gather_merge"):
        assert len(sp["per_rank_ms"][key]) == 2 and min(sp["per_rank_ms"][key]) > 0
    assert sp["predicted_speedup_over_one_gpu"] > 0 and sp["speedup_ceiling_if_the_sharded_part_vanished"] >= sp["predicted_speedup_over_one_gpu"]
    tr = two["timed_region"]
    assert tr["batch_launches_in_timed_region"] >= 2 and tr["launches_that_overflowed_a_work_buffer"] == tr["reruns_in_timed_region"] + tr["list_pass_reruns_in_timed_region"]
    rp = two["replicated"]        # beside it: every rank a full replica, 24 queries each (weak scaling)
    assert rp["ranks_in_collective"] == 2 and rp["scaling"] == "weak" and rp["queries_per_step"] == 48
    assert rp["equals_table_sharded_list"] is True
    assert sum(cfg["table_entries_per_rank"]) == rp["table_entries_per_rank"][0]   # shards add up to the replica
    one = _bench(["--gpus", "1", "--queries", "48"], dict(os.environ))
    assert one["n_gpus"] == 1 and one["recall"] == rp["recall"]      # same 48 queries, same answers
    qm = _bench(["--gpus", "2", "--shard", "query"], env)
    assert qm["scaling"] == "weak" and qm["config"]["queries_per_step"] == 48 and qm["recall"] == one["recall"]
    assert qm["table_sharded"]["merged_list_equals_single_table"] is True and qm["merged_list_equals_single_table"] is True
    if torch.cuda.device_count() >= 2:     # one process over both devices (sgtd_create_multi)
        mh = two["multi_device_handle"]
        assert mh["devices"] == 2 and mh["candidates_equal_headline_list"] is True


@pytest.mark.gpu
def test_sharded_search_loop_over_the_collective_backend():
    """ShardedMap.search_loop in two processes: over RCCL, one GPU per rank, whenever two devices are visible; on a
    one-GPU box both ranks share cuda:0 and the same collectives run over gloo.  Ranks are child processes of
    torch.distributed.run, started before anything here touches a GPU."""
    import socket
    n_dev = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True).stdout.strip() or 0)
    backend = "nccl" if n_dev >= 2 else "gloo"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SGTD_TEST_BACKEND=backend)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "tests", "_sharded_search_loop_worker.py")],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "sharded search_loop ok: 2 ranks over %s" % backend in out.stdout
