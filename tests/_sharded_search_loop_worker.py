"""One rank of tests/test_dist_gpu.py::test_sharded_search_loop_over_the_collective_backend (started under
torch.distributed.run before anything touches a GPU): ShardedMap.search_loop — shard sweep, all_gather + merge of
the top-50 (STDesc.cpp:423-433), candidate_verify on the owner, all_gather of (frame, score, pose), SearchLoop's
choice (STDesc.cpp:105-146) — against a single table held by rank 0.  RCCL when every rank has a GPU of its own,
gloo with all ranks on cuda:0 otherwise (SGTD_TEST_BACKEND)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from sgtd_amd import synth
    from sgtd_amd.dist import Map2D
    from sgtd_amd.manager import STDescManager
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("SGTD_TEST_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    F, N, Q = 240, 150, 12
    smap = synth.make_map(F, N, stream=171)
    qs = synth.make_queries(smap, Q, stream=171)
    # SGTD_TEST_RT table shards per query group (default: all ranks one group), SGTD_TEST_LISTS all | winners; every group is
    # handed the same queries here, so all ranks must end with the same result
    sm = Map2D(F, rank, world, r_t=int(os.environ.get("SGTD_TEST_RT", world)), device_id=local, lists=os.environ.get("SGTD_TEST_LISTS", "all"))
    sm.add_shard_frames(smap.xyz[sm.lo:sm.hi], smap.label[sm.lo:sm.hi])
    if os.environ.get("SGTD_TEST_ATTACHED") == "1":
        # a second map of the grid on this rank, borrowing the first one's table: it has a stream of ITS OWN, so the step's
        # collectives and torch ops must be ordered against that stream, not against the caller's current one
        owner = sm
        owner.mgr.query_frames(qs.xyz, qs.label)            # (the owner has a batch of its own in flight meanwhile)
        sm = Map2D(F, rank, world, r_t=owner.r_t, device_id=local, lists=owner.lists, attach_to=owner)
        assert sm.main.cuda_stream != torch.cuda.current_stream(dev).cuda_stream
    frames, votes, n_cand, scores, poses, bc, bf, bs = sm.search_loop(qs.xyz, qs.label)
    torch.cuda.synchronize()
    if os.environ.get("SGTD_TEST_EXPECT_REPAIR") == "1":
        assert sm.exchanges >= 2 and sm.mgr.stats()["reruns_total"] >= 1, "the batch was expected to overflow and be repaired (%d exchanges)" % sm.exchanges
    # every rank holds the same merged result
    probe = torch.cat([frames.double().flatten(), votes.double().flatten(), scores.flatten(), poses.flatten(), bf.double(), bs]).contiguous()
    ref = probe.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(ref, probe), "rank %d holds another merged result than rank 0" % rank
    ok = 1
    if rank == 0:
        single = None
        try:
            single = STDescManager(device_id=local)
            single.add_frames(smap.xyz, smap.label)
            want = single.query_frames(qs.xyz, qs.label)
            single.verify()
            w_bc, w_bf, w_bs = single.search_loop()
            cn = single.config_setting_["candidate_num"]
            for i in range(Q):
                nc = int(want.n_cand[i])
                assert int(n_cand[i]) == nc
                assert np.array_equal(frames[i, :nc].cpu().numpy(), want.cand_frame[i, :nc])
                assert np.array_equal(votes[i, :nc].cpu().numpy(), want.cand_votes[i, :nc])
                w_score, w_rot, w_t = single.result_verify(i)
                assert np.array_equal(scores[i].cpu().numpy(), w_score)
                got = poses[i].cpu().numpy()
                assert np.array_equal(got[:, :9].reshape(cn, 3, 3), w_rot) and np.array_equal(got[:, 9:], w_t)
            assert np.array_equal(bc.cpu().numpy(), w_bc) and np.array_equal(bf.cpu().numpy(), w_bf) and np.array_equal(bs.cpu().numpy(), w_bs)
            assert int((w_bf >= 0).sum()) >= Q // 2          # loops are found
        except Exception as exc:       # (anything: the broadcast and the barrier below must be reached)
            import traceback
            ok = 0
            print("MISMATCH", exc)
            traceback.print_exc()
        if single is not None:
            single.close()
    if os.environ.get("SGTD_FORCE_COLLECTIVE") == "1" and backend == "nccl":
        # what one call of the step's all-gather costs inside RCCL when nothing has to travel (a group of one): the packed
        # table of a 2048-query batch, event-timed on the side stream
        ints = 2 * 2048 * sm.cand_num + 4
        a, b = torch.zeros(ints, dtype=torch.int32, device=dev), torch.empty(world * ints, dtype=torch.int32, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        with torch.cuda.stream(sm.side):
            for _ in range(5):
                dist.all_gather_into_tensor(b, a)
            ev[0].record()
            for _ in range(50):
                dist.all_gather_into_tensor(b, a)
            ev[1].record()
        sm.side.synchronize()
        print("rccl all_gather_into_tensor, group of %d, %d KB: %.1f us per call on the stream" % (world, ints * 4 // 1024, ev[0].elapsed_time(ev[1]) * 1000 / 50))
    flag = torch.tensor([ok], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
    dist.broadcast(flag, src=0)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("sharded search_loop ok: %d ranks over %s, %d loops" % (world, backend, int((bf >= 0).sum())) if int(flag.item()) else "sharded search_loop FAILED")
    sys.exit(0 if int(flag.item()) else 1)


if __name__ == "__main__":
    main()
