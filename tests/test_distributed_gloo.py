"""N>1 path on CPU: two gloo ranks shard the pruned nodes like assignCoreNumbers and combine their
proposed moves with one all-gather; every rank must end with the same, improvement-sorted list."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maple_amd.parallel import gather_proposals, pack_proposals
    # synthetic per-rank search output: node ids dealt round-robin, some with no proposal
    rng = np.random.default_rng(7)
    n = 101
    placement_all = rng.integers(-1, 50, size=n)
    improvement_all = np.round(rng.random(n) * 5, 2)          # rounded -> ties exist
    mine = np.arange(n)[rank::world]
    rec = pack_proposals(mine, placement_all[mine], improvement_all[mine])
    moves = gather_proposals(rec)
    # ... and with the fixed-size blocks of a round (cap = the largest shard of searched nodes: ONE collective, no count exchange)
    moves_cap = gather_proposals(rec, cap=len(np.arange(n)[0::world]))
    assert moves_cap == moves
    empty = gather_proposals(np.zeros((0, 3)), cap=4)                     # (nobody proposes a move)
    assert empty == []
    q.put((rank, moves))
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_process():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]
    # reference semantics: concatenate per-worker lists (worker 0 first), then stable sort by improvement
    rng = np.random.default_rng(7)
    n = 101
    placement_all = rng.integers(-1, 50, size=n)
    improvement_all = np.round(rng.random(n) * 5, 2)
    concat = []
    for r in range(world):
        for v in np.arange(n)[r::world]:
            if placement_all[v] >= 0:
                concat.append((int(v), int(placement_all[v]), float(improvement_all[v])))
    concat.sort(key=lambda m: m[2])
    assert res[0] == concat
    assert all(a[2] <= b[2] for a, b in zip(concat, concat[1:]))


def _worker_level2(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maple_amd.parallel import allgather_interleaved, argmax_allreduce, shard_candidates
    rng = np.random.default_rng(3)
    n = 1001                                                    # not a multiple of the world size
    scores = np.round(rng.normal(size=n) * 3, 1)                # rounded -> exact ties exist
    scores[[17, 400, 923]] = scores.max() + 1.0                 # a three-way tie for the best
    flags = rng.integers(0, 3, size=n).astype(np.uint8)
    visit = rng.permutation(n)                                  # depth-first visit index of each candidate
    mine = shard_candidates(n, rank, world)
    full = allgather_interleaved(scores[mine], n)
    full_flags = allgather_interleaved(flags[mine], n)
    k = mine[np.lexsort((visit[mine], -scores[mine]))[0]]       # local best, earliest visit among local ties
    best = argmax_allreduce(float(scores[k]), int(visit[k]))
    q.put((rank, full.tolist(), full_flags.tolist(), best))
    dist.destroy_process_group()


def test_two_rank_candidate_sharding_level2():
    """One query, candidates sharded over the ranks: the all-gathered score vector equals the unsharded one on every
    rank, and the two-step arg-max all-reduce picks the best score with the earliest visit among exact ties."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_level2, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(3)
    n = 1001
    scores = np.round(rng.normal(size=n) * 3, 1)
    scores[[17, 400, 923]] = scores.max() + 1.0
    flags = rng.integers(0, 3, size=n).astype(np.uint8)
    visit = rng.permutation(n)
    want_k = min((17, 400, 923), key=lambda i: visit[i])
    for rank, full, full_flags, best in res:
        assert full == scores.tolist()
        assert full_flags == flags.tolist()
        assert best == (float(scores[want_k]), int(visit[want_k]))


class _StubDevice:
    """Stands in for runtime.Device in argmax_allreduce_native: the two ncclAllReduce of maple_argmax_allreduce_dev
    (maple_hip.hip: max of order-preserving 64-bit keys, then min of the offered indices among the ranks that hold the
    maximum) done over gloo on host memory, with the kernels' own key packing."""

    def __init__(self):
        self.inits = []

    def comm_unique_id(self):
        return np.arange(128, dtype=np.uint8)

    def comm_init(self, world, rank, uid):
        self.inits.append((world, rank, bytes(uid.tolist())))

    def argmax_allreduce_dev(self, n, score_ptr, idx_ptr, stream=0):
        import ctypes
        score = np.ctypeslib.as_array((ctypes.c_double * n).from_address(score_ptr))
        idx = np.ctypeslib.as_array((ctypes.c_int32 * n).from_address(idx_ptr))
        bits = score.view(np.uint64)
        key = np.where(bits >> np.uint64(63), ~bits, bits | np.uint64(1 << 63))        # k_argmax_pack: order-preserving
        k = torch.from_numpy(key.astype(np.int64) ^ np.int64(-(1 << 63)))              # (as a signed tensor with the same order)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        offer = torch.from_numpy(np.where((key.astype(np.int64) ^ np.int64(-(1 << 63))) == k.numpy(), idx.astype(np.int64),
                                          np.iinfo(np.int64).max))                      # k_argmax_offer
        dist.all_reduce(offer, op=dist.ReduceOp.MIN)
        best = (k.numpy() ^ np.int64(-(1 << 63))).astype(np.uint64)                      # k_argmax_unpack
        out_bits = np.where(best >> np.uint64(63), best & ~np.uint64(1 << 63), ~best)
        score[:] = out_bits.view(np.float64)
        idx[:] = offer.numpy().astype(np.int32)


def _argmax_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maple_amd.parallel import argmax_allreduce_native
    # 6 queries; each rank holds its shard's best score and the depth-first visit index of the branch that has it
    score = np.array([[-3.0, -1.0, -2.0, float("-inf"), -5.5, -0.0],
                      [-3.0, -4.0, -2.0, float("-inf"), -5.25, 0.0]])[rank].copy()
    idx = np.array([[40, 7, 9, 3, 11, 5], [12, 8, 90, 1, 2, 6]], dtype=np.int32)[rank].copy()
    dev = _StubDevice()
    argmax_allreduce_native(dev, torch.from_numpy(score), torch.from_numpy(idx))
    q.put((rank, score.tolist(), idx.tolist(), dev.inits))
    dist.destroy_process_group()


def test_argmax_allreduce_control_flow_and_tie_break_two_ranks():
    """Level 2 of the multi-GPU design through parallel.argmax_allreduce_native's own control flow (unique id from rank 0,
    broadcast, comm_init on every rank, one in-place reduction) with the RCCL calls replaced by gloo: the maximum score per
    query, exact ties to the SMALLEST visit index (the reference's strict >, M:7083 / 8065), -inf and -0.0 / 0.0 ordered as
    the 64-bit keys order them."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_argmax_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_score = [-3.0, -1.0, -2.0, float("-inf"), -5.25, 0.0]
    want_idx = [12, 7, 9, 1, 2, 6]
    for rank, score, idx, inits in got:
        assert score == want_score and idx == want_idx, (rank, score, idx)
        assert len(inits) == 1 and inits[0][0] == world and inits[0][1] == rank
        assert inits[0][2] == bytes(range(128))                     # rank 0's unique id reached every rank
