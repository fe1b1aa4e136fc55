"""Multi-GPU sharding of the SPR search round -- the counterpart of the reference's only parallel
strategy (MAPLEv0.7.5.4.py:12164-12195 assignCoreNumbers, 12283-12316 Pool.map + concat + sort).

One process per GPU (``torch.distributed``; backend ``nccl`` = RCCL over xGMI on the GPU box, ``gloo`` in the
CPU tests).  The tree mirror is replicated on every GPU; pruned nodes are dealt round-robin by pre-order
index exactly like ``coreNum``; the only exchange is one all-gather of fixed-size proposal records
``(improvement f64, node, placement)`` per (sub)round, after which every rank sorts them ascending by
improvement as the reference does before applying moves best-first (M:12312, 9476).
"""
from __future__ import annotations

import numpy as np


def shard_nodes(tree, rank: int, world: int, dirty=None, replacements=None, max_replacements=10):
    """Nodes this rank searches: dirty, not over-replaced, and ``coreNum[node] == rank`` (M:9619)."""
    core = tree.assign_core_numbers(world)
    out = []
    for v in tree.preorder():
        if dirty is not None and not dirty[v]:
            continue
        if replacements is not None and replacements[v] > max_replacements:
            continue
        if core[v] == rank:
            out.append(v)
    return out


def pack_proposals(nodes, placement, improvement):
    """Local search results -> float64 records [improvement, node, placement] of the proposed moves."""
    nodes = np.asarray(nodes)
    placement = np.asarray(placement)
    improvement = np.asarray(improvement, dtype=np.float64)
    keep = placement >= 0
    rec = np.zeros((int(keep.sum()), 3), dtype=np.float64)
    rec[:, 0] = improvement[keep]
    rec[:, 1] = nodes[keep]
    rec[:, 2] = placement[keep]
    return rec


def gather_proposals(local_records, device=None, cap=None):
    """All-gather the proposal records of every rank and return them sorted ascending by improvement
    (ties keep rank order, then local order -- the order of the reference's list concatenation).

    ONE collective per (sub)round: every rank sends a block of ``cap + 1`` fixed-size records -- a header row that
    holds its count, then its records, zero-padded.  ``cap`` must be the same on every rank and at least every rank's count:
    the callers pass the size of the largest shard of searched nodes (a search proposes at most one move), which every rank
    knows without asking, since the node list is the same everywhere.  Without ``cap`` the counts are exchanged first (a second,
    8-byte collective)."""
    import torch
    import torch.distributed as dist
    rec = np.ascontiguousarray(local_records, dtype=np.float64).reshape(-1, 3)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        a = rec
    else:
        world = dist.get_world_size()
        if cap is None:
            cnt = torch.tensor([rec.shape[0]], dtype=torch.int64)
            cnt = cnt.to(device) if device is not None else cnt
            dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
            cap = int(cnt.item())
        if rec.shape[0] > cap:
            raise ValueError(f"gather_proposals: {rec.shape[0]} records but cap {cap}")
        block = np.zeros((cap + 1, 3), dtype=np.float64)
        block[0, 0] = rec.shape[0]
        block[1: 1 + rec.shape[0]] = rec
        mine = torch.from_numpy(block)
        if device is not None:
            mine = mine.to(device)
        out = torch.empty((world * (cap + 1), 3), dtype=torch.float64, device=mine.device)
        dist.all_gather_into_tensor(out, mine)                      # the round's ONE collective: fixed-size records
        out = out.cpu().numpy().reshape(world, cap + 1, 3)
        a = np.concatenate([out[r, 1: 1 + int(out[r, 0, 0])] for r in range(world)], axis=0)
    if len(a) == 0:
        return []
    order = np.argsort(a[:, 0], kind="stable")
    a = a[order]
    return [(int(r[1]), int(r[2]), float(r[0])) for r in a]


# ---- level 2: ONE query, its candidate branches sharded over the GPUs ---------------------------
def shard_candidates(n_candidates: int, rank: int, world: int):
    """Indices (into the candidate list) this rank scores: interleaved, so every rank gets the same mix of list lengths."""
    return np.arange(rank, n_candidates, world)


def allgather_interleaved(local, n_total: int, device=None):
    """Inverse of shard_candidates for a per-candidate vector (scores f64 / flags u8): one all-gather of equal-size
    (padded) shards, re-interleaved so that element k is candidate k on every rank.  The depth-first replay of
    findBestParentForNewSample needs every score on its path (stop rules, M:8080-8093), not only the maximum, which is
    why the exchange is an all-gather and not a reduction."""
    import torch
    import torch.distributed as dist
    local = np.ascontiguousarray(local)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        assert len(local) == n_total
        return local
    world = dist.get_world_size()
    m = (n_total + world - 1) // world
    pad = torch.zeros(m, dtype=torch.from_numpy(local[:0]).dtype)
    pad[: len(local)] = torch.from_numpy(local)
    if device is not None:
        pad = pad.to(device)
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = np.zeros(n_total, dtype=local.dtype)
    for r, b in enumerate(bufs):
        cnt = len(range(r, n_total, world))
        out[r::world] = b[:cnt].cpu().numpy()
    return out


def argmax_allreduce(score: float, visit_index: int, device=None):
    """Best placement over the ranks' shards without moving the scores: max score, ties resolved to the SMALLEST visit
    index (the earliest depth-first visit wins under the reference's strict ``>``, M:7083, 8065).  Two all-reduces of
    one element each (16 bytes over xGMI); returns (score, visit_index) identical on every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(score), int(visit_index)
    s = torch.tensor([score], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.MAX)
    best = float(s.item())
    big = np.iinfo(np.int64).max
    i = torch.tensor([visit_index if score == best else big], dtype=torch.int64, device=device)
    dist.all_reduce(i, op=dist.ReduceOp.MIN)
    return best, int(i.item())


def argmax_allreduce_native(dev, score_tensor, idx_tensor, stream=0):
    """The same reduction for whole vectors that already sit in HBM, through the library's RCCL entry point
    (maple_argmax_allreduce_dev: two ncclAllReduce of 8-byte words, in place, asynchronous on `stream`).  The
    communicator is created on first use: rank 0's unique id is broadcast with torch.distributed."""
    import torch
    import torch.distributed as dist
    if not getattr(dev, "_comm_ready", False):
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        uid = torch.from_numpy(dev.comm_unique_id() if rank == 0 else np.zeros(128, dtype=np.uint8))
        if world > 1:
            holder = uid.to(score_tensor.device) if dist.get_backend() == "nccl" else uid
            dist.broadcast(holder, src=0)
            uid = holder.cpu()
        dev.comm_init(world, rank, uid.numpy())
        dev._comm_ready = True
    dev.argmax_allreduce_dev(score_tensor.numel(), score_tensor.data_ptr(), idx_tensor.data_ptr(), stream)


def sharded_spr_round(dev, nodes, search_kwargs, rank: int = 0, world: int = 1, device=None):
    """One SPR search (sub)round over several GPUs -- what the reference does with Pool.map over its cores
    (M:12283-12316): ``nodes`` (the dirty nodes in pre-order, identical on every rank) are dealt round-robin like
    coreNum, this rank searches its share on its GPU (maple_spr_search_batch), and the proposed moves of all ranks are
    combined with ONE all-gather and sorted by improvement.  Returns (moves, local_result) with moves =
    [(node, placement, improvement), ...] identical on every rank."""
    nodes = np.asarray(nodes)
    mine = nodes[rank::world]               # `nodes` in pre-order: this is coreNum[node] == rank (shard_nodes)
    res = dev.spr_search_batch(mine, **search_kwargs)
    bad = res["status"][res["status"] < -1]
    if len(bad):
        raise RuntimeError(f"SPR search could not finish some queries (status {sorted(set(bad.tolist()))})")
    rec = pack_proposals(mine, res["placement"], res["improvement"])
    return gather_proposals(rec, device=device, cap=len(nodes[0::world])), res
