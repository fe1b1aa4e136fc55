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


def gather_proposals(local_records, device=None):
    """All-gather the proposal records of every rank and return them sorted ascending by improvement
    (ties keep rank order, then local order -- the order of the reference's list concatenation)."""
    import torch
    import torch.distributed as dist
    rec = torch.as_tensor(np.ascontiguousarray(local_records, dtype=np.float64))
    if device is not None:
        rec = rec.to(device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        allrec = rec
    else:
        world = dist.get_world_size()
        cnt = torch.tensor([rec.shape[0]], dtype=torch.int64, device=rec.device)
        counts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(counts, cnt)
        counts = [int(c.item()) for c in counts]
        m = max(counts) if counts else 0
        pad = torch.zeros((m, 3), dtype=torch.float64, device=rec.device)
        pad[: rec.shape[0]] = rec
        bufs = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)                                  # ONE collective of fixed-size records
        allrec = torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
    a = allrec.cpu().numpy()
    order = np.argsort(a[:, 0], kind="stable")
    a = a[order]
    return [(int(r[1]), int(r[2]), float(r[0])) for r in a]
