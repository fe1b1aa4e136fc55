"""Host tree (MAPLE's struct-of-lists ``Tree``, M:331-376) -> device mirror.

``HostTree`` holds what the search rows need: topology, branch lengths, MAT branch mutations,
minor-sequence counts and the four genome lists of every node in the reference's tuple form.
``upload`` packs the lists into the arena and the topology into the device tree
(``maple_tree_upload``).
"""
from __future__ import annotations

import numpy as np

from .runtime import Device


class HostTree:
    def __init__(self, root, up, children, dist, mutations, n_minor, probVect, probVectUpRight, probVectUpLeft,
                 probVectTotUp):
        self.root = root
        self.up = list(up)
        self.children = [list(c) for c in children]
        self.dist = list(dist)
        self.mutations = [list(m) for m in mutations]
        self.n_minor = list(n_minor)
        self.probVect = probVect
        self.probVectUpRight = probVectUpRight
        self.probVectUpLeft = probVectUpLeft
        self.probVectTotUp = probVectTotUp
        self.n = len(self.up)

    def preorder(self):
        """Visit order of startTopologyUpdatesParallel (M:9615-9618): pop the last pushed child first."""
        order, stack = [], [self.root]
        while stack:
            v = stack.pop()
            order.append(v)
            stack.extend(self.children[v])
        return order

    def assign_core_numbers(self, num_cores):
        """assignCoreNumbers (M:12164-12195): pre-order (child 0 first) index modulo num_cores."""
        core = [None] * self.n
        cur = 0
        stack = [self.root]
        while stack:
            v = stack.pop()
            core[v] = cur
            cur = (cur + 1) % num_cores
            stack.extend(reversed(self.children[v]))
        return core

    def upload(self, dev: Device):
        def ids_for(lists):
            idx = [i for i, x in enumerate(lists) if x]
            out = -np.ones(self.n, dtype=np.int32)
            if idx:
                out[np.asarray(idx)] = dev.upload([[tuple(e) for e in lists[i]] for i in idx])
            return out
        self.id_lower = ids_for(self.probVect)
        self.id_upRight = ids_for(self.probVectUpRight)
        self.id_upLeft = ids_for(self.probVectUpLeft)
        self.id_totUp = ids_for(self.probVectTotUp)
        midx = [i for i, m in enumerate(self.mutations) if m]
        self.id_mut = -np.ones(self.n, dtype=np.int32)
        if midx:
            self.id_mut[np.asarray(midx)] = dev.upload_mutations([self.mutations[i] for i in midx])
        up = np.asarray([-1 if u is None else u for u in self.up], dtype=np.int32)
        c0 = np.asarray([c[0] if c else -1 for c in self.children], dtype=np.int32)
        c1 = np.asarray([c[1] if c else -1 for c in self.children], dtype=np.int32)
        is_tip = np.asarray([(not c) and (m == 0) for c, m in zip(self.children, self.n_minor)], dtype=np.uint8)
        dev.upload_tree(self.root, up, c0, c1, self.dist, is_tip, self.id_lower, self.id_upRight, self.id_upLeft,
                        self.id_totUp, self.id_mut)
        return self
