"""Device mirror of a tree's genome lists, built level-synchronously on the GPU.

Given a rooted binary topology with branch lengths and the tip genome lists, compute for every
node the four lists MAPLE keeps (M:331-376): ``probVect`` (lower), ``probVectUpRight`` /
``probVectUpLeft`` (upper, seen by child 0 / child 1) and ``probVectTotUp`` (mid-branch total) --
the two passes of reCalculateAllGenomeLists (M:6013-6347) -- as batches of mergeVectors /
rootVector / shorten launches, one batch per tree level.  No list ever leaves HBM.

This mirror is for trees without MAT local references (``tree.mutations`` all empty, the
reference's ``--noLocalRef`` layout); it is what ``bench.py`` scores candidates against.
"""
from __future__ import annotations

import numpy as np

from .runtime import Device


class TreeMirror:
    def __init__(self, dev: Device, parent, blen, tip_lists=None, *, tip_packed=None):
        """parent[n] (-1 root, parents precede children), blen[n]; the tips' lists either as ``tip_lists`` = {node: tuple-form
        list} or as ``tip_packed`` = (nodes, PackedLists) (host.tip_lists_packed: no Python object per entry)."""
        self.dev = dev
        self.parent = np.asarray(parent, dtype=np.int64)
        self.dist = np.asarray(blen, dtype=np.float64)
        n = len(self.parent)
        self.n_nodes = n
        # children in index order (child 0 = the smaller index); depths level by level (parents precede children)
        self.children = -np.ones((n, 2), dtype=np.int64)
        kids = np.nonzero(self.parent >= 0)[0]
        kids = kids[np.argsort(self.parent[kids], kind="stable")]
        par = self.parent[kids]
        firstk = np.ones(len(kids), dtype=bool)
        firstk[1:] = par[1:] != par[:-1]
        self.children[par[firstk], 0] = kids[firstk]
        self.children[par[~firstk], 1] = kids[~firstk]
        if (np.bincount(par, minlength=n) > 2).any():
            raise ValueError("TreeMirror needs a binary tree")
        self.is_tip = self.children[:, 0] < 0
        self.root = int(np.nonzero(self.parent < 0)[0][0])
        depth = np.zeros(n, dtype=np.int64)
        level = np.asarray([self.root], dtype=np.int64)
        d = 0
        while len(level):
            depth[level] = d
            ch = self.children[level].reshape(-1)
            level = ch[ch >= 0]
            d += 1
        self.depth = depth
        self.lower = -np.ones(n, dtype=np.int32)
        self.up_right = -np.ones(n, dtype=np.int32)
        self.up_left = -np.ones(n, dtype=np.int32)
        self.tot_up = -np.ones(n, dtype=np.int32)
        if tip_packed is not None:
            nodes, pl = tip_packed
            self.lower[np.asarray(nodes, dtype=np.int64)] = dev.upload_packed(pl)
        else:
            tips = sorted(tip_lists)
            ids = dev.upload([tip_lists[t] for t in tips])
            self.lower[np.asarray(tips, dtype=np.int64)] = ids
        self.launches = 0

    def build(self, max_restarts=64, native=True):
        """Build all lists; zero-length branches that turn out to be inconsistent with the data (mergeVectors returns
        None, M:4757-4762; the reference then re-estimates the branch with updateBLen, M:5377-5414) are given the
        length of one tenth of a mutation and the build is restarted.  ``native``: the level loop inside the library
        (maple_tree_rebuild_lists) instead of the Python one below (kept: the tests compare the two)."""
        mark = self.dev.mark()
        tip_ids = self.lower.copy()
        for _ in range(max_restarts):
            bad = self._build_native() if native else self._build_once()
            if bad is None:
                return self
            self.dist[bad] = np.maximum(self.dist[bad], 0.1 / self.dev.lRef)
            self.dev.release(mark)
            self.lower = tip_ids.copy()
            self.up_right[:] = -1
            self.up_left[:] = -1
            self.tot_up[:] = -1
        raise RuntimeError("inconsistent zero-length branches remain after restarts")

    def _build_native(self):
        dev = self.dev
        self.dist = np.ascontiguousarray(self.dist, dtype=np.float64)
        lo, ur, ul, tu, bad = dev.tree_rebuild_lists(self.root, self.parent, self.children[:, 0], self.children[:, 1], self.is_tip, None,
                                                     self.dist, self.lower, bump_len=0.1 / dev.lRef)
        if len(bad):
            return bad
        self.lower, self.up_right, self.up_left, self.tot_up = lo, ur, ul, tu
        return None

    def _build_once(self):
        dev, ch, dist = self.dev, self.children, self.dist
        internal = np.nonzero(~self.is_tip)[0]
        maxd = int(self.depth.max())
        # pass 1 (M:6031-6200): lower lists, deepest level first
        for d in range(maxd, -1, -1):
            nodes = internal[self.depth[internal] == d]
            if len(nodes) == 0:
                continue
            c0, c1 = ch[nodes, 0], ch[nodes, 1]
            out = dev.merge_batch(self.lower[c0], dist[c0], self.is_tip[c0], self.lower[c1], dist[c1], self.is_tip[c1],
                                  False)
            b = out < 0
            if b.any():
                # a child's branch length does not enter the child's own lower list: lengthen the two branches and merge
                # these pairs again instead of starting the whole build over
                bump = np.concatenate([c0[b], c1[b]])
                dist[bump] = np.maximum(dist[bump], 0.1 / dev.lRef)
                out[b] = dev.merge_batch(self.lower[c0[b]], dist[c0[b]], self.is_tip[c0[b]], self.lower[c1[b]], dist[c1[b]],
                                         self.is_tip[c1[b]], False)
                if (out < 0).any():
                    return np.concatenate([c0[out < 0], c1[out < 0]])
            self.lower[nodes] = dev.shorten_batch(out)
            self.launches += 2
        # pass 2 (M:6226-6345): upper lists from the root down
        r = self.root
        if not self.is_tip[r]:
            c0, c1 = ch[r]
            rv = dev.root_vector_batch([self.lower[c1], self.lower[c0]], [dist[c1], dist[c0]],
                                       [self.is_tip[c1], self.is_tip[c0]], [[], []])
            self.up_right[r], self.up_left[r] = rv[0], rv[1]
            self.launches += 1
        for d in range(1, maxd + 1):
            nodes = np.nonzero(self.depth == d)[0]
            if len(nodes) == 0:
                continue
            p = self.parent[nodes]
            first = ch[p, 0] == nodes
            vect_up = np.where(first, self.up_right[p], self.up_left[p]).astype(np.int32)
            nz = dist[nodes] != 0.0
            if nz.any():
                nn = nodes[nz]
                tu = dev.merge_batch(vect_up[nz], dist[nn] / 2, False, self.lower[nn], dist[nn] / 2, self.is_tip[nn], True)
                ok = tu >= 0
                sh = dev.shorten_batch(tu[ok])
                self.tot_up[nn[ok]] = sh
                self.launches += 2
            inner = ~self.is_tip[nodes]
            if inner.any():
                nn = nodes[inner]
                vu = vect_up[inner]
                c0, c1 = ch[nn, 0], ch[nn, 1]
                ur = dev.merge_batch(vu, dist[nn], False, self.lower[c1], dist[c1], self.is_tip[c1], True)
                ul = dev.merge_batch(vu, dist[nn], False, self.lower[c0], dist[c0], self.is_tip[c0], True)
                if (ur < 0).any() or (ul < 0).any():
                    return np.concatenate([nn[ur < 0], c1[ur < 0], nn[ul < 0], c0[ul < 0]])
                self.up_right[nn] = dev.shorten_batch(ur)
                self.up_left[nn] = dev.shorten_batch(ul)
                self.launches += 4
        return None

    def candidate_nodes(self, min_blen):
        """Nodes whose mid-branch total list is a placement candidate (dist > effectivelyNon0BLen, M:8012)."""
        ok = (self.tot_up >= 0) & (self.dist > min_blen) & (self.parent >= 0)
        return np.nonzero(ok)[0]

    def candidates_by_length(self, min_blen):
        """Candidate nodes ordered by the length of their mid-branch list (longest first).  A wavefront walks 64
        lists in lock-step and takes as long as its longest one, so neighbours should have similar lengths."""
        cand = self.candidate_nodes(min_blen)
        ne, _ = self.dev.sizes(self.tot_up[cand])
        return cand[np.argsort(-ne, kind="stable")]
