"""The apply phase of a parallel SPR round -- applySPRMovesParallel (M:9470-9484): the proposed moves, best first, are
re-searched on the CURRENT tree (traverseTreeForTopologyUpdate -> findBestParentTopology, M:9287-9464) and applied if the
search still proposes a move; after each applied move the genome lists around the change are repaired (updatePartials,
M:9249-9252) and the library's copy of the tree follows through maple_tree_patch.

Two drivers over the same steps:

* ``apply_sequential``: one re-search per move, as the reference does it.
* ``apply_batched``: the next K moves are re-searched in ONE call on the current tree (speculatively), then applied in
  order; the speculative result of a move is kept only if nothing its search may have read -- the branches the frontier
  tier expanded for it (maple_spr_search_visited) and their relatives -- was touched by the moves applied before it in the
  batch; at the first move that fails this test the batch ends and the rest is re-searched.  The applied sequence, every
  branch length and every list are those of the sequential driver (tests/test_hip_scale.py).

The tree edit itself is a STAND-IN for MAPLE's cutAndPasteNode / placeSubtreeOnTree (M:9188-9277, tree surgery of the
reference's host code, out of this repository's scope): the pruned node's parent moves with it onto the branch above the
best node, with the three branch lengths the search returned.  It has the shape and the locality of the reference's edit
(same nodes change hands, same lists are invalidated) -- good for driving and timing the path, not a re-implementation.
"""
from __future__ import annotations

import time

import numpy as np

from .runtime import MapleError

DEPTH_STEP = 1 << 12        # depths in units of 1/4096 of a level: a node put on a branch gets one in between


class SprApplier:
    def __init__(self, dev, root, parent, children, dist, is_tip, lower, up_right, up_left, tot_up):
        self.dev = dev
        self.root = int(root)
        n = len(parent)
        self.n = n
        self.up = np.asarray(parent, dtype=np.int32).copy()
        self.c0 = np.asarray(children[:, 0], dtype=np.int32).copy()
        self.c1 = np.asarray(children[:, 1], dtype=np.int32).copy()
        self.tip = np.asarray(is_tip, dtype=np.uint8).copy()
        self.dist = np.asarray(dist, dtype=np.float64).copy()
        self.mut = np.full(n, -1, dtype=np.int32)
        self.lower, self.up_right = np.asarray(lower, np.int32).copy(), np.asarray(up_right, np.int32).copy()
        self.up_left, self.tot_up = np.asarray(up_left, np.int32).copy(), np.asarray(tot_up, np.int32).copy()
        self.depth = np.zeros(n, dtype=np.int32)
        st = [self.root]
        while st:
            v = st.pop()
            if self.c0[v] >= 0:
                for ch in (int(self.c0[v]), int(self.c1[v])):
                    self.depth[ch] = self.depth[v] + DEPTH_STEP
                    st.append(ch)
        dev.upload_tree(self.root, self.up, self.c0, self.c1, self.dist, self.tip, self.lower, self.up_right, self.up_left,
                        self.tot_up, self.mut)
        self.times = dict(search=[], update=[], patch=[])
        self.applied = []              # (node, placement) in the order they were applied
        self.no_longer_proposed = 0
        self.skipped = 0
        self.patched = []
        self.batches = []              # (searched, kept) per batch of apply_batched
        self.degraded = 0              # batches of which only the first result could be used (a search left the frontier tier)
        self.whole_tree_searches = 0   # re-searches from a zero-length branch (apply_sequential: on freshly rebuilt tables)

    @classmethod
    def from_mirror(cls, dev, m):
        return cls(dev, m.root, m.parent, m.children, m.dist, m.is_tip, m.lower, m.up_right, m.up_left, m.tot_up)

    # ---- the stand-in edit + repair + patch; returns the touched nodes, or None where the edit does not apply ----------
    def _apply(self, node, b, blen):
        up, c0, c1, dist, depth = self.up, self.c0, self.c1, self.dist, self.depth
        top, bottom, app = (float(x) for x in blen)
        p = int(up[node])
        g = int(up[p]) if p >= 0 else -1
        bp = int(up[b])
        if g < 0 or bp < 0 or b == p or bp == p:
            return None                                  # (the stand-in edit does not re-root)
        s_ = int(c1[p] if c0[p] == node else c0[p])
        if b == s_:
            return None
        # prune: the sibling takes the parent's place
        if c0[g] == p:
            c0[g] = s_
        else:
            c1[g] = s_
        up[s_] = g
        dist[s_] = dist[s_] + dist[p]
        # regraft: the parent goes onto the branch above b
        bp = int(up[b])
        if c0[bp] == b:
            c0[bp] = p
        else:
            c1[bp] = p
        up[p], dist[p] = bp, top
        if c0[p] == node:
            c1[p] = b
        else:
            c0[p] = b
        up[b], dist[b], dist[node] = p, bottom, app
        depth[p] = (int(depth[bp]) + int(depth[b])) // 2
        if not (depth[bp] < depth[p] < depth[b]):
            raise RuntimeError("SprApplier: out of depth resolution on one branch (raise DEPTH_STEP)")
        if depth[node] <= depth[p]:                      # the moved clade goes deeper: shift its depths
            delta = int(depth[p]) + 1 - int(depth[node])
            st = [node]
            while st:
                w = st.pop()
                depth[w] += delta
                if c0[w] >= 0:
                    st.extend((int(c0[w]), int(c1[w])))
        t0 = time.perf_counter()
        self.dev.update_partials(self.root, up, c0, c1, self.tip, self.mut, depth, dist, self.lower, self.up_right, self.up_left,
                                 self.tot_up, [s_, b, node, p])
        self.times["update"].append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        touched = np.unique(np.concatenate([self.dev.update_partials_touched(), [g, s_, p, b, node, bp]])).astype(np.int32)
        self.dev.tree_patch(self.n, touched, up[touched], c0[touched], c1[touched], dist[touched], self.tip[touched],
                            self.lower[touched], self.up_right[touched], self.up_left[touched], self.tot_up[touched])
        self.times["patch"].append(time.perf_counter() - t0)
        self.patched.append(len(touched))
        return touched

    def _take(self, node, r, k):
        """Apply the k-th result of a search call to `node`; returns the touched nodes (None: nothing applied)."""
        b = int(r["placement"][k])
        if r["status"][k] != 0 or b < 0:
            self.no_longer_proposed += 1                 # an earlier move of the phase took the improvement away
            return None
        touched = self._apply(int(node), b, r["blen"][k])
        if touched is None:
            self.skipped += 1
            return None
        self.applied.append((int(node), b))
        return touched

    def apply_sequential(self, moves, kw):
        for node in moves:
            t0 = time.perf_counter()
            # A pruned node on a zero-length branch is searched against the whole tree when there is no error model (M:9644, 6663):
            # on the patched node records alone that is 10^5 items through the frontier tier and a walk of all of them by one
            # lane (hundreds of ms).  With the library's own budget the call brings the tree's tables up to date first (one
            # re-upload) and the search takes the witness filter + clade scan path of a round instead.
            whole_tree = (not self.dev.u) and self.dist[node] == 0.0
            r = self.dev.spr_search_batch(np.asarray([node], dtype=np.int32), wide_search_budget=0 if whole_tree else -1, **kw)
            self.whole_tree_searches += int(whole_tree)
            self.times["search"].append(time.perf_counter() - t0)
            self._take(node, r, 0)
        return self

    def apply_batched(self, moves, kw, batch=32):
        moves = [int(v) for v in moves]
        pos = 0
        while pos < len(moves):
            # A re-search from a zero-length branch without an error model is a whole-tree search (see apply_sequential): it goes
            # alone, with the library's own budget (tables brought up to date, witness filter + clade scan) -- on the patched node
            # records alone it is 10^5 items for ONE lane's walk, and in a speculative batch it made every batch that slow
            # (round 4: 78 ms per move against the sequential driver's 9.6).  A batch ends in front of the next such move.
            if (not self.dev.u) and self.dist[moves[pos]] == 0.0:
                t0 = time.perf_counter()
                r = self.dev.spr_search_batch(np.asarray([moves[pos]], dtype=np.int32), wide_search_budget=0, **kw)
                self.whole_tree_searches += 1
                self.times["search"].append(time.perf_counter() - t0)
                self._take(moves[pos], r, 0)
                self.batches.append((1, 1))
                pos += 1
                continue
            chunk = moves[pos:pos + batch]
            for k in range(1, len(chunk)):
                if (not self.dev.u) and self.dist[chunk[k]] == 0.0:
                    chunk = chunk[:k]
                    break
            t0 = time.perf_counter()
            r = self.dev.spr_search_batch(np.asarray(chunk, dtype=np.int32), wide_search_budget=-1, **kw)
            try:
                q, v = self.dev.spr_search_visited()
            except MapleError as e:
                if e.code != MapleError.ERR_STATE:
                    raise
                # one of the searches was handed to the one-lane kernel (it keeps no record of what it read): only the first
                # result of the batch -- searched on a clean tree -- can be used
                self.degraded += 1
                chunk = chunk[:1]
                q, v = np.zeros(0, np.int32), np.zeros(0, np.int32)
            self.times["search"].append(time.perf_counter() - t0)
            # what each search may have read: its expanded branches and their relatives (as the tree is NOW)
            order = np.argsort(q, kind="stable")
            qs, vs = q[order], v[order]
            cuts = np.searchsorted(qs, np.arange(len(chunk) + 1))
            dirty = np.zeros(self.n, dtype=bool)
            kept = 0
            for k, node in enumerate(chunk):
                mine = vs[cuts[k]:cuts[k + 1]]
                rel = np.concatenate([mine, self.up[mine], self.c0[mine], self.c1[mine], [node, self.up[node]]])
                rel = rel[rel >= 0]
                rel = np.concatenate([rel, self.up[rel], self.c0[rel], self.c1[rel]])
                rel = rel[rel >= 0]
                if dirty[rel].any():
                    break                                # an earlier move of the batch touched what this search read: search again
                touched = self._take(node, r, k)
                kept += 1
                if touched is not None:
                    dirty[touched] = True
            self.batches.append((len(chunk), kept))
            pos += kept
            if kept == 0:                                # (cannot happen: the first search of a batch read a clean tree)
                raise RuntimeError("apply_batched made no progress")
        return self
