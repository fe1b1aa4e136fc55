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
        self._cols = None
        self.dist = dist
        self.mutations = [list(m) for m in mutations]
        self.n_minor = list(n_minor)
        self.probVect = probVect
        self.probVectUpRight = probVectUpRight
        self.probVectUpLeft = probVectUpLeft
        self.probVectTotUp = probVectTotUp
        self.n = len(self.up)

    @property
    def dist(self):
        """Branch lengths as one float64 column (None / False of the reference's tree = 0.0); assign elements in place."""
        return self._dist

    @dist.setter
    def dist(self, value):
        if isinstance(value, np.ndarray):
            self._dist = np.ascontiguousarray(value, dtype=np.float64)
        else:
            self._dist = np.asarray([float(x or 0.0) for x in value], dtype=np.float64)

    def columns(self):
        """The topology as plain columns (up, child0, child1, isTip, depth: int32 / uint8 arrays, -1 = none) -- what the
        C ABI takes; cached until apply_topology changes the tree."""
        sig = (self.n, sum(self.n_minor))                 # (a sample placed as a minor sequence turns a tip into a non-tip)
        if self._cols is not None and getattr(self, "_cols_sig", None) != sig:
            self._cols = None
        if self._cols is None:
            self._cols_sig = sig
            up = np.asarray([-1 if u is None else u for u in self.up], dtype=np.int32)
            c0 = np.asarray([c[0] if c else -1 for c in self.children], dtype=np.int32)
            c1 = np.asarray([c[1] if c else -1 for c in self.children], dtype=np.int32)
            tip = np.asarray([(not c) and (m == 0) for c, m in zip(self.children, self.n_minor)], dtype=np.uint8)
            depth = np.zeros(self.n, dtype=np.int32)
            for v in self.preorder():
                if up[v] >= 0:
                    depth[v] = depth[up[v]] + 1
            self._cols = (up, c0, c1, tip, depth)
        return self._cols

    @classmethod
    def from_mirror(cls, mirror, dev: Device = None):
        """The GPU-built TreeMirror (lists already in the arena, no MAT mutations) as a HostTree; with ``dev`` the
        topology is also uploaded (maple_tree_upload) for the device-resident searches."""
        nn = mirror.n_nodes
        up = [None if p < 0 else int(p) for p in mirror.parent]
        children = [[] if mirror.children[v, 0] < 0 else [int(mirror.children[v, 0]), int(mirror.children[v, 1])]
                    for v in range(nn)]
        t = cls(mirror.root, up, children, [float(x) for x in mirror.dist], [[] for _ in range(nn)], [0] * nn,
                None, None, None, None)
        t.id_lower, t.id_upRight = mirror.lower.copy(), mirror.up_right.copy()
        t.id_upLeft, t.id_totUp = mirror.up_left.copy(), mirror.tot_up.copy()
        t.id_mut = -np.ones(nn, dtype=np.int32)
        if dev is not None:
            dev.upload_tree(mirror.root, mirror.parent, mirror.children[:, 0], mirror.children[:, 1], mirror.dist,
                            mirror.is_tip, t.id_lower, t.id_upRight, t.id_upLeft, t.id_totUp, t.id_mut)
        return t

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
        return self.upload_topology(dev)

    def upload_topology(self, dev: Device):
        """(Re-)upload the topology and the list ids (maple_tree_upload) -- after tree surgery or a repair of the lists."""
        up, c0, c1, is_tip, _ = self.columns()
        dev.upload_tree(self.root, up, c0, c1, self.dist, is_tip, self.id_lower, self.id_upRight, self.id_upLeft,
                        self.id_totUp, self.id_mut)
        self._sent = dict(root=self.root, up=up.copy(), c0=c0.copy(), c1=c1.copy(), tip=is_tip.copy(), dist=self.dist.copy(),
                          lower=self.id_lower.copy(), upRight=self.id_upRight.copy(), upLeft=self.id_upLeft.copy(),
                          totUp=self.id_totUp.copy(), mut=self.id_mut.copy())
        return self

    def sync(self, dev: Device):
        """Bring the library's copy of the tree up to date after a LOCAL change (a placed sample, the repair of the lists
        around it): the nodes whose record differs from what was last sent go through maple_tree_patch -- a few 4-byte
        writes instead of the whole tree -- and the next single-query placement search runs on the patched tree.  Falls
        back to the full upload when the root or a mutation list changed, or nothing was sent yet.  Returns the number of
        patched nodes (-1: full upload)."""
        sent = getattr(self, "_sent", None)
        up, c0, c1, is_tip, _ = self.columns()
        if sent is None or sent["root"] != self.root or len(sent["up"]) > self.n:
            self.upload_topology(dev)
            return -1
        n_old = len(sent["up"])
        if not np.array_equal(sent["mut"], self.id_mut[:n_old]) or (self.id_mut[n_old:] >= 0).any():
            self.upload_topology(dev)
            return -1
        now = dict(up=up, c0=c0, c1=c1, tip=is_tip, dist=self.dist, lower=self.id_lower, upRight=self.id_upRight,
                   upLeft=self.id_upLeft, totUp=self.id_totUp)
        diff = np.zeros(n_old, dtype=bool)
        for k, a in now.items():
            diff |= a[:n_old] != sent[k]
        touched = np.concatenate([np.nonzero(diff)[0], np.arange(n_old, self.n)]).astype(np.int32)
        if len(touched):
            dev.tree_patch(self.n, touched, up[touched], c0[touched], c1[touched], self.dist[touched], is_tip[touched],
                           self.id_lower[touched], self.id_upRight[touched], self.id_upLeft[touched], self.id_totUp[touched])
            for k, a in now.items():
                sent[k] = a.copy()
            sent["mut"] = self.id_mut.copy()
        return len(touched)

    def apply_topology(self, root, up, children, dist, n_minor, mutations=None, dev: Device = None):
        """Tree surgery done by the host (placeSampleOnTree / cutAndPasteNode stay host code): take over the new arrays,
        growing the per-node tables for nodes that did not exist (their lists are missing: id -1).  Returns the nodes
        whose parent, children, branch length or MAT mutation list changed -- what update_genome_lists needs to be told.
        ``mutations`` (with ``dev``): the tree's MAT mutation lists after the edit (a placement above a reference node hands
        that node's list to the new internal node); changed ones are uploaded."""
        n_new = len(up)
        changed = []
        for v in range(n_new):
            if v >= self.n:
                changed.append(v)
            elif self.up[v] != up[v] or list(self.children[v]) != list(children[v] or []) \
                    or float(self.dist[v] or 0.0) != float(dist[v] or 0.0):
                changed.append(v)
            elif mutations is not None and [list(m) for m in (mutations[v] or [])] != [list(m) for m in self.mutations[v]]:
                changed.append(v)
        grow = n_new - self.n
        if grow > 0:
            for name in ("id_lower", "id_upRight", "id_upLeft", "id_totUp", "id_mut"):
                setattr(self, name, np.concatenate([getattr(self, name), -np.ones(grow, dtype=np.int32)]))
            self.mutations = self.mutations + [[] for _ in range(grow)]
        self.root, self.up, self.children = root, list(up), [list(c) if c else [] for c in children]
        self.dist, self.n_minor, self.n = [float(x or 0.0) for x in dist], list(n_minor), n_new
        if mutations is not None:
            new_mut = [[list(m) for m in (ml or [])] for ml in mutations]
            for v in range(n_new):
                if new_mut[v] != [list(m) for m in self.mutations[v]]:
                    self.mutations[v] = new_mut[v]
                    self.id_mut[v] = dev.upload_mutations([new_mut[v]])[0] if new_mut[v] else -1
        self._cols = None
        return changed


def compact_arena(dev: Device, tree: HostTree, keep=()):
    """Give the room of every replaced genome list back (maple_arena_compact): the four lists of every node of ``tree`` --
    and the extra ids ``keep`` -- survive, renumbered; the tree is uploaded again.  For long runs of update_genome_lists /
    single-sample placements, whose every replaced list is bump-allocated (call it when Device.stats() shows the arena
    filling up).  Returns the new ids of ``keep``."""
    cols = [tree.id_lower, tree.id_upRight, tree.id_upLeft, tree.id_totUp]
    live = np.concatenate(cols + [np.asarray(keep, dtype=np.int32)])
    # (a list may be shared by two columns -- never in trees built here, but an id must not be named twice)
    uniq, first = np.unique(live[live >= 0], return_index=False), None
    new_of = dict(zip(uniq.tolist(), dev.arena_compact(uniq).tolist()))
    remap = np.vectorize(lambda i: new_of.get(int(i), -1) if i >= 0 else -1, otypes=[np.int32])
    n = tree.n
    tree.id_lower, tree.id_upRight = remap(cols[0]), remap(cols[1])
    tree.id_upLeft, tree.id_totUp = remap(cols[2]), remap(cols[3])
    tree.upload_topology(dev)
    return remap(np.asarray(keep, dtype=np.int32)) if len(keep) else np.zeros(0, np.int32)


def tree_log_likelihood(dev: Device, tree: HostTree):
    """calculateTreeLikelihood (M:9721-9779) on the uploaded tree: one merge launch (returnLK) over every internal
    node's two stored lower lists, one findProbRoot launch, summed in the reference's post-order."""
    order = []                                   # internal nodes in the order the reference adds their contribution
    stack = [(tree.root, False)]
    while stack:
        v, done = stack.pop()
        if not tree.children[v]:
            continue
        if done:
            order.append(v)
        else:
            stack.append((v, True))
            stack.append((tree.children[v][1], False))
            stack.append((tree.children[v][0], False))
    mark = dev.mark()
    try:
        c0 = np.asarray([tree.children[v][0] for v in order])
        c1 = np.asarray([tree.children[v][1] for v in order])
        l0, l1 = tree.id_lower[c0].copy(), tree.id_lower[c1].copy()
        for arr, ch in ((l0, c0), (l1, c1)):
            need = np.nonzero(tree.id_mut[ch] >= 0)[0]
            if len(need):
                arr[need] = dev.pass_branch_batch(arr[need], tree.id_mut[ch[need]], True)
        dist = np.asarray(tree.dist)
        tip = np.asarray([(not c) and (m == 0) for c, m in zip(tree.children, tree.n_minor)])
        nminor = np.asarray(tree.n_minor, dtype=np.int32)
        out, lk = dev.merge_batch(l0, dist[c0], tip[c0], l1, dist[c1], tip[c1], False, returnLK=True,
                                  numMinor1=nminor[c0], numMinor2=nminor[c1])
        if (out < 0).any():
            raise RuntimeError("inconsistent lower genome lists (the reference raises here, M:9761)")
        total = 0.0
        for x in lk.tolist():
            total += x
        root_list = tree.id_lower[tree.root]
        if tree.id_mut[tree.root] >= 0:
            root_list = dev.pass_branch_batch([root_list], [tree.id_mut[tree.root]], True)[0]
        root_lk = float(dev.root_prob_batch([root_list])[0])
        return total + root_lk, root_lk
    finally:
        dev.release(mark)


def rebuild_genome_lists(dev: Device, tree: HostTree, native=True):
    """reCalculateAllGenomeLists (M:6013-6347) on the device for a tree WITH MAT local references: given only the
    tips' lower lists (already uploaded by HostTree.upload) recompute probVect of every internal node (pass 1) and
    probVectUpRight / probVectUpLeft / probVectTotUp of every node (pass 2), level by level, as batches of
    passGenomeListThroughBranch / mergeVectors / rootVector / shorten launches.  Returns the four id arrays.
    ``native``: the level loop inside the library (maple_tree_rebuild_lists: one fused launch per level); the Python loop
    below is kept for the tests, which compare the two list for list."""
    n = tree.n
    if native:
        up_ = np.asarray([-1 if u is None else u for u in tree.up], dtype=np.int32)
        c0_ = np.asarray([c[0] if c else -1 for c in tree.children], dtype=np.int32)
        c1_ = np.asarray([c[1] if c else -1 for c in tree.children], dtype=np.int32)
        tip_ = np.asarray([(not c) and (m == 0) for c, m in zip(tree.children, tree.n_minor)], dtype=np.uint8)
        dist_ = np.ascontiguousarray(np.asarray(tree.dist, dtype=np.float64))
        lo, ur, ul, tu, _ = dev.tree_rebuild_lists(tree.root, up_, c0_, c1_, tip_, tree.id_mut, dist_, tree.id_lower)
        return lo, ur, ul, tu
    up = np.asarray([-1 if u is None else u for u in tree.up])
    c0 = np.asarray([c[0] if c else -1 for c in tree.children])
    c1 = np.asarray([c[1] if c else -1 for c in tree.children])
    dist = np.asarray(tree.dist, dtype=np.float64)
    tip = np.asarray([(not c) and (m == 0) for c, m in zip(tree.children, tree.n_minor)])
    mut = tree.id_mut
    order = tree.preorder()
    depth = np.zeros(n, dtype=np.int64)
    reach = np.zeros(n, dtype=bool)
    for v in order:
        reach[v] = True
        if up[v] >= 0:
            depth[v] = depth[up[v]] + 1
    lower = -np.ones(n, dtype=np.int32)
    leaves = np.nonzero(reach & (c0 < 0))[0]
    lower[leaves] = tree.id_lower[leaves]
    up_right = -np.ones(n, dtype=np.int32)
    up_left = -np.ones(n, dtype=np.int32)
    tot_up = -np.ones(n, dtype=np.int32)

    def passed(ids, nodes, direction_up):
        """lists `ids` moved across the branches above `nodes` (only where those carry mutations)"""
        ids = np.asarray(ids, dtype=np.int32).copy()
        need = np.nonzero(mut[nodes] >= 0)[0]
        if len(need):
            ids[need] = dev.pass_branch_batch(ids[need], mut[nodes[need]], direction_up)
        return ids

    internal = np.nonzero(reach & (c0 >= 0))[0]
    maxd = int(depth[reach].max())
    for d in range(maxd, -1, -1):                                   # pass 1, M:6031-6200
        nodes = internal[depth[internal] == d]
        if len(nodes) == 0:
            continue
        a, b = c0[nodes], c1[nodes]
        out = dev.merge_batch(passed(lower[a], a, True), dist[a], tip[a], passed(lower[b], b, True), dist[b], tip[b], False)
        if (out < 0).any():
            raise RuntimeError("inconsistent lower lists (the reference would call updateBLen here)")
        lower[nodes] = dev.shorten_batch(out)
    r = tree.root                                                   # pass 2, M:6226-6345
    if c0[r] >= 0:
        path = [[int(mut[r])] if mut[r] >= 0 else []] * 2
        kids = np.asarray([c1[r], c0[r]])
        rv = dev.root_vector_batch(passed(lower[kids], kids, True), dist[kids], tip[kids], path)
        up_right[r], up_left[r] = rv[0], rv[1]
    for d in range(1, maxd + 1):
        nodes = np.nonzero(reach & (depth == d))[0]
        if len(nodes) == 0:
            continue
        p = up[nodes]
        vect_up = passed(np.where(c0[p] == nodes, up_right[p], up_left[p]), nodes, False)
        nz = dist[nodes] != 0.0
        if nz.any():
            nn = nodes[nz]
            tu = dev.merge_batch(vect_up[nz], dist[nn] / 2, False, lower[nn], dist[nn] / 2, tip[nn], True)
            ok = tu >= 0
            tot_up[nn[ok]] = dev.shorten_batch(tu[ok])
        inner = c0[nodes] >= 0
        if inner.any():
            nn, vu = nodes[inner], vect_up[inner]
            a, b = c0[nn], c1[nn]
            ur = dev.merge_batch(vu, dist[nn], False, passed(lower[b], b, True), dist[b], tip[b], True)
            ul = dev.merge_batch(vu, dist[nn], False, passed(lower[a], a, True), dist[a], tip[a], True)
            if (ur < 0).any() or (ul < 0).any():
                raise RuntimeError("inconsistent upper lists (the reference would call updateBLen here)")
            up_right[nn] = dev.shorten_batch(ur)
            up_left[nn] = dev.shorten_batch(ul)
    return lower, up_right, up_left, tot_up


def update_genome_lists(dev: Device, tree: HostTree, changed, changed_dist=None, native=True):
    """updatePartials (M:5479-5815) in its GPU-native form: instead of the reference's one-node-at-a-time LIFO work
    list, the lists invalidated by a set of local changes are repaired level by level, every level one batch of
    passGenomeListThroughBranch / mergeVectors / areVectorsDifferent / shorten launches -- so that any number of
    simultaneous changes costs the same number of launches as one.

    ``changed``: nodes whose lower list (``tree.id_lower[v]``, already replaced by the caller) and/or branch length
    (``tree.dist[v]``) changed; ``changed_dist`` (default: the same nodes) those whose branch length did.  Phase A walks up: a node whose child changed gets a new lower list, and propagates while
    areVectorsDifferent(new, old) (M:5793).  Phase B walks down from every touched node: probVectTotUp is recomputed
    where the node's lower list, length or upper vector changed (M:5525-5557, 5700-5713), probVectUpRight /
    probVectUpLeft where the upper vector or the OTHER child changed (M:5559-5660, 5715-5735), and a child is visited
    only if its upper vector was replaced, i.e. areVectorsDifferent(old, new) (M:5645-5658) -- the reference's own stop
    rule, so the repaired region is the same up to that threshold.  A merge that comes out None between two zero-length
    branches re-estimates the branch above the changed child like updateBLen (M:5385-5414).
    ``tree.id_*`` and ``tree.dist`` are updated in place; returns the number of lists replaced.

    ``native`` (default): the level loop runs inside the library (maple_update_partials, maple_amd/csrc/update_host.h) on
    these same columns; the Python loop below is the same algorithm call for call and is kept as its cross-check."""
    if native and changed_dist is None:
        up_, c0_, c1_, tip_, depth_ = tree.columns()
        return dev.update_partials(tree.root, up_, c0_, c1_, tip_, tree.id_mut, depth_, tree.dist, tree.id_lower, tree.id_upRight,
                                   tree.id_upLeft, tree.id_totUp, changed)
    n = tree.n
    up = np.asarray([-1 if u is None else u for u in tree.up])
    c0 = np.asarray([c[0] if c else -1 for c in tree.children])
    c1 = np.asarray([c[1] if c else -1 for c in tree.children])
    tip = np.asarray([(not c) and (m == 0) for c, m in zip(tree.children, tree.n_minor)])
    mut = tree.id_mut
    dist = np.asarray(tree.dist, dtype=np.float64)
    lower, up_right, up_left, tot_up = tree.id_lower, tree.id_upRight, tree.id_upLeft, tree.id_totUp
    depth = np.zeros(n, dtype=np.int64)
    for v in tree.preorder():
        if up[v] >= 0:
            depth[v] = depth[up[v]] + 1
    replaced = 0

    def passed(ids, nodes, direction_up):
        ids = np.asarray(ids, dtype=np.int32).copy()
        nodes = np.asarray(nodes)
        need = np.nonzero(mut[nodes] >= 0)[0]
        if len(need):
            ids[need] = dev.pass_branch_batch(ids[need], mut[nodes[need]], direction_up)
        return ids

    def vect_up_of(nodes):
        """the upper vector seen by each node, in the node's own reference frame (M:5503-5513)"""
        p = up[nodes]
        return passed(np.where(c0[p] == nodes, up_right[p], up_left[p]), nodes, False)

    d_low = np.zeros(n, dtype=bool)          # own lower list or branch length changed
    d_up = np.zeros(n, dtype=bool)           # the upper vector this node sees changed
    d_child = np.zeros((n, 2), dtype=bool)   # child k's lower list or branch length changed
    d_dist = np.zeros(n, dtype=bool)         # own branch length changed (enters the node's probVectUpRight/UpLeft too)
    for v in (changed if changed_dist is None else changed_dist):
        d_dist[v] = True
    frontier = set()
    for v in changed:
        d_low[v] = True
        if up[v] >= 0:
            d_child[up[v], 0 if c0[up[v]] == v else 1] = True
            frontier.add(int(up[v]))

    # ---- phase A: lower lists, deepest first -------------------------------------------------------------------
    while frontier:
        dmax = max(depth[v] for v in frontier)
        nodes = np.asarray(sorted(v for v in frontier if depth[v] == dmax))
        frontier.difference_update(nodes.tolist())
        a, b = c0[nodes], c1[nodes]
        pa, pb = passed(lower[a], a, True), passed(lower[b], b, True)
        out = dev.merge_batch(pa, dist[a], tip[a], pb, dist[b], tip[b], False)
        for k in np.nonzero(out < 0)[0]:                                   # M:5689-5701
            p = int(nodes[k])
            if dist[a[k]] or dist[b[k]]:
                raise RuntimeError("None vector from non-zero distances in the lower merge (the reference raises too)")
            order = [0, 1] if d_child[p, 0] else [1, 0]
            for which in order:
                c = int(c0[p] if which == 0 else c1[p])
                t, is_false = dev.blen_batch(vect_up_of(np.asarray([c])), [lower[c]], [bool(tip[c])])
                dist[c] = 0.0 if is_false[0] else float(t[0])
                tree.dist[c] = float(dist[c])
                d_low[c] = d_dist[c] = True
                d_child[p, which] = True
                if dist[c]:
                    break
            o2 = dev.merge_batch(pa[k:k + 1], dist[a[k:k + 1]], tip[a[k:k + 1]], pb[k:k + 1], dist[b[k:k + 1]],
                                 tip[b[k:k + 1]], False)
            if o2[0] < 0:
                raise RuntimeError("None vector after updateBLen")
            out[k] = o2[0]
        new = dev.shorten_batch(out)
        old = lower[nodes]
        diff = np.ones(len(nodes), dtype=bool)
        has_old = old >= 0
        if has_old.any():
            diff[has_old] = dev.differ_batch(new[has_old], old[has_old]).astype(bool)   # (new, old), M:5793
        lower[nodes] = new
        replaced += len(nodes)
        for v, ch in zip(nodes, diff):
            if ch:
                d_low[v] = True
                if up[v] >= 0:
                    d_child[up[v], 0 if c0[up[v]] == v else 1] = True
                    frontier.add(int(up[v]))

    # ---- phase B: upper lists, shallowest first ----------------------------------------------------------------
    todo = set(np.nonzero(d_low | d_child.any(axis=1))[0].tolist())
    while todo:
        dmin = min(depth[v] for v in todo)
        nodes = np.asarray(sorted(v for v in todo if depth[v] == dmin))
        todo.difference_update(nodes.tolist())
        inner = c0[nodes] >= 0
        if nodes[0] == tree.root:
            r = tree.root
            if c0[r] >= 0:
                path = [[int(mut[r])] if mut[r] >= 0 else []]
                for which, store in ((1, up_right), (0, up_left)):         # upRight merges child 1, upLeft child 0
                    if not d_child[r, which]:
                        continue
                    kid = np.asarray([c1[r] if which == 1 else c0[r]])
                    nv = dev.root_vector_batch(passed(lower[kid], kid, True), dist[kid], tip[kid], path)
                    target = int(c0[r] if which == 1 else c1[r])
                    if store[r] < 0 or dev.differ_batch([store[r]], nv)[0]:
                        store[r] = nv[0]
                        replaced += 1
                        d_up[target] = True
                        todo.add(target)
            continue
        vect_up = vect_up_of(nodes)
        # probVectTotUp
        need = d_up[nodes] | d_low[nodes]
        if need.any():
            nn, vu = nodes[need], vect_up[need]
            nz = dist[nn] != 0.0
            tot_up[nn[~nz]] = -1
            if nz.any():
                m = nn[nz]
                tu = dev.merge_batch(vu[nz], dist[m] / 2, False, lower[m], dist[m] / 2, tip[m], True)
                if (tu < 0).any():
                    raise RuntimeError("None probVectTotUp on a branch of non-zero length")
                tot_up[m] = dev.shorten_batch(tu)
                replaced += len(m)
        # probVectUpRight (for child 0: upper vector + child 1) and probVectUpLeft (for child 1: upper vector + child 0)
        for which, store in ((1, up_right), (0, up_left)):
            need = inner & (d_up[nodes] | d_dist[nodes] | d_child[nodes, which])
            if not need.any():
                continue
            nn, vu = nodes[need], vect_up[need]
            kid = c1[nn] if which == 1 else c0[nn]
            nv = dev.merge_batch(vu, dist[nn], False, passed(lower[kid], kid, True), dist[kid], tip[kid], True)
            if (nv < 0).any():
                raise RuntimeError("None upper vector (the reference would call updateBLen here)")
            old = store[nn]
            diff = np.ones(len(nn), dtype=bool)
            has_old = old >= 0
            if has_old.any():
                diff[has_old] = dev.differ_batch(old[has_old], nv[has_old]).astype(bool)      # (old, new), M:5645
            if diff.any():
                sel = nn[diff]
                store[sel] = dev.shorten_batch(nv[diff])
                replaced += len(sel)
                target = c0[sel] if which == 1 else c1[sel]
                d_up[target] = True
                todo.update(target.tolist())
    return replaced


def optimize_branch_lengths_fast_pass(dev: Device, tree: HostTree, effectivelyNon0BLen, dirty=None):
    """traverseTreeToOptimizeBranchLengths(tree, root, fastPass=True) (M:8727-8893): the two branches below the root
    share their total length by a grid search on the merged root vector's likelihood (M:8743-8780, one
    mergeVectors(returnLK) + findProbRoot launch over the whole grid), every other dirty branch gets
    estimateBranchLengthWithDerivative(upper vector, lower list) (M:8821-8833) -- all from the same frozen lists, hence
    ONE estimateBranchLength launch for the whole tree.  A length is replaced when it moves by more than 1 % (M:8873).
    ``tree.dist`` is updated in place (the lists are not: call update_genome_lists / rebuild_genome_lists afterwards,
    as the reference's caller does after a fast pass).  Returns (number of updates, dirty flags)."""
    root = tree.root
    if not tree.children[root]:
        return 0, dirty
    n = tree.n
    mut = tree.id_mut
    tip = np.asarray([(not c) and (m == 0) for c, m in zip(tree.children, tree.n_minor)])
    dist = [float(x or 0.0) for x in tree.dist]
    dirty = [True] * n if dirty is None else list(dirty)
    l_ref = dev.lRef
    mark = dev.mark()
    try:
        a, b = tree.children[root]
        if dist[a] > effectivelyNon0BLen or dist[b] > effectivelyNon0BLen:
            tot = (dist[a] + dist[b]) * l_ref
            grid = []
            for i in range(max(1, round(tot)) * 2 + 1):
                b1 = min(tot, float(i) / 2)
                b2 = max(tot - b1, 0.0)
                grid.append((b1 / l_ref, b2 / l_ref))
            pv = []
            for c in (a, b):
                lid = tree.id_lower[c]
                if mut[c] >= 0:
                    lid = dev.pass_branch_batch([lid], [mut[c]], True)[0]
                pv.append(int(lid))
            k = len(grid)
            out, lk = dev.merge_batch([pv[0]] * k, [g[0] for g in grid], [bool(tip[a])] * k, [pv[1]] * k,
                                      [g[1] for g in grid], [bool(tip[b])] * k, False, returnLK=True)
            if (out < 0).any():
                raise RuntimeError("None root vector in the root branch-length grid (the reference fails here too)")
            if mut[root] >= 0:
                out = dev.pass_branch_batch(out, [mut[root]] * k, True)
            cost = lk + dev.root_prob_batch(out)
            best, best_cost = None, float("-inf")
            for i in range(k):
                if cost[i] > best_cost:
                    best_cost, best = float(cost[i]), grid[i][0]
            both = dist[a] + dist[b]
            dist[a] = best
            dist[b] = max(both - best, 0.0)
        # every other branch: the reference starts from the root's grandchildren (M:8812-8819)
        nodes = []
        # same visiting order as the reference (children of child 0, then of child 1, LIFO), M:8812-8822
        stack = ([*tree.children[a]] if tree.children[a] else []) + ([*tree.children[b]] if tree.children[b] else [])
        while stack:
            v = stack.pop()
            nodes.append(v)
            stack.extend(tree.children[v])
        nodes = np.asarray([v for v in nodes if dirty[v]], dtype=np.int64)
        updates = 0
        if len(nodes):
            p = np.asarray([tree.up[v] for v in nodes])
            first = np.asarray([tree.children[u][0] for u in p]) == nodes
            up_vect = np.where(first, tree.id_upRight[p], tree.id_upLeft[p]).astype(np.int32)
            need = np.nonzero(mut[nodes] >= 0)[0]
            if len(need):
                up_vect[need] = dev.pass_branch_batch(up_vect[need], mut[nodes[need]], False)
            t, is_false = dev.blen_batch(up_vect, tree.id_lower[nodes], tip[nodes])
            for v, tv, f in zip(nodes.tolist(), t.tolist(), is_false.tolist()):
                best = 0.0 if f else tv
                if best or dist[v]:
                    if (not best) or (not dist[v]) or dist[v] / best > 1.01 or dist[v] / best < 0.99:   # M:8873
                        dist[v] = best
                        updates += 1
                    else:
                        dirty[v] = False
                else:
                    dirty[v] = False
        tree.dist = dist
        return updates, dirty
    finally:
        dev.release(mark)


def optimize_branch_lengths(dev: Device, tree: HostTree, effectivelyNon0BLen, dirty=None, batch=256):
    """traverseTreeToOptimizeBranchLengths(tree, root) with the reference's DEFAULT arguments (fastPass=False,
    M:8727-8893; every call site of the reference uses this form: M:11059, 11728, 11806, 11854, 11864, 11900, 11906,
    11979, 12256): a Gauss-Seidel sweep in the reference's visiting order -- every dirty branch is re-estimated with
    estimateBranchLengthWithDerivative(upper vector, lower list) from lists that already reflect every earlier change,
    and a branch that moves by more than 1 % (M:8873) is followed at once by the repair of the genome lists around it
    (updatePartials, M:8875-8878: here update_genome_lists).

    GPU form: the estimates of the next `batch` branches in visiting order are produced by ONE
    estimateBranchLength launch; an estimate stays valid for as long as neither list it was computed from has been
    replaced by a repair since (list ids are compared), so the sweep only goes back to the GPU for estimates when a
    repair reached a branch that was already estimated -- the result is that of the one-at-a-time sweep.
    ``tree.dist`` and the genome lists are updated in place.  Returns (number of updates, dirty flags, nodes updated in
    order)."""
    root = tree.root
    if not tree.children[root]:
        return 0, dirty, []
    n = tree.n
    mut = tree.id_mut
    tip = np.asarray([(not c) and (m == 0) for c, m in zip(tree.children, tree.n_minor)])
    dirty = [True] * n if dirty is None else list(dirty)
    l_ref = dev.lRef
    a, b = tree.children[root]
    dist = tree.dist
    for v in range(n):
        dist[v] = float(dist[v] or 0.0)
    if dist[a] > effectivelyNon0BLen or dist[b] > effectivelyNon0BLen:                  # M:8743-8810
        mark = dev.mark()
        tot = (dist[a] + dist[b]) * l_ref
        grid = []
        for i in range(max(1, round(tot)) * 2 + 1):
            b1 = min(tot, float(i) / 2)
            b2 = max(tot - b1, 0.0)
            grid.append((b1 / l_ref, b2 / l_ref))
        pv = []
        for c in (a, b):
            lid = tree.id_lower[c]
            if mut[c] >= 0:
                lid = dev.pass_branch_batch([lid], [mut[c]], True)[0]
            pv.append(int(lid))
        k = len(grid)
        out, lk = dev.merge_batch([pv[0]] * k, [g[0] for g in grid], [bool(tip[a])] * k, [pv[1]] * k,
                                  [g[1] for g in grid], [bool(tip[b])] * k, False, returnLK=True)
        if (out < 0).any():
            raise RuntimeError("None root vector in the root branch-length grid (the reference fails here too)")
        if mut[root] >= 0:
            out = dev.pass_branch_batch(out, [mut[root]] * k, True)
        cost = lk + dev.root_prob_batch(out)
        best, best_cost = None, float("-inf")
        for i in range(k):
            if cost[i] > best_cost:
                best_cost, best = float(cost[i]), grid[i][0]
        dev.release(mark)
        both = dist[a] + dist[b]
        if best != dist[a]:
            dist[a] = best
            update_genome_lists(dev, tree, [a])
        b2 = max(both - best, 0.0)
        if b2 != dist[b]:
            dist[b] = b2
            update_genome_lists(dev, tree, [b])
    order = []
    stack = ([*tree.children[a]] if tree.children[a] else []) + ([*tree.children[b]] if tree.children[b] else [])
    while stack:                                                       # the reference's visiting order, M:8812-8822 / 8890
        v = stack.pop()
        order.append(v)
        stack.extend(tree.children[v])
    todo = [v for v in order if dirty[v]]
    updates, updated_nodes = 0, []
    pos = 0
    while pos < len(todo):
        chunk = np.asarray(todo[pos:pos + batch], dtype=np.int64)
        mark = dev.mark()
        p = np.asarray([tree.up[v] for v in chunk])
        first = np.asarray([tree.children[u][0] for u in p]) == chunk
        up_id = np.where(first, tree.id_upRight[p], tree.id_upLeft[p]).astype(np.int32)
        low_id = tree.id_lower[chunk].copy()
        up_vect = up_id.copy()
        need = np.nonzero(mut[chunk] >= 0)[0]
        if len(need):
            up_vect[need] = dev.pass_branch_batch(up_vect[need], mut[chunk[need]], False)
        t, is_false = dev.blen_batch(up_vect, low_id, tip[chunk])
        dev.release(mark)                                             # (the passed copies were only needed for the estimates)
        done = 0
        for k, v in enumerate(chunk.tolist()):
            u = tree.up[v]
            cur_up = tree.id_upRight[u] if tree.children[u][0] == v else tree.id_upLeft[u]
            if cur_up != up_id[k] or tree.id_lower[v] != low_id[k]:
                break                                                 # a repair replaced one of its inputs: estimate again
            best = 0.0 if is_false[k] else float(t[k])
            if best or dist[v]:
                if (not best) or (not dist[v]) or dist[v] / best > 1.01 or dist[v] / best < 0.99:   # M:8873
                    dist[v] = best
                    updates += 1
                    updated_nodes.append(v)
                    update_genome_lists(dev, tree, [v])
                else:
                    dirty[v] = False
            else:
                dirty[v] = False
            done += 1
        pos += max(done, 0)
        if done == 0:                                                 # cannot happen (a fresh estimate is always valid)
            raise RuntimeError("branch-length sweep made no progress")
    return updates, dirty, updated_nodes


def find_best_root(dev: Device, tree: HostTree, *, strictTopologyStopRules, allowedFailsTopology, thresholdLogLKtopology,
                   thresholdLogLKoptimizationTopology, thresholdLogLKconsecutivePlacement):
    """The search of findBestRoot (M:7730-7840): the relative log-likelihood of re-rooting the tree on each branch.

    The score of a branch depends only on the path from the current root to it (the partials passed down, the
    likelihood to remove), not on the search's running best, so the scores of ALL branches are produced level by level
    -- per level three mergeVectors(returnLK) launches, the passes into the root frame and one findProbRoot launch --
    and the reference's depth-first traversal with its stop rules (M:7800-7830) is replayed on the host over them.
    Returns (bestNode, bestLKdiff, bestNodes {node: score within thresholdLogLKoptimizationTopology of the best},
    nodesVisitedRoot) as the reference has them when its search loop ends; re-rooting itself (reRootTree, M:7842-7873)
    is tree surgery and stays with the caller."""
    root = tree.root
    ch = tree.children
    mut = tree.id_mut
    n_minor = np.asarray(tree.n_minor, dtype=np.int32)
    tip = np.asarray([(not c) and (m == 0) for c, m in zip(ch, tree.n_minor)])
    dist = np.asarray([float(x or 0.0) for x in tree.dist])
    up = tree.up
    mark = dev.mark()
    dev.set_fatal_policy(True)            # the reference tries each rooting inside try/except (M:7793-7828)
    try:
        def passed(ids, nodes, direction_up):
            ids = np.asarray(ids, dtype=np.int32).copy()
            nodes = np.asarray(nodes)
            need = np.nonzero(mut[nodes] >= 0)[0]
            if len(need):
                ids[need] = dev.pass_branch_batch(ids[need], mut[nodes[need]], direction_up)
            return ids

        def root_prob_of(lists, nodes):
            """findProbRoot(list, node=t1, ...): through every mutated branch between t1 and the root, then the root sum"""
            lists = np.asarray(lists, dtype=np.int32).copy()
            cur = np.asarray(nodes).copy()
            alive = np.ones(len(cur), dtype=bool)
            while alive.any():
                idx = np.nonzero(alive)[0]
                need = idx[mut[cur[idx]] >= 0]
                if len(need):
                    lists[need] = dev.pass_branch_batch(lists[need], mut[cur[need]], True)
                nxt = np.asarray([-1 if up[v] is None else up[v] for v in cur[idx]])
                cur[idx] = np.where(nxt >= 0, nxt, cur[idx])
                alive[idx] = nxt >= 0
            return dev.root_prob_batch(lists)

        score = {}            # (t1, i) -> score of rooting on the branch above children[t1][i]
        if not ch[root]:
            return root, 0.0, {root: 0.0}, 1
        c1, c2 = ch[root]
        v1 = passed([tree.id_lower[c2]], [c2], True)     # what child1 sees from above: the other child's lower list
        v2 = passed([tree.id_lower[c1]], [c1], True)
        root_list = tree.id_lower[root]
        original = float(root_prob_of([root_list], [root])[0])
        _, lk0 = dev.merge_batch(v1, [dist[c2]], [tip[c2]], v2, [dist[c1]], [tip[c1]], False, returnLK=True,
                                 numMinor1=[n_minor[c2]], numMinor2=[n_minor[c1]])
        original += float(lk0[0])
        items = []            # (t1, passed list, distance, isTip, numMinor, LKtoRemove)
        if ch[c1]:
            items.append((c1, int(passed(v1, [c1], False)[0]), dist[c1] + dist[c2], bool(tip[c2]), int(n_minor[c2]), original))
        if ch[c2]:
            items.append((c2, int(passed(v2, [c2], False)[0]), dist[c2] + dist[c1], bool(tip[c1]), int(n_minor[c1]), original))
        to_pass = {}          # (t1, i) -> (list to hand to child i, its LKtoRemove)
        while items:
            t = np.asarray([it[0] for it in items])
            pas = np.asarray([it[1] for it in items], dtype=np.int32)
            dis = np.asarray([it[2] for it in items])
            ptip = np.asarray([it[3] for it in items])
            pmin = np.asarray([it[4] for it in items], dtype=np.int32)
            rem = np.asarray([it[5] for it in items])
            k0 = np.asarray([ch[v][0] for v in t])
            k1 = np.asarray([ch[v][1] for v in t])
            pv = [passed(tree.id_lower[k0], k0, True), passed(tree.id_lower[k1], k1, True)]
            kids = [k0, k1]
            _, lkc = dev.merge_batch(pv[0], dist[k0], tip[k0], pv[1], dist[k1], tip[k1], False, returnLK=True,
                                     numMinor1=n_minor[k0], numMinor2=n_minor[k1])
            new_rem = rem + lkc
            nxt = []
            for i in (0, 1):
                a, b = kids[1 - i], kids[i]
                up_vect, lk = dev.merge_batch(pv[1 - i], dist[a], tip[a], pas, dis, ptip, False, returnLK=True,
                                              numMinor1=n_minor[a], numMinor2=pmin)
                ok = up_vect >= 0
                new_root, lk_root = dev.merge_batch(np.where(ok, up_vect, pv[i]), dist[b] / 2, False, pv[i], dist[b] / 2, tip[b],
                                                    False, returnLK=True, numMinor1=np.zeros(len(t), np.int32), numMinor2=n_minor[b])
                ok &= new_root >= 0
                rp = root_prob_of(np.where(ok, new_root, pv[i]), t)
                sc = rp + lk_root + lk - new_rem
                vect = passed(np.where(ok, up_vect, pv[i]), b, False)
                changed = np.nonzero(ok & (mut[b] >= 0))[0]
                if len(changed):
                    vect[changed] = dev.shorten_batch(vect[changed])             # M:7834-7835
                for j in range(len(t)):
                    if not ok[j]:
                        continue                                                  # "Stopping root search at node ..." (M:7827)
                    score[(int(t[j]), i)] = float(sc[j])
                    if ch[int(b[j])]:
                        nxt.append((int(b[j]), int(vect[j]), float(dist[b[j]]), False, 0, float(new_rem[j] - lk[j])))
            items = nxt
        # ---- the reference's traversal over the scores, M:7786-7838
        best_node, best, visited = root, 0.0, 1
        best_nodes = {root: 0.0}
        stack = []
        if ch[c1]:
            stack.append((c1, 0.0, 0))
        if ch[c2]:
            stack.append((c2, 0.0, 0))
        while stack:
            visited += 1
            t1, last_lk, fails = stack.pop()
            for i in (0, 1):
                if (t1, i) not in score:
                    continue
                sc = score[(t1, i)]
                kid = ch[t1][i]
                fails_new = fails
                if sc > best:
                    best, best_node, fails_new = sc, kid, 0
                elif sc < last_lk - thresholdLogLKconsecutivePlacement:
                    fails_new += 1
                if sc >= best - thresholdLogLKoptimizationTopology:
                    best_nodes[kid] = sc
                go = False
                if ch[kid]:
                    within = sc > best - thresholdLogLKtopology
                    go = (fails_new <= allowedFailsTopology and within) if strictTopologyStopRules else \
                         (fails_new <= allowedFailsTopology or within)
                if go:
                    stack.append((kid, sc, fails_new))
        return best_node, best, best_nodes, visited
    finally:
        dev.set_fatal_policy(False)
        dev.release(mark)
