"""Placement search for a new sample -- the host counterpart of findBestParentForNewSample
(MAPLEv0.7.5.4.py:7912-8292) with every genome-list evaluation done on the GPU.

One query is searched at a time (the tree changes between samples), so the parallel axis is the
candidate set: the query is scored against EVERY branch of the tree in one launch (a superset of
what the reference's depth-first search visits), then the reference's LIFO traversal, stop rules,
tie-breaks and short-list refinement are replayed over the cached scores.  Results (node,
score, branch lengths, bestDiffs) are those of the reference; the number of appendProbNode
evaluations the *reference* would have issued is reported as ``n_append``.

``find_placement_supports`` is the computePlacementSupportOnly=True exit (M:8101-8290).
Not implemented (off by default, outside BASELINE's configs): HnZ, time trees, --deeperSearchForLongBranches.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .runtime import Device
from .tree_host import HostTree


@dataclass
class PlacementParams:
    oneMutBLen: float
    effectivelyNon0BLen: float
    thresholdLogLK: float                       # already x log(lRef), M:3613
    thresholdLogLKoptimization: float           # already x log(lRef), M:3610
    thresholdLogLKconsecutivePlacement: float   # M:63
    allowedFails: int = 5
    strictStopRules: bool = True
    onlyFindIdentical: bool = False             # M:7931 / 7976: any error-model flag, supportFor0Branches or HnZ


class _Diffs:
    """A query genome list as the reference sees it: one mutable object per (frame, creation)."""
    __slots__ = ("lid", "frame", "shortened")

    def __init__(self, lid, frame):
        self.lid, self.frame, self.shortened = int(lid), frame, False


class PlacementSearcher:
    """``rank`` / ``world`` (default: one GPU): level 2 of SURVEY section 8e -- the candidate branches and the leaves of
    ONE query are sharded over the GPUs (maple_amd.parallel.shard_candidates), each rank scores its shard in one launch
    and the score / minor-test vectors are all-gathered, after which every rank replays the same traversal."""

    def __init__(self, dev: Device, tree: HostTree, params: PlacementParams, rank: int = 0, world: int = 1, device=None):
        self.dev, self.tree, self.p = dev, tree, params
        self.rank, self.world, self.coll_device = rank, world, device
        t = tree
        n = t.n
        self.has_mut = np.asarray([bool(m) for m in t.mutations])
        # frame of a node = nearest ancestor-or-self carrying MAT mutations (-1 = the root frame)
        order = []
        stack = [t.root]
        while stack:
            v = stack.pop()
            order.append(v)
            stack.extend(t.children[v])
        self.frame = -np.ones(n, dtype=np.int64)
        self.frame_depth = {}
        for v in order:
            u = t.up[v]
            pf = -1 if u is None else self.frame[u]
            if self.has_mut[v]:
                self.frame[v] = v
                self.frame_depth[v] = 0 if pf < 0 else self.frame_depth[pf] + 1
            else:
                self.frame[v] = pf
        self.frame_parent = {f: (-1 if t.up[f] is None else int(self.frame[t.up[f]])) for f in self.frame_depth}
        levels = {}
        for f, d in self.frame_depth.items():
            levels.setdefault(d, []).append(f)
        self.frame_levels = [levels[d] for d in sorted(levels)]
        up = np.asarray([-1 if u is None else u for u in t.up])
        dist = np.asarray(t.dist)
        self.cand = np.nonzero((up >= 0) & (dist > params.effectivelyNon0BLen) & (t.id_totUp >= 0))[0]
        # scored in order of list length, so that the 64 lanes of a wavefront finish together
        self.cand = self.cand[np.argsort(dev.sizes(t.id_totUp[self.cand])[0], kind="stable")]
        self.leaves = np.asarray([v for v in order if not t.children[v]], dtype=np.int64)   # reachable ones: surgery leaves dead slots
        # rootVector(probVect[root], False, False, tree, root) does not depend on the query (M:7958)
        path = [t.id_mut[t.root]] if t.id_mut[t.root] >= 0 else []
        self.root_vect = int(dev.root_vector_batch([t.id_lower[t.root]], [0.0], [False], [path])[0])
        self.is_tip = np.asarray([(not c) and (m == 0) for c, m in zip(t.children, t.n_minor)])
        # resident candidate sets: frame index 0 is the root frame, then the frames in a fixed order
        self.frame_order = [-1] + [f for fl in self.frame_levels for f in fl]
        fidx = {f: i for i, f in enumerate(self.frame_order)}
        from .parallel import shard_candidates
        self.my_cand = self.cand[shard_candidates(len(self.cand), rank, world)]
        self.my_leaves = self.leaves[shard_candidates(len(self.leaves), rank, world)]
        cand_frames = [fidx[int(self.frame[v])] for v in self.my_cand] + [fidx[int(self.frame[t.root])]]
        self.cset_cand = dev.candset_create(np.concatenate([t.id_totUp[self.my_cand], [self.root_vect]]), cand_frames,
                                            len(self.frame_order))
        self.cset_leaf = None
        if len(self.my_leaves):
            self.cset_leaf = dev.candset_create(t.id_lower[self.my_leaves],
                                                [fidx[int(self.frame[v])] for v in self.my_leaves], len(self.frame_order))

    def close(self):
        """Free the resident candidate sets (one PlacementSearcher per tree state: the sequential placement loop makes many)."""
        for cs in (self.cset_cand, self.cset_leaf):
            if cs is not None:
                self.dev.candset_destroy(cs)
        self.cset_cand = self.cset_leaf = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------------------------------------
    def _frame_lists(self, q_id):
        """Query list in every frame (U), its shortened form (S) -- all on the device."""
        dev, t = self.dev, self.tree
        U = {-1: int(q_id)}
        for fl in self.frame_levels:
            src = [U[self.frame_parent[f]] for f in fl]
            out = dev.pass_branch_batch(src, [t.id_mut[f] for f in fl], False)
            for f, o in zip(fl, out):
                U[f] = int(o)
        keys = list(U)
        s_ids = dev.shorten_batch([U[k] for k in keys])
        ne_u, _ = dev.sizes([U[k] for k in keys])
        ne_s, _ = dev.sizes(s_ids)
        S = {k: (int(s) if a != b else U[k]) for k, s, a, b in zip(keys, s_ids, ne_u, ne_s)}
        return U, S

    def _prepare_native(self):
        if getattr(self, "_prepared", False) or not getattr(self.dev, "n_nodes", 0):
            return
        p = self.p
        self.dev.placement_prepare(
            oneMutBLen=p.oneMutBLen, effectivelyNon0BLen=p.effectivelyNon0BLen, thresholdLogLK=p.thresholdLogLK,
            thresholdLogLKoptimization=p.thresholdLogLKoptimization,
            thresholdLogLKconsecutivePlacement=p.thresholdLogLKconsecutivePlacement, allowedFails=p.allowedFails,
            strictStopRules=p.strictStopRules, onlyFindIdentical=p.onlyFindIdentical)
        self._prepared = True

    def find_best_parent_batch(self, diffs_list):
        """Many queries against the frozen tree in one native call (maple_placement_search_batch): scoring, the
        reference's traversal (on the device, one lane per query) and the short-list refinement are all batched.
        Returns a list of (bestNode, bestScore, bestBranchLengths, bestDiffs, info) like the single-query form."""
        dev, p = self.dev, self.p
        self._prepare_native()                                       # per-tree tables and root vector, before the mark
        mark = dev.mark()
        try:
            q_ids = dev.upload(list(diffs_list))
            out = dev.placement_search_batch(
                q_ids, oneMutBLen=p.oneMutBLen, effectivelyNon0BLen=p.effectivelyNon0BLen, thresholdLogLK=p.thresholdLogLK,
                thresholdLogLKoptimization=p.thresholdLogLKoptimization,
                thresholdLogLKconsecutivePlacement=p.thresholdLogLKconsecutivePlacement, allowedFails=p.allowedFails,
                strictStopRules=p.strictStopRules, onlyFindIdentical=p.onlyFindIdentical)
            if (out["status"] < 0).any():
                raise RuntimeError("placement search overflow (status %s)" % sorted(set(out["status"][out["status"] < 0])))
            lists = dev.download(out["bestDiffs"])
            res = []
            for k in range(len(q_ids)):
                minor = out["status"][k] == 1
                res.append((int(out["bestNode"][k]), float(out["bestScore"][k]),
                            None if minor else tuple(float(x) for x in out["blen"][k]), lists[k],
                            dict(n_append=int(out["nAppend"][k]), minor=bool(minor))))
            return res
        finally:
            dev.release(mark)

    def find_placement_supports(self, diffs_list, thresholdLogLKoptimizationTopology, minBranchSupport=0.01):
        """findBestParentForNewSample(tree, root, diffs, sample, computePlacementSupportOnly=True) for many samples on the
        frozen tree (the call of process_chunk, M:11200): per sample (possiblePlacements, bestPlacementTotalLh) as the
        reference returns them (M:8264-8290); possiblePlacements = [(node, support, (top, bottom, appending)), ...]."""
        dev, p = self.dev, self.p
        self._prepare_native()
        mark = dev.mark()
        try:
            q_ids = dev.upload(list(diffs_list))
            res, status = dev.placement_supports_batch(
                q_ids, oneMutBLen=p.oneMutBLen, effectivelyNon0BLen=p.effectivelyNon0BLen, thresholdLogLK=p.thresholdLogLK,
                thresholdLogLKoptimization=p.thresholdLogLKoptimization,
                thresholdLogLKconsecutivePlacement=p.thresholdLogLKconsecutivePlacement,
                thresholdLogLKoptimizationTopology=thresholdLogLKoptimizationTopology, minBranchSupport=minBranchSupport,
                allowedFails=p.allowedFails, strictStopRules=p.strictStopRules, onlyFindIdentical=p.onlyFindIdentical)
            if (status < 0).any():
                raise RuntimeError("placement search overflow (status %s)" % sorted(set(status[status < 0])))
            ids = [b for _, b in res]
            lists = dev.download(ids)
            return [(pl, [] if lst is None else lst) for (pl, _), lst in zip(res, lists)]
        finally:
            dev.release(mark)

    def find_best_parent_for_new_sample(self, diffs, sample=None):
        """Returns (bestNode, bestScore, bestBranchLengths, bestDiffs, info); the first four as the reference.
        On one GPU this is one native call (maple_placement_search_batch with a single query: scoring on the GPU, the
        traversal on the host for so small a batch); with the candidates sharded over several GPUs (world > 1) the
        traversal is the Python replay below, after the all-gather of the score shards."""
        if self.world == 1:
            return self.find_best_parent_batch([diffs])[0]
        return self.find_best_parent_host_replay(diffs, sample)

    def find_best_parent_host_replay(self, diffs, sample=None):
        """The same search with the reference's traversal replayed in Python over the all-branch scores (the form the
        multi-GPU candidate sharding uses; also an independent implementation for the tests)."""
        dev = self.dev
        mark = dev.mark()
        try:
            return self._search(diffs, sample)
        finally:
            dev.release(mark)

    def _search(self, diffs, sample):
        dev, t, p = self.dev, self.tree, self.p
        q_id = dev.upload([diffs])[0]
        U, S = self._frame_lists(q_id)
        fr = self.frame
        # one launch: the query against every candidate branch (+ the root), one launch: minor test on every leaf
        cand = self.cand
        frame_lists = [U[f] for f in self.frame_order]
        from .parallel import allgather_interleaved
        sc = dev.append_candset(self.cset_cand, frame_lists, True, p.oneMutBLen)     # this rank's shard (+ the root)
        score = dict(zip(cand.tolist(), allgather_interleaved(sc[:-1], len(cand), self.coll_device).tolist()))
        minor = {}
        if len(self.leaves):
            mres = (dev.minor_candset(self.cset_leaf, frame_lists, p.onlyFindIdentical) if self.cset_leaf is not None
                    else np.zeros(0, dtype=np.uint8))
            minor = dict(zip(self.leaves.tolist(), allgather_interleaved(mres, len(self.leaves), self.coll_device).tolist()))
        n_append = 1

        def shorten(obj):
            if obj.shortened:
                return
            if obj.lid == U[obj.frame]:
                obj.lid = S[obj.frame]
            else:
                obj.lid = int(dev.shorten_batch([obj.lid])[0])
            obj.shortened = True

        def pass_to(obj, c):
            if not self.has_mut[c]:
                return obj
            if obj.lid == U[obj.frame]:
                return _Diffs(U[c], c)
            return _Diffs(dev.pass_branch_batch([obj.lid], [t.id_mut[c]], False)[0], c)

        root = t.root
        d0 = _Diffs(U[int(fr[root])], int(fr[root]))
        best_nodes = []
        best_node = root
        best_blens = (False, False, p.oneMutBLen)
        best_diffs = d0
        missed = 0
        if not t.children[root]:
            if minor.get(root) == 1:
                return root, 1.0, None, dev.download([d0.lid])[0], dict(n_append=0, minor=True)
        best_lk = float(sc[-1])
        original_lk = best_lk
        stack = [(c, best_lk, 0, pass_to(d0, c)) for c in t.children[root]]
        while stack:                                                     # M:7972-8100
            t1, parent_lk, fails, dobj = stack.pop()
            if not t.children[t1]:
                cmpv = minor[t1]
                if cmpv == 1:
                    return t1, 1.0, None, dev.download([dobj.lid])[0], dict(n_append=n_append, minor=True)
                if cmpv == 2:
                    missed += 1
            if t1 in score:                                              # dist > effectivelyNon0BLen and up != None
                lk = score[t1]
                n_append += 1
                half = t.dist[t1] / 2
                if lk >= best_lk:                                        # M:8065-8073
                    shorten(dobj)
                    best_lk = lk
                    best_node = t1
                    fails = 0
                    best_nodes.append((t1, lk, dobj))
                    best_diffs = dobj
                    best_blens = (half, half / 2, p.oneMutBLen)
                elif lk > best_lk - p.thresholdLogLKoptimization:
                    best_nodes.append((t1, lk, dobj))
                if lk < (parent_lk - p.thresholdLogLKconsecutivePlacement):
                    fails += 1
            else:
                lk = parent_lk
            if p.strictStopRules:
                go = fails <= p.allowedFails and lk > (best_lk - p.thresholdLogLK)
            else:
                go = fails <= p.allowedFails or lk > (best_lk - p.thresholdLogLK)
            if go:
                for c in t.children[t1]:
                    stack.append((c, lk, fails, pass_to(dobj, c)))
        # short-list refinement, M:8101-8187
        best_score = best_lk
        todo = [(nd, s, d) for nd, s, d in best_nodes if s >= best_lk - p.thresholdLogLKoptimization]
        if todo:
            nodes = [nd for nd, _, _ in todo]
            up_ids = []
            for nd in nodes:
                u = t.up[nd]
                uid = t.id_upRight[u] if t.children[u][0] == nd else t.id_upLeft[u]
                up_ids.append(int(uid))
            need = [i for i, nd in enumerate(nodes) if self.has_mut[nd]]
            if need:
                passed = dev.pass_branch_batch([up_ids[i] for i in need], [t.id_mut[nodes[i]] for i in need], False)
                for i, pid in zip(need, passed):
                    up_ids[i] = int(pid)
            q_ids = [d.lid for _, _, d in todo]
            tips = [bool(self.is_tip[nd]) for nd in nodes]
            dists = [t.dist[nd] for nd in nodes]
            ev = dev.evaluate_placement_batch(t.id_totUp[nodes], t.id_lower[nodes], up_ids, dists, q_ids, True, tips)
            k = len(nodes)
            comp = dev.append_batch(up_ids + up_ids, list(t.id_lower[nodes]) * 2, tips + tips,
                                    dists + [float(ev[i, 1] + ev[i, 2]) for i in range(k)])
            n_append += 3 * k
            for i, (nd, _, dobj) in enumerate(todo):
                cost, bottom, top, app = (float(x) for x in ev[i])
                optimized = cost + float(comp[k + i]) - float(comp[i])
                if optimized >= best_score:
                    best_node, best_score = nd, optimized
                    best_blens = (top, bottom, app)
                    best_diffs = dobj
        if best_score == float("-inf"):
            best_score = original_lk
        return (best_node, best_score, best_blens, dev.download([best_diffs.lid])[0],
                dict(n_append=n_append, minor=False, missed_minors=missed, candidates_scored=len(cand) + 1))
