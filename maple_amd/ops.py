"""Host-side mirror of the reference's operator interface for the placement path.

Same names, argument meaning and return conventions as the top-level functions
of MAPLEv0.7.5.4.py that the search loops call:
``None`` for an impossible merge (M:4758), ``-inf`` for an impossible
attachment (M:6663), ``False`` for a zero branch length (M:5301).  Every call
goes through the C ABI into the HIP kernels; the ``*_many`` forms are the
batched shape the searches actually use (thousands of candidates per launch).
"""
from __future__ import annotations

import numpy as np

from .runtime import Device


class GenomeOps:
    def __init__(self, dev: Device):
        self.dev = dev

    # ---- scoring --------------------------------------------------------------------------
    def appendProbNode(self, probVectP, probVectC, isTipC, bLen):
        """M:6505-6785."""
        return self.appendProbNode_many([probVectP], [probVectC], [isTipC], [bLen])[0]

    def appendProbNode_many(self, parents, children, isTipC, bLen):
        d = self.dev
        mark = d.mark()
        try:
            ids = d.upload(list(parents) + list(children))
            n = len(parents)
            out = d.append_batch(ids[:n], ids[n:], np.asarray(isTipC, dtype=bool), np.asarray(bLen, dtype=float))
        finally:
            d.release(mark)
        return [float(x) for x in out]

    # ---- merging ----------------------------------------------------------------------------
    def mergeVectors(self, probVect1, bLen1, fromTip1, probVect2, bLen2, fromTip2, returnLK=False, isUpDown=False,
                     numMinor1=0, numMinor2=0):
        """M:4446-4859 (isUpDown=True is the up-down merge)."""
        d = self.dev
        mark = d.mark()
        try:
            ids = d.upload([probVect1, probVect2])
            res = d.merge_batch(ids[:1], [bLen1 or 0.0], [fromTip1], ids[1:], [bLen2 or 0.0], [fromTip2], [isUpDown],
                                returnLK=returnLK, numMinor1=[numMinor1], numMinor2=[numMinor2])
            if returnLK:
                out, lk = res
                return d.download(out)[0], float(lk[0])
            return d.download(res)[0]
        finally:
            d.release(mark)

    def estimateBranchLengthWithDerivative(self, probVectP, probVectC, fromTipC=False):
        """M:5040-5358."""
        d = self.dev
        mark = d.mark()
        try:
            ids = d.upload([probVectP, probVectC])
            t, f = d.blen_batch(ids[:1], ids[1:], [fromTipC])
        finally:
            d.release(mark)
        return False if f[0] else float(t[0])

    def areVectorsDifferent(self, probVect1, probVect2):
        """M:5419-5472."""
        if probVect2 is None:
            return True
        d = self.dev
        mark = d.mark()
        try:
            ids = d.upload([probVect1, probVect2])
            return bool(d.differ_batch(ids[:1], ids[1:])[0])
        finally:
            d.release(mark)

    def passGenomeListThroughBranch(self, probVect, mutations, dirIsUp=False):
        """M:3749-3877."""
        d = self.dev
        mark = d.mark()
        try:
            ids = d.upload([probVect])
            mids = d.upload_mutations([mutations])
            return d.download(d.pass_branch_batch(ids, mids, [dirIsUp]))[0]
        finally:
            d.release(mark)

    def shorten(self, vec):
        """M:3721-3745.  Returns the shortened list (the reference edits in place)."""
        d = self.dev
        mark = d.mark()
        try:
            return d.download(d.shorten_batch(d.upload([vec])))[0]
        finally:
            d.release(mark)

    def rootVector(self, probVect, bLen, isFromTip, pathMutations):
        """M:4916-4996; ``pathMutations`` = tree.mutations[] on the walk node -> root (node first)."""
        d = self.dev
        mark = d.mark()
        try:
            ids = d.upload([probVect])
            mids = d.upload_mutations(pathMutations)
            return d.download(d.root_vector_batch(ids, [bLen or 0.0], [isFromTip], [mids]))[0]
        finally:
            d.release(mark)

    def evaluatePlacement(self, midTot, downVect, upVect, distance, removedPartials, isRemovedTip, fromTip1):
        """M:6790-6806 -> (appendingCost, bestBottomLength, bestTopLength, bestAppendingLength)."""
        d = self.dev
        mark = d.mark()
        try:
            ids = d.upload([midTot, downVect, upVect, removedPartials])
            out = d.evaluate_placement_batch(ids[0:1], ids[1:2], ids[2:3], [distance], ids[3:4], [isRemovedTip],
                                             [fromTip1])
        finally:
            d.release(mark)
        return tuple(float(x) for x in out[0])


class TreeOps:
    """The tree-level functions of the path under the reference's own names, on a ``HostTree`` whose lists live in the
    device arena (``tree.upload(dev)`` or ``HostTree.from_mirror``).  Thin delegation: the work is in
    ``maple_spr_search_batch`` / ``maple_placement_search_batch`` and the level-synchronous batches of ``tree_host``."""

    def __init__(self, dev: Device, tree):
        self.dev, self.tree = dev, tree
        self._searcher = None
        self._searcher_key = None

    def upload_topology(self):
        """(Re)send the topology and list ids after the host changed the tree (maple_tree_upload)."""
        self.tree.upload_topology(self.dev)
        self._searcher = None

    # M:7912 -- returns (bestNode, bestScore, bestBranchLengths, bestDiffs) like the reference; with
    # computePlacementSupportOnly=True (possiblePlacements, bestPlacementTotalLh), M:8264-8290
    def findBestParentForNewSample(self, diffs, *, oneMutBLen, effectivelyNon0BLen, thresholdLogLK, thresholdLogLKoptimization,
                                   thresholdLogLKconsecutivePlacement, allowedFails=5, strictStopRules=True,
                                   onlyFindIdentical=False, computePlacementSupportOnly=False,
                                   thresholdLogLKoptimizationTopology=None, minBranchSupport=0.01):
        from .search import PlacementParams, PlacementSearcher
        key = (oneMutBLen, effectivelyNon0BLen, thresholdLogLK, thresholdLogLKoptimization,
               thresholdLogLKconsecutivePlacement, allowedFails, strictStopRules, onlyFindIdentical)
        if self._searcher is None or self._searcher_key != key:
            self._searcher = PlacementSearcher(self.dev, self.tree, PlacementParams(
                oneMutBLen=oneMutBLen, effectivelyNon0BLen=effectivelyNon0BLen, thresholdLogLK=thresholdLogLK,
                thresholdLogLKoptimization=thresholdLogLKoptimization,
                thresholdLogLKconsecutivePlacement=thresholdLogLKconsecutivePlacement, allowedFails=allowedFails,
                strictStopRules=strictStopRules, onlyFindIdentical=onlyFindIdentical))
            self._searcher_key = key
        if computePlacementSupportOnly:
            thr = thresholdLogLKoptimization if thresholdLogLKoptimizationTopology is None else thresholdLogLKoptimizationTopology
            return self._searcher.find_placement_supports([diffs], thr, minBranchSupport)[0]
        return self._searcher.find_best_parent_for_new_sample(diffs)[:4]

    # M:9580 -- the worker body for `nodes`; returns the list of (node, placement, improvement) the reference's worker returns
    def startTopologyUpdatesParallel(self, nodes, *, strictTopologyStopRules, allowedFailsTopology, thresholdLogLKtopology,
                                     thresholdTopologyPlacement, thresholdLogLKoptimizationTopology,
                                     thresholdLogLKconsecutivePlacement, effectivelyNon0BLen):
        res = self.dev.spr_search_batch(nodes, strict=strictTopologyStopRules, allowedFails=allowedFailsTopology,
                                        thresholdLogLKtopology=thresholdLogLKtopology,
                                        thresholdTopologyPlacement=thresholdTopologyPlacement,
                                        thresholdLogLKoptimizationTopology=thresholdLogLKoptimizationTopology,
                                        thresholdLogLKconsecutivePlacement=thresholdLogLKconsecutivePlacement,
                                        effectivelyNon0BLen=effectivelyNon0BLen)
        bad = res["status"][res["status"] < -1]
        if len(bad):      # -1 is the reference's swallowed exception (M:9703); anything below is a capacity problem, not "no move"
            raise RuntimeError(f"SPR search could not finish some queries (status {sorted(set(bad.tolist()))}: workspace, "
                               "removed-partials pool or short-list capacity)")
        return [(int(n), int(p), float(i)) for n, p, i in zip(nodes, res["placement"], res["improvement"]) if p >= 0]

    def reCalculateAllGenomeLists(self):                                   # M:6013
        from .tree_host import rebuild_genome_lists
        t = self.tree
        t.id_lower, t.id_upRight, t.id_upLeft, t.id_totUp = rebuild_genome_lists(self.dev, t)

    def sync(self):
        """After a LOCAL change (one placed sample + updatePartials): only the touched nodes go to the library
        (maple_tree_patch); falls back to upload_topology when that is not possible.  Returns the number of patched nodes."""
        n = self.tree.sync(self.dev)
        if self._searcher is not None:
            if n == -1 or self._searcher.world != 1:
                self._searcher = None                                  # (its candidate shards describe the old tree)
            else:
                self._searcher._prepared = False                       # the root vector may have changed: prepare again
        return n

    def updatePartials(self, changed_nodes):                               # M:5479 (level-synchronous, any number of changes)
        from .tree_host import update_genome_lists
        return update_genome_lists(self.dev, self.tree, list(changed_nodes))

    def calculateTreeLikelihood(self):                                     # M:9721
        from .tree_host import tree_log_likelihood
        return tree_log_likelihood(self.dev, self.tree)[0]

    def traverseTreeToOptimizeBranchLengths(self, effectivelyNon0BLen, fastPass=False):  # M:8727 (default: the Gauss-Seidel sweep)
        from .tree_host import optimize_branch_lengths, optimize_branch_lengths_fast_pass
        if fastPass:
            return optimize_branch_lengths_fast_pass(self.dev, self.tree, effectivelyNon0BLen)[0]
        return optimize_branch_lengths(self.dev, self.tree, effectivelyNon0BLen)[0]

    def findBestRoot(self, **kw):                                          # M:7730 (the search; re-rooting stays with the caller)
        from .tree_host import find_best_root
        return find_best_root(self.dev, self.tree, **kw)
