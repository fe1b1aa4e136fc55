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
