"""Local references (MAT frames) for a tree built without them -- what the reference's setUpMAT does when it reads an
input tree (MAPLEv0.7.5.4.py:4148-4290, rule at M:6152-6164): a node with at least ``max_desc`` positive-length
descendant branches below it (counted since the last such node) becomes a reference node; its branch carries the
mutation list (position, nucleotide in the parent frame, nucleotide in the new frame) of every site where its lower
genome list shows a definite non-reference nucleotide, and every list in its clade is expressed against the new
reference.  Used to put the synthetic benchmark trees into the form real MAPLE trees have (local references are on by
default, M:166); the lists themselves are rebuilt on the GPU (tree_host.rebuild_genome_lists)."""
from __future__ import annotations

import numpy as np

from .runtime import Device
from .tree_host import HostTree, rebuild_genome_lists


def add_local_references(dev: Device, tree: HostTree, max_desc: int = 50):
    """``tree``: a HostTree WITHOUT mutations whose lower lists (id_lower) are all in the root frame.  Chooses the
    reference nodes, uploads their mutation lists, re-expresses the tips and rebuilds all four lists of every node.
    Returns the number of reference nodes."""
    n = tree.n
    order = tree.preorder()
    # nDesc as the reference accumulates it bottom-up (M:6152-6164)
    n_desc = np.zeros(n, dtype=np.int64)
    is_ref = np.zeros(n, dtype=bool)
    for v in reversed(order):
        ch = tree.children[v]
        if not ch:
            continue
        for c in ch:
            if tree.children[c] and not is_ref[c]:
                n_desc[v] += n_desc[c]
            if tree.dist[c]:
                n_desc[v] += 1
        if n_desc[v] >= max_desc and tree.dist[v] and v != tree.root:
            n_desc[v] = 0
            is_ref[v] = True
    frames = [v for v in order if is_ref[v]]            # pre-order: an outer frame always comes before its inner ones
    tree.id_mut = -np.ones(n, dtype=np.int32)
    tree.mutations = [[] for _ in range(n)]
    # the path of frame nodes above each node
    frame_of = {}
    for v in order:
        u = tree.up[v]
        above = frame_of[u] if u is not None else ()
        frame_of[v] = above + (v,) if is_ref[v] else above
    for v in frames:
        lid = int(tree.id_lower[v])
        for f in frame_of[v][:-1]:                      # into the frame of v's parent
            lid = int(dev.pass_branch_batch([lid], [tree.id_mut[f]], False)[0])
        pos = 0
        muts = []
        for e in dev.download([lid])[0]:
            if e[0] in (4, 5):
                pos = e[1]
            else:
                pos += 1
                if e[0] < 4:
                    muts.append((pos, int(e[1]), int(e[0])))
        if not muts:
            is_ref[v] = False
            frame_of_v = frame_of[v]
            for w in order:                              # rare: nothing to re-reference
                if frame_of[w][: len(frame_of_v)] == frame_of_v:
                    frame_of[w] = frame_of[w][: len(frame_of_v) - 1] + frame_of[w][len(frame_of_v):]
            continue
        tree.mutations[v] = muts
        tree.id_mut[v] = dev.upload_mutations([muts])[0]
    # tips into their local frames, level by level of frame nesting
    tips = [v for v in order if not tree.children[v]]
    cur = {v: int(tree.id_lower[v]) for v in tips}
    depth = max((len(frame_of[v]) for v in tips), default=0)
    for d in range(depth):
        todo = [v for v in tips if len(frame_of[v]) > d]
        if todo:
            out = dev.pass_branch_batch([cur[v] for v in todo], [tree.id_mut[frame_of[v][d]] for v in todo], False)
            for v, o in zip(todo, out):
                cur[v] = int(o)
    shortened = dev.shorten_batch([cur[v] for v in tips])
    for v, o in zip(tips, shortened):
        tree.id_lower[v] = o
    lower, up_right, up_left, tot_up = rebuild_genome_lists(dev, tree)
    tree.id_lower, tree.id_upRight, tree.id_upLeft, tree.id_totUp = lower, up_right, up_left, tot_up
    return int(is_ref.sum())
