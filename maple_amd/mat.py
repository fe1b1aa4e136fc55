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


def add_local_references_one_by_one(dev: Device, tree: HostTree, max_desc: int = 50):
    """``tree``: a HostTree WITHOUT mutations whose lower lists (id_lower) are all in the root frame.  Chooses the
    reference nodes, uploads their mutation lists, re-expresses the tips and rebuilds all four lists of every node.
    Returns the number of reference nodes.  (The first form of add_local_references, one reference node at a time: kept as
    the check of the batched one below.)"""
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


def add_local_references(dev: Device, tree: HostTree, max_desc: int = 50):
    """The same as add_local_references_one_by_one -- same reference nodes, mutation lists and genome lists (tested) -- with
    every step batched: the descendant counts level by level, the reference nodes' own lists through the frames that enclose
    them one nesting depth at a time, the tips likewise.  (At 1 000 000 tips the tree has ~20 000 reference nodes.)"""
    n = tree.n
    up = np.asarray([-1 if u is None else u for u in tree.up], dtype=np.int64)
    c0 = np.asarray([c[0] if c else -1 for c in tree.children], dtype=np.int64)
    c1 = np.asarray([c[1] if c else -1 for c in tree.children], dtype=np.int64)
    dist = np.asarray(tree.dist, dtype=np.float64)
    root = int(tree.root)
    # depth of every node reachable from the root, level by level
    depth = -np.ones(n, dtype=np.int64)
    levels = []
    lev = np.asarray([root], dtype=np.int64)
    while len(lev):
        depth[lev] = len(levels)
        levels.append(lev)
        ch = np.concatenate([c0[lev], c1[lev]])
        lev = ch[ch >= 0]
    # nDesc as the reference accumulates it bottom-up (M:6152-6164), deepest level first
    n_desc = np.zeros(n, dtype=np.int64)
    is_ref = np.zeros(n, dtype=bool)
    for lev in reversed(levels):
        inner = lev[c0[lev] >= 0]
        if len(inner) == 0:
            continue
        add = np.zeros(len(inner), dtype=np.int64)
        for ch in (c0[inner], c1[inner]):
            add += np.where((c0[ch] >= 0) & ~is_ref[ch], n_desc[ch], 0) + (dist[ch] != 0)
        n_desc[inner] = add
        ref = (add >= max_desc) & (dist[inner] != 0) & (inner != root)
        n_desc[inner[ref]] = 0
        is_ref[inner[ref]] = True
    # the innermost frame every node lies in (the reference node itself included), -1 = the root's frame; nesting depth of frames
    frame_node = -np.ones(n, dtype=np.int64)
    for lev in levels[1:]:
        frame_node[lev] = np.where(is_ref[lev], lev, frame_node[up[lev]])
    tree.id_mut = -np.ones(n, dtype=np.int32)
    tree.mutations = [[] for _ in range(n)]
    frames = np.nonzero(is_ref)[0]
    frames = frames[np.argsort(depth[frames], kind="stable")]          # outer frames before the ones nested in them
    outer = np.where(frames >= 0, frame_node[up[frames]], -1)           # the frame that encloses each frame (-1: the root's)
    outer_of = dict(zip(frames.tolist(), outer.tolist()))
    nest = {}
    for f in frames.tolist():                                           # (frames are few: ~1 per 100 nodes)
        nest[f] = 0 if outer_of[f] < 0 else nest[outer_of[f]] + 1
    nest_arr = np.asarray([nest[f] for f in frames.tolist()], dtype=np.int64)
    dropped = set()

    def chain_of(nodes_frame, k):
        """for every entry of nodes_frame (a frame node or -1) the frame at nesting depth k on its chain, or -1"""
        out = np.asarray(nodes_frame, dtype=np.int64).copy()
        for i, f in enumerate(out.tolist()):
            while f >= 0 and nest[f] > k:
                f = outer_of[f]
            out[i] = f if (f >= 0 and nest[f] == k) else -1
        return out

    max_nest = int(nest_arr.max()) if len(frames) else -1
    for k in range(max_nest + 1):
        cur_frames = frames[nest_arr == k]
        lids = np.asarray(tree.id_lower[cur_frames], dtype=np.int32).copy()
        encl = outer[nest_arr == k]                                    # into the frame of each reference node's parent
        for j in range(k):                                              # through the enclosing frames, outermost first
            fj = chain_of(encl, j)
            need = np.nonzero((fj >= 0) & (tree.id_mut[np.maximum(fj, 0)] >= 0))[0]
            if len(need):
                lids[need] = dev.pass_branch_batch(lids[need], tree.id_mut[fj[need]], False)
        mut_lists = []
        for v, gl in zip(cur_frames.tolist(), dev.download(lids)):
            pos = 0
            muts = []
            for e in gl:
                if e[0] in (4, 5):
                    pos = e[1]
                else:
                    pos += 1
                    if e[0] < 4:
                        muts.append((pos, int(e[1]), int(e[0])))
            mut_lists.append(muts)
        keep = [i for i, m in enumerate(mut_lists) if m]
        if keep:
            ids = dev.upload_mutations([mut_lists[i] for i in keep])
            for i, mid in zip(keep, ids):
                v = int(cur_frames[i])
                tree.mutations[v] = mut_lists[i]
                tree.id_mut[v] = mid
        for i, m in enumerate(mut_lists):                                # rare: nothing to re-reference (a pass through it is the identity)
            if not m:
                is_ref[int(cur_frames[i])] = False
                dropped.add(int(cur_frames[i]))
    # tips into their local frames, one nesting depth at a time (outermost frame first)
    tips = np.nonzero((depth >= 0) & (c0 < 0))[0]
    cur = np.asarray(tree.id_lower[tips], dtype=np.int32).copy()
    tip_frame = frame_node[tips]
    if len(frames):
        uniq, inv = np.unique(tip_frame, return_inverse=True)
        for k in range(max_nest + 1):
            fk = chain_of(uniq, k)[inv]
            need = np.nonzero((fk >= 0) & (tree.id_mut[np.maximum(fk, 0)] >= 0))[0]
            if len(need):
                cur[need] = dev.pass_branch_batch(cur[need], tree.id_mut[fk[need]], False)
    tree.id_lower[tips] = dev.shorten_batch(cur)
    lower, up_right, up_left, tot_up = rebuild_genome_lists(dev, tree)
    tree.id_lower, tree.id_upRight, tree.id_upLeft, tree.id_totUp = lower, up_right, up_left, tot_up
    return int(is_ref.sum())
