// maple_amd/csrc/rebuild_host.h -- reCalculateAllGenomeLists (MAPLEv0.7.5.4.py:6013-6347) level by level inside the library.
//
// Given the tips' lower lists, every internal node's probVect (pass 1, M:6031-6200: deepest level first) and every node's
// probVectUpRight / probVectUpLeft / probVectTotUp (pass 2, M:6226-6345: from the root down), each level ONE fused launch of
// k_update_items (mergeVectors -> shorten, update_host.h) on the caller's own tree columns: no Python in the loop, one
// synchronisation per level (the level's new list ids feed the next one).  Lists that cross a branch with MAT mutations go
// through passGenomeListThroughBranch first (upd_passed).  Included by maple_hip.hip after update_host.h.
#pragma once

// bumpLen = 0: a None between two zero-length branches is fatal, as in the reference (M:6087 / 6279: it would call updateBLen);
// bumpLen > 0 (building a synthetic tree): in pass 1 the two child branches are lengthened to bumpLen and merged again; a None
// that remains, or one in pass 2, is reported in badNodes (the caller lengthens those branches and starts over): *nBad > 0.
extern "C" int maple_tree_rebuild_lists(maple_ctx *c, int32_t n, int32_t root, const int32_t *up, const int32_t *c0, const int32_t *c1,
                                        const uint8_t *tip, const int32_t *mut, double *dist, int32_t *lower, int32_t *upRight,
                                        int32_t *upLeft, int32_t *totUp, double bumpLen, int32_t capBad, int32_t *badNodes, int32_t *nBad)
{
    if (!c || n <= 0 || !up || !c0 || !c1 || !tip || !dist || !lower || !upRight || !upLeft || !totUp) return MAPLE_ERR_ARG;
    if (root < 0 || root >= n) return fail(c, MAPLE_ERR_ARG, "root %d is not a node", root);
    if (bumpLen > 0.0 && (!badNodes || !nBad || capBad < 4)) return MAPLE_ERR_ARG;
    TRY(need_model(c));
    if (nBad) *nBad = 0;
    std::vector<int32_t> noMut;
    if (!mut) { noMut.assign((size_t)n, -1); mut = noMut.data(); }
    // the nodes reachable from the root, by depth
    std::vector<int32_t> depth((size_t)n, -1), order;
    order.reserve((size_t)n);
    order.push_back(root);
    depth[root] = 0;
    int maxd = 0;
    for (size_t h = 0; h < order.size(); h++) {
        const int v = order[h];
        for (int k = 0; k < 2; k++) {
            const int ch = k == 0 ? c0[v] : c1[v];
            if (ch < 0) continue;
            if (ch >= n || depth[ch] >= 0) return fail(c, MAPLE_ERR_ARG, "node %d: child %d out of range or reached twice", v, ch);
            if (up[ch] != v) return fail(c, MAPLE_ERR_ARG, "node %d is a child of %d but its `up` entry says %d", ch, v, up[ch]);   // (pass 2 reads `up`)
            if ((c0[v] < 0) != (c1[v] < 0)) return fail(c, MAPLE_ERR_ARG, "node %d has one child", v);
            depth[ch] = depth[v] + 1;
            maxd = std::max(maxd, depth[ch]);
            order.push_back(ch);
        }
    }
    std::vector<std::vector<int32_t>> byDepth((size_t)maxd + 1);
    for (int v : order) byDepth[depth[v]].push_back(v);                    // (breadth-first: ascending depth; sorted below)
    for (auto &lv : byDepth) std::sort(lv.begin(), lv.end());
    for (int v : order) {
        if (c0[v] >= 0) lower[v] = -1;
        else if (lower[v] < 0) return fail(c, MAPLE_ERR_ARG, "tip %d has no lower list", v);
        upRight[v] = upLeft[v] = totUp[v] = -1;
    }
    // (*nBad counts every bad node, also those beyond capBad: a caller that sees *nBad > capBad knows the list is cut short)
    auto bad = [&](int v) { if (nBad) { if (*nBad < capBad) badNodes[*nBad] = v; (*nBad)++; } };
    std::vector<int32_t> nodes, a, b, pa, pb, out, old;
    std::vector<double> da, db;
    std::vector<uint8_t> ta, tb, ud, mode, none, diff;
    // ---- pass 1: lower lists, deepest level first ------------------------------------------------------------------------------
    for (int d = maxd; d >= 0; d--) {
        nodes.clear();
        for (int v : byDepth[d]) if (c0[v] >= 0) nodes.push_back(v);
        const size_t m = nodes.size();
        if (!m) continue;
        a.resize(m); b.resize(m); pa.resize(m); pb.resize(m); da.resize(m); db.resize(m); ta.resize(m); tb.resize(m);
        for (size_t k = 0; k < m; k++) {
            a[k] = c0[nodes[k]]; b[k] = c1[nodes[k]];
            pa[k] = lower[a[k]]; pb[k] = lower[b[k]];
            da[k] = dist[a[k]]; db[k] = dist[b[k]]; ta[k] = tip[a[k]]; tb[k] = tip[b[k]];
        }
        TRY(upd_passed(c, mut, pa, a, true));
        TRY(upd_passed(c, mut, pb, b, true));
        ud.assign(m, 0); mode.assign(m, 1); old.assign(m, -1); out.assign(m, -1); none.assign(m, 0); diff.assign(m, 0);
        TRY(update_items(c, (int32_t)m, pa.data(), da.data(), ta.data(), pb.data(), db.data(), tb.data(), ud.data(), mode.data(), old.data(),
                         out.data(), none.data(), diff.data()));
        for (size_t k = 0; k < m; k++) {
            if (!none[k]) continue;
            if (bumpLen <= 0.0)
                return fail(c, MAPLE_ERR_FATAL, "inconsistent lower lists at node %d (the reference would call updateBLen here)", nodes[k]);
            // (a child's branch length does not enter the child's own lower list: lengthen the two branches, merge this pair again)
            dist[a[k]] = std::max(dist[a[k]], bumpLen); dist[b[k]] = std::max(dist[b[k]], bumpLen);
            da[k] = dist[a[k]]; db[k] = dist[b[k]];
            const uint8_t z = 0, one = 1;
            const int32_t o = -1;
            TRY(update_items(c, 1, &pa[k], &da[k], &ta[k], &pb[k], &db[k], &tb[k], &z, &one, &o, &out[k], &none[k], &diff[k]));
            if (none[k]) { bad(a[k]); bad(b[k]); }
        }
        if (nBad && *nBad) return MAPLE_OK;
        for (size_t k = 0; k < m; k++) lower[nodes[k]] = out[k];
    }
    // ---- pass 2: upper lists from the root down ---------------------------------------------------------------------------------
    if (c0[root] >= 0) {
        const int r = root;
        std::vector<int32_t> kids{c1[r], c0[r]}, pl{lower[c1[r]], lower[c0[r]]};
        TRY(upd_passed(c, mut, pl, kids, true));
        const int np = mut[r] >= 0 ? 1 : 0;
        const int64_t pathOff[3] = {0, np, 2 * np};
        const int32_t pathMut[2] = {mut[r] >= 0 ? mut[r] : 0, mut[r] >= 0 ? mut[r] : 0};
        const double bl[2] = {dist[c1[r]], dist[c0[r]]};
        const uint8_t tp[2] = {tip[c1[r]], tip[c0[r]]};
        int32_t rv[2] = {-1, -1};
        TRY(maple_root_vector_batch(c, 2, pl.data(), bl, tp, pathOff, pathMut, rv));
        upRight[r] = rv[0]; upLeft[r] = rv[1];
    }
    std::vector<int32_t> vu, iL1, iL2, iNode, iKid;
    std::vector<double> iB1, iB2;
    std::vector<uint8_t> iT1, iT2, iKind;
    for (int d = 1; d <= maxd; d++) {
        const std::vector<int32_t> &lv = byDepth[d];
        if (lv.empty()) continue;
        vu.resize(lv.size());
        for (size_t k = 0; k < lv.size(); k++) { const int p = up[lv[k]]; vu[k] = c0[p] == lv[k] ? upRight[p] : upLeft[p]; }
        TRY(upd_passed(c, mut, vu, lv, false));
        iL1.clear(); iL2.clear(); iNode.clear(); iKid.clear(); iB1.clear(); iB2.clear(); iT2.clear(); iKind.clear();
        // (kind by kind: the new lists take their room in item order, so the level's probVectTotUp lists -- the candidate lists of
        // every search -- end up next to each other in the arena, and a download of one kind moves in long runs)
        for (size_t k = 0; k < lv.size(); k++) {
            const int v = lv[k];
            if (dist[v] != 0.0) {                                           // probVectTotUp, M:6262-6275
                iL1.push_back(vu[k]); iB1.push_back(dist[v] / 2); iL2.push_back(lower[v]); iB2.push_back(dist[v] / 2); iT2.push_back(tip[v]);
                iNode.push_back(v); iKid.push_back(-1); iKind.push_back(0);
            }
        }
        for (int which = 1; which >= 0; which--)                            // probVectUpRight (child 1), then probVectUpLeft (child 0)
            for (size_t k = 0; k < lv.size(); k++) {
                const int v = lv[k];
                if (c0[v] < 0) continue;
                const int kd = which == 1 ? c1[v] : c0[v];
                iL1.push_back(vu[k]); iB1.push_back(dist[v]); iL2.push_back(lower[kd]); iB2.push_back(dist[kd]); iT2.push_back(tip[kd]);
                iNode.push_back(v); iKid.push_back(kd); iKind.push_back(which == 1 ? 1 : 2);
            }
        const size_t m = iL1.size();
        if (!m) continue;
        {   // the children's lower lists go through their own branches first where those carry mutations
            std::vector<int32_t> kids, ids, where;
            for (size_t i = 0; i < m; i++) if (iKid[i] >= 0 && mut[iKid[i]] >= 0) { kids.push_back(iKid[i]); ids.push_back(iL2[i]); where.push_back((int32_t)i); }
            if (!kids.empty()) {
                TRY(upd_passed(c, mut, ids, kids, true));
                for (size_t i = 0; i < where.size(); i++) iL2[where[i]] = ids[i];
            }
        }
        iT1.assign(m, 0); ud.assign(m, 1); mode.assign(m, 1); old.assign(m, -1); out.assign(m, -1); none.assign(m, 0); diff.assign(m, 0);
        TRY(update_items(c, (int32_t)m, iL1.data(), iB1.data(), iT1.data(), iL2.data(), iB2.data(), iT2.data(), ud.data(), mode.data(), old.data(),
                         out.data(), none.data(), diff.data()));
        for (size_t i = 0; i < m; i++) {
            const int v = iNode[i];
            if (iKind[i] == 0) { totUp[v] = none[i] ? -1 : out[i]; continue; }   // (a None probVectTotUp stays None, M:6270)
            if (none[i]) {
                if (bumpLen <= 0.0)
                    return fail(c, MAPLE_ERR_FATAL, "inconsistent upper lists at node %d (the reference would call updateBLen here)", v);
                bad(v); bad(iKid[i]);
                continue;
            }
            (iKind[i] == 1 ? upRight : upLeft)[v] = out[i];
        }
        // (bad nodes of this level: the levels below a node without its upper lists cannot be made, but every other node of THIS
        // level has been looked at -- the caller lengthens all of them before it starts over)
        if (nBad && *nBad) return MAPLE_OK;
    }
    return MAPLE_OK;
}
