// maple_amd/csrc/update_host.h -- updatePartials (MAPLEv0.7.5.4.py:5479-5815) for a set of local changes, level by level.
//
// The reference repairs the genome lists around ONE change with a LIFO work list of (node, direction) items, one
// mergeVectors at a time.  Here the lists invalidated by ANY number of simultaneous changes are repaired level by level
// -- phase A up (lower lists, deepest level first), phase B down (probVectTotUp / probVectUpRight / probVectUpLeft,
// shallowest first) -- every level a handful of batched launches, with the reference's own stop rule
// (areVectorsDifferent, M:5645-5658 / 5793) deciding where the repair ends.  The level loop runs here, on the host side of
// the library, on the caller's own tree arrays (plain int32 / double columns, updated in place): no Python in the loop.
// Included by maple_hip.hip (it uses the batch entry points defined there).
#pragma once

namespace {

struct UpdateScratch {                 // per-context, persistent: flags per node, cleared through the touched list
    std::vector<uint8_t> dLow, dUp, dDist, dCh0, dCh1, inFrontier, inTodo;
    std::vector<int32_t> touched;
    std::vector<int32_t> replacedNodes;   // nodes one of whose four lists (or whose branch length) the last call replaced
    void fit(size_t n)
    {
        if (dLow.size() < n) {
            dLow.assign(n, 0); dUp.assign(n, 0); dDist.assign(n, 0); dCh0.assign(n, 0); dCh1.assign(n, 0);
            inFrontier.assign(n, 0); inTodo.assign(n, 0);
            touched.clear();
        }
    }
    void touch(int v) { touched.push_back(v); }
    void clear()
    {
        for (int v : touched) dLow[v] = dUp[v] = dDist[v] = dCh0[v] = dCh1[v] = inFrontier[v] = inTodo[v] = 0;
        touched.clear();
    }
};

}  // namespace

static UpdateScratch &update_scratch(maple_ctx *c)
{
    if (!c->upd) c->upd = new UpdateScratch();
    return *(UpdateScratch *)c->upd;
}

static void update_scratch_free(maple_ctx *c)
{
    delete (UpdateScratch *)c->upd;
    c->upd = nullptr;
}

// lists[i] passed through the branch above nodes[i] (up or down) where that branch carries mutations, M:3749-3877
static int upd_passed(maple_ctx *c, const int32_t *mut, std::vector<int32_t> &ids, const std::vector<int32_t> &nodes, bool dirUp)
{
    std::vector<int32_t> src, ml, where;
    for (size_t i = 0; i < nodes.size(); i++)
        if (mut[nodes[i]] >= 0 && ids[i] >= 0) { src.push_back(ids[i]); ml.push_back(mut[nodes[i]]); where.push_back((int32_t)i); }
    if (src.empty()) return MAPLE_OK;
    std::vector<uint8_t> dir(src.size(), dirUp ? 1 : 0);
    std::vector<int32_t> out(src.size());
    TRY(maple_pass_branch_batch(c, (int32_t)src.size(), src.data(), ml.data(), dir.data(), out.data()));
    for (size_t i = 0; i < where.size(); i++) ids[where[i]] = out[i];
    return MAPLE_OK;
}

// One level's worth of items through k_update_items: merge, then shorten / compare as the item's mode says (see the
// kernel).  outList[i]: the new list (-1: None from the merge, or mode 2 and not different: nothing to replace);
// outNone[i] = 1 if the merge itself came out None; outDiff[i] = the areVectorsDifferent verdict (1 where none was asked).
static int update_items(maple_ctx *c, int32_t n, const int32_t *l1, const double *b1, const uint8_t *t1, const int32_t *l2,
                        const double *b2, const uint8_t *t2, const uint8_t *ud, const uint8_t *mode, const int32_t *old,
                        int32_t *outList, uint8_t *outNone, uint8_t *outDiff)
{
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(check_ids(c, n, l1, false, "list1"));
    TRY(check_ids(c, n, l2, false, "list2"));
    TRY(check_ids(c, n, old, true, "old list"));
    std::vector<int64_t> woff(n), cap(n), woffB(n), aoffB(n);
    int64_t tot = 0;
    for (int i = 0; i < n; i++) {
        cap[i] = (int64_t)c->h_n_ent[l1[i]] + c->h_n_ent[l2[i]];
        woff[i] = tot; woffB[i] = tot + cap[i]; aoffB[i] = 5 * woffB[i];
        tot += 2 * cap[i];
    }
    HIPCK(c, c->s_words.reserve((size_t)tot));
    HIPCK(c, c->s_aux.reserve((size_t)(5 * tot)));
    TRY(stage_begin(c, (size_t)n * 128 + 1024));
    STAGE(dl1, c, l1, n); STAGE(dl2, c, l2, n); STAGE(db1, c, b1, n); STAGE(db2, c, b2, n);
    STAGE(dt1, c, t1, n); STAGE(dt2, c, t2, n); STAGE(dud, c, ud, n); STAGE(dmode, c, mode, n); STAGE(dold, c, old, n);
    STAGE(dwo, c, woff.data(), n); STAGE(dcap, c, cap.data(), n); STAGE(dwoB, c, woffB.data(), n); STAGE(daoB, c, aoffB.data(), n);
    TRY(stage_flush(c));
    HIPCK(c, c->s_i32[2].reserve((size_t)3 * n));
    int32_t *res3 = c->s_i32[2].p;
    // a level with few items waits for ONE item's latency: a wavefront per item (wave_update.h); many items: a lane each
    if (n <= wave_item_max(c, 4096))
        DISPATCH3(c, k_update_items_wave, <<<n, 64, 0, c->stream>>>(c->d_model, view(c), n, dl1, db1, dt1, dl2, db2, dt2, dud, dmode, dold,
                                                                     c->s_words.p, c->s_aux.p, dwo, dcap, res3));
    else
        DISPATCH3(c, k_update_items, <<<grid_for(n), MAPLE_BLOCK, 0, c->stream>>>(c->d_model, view(c), n, dl1, db1, dt1, dl2, db2, dt2, dud,
                                                                                    dmode, dold, c->s_words.p, c->s_aux.p, dwo, dcap, res3));
    HIPCK(c, hipGetLastError());
    HIPCK(c, c->pin_res.reserve((size_t)3 * n * sizeof(int32_t)));
    const int32_t *h3 = (const int32_t *)c->pin_res.p;
    HIPCK(c, hipMemcpyAsync(c->pin_res.p, res3, (size_t)3 * n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    std::vector<int32_t> ne(n), na(n);
    for (int i = 0; i < n; i++) {
        const int r = h3[i], flag = h3[(size_t)2 * n + i];
        outNone[i] = r == -1;
        outDiff[i] = (uint8_t)(flag != 0);
        ne[i] = (r >= 0 && !flag && mode[i] == 2) ? -1 : r;                // mode 2 and not different: nothing to commit
        na[i] = h3[(size_t)n + i];
    }
    return commit_known(c, n, dwoB, daoB, res3, res3 + n, ne, na, outList, nullptr, nullptr);
}

extern "C" int maple_update_partials(maple_ctx *c, int32_t n, int32_t root, const int32_t *up, const int32_t *c0, const int32_t *c1,
                                     const uint8_t *tip, const int32_t *mut, const int32_t *depth, double *dist, int32_t *lower,
                                     int32_t *upRight, int32_t *upLeft, int32_t *totUp, int32_t nChanged, const int32_t *changed,
                                     int32_t *nReplaced)
{
    if (!c || n <= 0 || !up || !c0 || !c1 || !tip || !mut || !depth || !dist || !lower || !upRight || !upLeft || !totUp
        || nChanged < 0 || (nChanged && !changed) || !nReplaced)
        return MAPLE_ERR_ARG;
    if (root < 0 || root >= n) return fail(c, MAPLE_ERR_ARG, "root %d is not a node", root);
    for (int i = 0; i < nChanged; i++)
        if (changed[i] < 0 || changed[i] >= n) return fail(c, MAPLE_ERR_ARG, "changed[%d] = %d is not a node", i, changed[i]);
    TRY(need_model(c));
    UpdateScratch &S = update_scratch(c);
    S.fit((size_t)n);
    S.clear();
    S.replacedNodes.clear();
    int replaced = 0;
    // (the caller's columns are trusted only as far as they are read: every relative that enters the work lists is range-checked)
    bool badNode = false;
    auto okNode = [&](int v) { if (v < -1 || v >= n) { badNode = true; return false; } return v >= 0; };
    auto whichChild = [&](int p, int v) { return c0[p] == v ? 0 : 1; };
    auto markChild = [&](int p, int which) { (which == 0 ? S.dCh0 : S.dCh1)[p] = 1; S.touch(p); };
    // the upper vector seen by each node, in the node's own reference frame (M:5503-5513)
    auto vectUpOf = [&](const std::vector<int32_t> &nodes, std::vector<int32_t> &out) -> int {
        out.resize(nodes.size());
        for (size_t i = 0; i < nodes.size(); i++) { const int p = up[nodes[i]]; out[i] = c0[p] == nodes[i] ? upRight[p] : upLeft[p]; }
        return upd_passed(c, mut, out, nodes, false);
    };
    int roundNo = 0;
    std::vector<int32_t> redoNow, redoNext;                                // nodes whose upper vectors an earlier round left undone
    auto repair = [&](const std::vector<int32_t> &chg, std::vector<int32_t> &next) -> int {
    std::vector<int32_t> frontier;
    for (size_t i = 0; i < chg.size(); i++) {
        const int v = chg[i];
        S.dLow[v] = 1; S.dDist[v] = 1; S.touch(v);
        if (okNode(up[v])) {
            markChild(up[v], whichChild(up[v], v));
            if (!S.inFrontier[up[v]]) { S.inFrontier[up[v]] = 1; frontier.push_back(up[v]); }
        }
    }
    // ---- phase A: lower lists, deepest first --------------------------------------------------------------------------
    std::vector<int32_t> nodes, a, b, pa, pb, out, fresh, ids1, ids2;
    std::vector<double> da, db;
    std::vector<uint8_t> ta, tb, flags, ud;
    while (!frontier.empty()) {
        int dmax = -1;
        for (int v : frontier) dmax = std::max(dmax, depth[v]);
        nodes.clear();
        std::vector<int32_t> rest;
        for (int v : frontier) (depth[v] == dmax ? nodes : rest).push_back(v);
        frontier.swap(rest);
        std::sort(nodes.begin(), nodes.end());
        for (int v : nodes) S.inFrontier[v] = 0;
        const size_t m = nodes.size();
        a.resize(m); b.resize(m); pa.resize(m); pb.resize(m); da.resize(m); db.resize(m); ta.resize(m); tb.resize(m);
        for (size_t k = 0; k < m; k++) {
            a[k] = c0[nodes[k]]; b[k] = c1[nodes[k]];
            if (a[k] < 0 || b[k] < 0 || a[k] >= n || b[k] >= n) return fail(c, MAPLE_ERR_ARG, "node %d has a changed child but no two (valid) children", nodes[k]);
            pa[k] = lower[a[k]]; pb[k] = lower[b[k]];
            da[k] = dist[a[k]]; db[k] = dist[b[k]]; ta[k] = tip[a[k]]; tb[k] = tip[b[k]];
        }
        TRY(upd_passed(c, mut, pa, a, true));
        TRY(upd_passed(c, mut, pb, b, true));
        ud.assign(m, 0);
        out.resize(m);
        std::vector<uint8_t> modeA(m, 0), none(m), diff(m);
        std::vector<int32_t> oldA(m);
        for (size_t k = 0; k < m; k++) oldA[k] = lower[nodes[k]];
        TRY(update_items(c, (int32_t)m, pa.data(), da.data(), ta.data(), pb.data(), db.data(), tb.data(), ud.data(), modeA.data(),
                         oldA.data(), out.data(), none.data(), diff.data()));
        for (size_t k = 0; k < m; k++) {
            if (!none[k]) continue;
            // None between two zero-length branches: re-estimate the branch above the changed child, like updateBLen
            // (M:5385-5414, 5689-5701)
            const int p = nodes[k];
            if (da[k] != 0.0 || db[k] != 0.0)
                return fail(c, MAPLE_ERR_FATAL, "None vector from non-zero distances in the lower merge at node %d (the reference raises too)", p);
            const int order[2] = {S.dCh0[p] ? 0 : 1, S.dCh0[p] ? 1 : 0};
            for (int oi = 0; oi < 2; oi++) {
                const int which = order[oi];
                const int ch = which == 0 ? c0[p] : c1[p];
                std::vector<int32_t> one{ch}, vu1;
                TRY(vectUpOf(one, vu1));
                double t = 0.0;
                uint8_t isFalse = 0;
                const uint8_t tipc = tip[ch];
                TRY(maple_blen_batch(c, 1, vu1.data(), &lower[ch], &tipc, &t, &isFalse));
                dist[ch] = isFalse ? 0.0 : t;
                S.replacedNodes.push_back(ch);
                S.dLow[ch] = 1; S.dDist[ch] = 1; S.touch(ch);
                markChild(p, which);
                if (dist[ch] != 0.0) break;
            }
            da[k] = dist[a[k]]; db[k] = dist[b[k]];
            const uint8_t z = 0;
            TRY(update_items(c, 1, &pa[k], &da[k], &ta[k], &pb[k], &db[k], &tb[k], &z, &z, &oldA[k], &out[k], &none[k], &diff[k]));
            if (none[k]) return fail(c, MAPLE_ERR_FATAL, "None vector after updateBLen at node %d", p);
        }
        for (size_t k = 0; k < m; k++) {
            const int v = nodes[k];
            lower[v] = out[k];
            S.replacedNodes.push_back(v);
            replaced++;
            if (!diff[k]) continue;
            S.dLow[v] = 1; S.touch(v);
            if (okNode(up[v])) {
                markChild(up[v], whichChild(up[v], v));
                if (!S.inFrontier[up[v]]) { S.inFrontier[up[v]] = 1; frontier.push_back(up[v]); }
            }
        }
    }
    // ---- phase B: upper lists, shallowest first -------------------------------------------------------------------------
    std::vector<int32_t> todo;
    {
        std::vector<int32_t> seen(S.touched);
        std::sort(seen.begin(), seen.end());
        seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
        for (int v : seen)
            if ((S.dLow[v] || S.dCh0[v] || S.dCh1[v]) && !S.inTodo[v]) { S.inTodo[v] = 1; todo.push_back(v); }
    }
    auto addTodo = [&](int v) { if (!S.inTodo[v]) { S.inTodo[v] = 1; S.touch(v); todo.push_back(v); } };
    for (int v : redoNow) { S.dUp[v] = 1; S.touch(v); addTodo(v); }
    std::vector<int32_t> vu, sel, kid, nv, pk;
    while (!todo.empty()) {
        int dmin = 1 << 30;
        for (int v : todo) dmin = std::min(dmin, depth[v]);
        nodes.clear();
        std::vector<int32_t> rest;
        for (int v : todo) (depth[v] == dmin ? nodes : rest).push_back(v);
        todo.swap(rest);
        std::sort(nodes.begin(), nodes.end());
        for (int v : nodes) S.inTodo[v] = 0;
        if (nodes[0] == root) {
            const int r = root;
            if (c0[r] >= 0) {
                const int64_t pathOff[2] = {0, mut[r] >= 0 ? 1 : 0};
                const int32_t pathMut[1] = {mut[r] >= 0 ? mut[r] : 0};
                for (int pass = 0; pass < 2; pass++) {                     // upRight merges child 1, upLeft child 0
                    const int which = pass == 0 ? 1 : 0;
                    int32_t *store = pass == 0 ? upRight : upLeft;
                    if (!(which == 0 ? S.dCh0[r] : S.dCh1[r])) continue;
                    const int k = which == 1 ? c1[r] : c0[r];
                    std::vector<int32_t> one{k}, pl{lower[k]};
                    TRY(upd_passed(c, mut, pl, one, true));
                    int32_t res = -1;
                    const uint8_t tk = tip[k];
                    TRY(maple_root_vector_batch(c, 1, pl.data(), &dist[k], &tk, pathOff, pathMut, &res));
                    const int target = which == 1 ? c0[r] : c1[r];
                    uint8_t df = 1;
                    if (store[r] >= 0) TRY(maple_differ_batch(c, 1, &store[r], &res, &df));
                    if (df) {
                        store[r] = res;
                        S.replacedNodes.push_back(r);
                        replaced++;
                        S.dUp[target] = 1; S.touch(target);
                        addTodo(target);
                    }
                }
            }
            if (nodes.size() == 1) continue;
            nodes.erase(nodes.begin());                                    // (only the root has depth 0)
        }
        for (int v : nodes) if (up[v] < 0 || up[v] >= n) return fail(c, MAPLE_ERR_ARG, "node %d: parent index out of range", v);
        TRY(vectUpOf(nodes, vu));
        // One fused launch for the level: probVectTotUp where the node's lower list, length or upper vector changed,
        // probVectUpRight (upper vector + child 1, for child 0) and probVectUpLeft (upper vector + child 0, for child 1) where
        // the upper vector, the length or that child changed
        std::vector<int32_t> iL1, iL2, iOld, iNode, iKid;
        std::vector<double> iB1, iB2;
        std::vector<uint8_t> iT2, iMode, iKind;                            // kind: 0 totUp, 1 upRight, 2 upLeft
        for (size_t k = 0; k < nodes.size(); k++) {
            const int v = nodes[k];
            if (S.dUp[v] || S.dLow[v]) {
                if (dist[v] == 0.0) { if (totUp[v] != -1) S.replacedNodes.push_back(v); totUp[v] = -1; }
                else {
                    iL1.push_back(vu[k]); iB1.push_back(dist[v] / 2); iL2.push_back(lower[v]); iB2.push_back(dist[v] / 2);
                    iT2.push_back(tip[v]); iMode.push_back(1); iOld.push_back(-1); iNode.push_back(v); iKid.push_back(-1); iKind.push_back(0);
                }
            }
            if (c0[v] < 0) continue;
            if (c0[v] >= n || c1[v] < 0 || c1[v] >= n) return fail(c, MAPLE_ERR_ARG, "node %d: child index out of range", v);
            for (int which = 1; which >= 0; which--) {
                if (!(S.dUp[v] || S.dDist[v] || (which == 0 ? S.dCh0[v] : S.dCh1[v]))) continue;
                const int kd = which == 1 ? c1[v] : c0[v];
                iL1.push_back(vu[k]); iB1.push_back(dist[v]); iL2.push_back(lower[kd]); iB2.push_back(dist[kd]); iT2.push_back(tip[kd]);
                iMode.push_back(2); iOld.push_back(which == 1 ? upRight[v] : upLeft[v]); iNode.push_back(v); iKid.push_back(kd);
                iKind.push_back(which == 1 ? 1 : 2);
            }
        }
        const size_t m = iL1.size();
        if (m == 0) continue;
        {   // the children's lower lists go through their own branches first where those carry mutations
            std::vector<int32_t> kids, ids, where;
            for (size_t i = 0; i < m; i++) if (iKid[i] >= 0 && mut[iKid[i]] >= 0) { kids.push_back(iKid[i]); ids.push_back(iL2[i]); where.push_back((int32_t)i); }
            if (!kids.empty()) {
                TRY(upd_passed(c, mut, ids, kids, true));
                for (size_t i = 0; i < where.size(); i++) iL2[where[i]] = ids[i];
            }
        }
        std::vector<uint8_t> iT1(m, 0), iUd(m, 1), none(m), diff(m);
        out.resize(m);
        TRY(update_items(c, (int32_t)m, iL1.data(), iB1.data(), iT1.data(), iL2.data(), iB2.data(), iT2.data(), iUd.data(), iMode.data(),
                         iOld.data(), out.data(), none.data(), diff.data()));
        std::vector<int32_t> deferred;                                     // nodes of this level that met an inconsistency
        for (size_t i = 0; i < m; i++) {
            const int v = iNode[i];
            if (none[i]) {
                // An inconsistency between zero-length branches (M:5522-5528, 5568-5600): the reference re-estimates the branch
                // above the node (updateBLen, M:5385-5414) and, if that stays at zero, the branch above the child that was being
                // merged, then repairs from there.  Here: the same estimates, and the node whose length changed starts another
                // round of the level loops.
                const int kd = iKid[i];
                if (c->tuning.verbose)
                    fprintf(stderr, "[maple] updatePartials round %d: None %s at node %d (length %.3g), child %d (length %.3g), merged with b1 %.3g b2 %.3g\n",
                            roundNo, iKind[i] == 0 ? "probVectTotUp" : (iKind[i] == 1 ? "probVectUpRight" : "probVectUpLeft"), v, dist[v], kd,
                            kd >= 0 ? dist[kd] : -1.0, iB1[i], iB2[i]);
                bool done = false;
                for (int32_t w : deferred) if (w == v) done = true;          // (its other merge already dealt with it)
                if (done) continue;
                if (iKind[i] != 0 && (dist[v] != 0.0 || dist[kd] != 0.0))
                    return fail(c, MAPLE_ERR_FATAL, "None upper vector from non-zero distances at node %d (the reference raises too)", v);
                deferred.push_back(v);
                redoNext.push_back(v);
                const int cand[2] = {v, kd};
                for (int ci = 0; ci < 2 && !done; ci++) {
                    const int w = cand[ci];
                    if (w < 0 || up[w] < 0) continue;
                    std::vector<int32_t> one{w}, vu1;
                    TRY(vectUpOf(one, vu1));
                    double t = 0.0;
                    uint8_t isFalse = 0;
                    const uint8_t tipc = tip[w];
                    TRY(maple_blen_batch(c, 1, vu1.data(), &lower[w], &tipc, &t, &isFalse));
                    dist[w] = isFalse ? 0.0 : t;
                    S.replacedNodes.push_back(w);
                    if (dist[w] != 0.0) { next.push_back(w); done = true; }
                }
                if (!done) return fail(c, MAPLE_ERR_FATAL, "None upper vector at node %d and no branch length to lengthen", v);
                continue;
            }
        }
        // (the reference recomputes ALL of such a node's vectors with the new length before anything goes on to its children,
        // M:5575-5600: none of what this level computed for it with the old length is used; the next round redoes the node)
        for (size_t i = 0; i < m; i++) {
            const int v = iNode[i];
            if (none[i] || std::find(deferred.begin(), deferred.end(), v) != deferred.end()) continue;
            if (iKind[i] == 0) { totUp[v] = out[i]; S.replacedNodes.push_back(v); replaced++; continue; }
            if (!diff[i]) continue;
            (iKind[i] == 1 ? upRight : upLeft)[v] = out[i];
            S.replacedNodes.push_back(v);
            replaced++;
            const int target = iKind[i] == 1 ? c0[v] : c1[v];
            S.dUp[target] = 1; S.touch(target);
            addTodo(target);
        }
    }
    if (badNode) return fail(c, MAPLE_ERR_ARG, "a relative index of a touched node is out of range");
    return MAPLE_OK;
    };
    std::vector<int32_t> chg(changed, changed + nChanged), next;
    for (int round = 0; !chg.empty() || !redoNext.empty(); round++) {
        if (round >= 64) return fail(c, MAPLE_ERR_FATAL, "updatePartials keeps finding inconsistent zero-length branches");
        next.clear();
        roundNo = round;
        redoNow.swap(redoNext);
        redoNext.clear();
        TRY(repair(chg, next));
        S.clear();
        chg.swap(next);
    }
    S.clear();
    *nReplaced = replaced;
    return MAPLE_OK;
}

// the nodes whose lists (or branch length) the last maple_update_partials replaced, each once, ascending: what a caller
// has to tell maple_tree_patch about besides its own tree edit
extern "C" int maple_update_partials_touched(maple_ctx *c, int32_t cap, int32_t *nodes, int32_t *n)
{
    if (!c || cap < 0 || !n || (cap && !nodes)) return MAPLE_ERR_ARG;
    UpdateScratch &S = update_scratch(c);
    std::vector<int32_t> u(S.replacedNodes);
    std::sort(u.begin(), u.end());
    u.erase(std::unique(u.begin(), u.end()), u.end());
    *n = (int32_t)u.size();
    if ((int32_t)u.size() > cap) return fail(c, MAPLE_ERR_ARG, "%zu touched nodes do not fit in %d", u.size(), cap);
    for (size_t i = 0; i < u.size(); i++) nodes[i] = u[i];
    return MAPLE_OK;
}
