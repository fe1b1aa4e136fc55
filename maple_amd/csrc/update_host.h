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
    int replaced = 0;
    auto whichChild = [&](int p, int v) { return c0[p] == v ? 0 : 1; };
    auto markChild = [&](int p, int which) { (which == 0 ? S.dCh0 : S.dCh1)[p] = 1; S.touch(p); };
    // the upper vector seen by each node, in the node's own reference frame (M:5503-5513)
    auto vectUpOf = [&](const std::vector<int32_t> &nodes, std::vector<int32_t> &out) -> int {
        out.resize(nodes.size());
        for (size_t i = 0; i < nodes.size(); i++) { const int p = up[nodes[i]]; out[i] = c0[p] == nodes[i] ? upRight[p] : upLeft[p]; }
        return upd_passed(c, mut, out, nodes, false);
    };
    std::vector<int32_t> frontier;
    for (int i = 0; i < nChanged; i++) {
        const int v = changed[i];
        S.dLow[v] = 1; S.dDist[v] = 1; S.touch(v);
        if (up[v] >= 0) {
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
            if (a[k] < 0 || b[k] < 0) return fail(c, MAPLE_ERR_ARG, "node %d has a changed child but no two children", nodes[k]);
            pa[k] = lower[a[k]]; pb[k] = lower[b[k]];
            da[k] = dist[a[k]]; db[k] = dist[b[k]]; ta[k] = tip[a[k]]; tb[k] = tip[b[k]];
        }
        TRY(upd_passed(c, mut, pa, a, true));
        TRY(upd_passed(c, mut, pb, b, true));
        ud.assign(m, 0);
        out.resize(m);
        TRY(maple_merge_batch(c, (int32_t)m, pa.data(), da.data(), ta.data(), pb.data(), db.data(), tb.data(), ud.data(), nullptr,
                              nullptr, out.data(), nullptr));
        for (size_t k = 0; k < m; k++) {
            if (out[k] >= 0) continue;
            // None between two zero-length branches: re-estimate the branch above the changed child, like updateBLen
            // (M:5385-5414, 5689-5701)
            const int p = nodes[k];
            if (da[k] != 0.0 || db[k] != 0.0)
                return fail(c, MAPLE_ERR_FATAL, "None vector from non-zero distances in the lower merge at node %d (the reference raises too)", p);
            const int order[2] = {S.dCh0[p] ? 0 : 1, S.dCh0[p] ? 1 : 0};
            for (int oi = 0; oi < 2; oi++) {
                const int which = order[oi];
                const int ch = which == 0 ? c0[p] : c1[p];
                std::vector<int32_t> one{ch}, vu;
                TRY(vectUpOf(one, vu));
                double t = 0.0;
                uint8_t isFalse = 0;
                const uint8_t tipc = tip[ch];
                TRY(maple_blen_batch(c, 1, vu.data(), &lower[ch], &tipc, &t, &isFalse));
                dist[ch] = isFalse ? 0.0 : t;
                S.dLow[ch] = 1; S.dDist[ch] = 1; S.touch(ch);
                markChild(p, which);
                if (dist[ch] != 0.0) break;
            }
            da[k] = dist[a[k]]; db[k] = dist[b[k]];
            int32_t o2 = -1;
            const uint8_t z = 0;
            TRY(maple_merge_batch(c, 1, &pa[k], &da[k], &ta[k], &pb[k], &db[k], &tb[k], &z, nullptr, nullptr, &o2, nullptr));
            if (o2 < 0) return fail(c, MAPLE_ERR_FATAL, "None vector after updateBLen at node %d", p);
            out[k] = o2;
        }
        fresh.resize(m);
        TRY(maple_shorten_batch(c, (int32_t)m, out.data(), fresh.data()));
        ids1.clear(); ids2.clear();
        std::vector<int32_t> whereOld;
        for (size_t k = 0; k < m; k++)
            if (lower[nodes[k]] >= 0) { ids1.push_back(fresh[k]); ids2.push_back(lower[nodes[k]]); whereOld.push_back((int32_t)k); }
        flags.assign(ids1.size(), 1);
        if (!ids1.empty()) TRY(maple_differ_batch(c, (int32_t)ids1.size(), ids1.data(), ids2.data(), flags.data()));   // (new, old), M:5793
        std::vector<uint8_t> diff(m, 1);
        for (size_t i = 0; i < whereOld.size(); i++) diff[whereOld[i]] = flags[i];
        for (size_t k = 0; k < m; k++) {
            const int v = nodes[k];
            lower[v] = fresh[k];
            replaced++;
            if (!diff[k]) continue;
            S.dLow[v] = 1; S.touch(v);
            if (up[v] >= 0) {
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
                        replaced++;
                        S.dUp[target] = 1; S.touch(target);
                        addTodo(target);
                    }
                }
            }
            if (nodes.size() == 1) continue;
            nodes.erase(nodes.begin());                                    // (only the root has depth 0)
        }
        TRY(vectUpOf(nodes, vu));
        // probVectTotUp
        sel.clear();
        for (size_t k = 0; k < nodes.size(); k++) if (S.dUp[nodes[k]] || S.dLow[nodes[k]]) sel.push_back((int32_t)k);
        if (!sel.empty()) {
            ids1.clear(); ids2.clear(); da.clear(); db.clear(); ta.clear(); tb.clear();
            std::vector<int32_t> who;
            for (int32_t k : sel) {
                const int v = nodes[k];
                if (dist[v] == 0.0) { totUp[v] = -1; continue; }
                who.push_back(v);
                ids1.push_back(vu[k]); ids2.push_back(lower[v]); da.push_back(dist[v] / 2); db.push_back(dist[v] / 2);
                ta.push_back(0); tb.push_back(tip[v]);
            }
            if (!who.empty()) {
                ud.assign(who.size(), 1);
                out.resize(who.size());
                TRY(maple_merge_batch(c, (int32_t)who.size(), ids1.data(), da.data(), ta.data(), ids2.data(), db.data(), tb.data(),
                                      ud.data(), nullptr, nullptr, out.data(), nullptr));
                for (size_t i = 0; i < who.size(); i++)
                    if (out[i] < 0) return fail(c, MAPLE_ERR_FATAL, "None probVectTotUp on a branch of non-zero length (node %d)", who[i]);
                fresh.resize(who.size());
                TRY(maple_shorten_batch(c, (int32_t)who.size(), out.data(), fresh.data()));
                for (size_t i = 0; i < who.size(); i++) totUp[who[i]] = fresh[i];
                replaced += (int)who.size();
            }
        }
        // probVectUpRight (for child 0: upper vector + child 1) and probVectUpLeft (for child 1: upper vector + child 0)
        for (int pass = 0; pass < 2; pass++) {
            const int which = pass == 0 ? 1 : 0;
            int32_t *store = pass == 0 ? upRight : upLeft;
            std::vector<int32_t> who;
            ids1.clear(); da.clear(); kid.clear();
            for (size_t k = 0; k < nodes.size(); k++) {
                const int v = nodes[k];
                if (c0[v] < 0) continue;
                if (!(S.dUp[v] || S.dDist[v] || (which == 0 ? S.dCh0[v] : S.dCh1[v]))) continue;
                who.push_back(v); ids1.push_back(vu[k]); da.push_back(dist[v]);
                kid.push_back(which == 1 ? c1[v] : c0[v]);
            }
            if (who.empty()) continue;
            const size_t m = who.size();
            pk.resize(m); db.resize(m); ta.assign(m, 0); tb.resize(m);
            for (size_t i = 0; i < m; i++) { pk[i] = lower[kid[i]]; db[i] = dist[kid[i]]; tb[i] = tip[kid[i]]; }
            TRY(upd_passed(c, mut, pk, kid, true));
            ud.assign(m, 1);
            nv.resize(m);
            TRY(maple_merge_batch(c, (int32_t)m, ids1.data(), da.data(), ta.data(), pk.data(), db.data(), tb.data(), ud.data(), nullptr,
                                  nullptr, nv.data(), nullptr));
            for (size_t i = 0; i < m; i++)
                if (nv[i] < 0) return fail(c, MAPLE_ERR_FATAL, "None upper vector at node %d (the reference would call updateBLen here)", who[i]);
            std::vector<int32_t> o1, o2, whereOld;
            for (size_t i = 0; i < m; i++)
                if (store[who[i]] >= 0) { o1.push_back(store[who[i]]); o2.push_back(nv[i]); whereOld.push_back((int32_t)i); }
            flags.assign(o1.size(), 1);
            if (!o1.empty()) TRY(maple_differ_batch(c, (int32_t)o1.size(), o1.data(), o2.data(), flags.data()));     // (old, new), M:5645
            std::vector<uint8_t> diff(m, 1);
            for (size_t i = 0; i < whereOld.size(); i++) diff[whereOld[i]] = flags[i];
            std::vector<int32_t> keep, keepWho;
            for (size_t i = 0; i < m; i++) if (diff[i]) { keep.push_back(nv[i]); keepWho.push_back(who[i]); }
            if (keep.empty()) continue;
            fresh.resize(keep.size());
            TRY(maple_shorten_batch(c, (int32_t)keep.size(), keep.data(), fresh.data()));
            for (size_t i = 0; i < keep.size(); i++) {
                const int v = keepWho[i];
                store[v] = fresh[i];
                const int target = which == 1 ? c0[v] : c1[v];
                S.dUp[target] = 1; S.touch(target);
                addTodo(target);
            }
            replaced += (int)keep.size();
        }
    }
    S.clear();
    *nReplaced = replaced;
    return MAPLE_OK;
}
