// maple_placement_search_batch: findBestParentForNewSample (MAPLEv0.7.5.4.py:7912-8292) for MANY query samples on one
// frozen tree (the shape of --findSamplePlacements / --lineageRefs, M:11190-11220, and of online batches).
// Included by maple_hip.hip ahead of the tree-mirror section; composes the library's own batch entry points plus three kernels:
//   k_place_score  - every query against every candidate branch, each in the candidate's MAT reference frame
//   k_place_minor  - isMinorSequence of every query against every leaf
//   k_place_replay - the reference's depth-first traversal over those scores, one lane per query (a forward scan of the tree
//                    laid out in traversal order)
#pragma once

// same scheduling as k_append_queries (dynamic 64-candidate tiles per wavefront, candidate-chunk-major, candidates sorted by
// list length), with the query taken in each candidate's MAT reference frame
template <bool RV, bool U, bool SS>
__global__ MAPLE_APPEND_ATTR void k_place_score(const DevModel *__restrict__ mp, ArenaView av, int nQ, int nF,
                                                const int32_t *qFrameLists, int nC, const int32_t *cand,
                                                const int32_t *candFrame, int isTip, double bLen, double *out, long long ldOut,
                                                const int32_t *outCol, const uint8_t *qTip, const double *qBLen, int *counter)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const int lane = threadIdx.x & 63;
    const int nChunks = (nC + 63) / 64;
    const long long tiles = (long long)nQ * nChunks;
    for (;;) {
        int j = 0;
        if (lane == 0) j = atomicAdd(counter, 1);
        j = __builtin_amdgcn_readfirstlane(j);
        if (j >= tiles) break;
        const int ch = j / nQ;
        const int q = j - ch * nQ;
        const int k = ch * 64 + lane;
        if (k < nC) {
            const int ql = qFrameLists[(long long)q * nF + candFrame[k]];
            if (ql >= 0)
                out[(long long)q * ldOut + (outCol ? outCol[k] : k)] =
                    append_walk(c, list_ref(av, cand[k]), list_ref(av, ql), qTip ? qTip[q] != 0 : isTip != 0, qBLen ? qBLen[q] : bLen);
        }
    }
}

// one launch of k_place_score on the context's stream (timed): out[q * ldOut + (outCol ? outCol[k] : k)]
int launch_place_score(maple_ctx *c, int nQ, int nF, const int32_t *qFrameLists, int nC, const int32_t *cand,
                       const int32_t *candFrame, int isTip, double bLen, double *out, long long ldOut,
                       const int32_t *outCol, const uint8_t *qTip, const double *qBLen, int kind, double algBytes)
{
    const long long tiles = (long long)nQ * ((nC + 63) / 64);
    if (tiles > 0x7fffffffLL - (1 << 20)) return fail(c, MAPLE_ERR_ARG, "nQ x nC too large for one launch");
    if (!c->d_tile_counters) HIPCK(c, hipMalloc((void **)&c->d_tile_counters, 64 * sizeof(int32_t)));
    int32_t *counter = c->d_tile_counters + (c->tile_counter_next++ & 63);
    HIPCK(c, hipMemsetAsync(counter, 0, sizeof(int32_t), c->stream));
    const long long waves = (tiles + 3) / 4;
    const int grid = waves < 256 * MAPLE_APPEND_WAVES ? (int)waves : 256 * MAPLE_APPEND_WAVES;
    hipEvent_t e0, e1;
    TRY(ev_pair(c, &e0, &e1, kind, (double)nQ * (double)nC, algBytes));
    HIPCK(c, hipEventRecord(e0, c->stream));
    DISPATCH3(c, k_place_score, <<<grid, MAPLE_BLOCK, 0, c->stream>>>(c->d_model, view(c), nQ, nF, qFrameLists, nC, cand, candFrame, isTip,
                                                                        bLen, out, ldOut, outCol, qTip, qBLen, counter));
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipEventRecord(e1, c->stream));
    return MAPLE_OK;
}

// out[q * ldOut + (outCol ? outCol[k] : k)]; leafFrame null: one reference frame
__global__ __launch_bounds__(MAPLE_BLOCK) void k_place_minor(int lRef, ArenaView av, int nQ, int nF, const int32_t *qFrameLists,
                                                             int nL, const int32_t *leaf, const int32_t *leafFrame,
                                                             int onlyIdentical, uint8_t *out, long long ldOut, const int32_t *outCol)
{
    const long long tot = (long long)nQ * nL;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i / nL), k = (int)(i - (long long)q * nL);
        out[(long long)q * ldOut + (outCol ? outCol[k] : k)] =
            (uint8_t)minor_walk(lRef, list_ref(av, leaf[k]), list_ref(av, qFrameLists[(long long)q * nF + (leafFrame ? leafFrame[k] : 0)]),
                                onlyIdentical != 0);
    }
}

static int place_meta(maple_ctx *c, double effNon0)
{
    PlaceMeta &M = *c->place;
    if (M.valid && M.effNon0 == effNon0) return MAPLE_OK;
    if (c->ahead) c->ahead->active = false;                               // (the columns are numbered anew: rows made ahead are void)
    M.d_pn.release();
    const int32_t n = c->dtree.n, root = c->dtree.root;
    const auto &up = c->h_tree_up;
    const auto &c0 = c->h_tree_c0, &totUp = c->h_tree_totUp, &lower = c->h_tree_lower;
    const std::vector<int32_t> &order = M.order;                          // compute_frames(), at maple_tree_upload
    M.cand.clear();
    M.leaves.clear();
    std::vector<int32_t> &candIdx = M.h_candIdx, &leafIdx = M.h_leafIdx;
    candIdx.assign(n, -1);
    leafIdx.assign(n, -1);
    std::vector<int32_t> &candList = M.h_candList, &candFrame = M.h_candFrame, &leafList = M.h_leafList, &leafFrame = M.h_leafFrame;
    candList.clear(); candFrame.clear(); leafList.clear(); leafFrame.clear();
    for (int32_t v : order)
        if (v != root && up[v] >= 0 && c->h_tree_dist[v] > effNon0 && totUp[v] >= 0) M.cand.push_back(v);      // M:8049
    // columns sorted by list length: the 64 lanes of a scoring wavefront then finish together
    {   // (a stable counting pass: the keys are list lengths, a few hundred values; a comparison sort of 2 M columns through two
        // tables was most of a second at 1 000 000 tips)
        std::vector<int32_t> key(M.cand.size());
        int32_t maxKey = 0;
        for (size_t i = 0; i < M.cand.size(); i++) { key[i] = c->h_n_ent[totUp[M.cand[i]]]; maxKey = std::max(maxKey, key[i]); }
        std::vector<int64_t> start((size_t)maxKey + 2, 0);
        for (int32_t k : key) start[(size_t)k + 1]++;
        for (int32_t k = 0; k <= maxKey; k++) start[(size_t)k + 1] += start[k];
        std::vector<int32_t> sorted(M.cand.size());
        for (size_t i = 0; i < M.cand.size(); i++) sorted[(size_t)start[key[i]]++] = M.cand[i];
        M.cand.swap(sorted);
    }
    for (size_t i = 0; i < M.cand.size(); i++) {
        const int32_t v = M.cand[i];
        candIdx[v] = (int32_t)i;
        candList.push_back(totUp[v]);
        candFrame.push_back(M.frameOf[v]);
    }
    for (int32_t v : order) {
        if (c0[v] < 0) {
            if (lower[v] < 0) return fail(c, MAPLE_ERR_STATE, "leaf %d has no lower genome list", v);
            leafIdx[v] = (int32_t)M.leaves.size();
            M.leaves.push_back(v);
            leafList.push_back(lower[v]);
            leafFrame.push_back(M.frameOf[v]);
        }
    }
    {   // the tree in traversal order: clade sizes and depths over compute_frames()'s depth-first order
        const size_t nr = order.size();
        std::vector<int32_t> size(n, 1), depth(n, 0);
        for (size_t i = nr; i-- > 0;) { const int32_t v = order[i]; if (v != root && up[v] >= 0) size[up[v]] += size[v]; }
        M.h_scan.assign(nr, ScanRec{});
        for (size_t i = 0; i < nr; i++) {
            const int32_t v = order[i];
            if (v != root && up[v] >= 0) depth[v] = depth[up[v]] + 1;
            ScanRec &r = M.h_scan[i];
            r.node = v; r.size = size[v]; r.depth = depth[v];
            r.candCol = candIdx[v]; r.leafCol = leafIdx[v]; r.frame = M.frameOf[v];
            r.childFrame[0] = (c0[v] >= 0 && M.frameOf[c0[v]] != M.frameOf[v]) ? M.frameOf[c0[v]] : -1;
            r.childFrame[1] = (c0[v] >= 0 && M.frameOf[c->h_tree_c1[v]] != M.frameOf[v]) ? M.frameOf[c->h_tree_c1[v]] : -1;
        }
        TRY(h2d(c, M.d_scan, M.h_scan.data(), nr));
    }
    // reserve one extra candidate column for the root vector (its list id is filled in per call)
    candList.push_back(-1);
    candFrame.push_back(M.frameOf[root]);
    TRY(h2d(c, M.d_frameOf, M.frameOf.data(), (size_t)n));
    TRY(h2d(c, M.d_candIdx, candIdx.data(), (size_t)n));
    TRY(h2d(c, M.d_leafIdx, leafIdx.data(), (size_t)n));
    TRY(h2d(c, M.d_candList, candList.data(), candList.size()));
    TRY(h2d(c, M.d_candFrame, candFrame.data(), candFrame.size()));
    TRY(h2d(c, M.d_leafList, leafList.data(), leafList.size()));
    TRY(h2d(c, M.d_leafFrame, leafFrame.data(), leafFrame.size()));
    HIPCK(c, hipStreamSynchronize(c->stream));
    M.h_pn.resize((size_t)n);
    for (int32_t v = 0; v < n; v++) M.h_pn[v] = PlaceMeta::PNode{candIdx[v], leafIdx[v], c->h_tree_c0[v], c->h_tree_c1[v]};
    M.effNon0 = effNon0;
    M.valid = true;
    M.scanStale = false;
    return MAPLE_OK;
}

// The same traversal as place_replay_one (placement_dev.h), over the host's own tree columns instead of the scan array: the
// single-query calls of the sequential placement phase use it, so that a tree changed through maple_tree_patch needs no
// re-linearisation.  Children in the order of compute_frames' depth-first order: child 1's clade first.
// sparse: the row holds PLACE_NO_SCORE in the columns nobody scored (rows by expansion): a visit that needs one ends the
// traversal with status -7 and the caller scores the whole row
static void place_replay_ptr(const maple_ctx *c, const PlaceMeta &M, const PlaceParams &P, const double *sc, int rootCol,
                             const uint8_t *mn, int nF, const PlaceOut &o, bool sparse = false, int32_t *visitEpoch = nullptr,
                             int32_t epoch = 0)
{
    const int32_t root = c->dtree.root;
    const PlaceMeta::PNode *const pn = M.h_pn.data();
    std::vector<uint32_t> frameBits((size_t)(nF + 31) >> 5, 0u);
    if (o.fromBits) for (int i = 0; i < (nF + 31) >> 5; i++) o.fromBits[i] = 0u;
    int32_t *slN = o.slNode;
    double *slL = o.slLK;
    int nSl = 0, status = 0, minorNode = -1, missed = 0, nAppend = 1;
    double bestLK = sc[rootCol];
    const double originalLK = bestLK;
    int bestNode = root;
    struct It { int32_t node; int32_t fails; double parentLK; };
    std::vector<It> st;
    if (visitEpoch) visitEpoch[root] = epoch;                             // (a speculative traversal notes what it read: ahead_spec_*)
    if (!P.supportOnly && pn[root].leafCol >= 0 && mn[pn[root].leafCol] == 1) { status = 1; minorNode = root; nAppend = 0; }
    if (pn[root].c0 >= 0) { st.push_back(It{pn[root].c0, 0, bestLK}); st.push_back(It{pn[root].c1, 0, bestLK}); }
    while (!st.empty() && status == 0) {                                  // M:7972-8100
        const It it = st.back();
        st.pop_back();
        const int t1 = it.node;
        const PlaceMeta::PNode me = pn[t1];                                // (asked for when the node was pushed)
        if (visitEpoch) visitEpoch[t1] = epoch;
        const int candCol = me.candCol, leafCol = me.leafCol;
        // (the visit after this one -- unless this node pushes children -- is the node now on top: its record is here since it
        // was pushed; its score and its minor flag are asked for now, a visit ahead)
        if (!st.empty()) {
            const PlaceMeta::PNode nx = pn[st.back().node];
            if (nx.candCol >= 0) __builtin_prefetch(sc + nx.candCol);
            if (nx.leafCol >= 0) __builtin_prefetch(mn + nx.leafCol);
        }
        int fails = it.fails;
        if (leafCol >= 0) {
            const int cmp = mn[leafCol];
            if (cmp == 1) { if (!P.supportOnly) { status = 1; minorNode = t1; break; } }   // M:7986-8003
            else if (cmp == 2) missed++;
        }
        double lk = it.parentLK;
        if (candCol >= 0) {
            lk = sc[candCol];
            if (sparse) {
                unsigned long long bits;
                memcpy(&bits, &lk, sizeof bits);
                if (bits == 0x7ff8dead0badc0deull) { status = -7; break; }
            }
            nAppend++;
            bool keep = false;
            if (lk >= bestLK) {                                           // M:8065-8073
                const int f = M.frameOf[t1];
                frameBits[f >> 5] |= 1u << (f & 31);
                bestLK = lk; bestNode = t1; fails = 0; keep = true;
            } else if (lk > bestLK - P.thrOpt) keep = true;               // M:8074-8075
            if (keep) {
                if (nSl == MAPLE_PLACE_SHORTLIST) {
                    int k = 0;
                    for (int i = 0; i < nSl; i++)
                        if (slL[i] >= bestLK - P.thrFilter) { slN[k] = slN[i]; slL[k] = slL[i]; k++; }
                    nSl = k;
                }
                if (nSl == MAPLE_PLACE_SHORTLIST) { status = -6; break; }
                slN[nSl] = t1; slL[nSl] = lk; nSl++;
            }
            if (lk < it.parentLK - P.thrConsec) fails++;                  // M:8076-8077
        }
        const bool within = lk > bestLK - P.thrLK;
        const bool go = P.strict ? (fails <= P.allowedFails && within) : (fails <= P.allowedFails || within);   // M:8080-8093
        if (go && me.c0 >= 0) {
            st.push_back(It{me.c0, fails, lk}); st.push_back(It{me.c1, fails, lk});
            __builtin_prefetch(pn + me.c0); __builtin_prefetch(pn + me.c1);
            if (o.fromBits) {
                const int f = M.frameOf[t1];
                if ((frameBits[f >> 5] >> (f & 31)) & 1u)                  // (see place_replay_one)
                    for (int32_t ch : {me.c0, me.c1})
                        if (M.frameOf[ch] != f) o.fromBits[M.frameOf[ch] >> 5] |= 1u << (M.frameOf[ch] & 31);
            }
        }
    }
    int k = 0;
    for (int i = 0; i < nSl; i++)
        if (slL[i] >= bestLK - P.thrFilter) { slN[k] = slN[i]; slL[k] = slL[i]; k++; }
    nSl = k;
    for (int i = 0; i < nSl; i++) { const int f = M.frameOf[slN[i]]; o.slShort[i] = (frameBits[f >> 5] >> (f & 31)) & 1u; }
    const int fb = M.frameOf[status == 1 ? minorNode : bestNode];
    o.bestShort[0] = (frameBits[fb >> 5] >> (fb & 31)) & 1u;
    o.status[0] = status; o.minorNode[0] = minorNode; o.bestNode[0] = bestNode;
    o.bestLK[0] = bestLK; o.originalLK[0] = originalLK;
    o.nAppend[0] = nAppend; o.missed[0] = missed; o.nShort[0] = nSl;
}

template <class T> static int d2h_vec(maple_ctx *c, std::vector<T> &dst, const T *src, size_t n)
{
    dst.resize(n);
    if (n) HIPCK(c, hipMemcpyAsync(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    return MAPLE_OK;
}

extern "C" int maple_placement_prepare(maple_ctx *c, const maple_placement_params *pp)
{
    if (!c || !pp) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    if (!c->tree_set) return fail(c, MAPLE_ERR_STATE, "maple_tree_upload has not been called");
    if (c->tree_stale && (!c->place->valid || c->place->effNon0 != pp->effectivelyNon0BLen)) TRY(tree_rebuild_from_host(c));
    TRY(place_meta(c, pp->effectivelyNon0BLen));
    PlaceMeta &M = *c->place;
    if (M.rootVect < 0) {
        const int32_t root = c->dtree.root;
        const double zero = 0.0;
        const uint8_t nt = 0;
        const int64_t off[2] = {0, c->h_tree_mut[root] >= 0 ? 1 : 0};
        const int32_t path[1] = {c->h_tree_mut[root]};
        TRY(maple_root_vector_batch(c, 1, &c->h_tree_lower[root], &zero, &nt, off, path, &M.rootVect));
    }
    return MAPLE_OK;
}

// ---- score rows made AHEAD of a serial placement loop (M:11692-11752) --------------------------------------------------
// The loop places one sample, edits the tree, places the next: every search scores its sample against every branch of the
// tree as it is then -- one launch of the one-lane-per-branch kernel per sample, 1.6 ms at 1 000 000 tips, most of it the
// latency of a kernel that holds one query.  But a placement changes a handful of lists (updatePartials stops where the lists
// stop changing: 6.5 nodes per sample on the 1 000 000-tip tree), and a branch whose list did not change scores what it scored
// before.  maple_placement_ahead scores the next K samples against the current tree in ONE launch of the batch kernel
// (k_append_queries_lds: a tile of 64 candidate lists staged in LDS for 512 queries) and the minor-sequence tests likewise,
// into page-locked host tables the kernels write directly; maple_tree_patch notes the columns whose list changed and the
// columns it adds; the next single-query search of one of those samples first brings the rows of ALL samples still waiting up
// to date (one small launch: waiting rows x changed columns), then runs the reference's traversal over its own row -- the same
// scores the search would have computed, so the same result (tests/test_hip_search.py compares the two loops).
// Only on trees without MAT reference frames (one frame: the query needs no re-expression per frame).
// ---- the rows of maple_placement_ahead by EXPANSION instead of by scoring every branch --------------------------------------
// The traversal of a placement search (M:7972-8100) visits ~10 000 of the 1 170 000 branches of the 1 000 000-tip tree; scoring
// every branch for every sample is a hundred times the work that is read.  What a traversal visits depends on the running
// best -- but only through rules that are MONOTONE in it: with `pathBest` = the best score among a node's ancestors (never
// above the real running best when the node is visited) and failedPasses reset whenever a score reaches pathBest, every node
// the reference's traversal visits is visited (the frontier tier of the SPR search, frontier.hip, rests on the same argument).
// So: all samples of a batch are walked down the tree together, level by level, an item = (sample, node, score of the parent,
// failedPasses, pathBest); an item scores its branch (one lane, append_walk -- the scoring kernels' arithmetic) into the
// sample's row and pushes the node's children if the permissive rule lets it.  Columns nobody reached keep PLACE_NO_SCORE; the
// host traversal that meets one -- the tree changed under the batch in a way that leads it elsewhere -- has the row scored in
// full (PlaceAhead::fallbacks).
#define PLACE_NO_SCORE_BITS 0x7ff8dead0badc0deull
struct PEItem { int32_t r, node, fails, pad; double parentLK, pathBest; };
struct PECtr { unsigned long long lo, hi, used, cap, overflow, pad[3]; };

__global__ __launch_bounds__(256) void k_pe_fill(unsigned long long *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = PLACE_NO_SCORE_BITS;
}
// the two children of the root for every sample (M:7958-7970); the root vector's score is in the row already ([ld - 1])
__global__ __launch_bounds__(256) void k_pe_seed(int K, const int32_t *pn, int root, const double *table, long long ld, PEItem *items, PECtr *ctr)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= K) return;
    const int c0 = pn[4 * root + 2], c1 = pn[4 * root + 3];
    if (c0 < 0) return;
    const double lk = table[(long long)r * ld + (ld - 1)];
    const unsigned long long at = atomicAdd(&ctr->used, 2ull);
    if (at + 2 > ctr->cap) { ctr->overflow = 1; return; }
    items[at] = PEItem{r, c0, 0, 0, lk, lk};
    items[at + 1] = PEItem{r, c1, 0, 0, lk, lk};
}
__global__ void k_pe_snap(PECtr *ctr)
{
    ctr->lo = ctr->hi;
    ctr->hi = ctr->used < ctr->cap ? ctr->used : ctr->cap;
}
template <bool RV, bool U, bool SS>
__global__ MAPLE_APPEND_ATTR void k_pe_level(const DevModel *__restrict__ mp, ArenaView av, const int32_t *qList, const int32_t *pn,
                                             const int32_t *candList, PlaceParams P, double bLen, PEItem *items, PECtr *ctr, double *table,
                                             long long ld)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const unsigned long long lo = ctr->lo, hi = ctr->hi;
    for (unsigned long long i = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (unsigned long long)gridDim.x * blockDim.x) {
        const PEItem it = items[i];
        const int4 me = *reinterpret_cast<const int4 *>(pn + 4 * (long long)it.node);     // candCol, leafCol, c0, c1
        double lk = it.parentLK, pathBest = it.pathBest;
        int fails = it.fails;
        if (me.x >= 0) {
            lk = append_walk(c, list_ref(av, candList[me.x]), list_ref(av, qList[it.r]), true, bLen);
            table[(long long)it.r * ld + me.x] = lk;
            if (lk >= pathBest) { pathBest = lk; fails = 0; }              // (the reference resets on the REAL best, which is no lower: M:8065)
            else if (lk < it.parentLK - P.thrConsec) fails++;             // M:8076-8077
        }
        const bool within = lk > pathBest - P.thrLK;
        const bool go = P.strict ? (fails <= P.allowedFails && within) : (fails <= P.allowedFails || within);   // M:8080-8093
        const bool push = go && me.z >= 0;
        const unsigned long long pm = __ballot(push);                      // (one atomic per wavefront, not per item)
        if (push) {
            const int lane = threadIdx.x & 63, leader = (int)__ffsll((long long)pm) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(&ctr->used, 2ull * (unsigned long long)__popcll(pm));
            base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), leader, 64) << 32) | (uint32_t)__shfl((int)base, leader, 64);
            const unsigned long long at = base + 2ull * (unsigned long long)__popcll(pm & ((1ull << lane) - 1ull));
            if (at + 2 > ctr->cap) ctr->overflow = 1;
            else {
                items[at] = PEItem{it.r, me.z, fails, 0, lk, pathBest};
                items[at + 1] = PEItem{it.r, me.w, fails, 0, lk, pathBest};
            }
        }
    }
}


// ---- the traversal of the next announced sample, ahead of its search (PlaceAhead::Spec) -------------------------------------
// one lane per changed column: out[k] = appendProbNode(list[k], the sample's list) -- no tile counters, no timing slots: nothing
// of the context is touched, the launch comes from the speculating thread
template <bool RV, bool U, bool SS>
__global__ MAPLE_APPEND_ATTR void k_ahead_cols(const DevModel *__restrict__ mp, ArenaView av, const int32_t *qList, int n, const int32_t *lists,
                                               double bLen, double *out)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const ListRef q = list_ref(av, qList[0]);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) out[k] = append_walk(c, list_ref(av, lists[k]), q, true, bLen);
}
#define MAPLE_SPEC_CAP 32768             // changed columns / leaves a speculative traversal takes (more: no speculation)
static void ahead_spec_body(maple_ctx *c, PlaceParams P, double bLen, int onlyIdentical, bool rv_, bool u_, bool ss_)
{
    PlaceAhead &A = *c->ahead;
    PlaceAhead::Spec &S = A.spec;
    const PlaceMeta &M = *c->place;
    S.status = -100;
    if (hipSetDevice(c->device) != hipSuccess) return;
    const size_t n = S.cols.size(), nl = S.leafCols.size();
    uint8_t *const hPatchM = (uint8_t *)(A.hSpecPatch + n), *const dPatchM = (uint8_t *)(A.dSpecPatch + n);
    if (n) {
        if (hipMemcpyAsync(A.dSpecLists.p, S.lists.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, A.specStream) != hipSuccess) return;
        const int grid = (int)std::min<size_t>(1024, (n + MAPLE_BLOCK - 1) / MAPLE_BLOCK);
#define SPEC_COLS(RV, U, SS) k_ahead_cols<RV, U, SS><<<grid, MAPLE_BLOCK, 0, A.specStream>>>(c->d_model, view(c), A.dQ.p + S.row, (int)n, A.dSpecLists.p, bLen, A.dSpecPatch)
        if (!rv_ && !u_) SPEC_COLS(false, false, false);
        else if (rv_ && !u_) SPEC_COLS(true, false, false);
        else if (!rv_ && u_ && !ss_) SPEC_COLS(false, true, false);
        else if (!rv_ && u_ && ss_) SPEC_COLS(false, true, true);
        else if (rv_ && u_ && !ss_) SPEC_COLS(true, true, false);
        else SPEC_COLS(true, true, true);
#undef SPEC_COLS
    }
    if (nl) {
        if (hipMemcpyAsync(A.dSpecLeaf.p, S.leafLists.data(), nl * sizeof(int32_t), hipMemcpyHostToDevice, A.specStream) != hipSuccess) return;
        hipLaunchKernelGGL(k_place_minor, dim3((int)std::min<size_t>(1024, (nl + MAPLE_BLOCK - 1) / MAPLE_BLOCK)), dim3(MAPLE_BLOCK), 0, A.specStream, c->lRef,
                           view(c), 1, 1, A.dQ.p + S.row, (int)nl, A.dSpecLeaf.p, (const int32_t *)nullptr, onlyIdentical, dPatchM, (long long)nl,
                           (const int32_t *)nullptr);
    }
    if (hipGetLastError() != hipSuccess) return;
    if (hipStreamSynchronize(A.specStream) != hipSuccess) return;
    if (hipStreamSynchronize(A.copyStream) != hipSuccess) return;         // (the row itself: set off by the search before)
    double *const row = A.hRow[S.buf];
    for (size_t i = 0; i < n; i++) row[S.cols[i]] = A.hSpecPatch[i];
    uint8_t *const mrow = (uint8_t *)A.hMinor + (size_t)S.row * A.ldL;
    for (size_t i = 0; i < nl; i++) mrow[S.leafCols[i]] = hPatchM[i];
    const size_t SL = MAPLE_PLACE_SHORTLIST;
    S.hi.assign(6 + SL, 0); S.hf.assign(2 + SL, 0.0); S.hb.assign(1 + SL, 0);
    PlaceOut o;
    int32_t *ib = S.hi.data();
    o.status = ib; o.minorNode = ib + 1; o.bestNode = ib + 2; o.nAppend = ib + 3; o.missed = ib + 4; o.nShort = ib + 5; o.slNode = ib + 6;
    o.bestLK = S.hf.data(); o.originalLK = S.hf.data() + 1; o.slLK = S.hf.data() + 2;
    o.bestShort = S.hb.data(); o.slShort = S.hb.data() + 1;
    o.fromBits = nullptr;
    place_replay_ptr(c, M, P, row, (int)(A.ld - 1), mrow, 1, o, A.sparse, A.visitEpoch.data(), S.id);
    S.status = o.status[0] == -7 ? -7 : 0;
}
// main thread, at the end of a search that took a row: the traversal of the sample after it sets off
// (`row`: the sample after the one being searched -- the thread runs next to that search's own traversal, if it makes one, and
// to everything the caller does until its maple_tree_patch)
static void ahead_spec_kick(maple_ctx *c, int32_t row, const PlaceParams &P, const maple_placement_params *pp)
{
    PlaceAhead &A = *c->ahead;
    const PlaceMeta &M = *c->place;
    A.join();
    A.spec.row = -1;
    if (!A.active || row >= A.K || c->tuning.noAheadSpeculation || A.rootDirty || !A.specStream || !A.hSpecPatch) return;
    const int buf = A.buf_of(row);
    if (buf < 0) return;                                                  // (its row is not on its way)
    auto uniq = [](std::vector<int32_t> &v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); };
    uniq(A.dirtyCols); uniq(A.dirtyLeaves);
    if (A.dirtyCols.size() > MAPLE_SPEC_CAP || A.dirtyLeaves.size() > MAPLE_SPEC_CAP) return;
    PlaceAhead::Spec &S = A.spec;
    S.cols = A.dirtyCols; S.leafCols = A.dirtyLeaves;
    S.lists.resize(S.cols.size()); S.leafLists.resize(S.leafCols.size());
    for (size_t i = 0; i < S.cols.size(); i++) S.lists[i] = M.h_candList[S.cols[i]];
    for (size_t i = 0; i < S.leafCols.size(); i++) S.leafLists[i] = M.h_leafList[S.leafCols[i]];
    S.touched.clear(); S.rootTouched = false;
    S.row = row; S.buf = buf; S.id = ++A.specSeq; S.status = -1;
    if (A.visitEpoch.size() < M.h_pn.size()) A.visitEpoch.resize(M.h_pn.size(), 0);
    try {
        S.th = std::thread(ahead_spec_body, c, P, pp->oneMutBLen, (int)pp->onlyFindIdentical, c->dm.useRateVariation != 0, c->dm.usingErrorRate != 0,
                           c->dm.errorRateSiteSpecific != 0);
    } catch (...) { S.row = -1; }                                         // (no thread to be had: the search makes its traversal itself)
}
// ... and at the start of the next search: is the traversal made for this row good as it is?
static bool ahead_spec_usable(maple_ctx *c)
{
    PlaceAhead &A = *c->ahead;
    A.join();
    PlaceAhead::Spec &S = A.spec;
    if (S.row != A.next) return false;
    bool ok = S.status == 0 && !S.rootTouched && !A.rootDirty;
    for (size_t i = 0; ok && i < S.touched.size(); i++) {
        const int32_t v = S.touched[i];
        if ((size_t)v < A.visitEpoch.size() && A.visitEpoch[v] == S.id) ok = false;   // (the placement in between changed a node it had visited)
    }
    if (ok) A.specUsed++; else A.specDropped++;
    return ok;
}

// The row of the sample searched NOW (A.next), current: *rowOut = that row in page-locked host memory.  The device table keeps
// the rows as they were made (the tree of the batch's start); what changed since -- every column maple_tree_patch noted since the
// batch began, a few per placement -- is scored for THIS sample in one launch (1 x changed columns, straight into a page-locked
// patch buffer) and written over the host copy of its row, which the copy engine brought over during the placement before.
static int ahead_refresh(maple_ctx *c, const double **rowOut, bool rowIsCurrent = false)
{
    PlaceAhead &A = *c->ahead;
    PlaceMeta &M = *c->place;
    // the rows of the two samples after this one set off (the device rows never change: no order to keep); a buffer is free if it
    // holds neither this sample's row nor the next one's
    auto prefetch = [&]() -> int {
        for (int32_t r = A.next + 1; r <= A.next + 2 && r < A.K; r++) {
            if (A.buf_of(r) >= 0) continue;
            const int nb = A.free_buf(A.next, A.next + 1);
            if (nb < 0) break;
            HIPCK(c, hipMemcpyAsync(A.hRow[nb], A.dTable.p + (size_t)r * A.ld, (size_t)A.ld * sizeof(double), hipMemcpyDeviceToHost, A.copyStream));
            A.rowInBuf[nb] = r;
        }
        return MAPLE_OK;
    };
    if (rowIsCurrent) {                                                    // (a traversal made ahead has used the row already: only what follows it)
        TRY(prefetch());
        *rowOut = A.hRow[A.spec.buf];
        return MAPLE_OK;
    }
    auto uniq = [](std::vector<int32_t> &v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); };
    uniq(A.dirtyCols); uniq(A.dirtyLeaves);
    int buf = A.buf_of(A.next);
    const bool prefetched = buf >= 0;
    if (!prefetched) buf = 0;
    std::vector<int32_t> cols(A.dirtyCols), lists(A.dirtyCols.size());
    for (size_t i = 0; i < lists.size(); i++) lists[i] = M.h_candList[cols[i]];
    if (A.rootDirty) { cols.push_back((int32_t)(A.ld - 1)); lists.push_back(M.rootVect); }   // (the root's list changed: its new root vector)
    const size_t n = cols.size(), nl = A.dirtyLeaves.size();
    if (n * sizeof(double) + nl + 64 > A.capPatch) {
        if (A.hPatch) (void)hipHostFree(A.hPatch);
        A.hPatch = nullptr; A.capPatch = 0;
        const size_t want = 2 * (n * sizeof(double) + nl) + (1 << 16);
        HIPCK(c, hipHostMalloc((void **)&A.hPatch, want, hipHostMallocDefault));
        void *dp = nullptr;
        HIPCK(c, hipHostGetDevicePointer(&dp, A.hPatch, 0));
        A.dPatch = (double *)dp; A.capPatch = want;
    }
    uint8_t *const hPatchM = (uint8_t *)(A.hPatch + n), *const dPatchM = (uint8_t *)(A.dPatch + n);
    if (n) {
        TRY(h2d(c, A.dLists, lists.data(), n));
        TRY(launch_append_queries(c, c->stream, 1, A.dQ.p + A.next, (int)n, A.dLists.p, 1, A.pp.oneMutBLen, A.dPatch, (long long)n, nullptr,
                                  nullptr, nullptr, MAPLE_K_PLACE_SCORE, 0.0));
        A.refreshes++; A.refreshedPairs += (long long)n;
    }
    if (nl) {
        std::vector<int32_t> ll(nl);
        for (size_t i = 0; i < nl; i++) ll[i] = M.h_leafList[A.dirtyLeaves[i]];
        TRY(h2d(c, A.dCols, ll.data(), nl));
        hipLaunchKernelGGL(k_place_minor, dim3(grid_for((int)nl)), dim3(MAPLE_BLOCK), 0, c->stream, c->lRef, view(c), 1, 1, A.dQ.p + A.next, (int)nl,
                           A.dCols.p, (const int32_t *)nullptr, A.pp.onlyFindIdentical, dPatchM, (long long)nl, (const int32_t *)nullptr);
        HIPCK(c, hipGetLastError());
    }
    if (!prefetched) {                                                      // (the first row of a batch: copied now)
        HIPCK(c, hipMemcpyAsync(A.hRow[buf], A.dTable.p + (size_t)A.next * A.ld, (size_t)A.ld * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        A.rowInBuf[buf] = A.next;
    }
    if (n || nl || !prefetched) HIPCK(c, hipStreamSynchronize(c->stream));
    if (prefetched) HIPCK(c, hipStreamSynchronize(A.copyStream));           // (long done: it was queued a whole placement ago)
    double *const row = A.hRow[buf];
    for (size_t i = 0; i < n; i++) row[cols[i]] = A.hPatch[i];
    uint8_t *const mrow = (uint8_t *)A.hMinor + (size_t)A.next * A.ldL;
    for (size_t i = 0; i < nl; i++) mrow[A.dirtyLeaves[i]] = hPatchM[i];
    TRY(prefetch());
    *rowOut = row;
    return MAPLE_OK;
}

// rows [0, K) of the table by expansion (see above); false in *done: not possible here (no room for the device copy of the node
// records) -- the caller scores every branch instead
static int ahead_expand(maple_ctx *c, const maple_placement_params *pp, bool *done)
{
    PlaceAhead &A = *c->ahead;
    PlaceMeta &M = *c->place;
    *done = false;
    const int32_t n = c->dtree.n, K = A.K;
    if (!M.d_pn.cap) {                                                    // (room for the nodes the batches to come add)
        HIPCK(c, M.d_pn.reserve_exact((size_t)4 * ((size_t)n + 262144)));
        HIPCK(c, hipMemcpyAsync(M.d_pn.p, M.h_pn.data(), (size_t)n * sizeof(PlaceMeta::PNode), hipMemcpyHostToDevice, c->stream));
    }
    if ((size_t)4 * ((size_t)n + 2 * (size_t)K) + 4 >= M.d_pn.cap) return MAPLE_OK;
    static_assert(sizeof(PlaceMeta::PNode) == 16, "PNode");
    const size_t capItems = std::max<size_t>((size_t)1 << 20, (size_t)K * 65536);
    HIPCK(c, A.dItems.reserve_exact(capItems * sizeof(PEItem)));
    HIPCK(c, A.dCtr.reserve(sizeof(PECtr) / 8));
    PECtr h{};
    h.cap = capItems;
    HIPCK(c, hipMemcpyAsync(A.dCtr.p, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
    k_pe_fill<<<4096, 256, 0, c->stream>>>((unsigned long long *)A.dTable.p, (size_t)K * (size_t)A.ld);
    HIPCK(c, hipGetLastError());
    // the root vector's score of every sample, at [ld - 1]
    const int32_t rootCol = (int32_t)(A.ld - 1), rootVect = M.rootVect;
    TRY(h2d(c, A.dCols, &rootCol, 1));
    TRY(h2d(c, A.dLists, &rootVect, 1));
    TRY(launch_append_queries(c, c->stream, K, A.dQ.p, 1, A.dLists.p, 1, pp->oneMutBLen, A.dTable.p, A.ld, A.dCols.p, nullptr, nullptr,
                              MAPLE_K_PLACE_SCORE, 0.0, nullptr, nullptr, nullptr, 0, 1, nullptr, true));
    PlaceParams P{};
    P.thrLK = pp->thresholdLogLK; P.thrOpt = pp->thresholdLogLKoptimization; P.thrConsec = pp->thresholdLogLKconsecutivePlacement;
    P.allowedFails = pp->allowedFails; P.strict = pp->strictStopRules;
    PEItem *items = (PEItem *)A.dItems.p;
    PECtr *ctr = (PECtr *)A.dCtr.p;
    k_pe_seed<<<(K + 255) / 256, 256, 0, c->stream>>>(K, M.d_pn.p, c->dtree.root, A.dTable.p, A.ld, items, ctr);
    HIPCK(c, hipGetLastError());
    // as many levels as the tree is deep (the placements since the tables were made add at most one level each: counted in)
    // (noAheadExpansion = 2, for the tests: the expansion stops after six levels, so that nearly every traversal meets a column
    // without a score and takes the full-row path)
    const int levels = c->tuning.noAheadExpansion == 2 ? 6
                       : M.maxDepth + 8 + (int)std::min<size_t>(4096, M.h_pn.size() - std::min(M.h_pn.size(), M.order.size()));
    int launchedLevels = 0;
    for (int l = 0; l < levels; l++) {
        k_pe_snap<<<1, 1, 0, c->stream>>>(ctr);
        DISPATCH3(c, k_pe_level, <<<2048, MAPLE_BLOCK, 0, c->stream>>>(c->d_model, view(c), A.dQ.p, M.d_pn.p, M.d_candList.p, P, pp->oneMutBLen, items, ctr,
                                                                        A.dTable.p, A.ld));
        launchedLevels++;
        if ((l & 15) == 15) {                                             // (the host looks every 16 levels whether anything is left)
            HIPCK(c, hipMemcpyAsync(&h, A.dCtr.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
            HIPCK(c, hipStreamSynchronize(c->stream));
            if (h.used <= h.hi || h.overflow) break;                        // nothing was pushed by the last level that ran
        }
    }
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(&h, A.dCtr.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    A.expanded += (long long)std::min(h.used, h.cap);
    if (c->tuning.verbose > 1)
        fprintf(stderr, "[maple] rows ahead by expansion: %d samples, %llu items (%.0f per sample), %d levels launched%s, last level %llu items\n", K,
                (unsigned long long)h.used, (double)h.used / K, launchedLevels, h.overflow ? ", item pool ran over" : "", (unsigned long long)(h.hi - h.lo));
    *done = true;
    return MAPLE_OK;
}

extern "C" int maple_placement_ahead(maple_ctx *c, int32_t nQ, const int32_t *qLists, const maple_placement_params *pp, int32_t *nTaken)
{
    if (!c || nQ < 0 || (nQ && !qLists) || !pp || !nTaken) return MAPLE_ERR_ARG;
    *nTaken = 0;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    if (!c->tree_set) return fail(c, MAPLE_ERR_STATE, "maple_tree_upload has not been called");
    if (c->ahead) {
        c->ahead->join();
        c->ahead->spec.row = -1;
        if (c->ahead->copyStream) HIPCK(c, hipStreamSynchronize(c->ahead->copyStream));
        c->ahead->active = false;
    }
    if (nQ == 0) return MAPLE_OK;
    TRY(check_ids(c, nQ, qLists, false, "qLists"));
    TRY(maple_placement_prepare(c, pp));                                 // (tables of the current tree, the root vector)
    PlaceMeta &M = *c->place;
    if (M.nF != 1) return MAPLE_OK;                                      // (reference frames: the single-query path as it is; nTaken = 0)
    if (!c->ahead) c->ahead = new PlaceAhead();
    PlaceAhead &A = *c->ahead;
    const int32_t nC = (int32_t)M.cand.size(), nCols = nC + 1, nL = (int32_t)M.leaves.size();
    // rows: what fits in a quarter of the free device memory (1 000 000 tips: 2 M columns, 16 MB per row -- 512 rows are 8.4 GB)
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) freeB = (size_t)8 << 30;
    int32_t K = nQ;
    const int64_t room = (int64_t)((freeB + A.dTable.cap * sizeof(double)) / 4);
    for (;;) {
        A.ld = ((int64_t)nCols + 2 * (int64_t)K + 64 + 15) / 16 * 16;
        A.ldL = ((int64_t)nL + (int64_t)K + 64 + 63) / 64 * 64;
        if (K <= 1 || (int64_t)K * A.ld * 8 <= room) break;
        K = std::max(1, K / 2);
    }
    // (with room: the columns grow by ~2 per placement, and a table that grows by a sliver per batch would be 5 GB freed and
    // allocated again every 512 samples)
    if ((size_t)K * (size_t)A.ld > A.dTable.cap) HIPCK(c, A.dTable.reserve_exact((size_t)K * ((size_t)A.ld + (size_t)A.ld / 8 + 65536)));
    if ((size_t)A.ld > A.capRow) {
        for (double *&r : A.hRow) { if (r) (void)hipHostFree(r); r = nullptr; }
        A.capRow = 0;
        const size_t want = (size_t)A.ld + (size_t)A.ld / 8 + 4096;
        for (double *&r : A.hRow) HIPCK(c, hipHostMalloc((void **)&r, want * sizeof(double), hipHostMallocDefault));
        A.capRow = want;
    }
    const size_t needM = (size_t)K * (size_t)A.ldL;
    if (needM > A.capMinor) {
        if (A.hMinor) (void)hipHostFree(A.hMinor);
        A.hMinor = nullptr; A.capMinor = 0;
        HIPCK(c, hipHostMalloc(&A.hMinor, needM + needM / 8, hipHostMallocDefault));
        A.capMinor = needM + needM / 8;
    }
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, A.hMinor, 0) != hipSuccess || !dp) { (void)hipGetLastError(); return MAPLE_OK; }   // (no zero copy: no rows ahead)
    A.dMinor = (uint8_t *)dp;
    if (!A.copyStream) HIPCK(c, hipStreamCreateWithFlags(&A.copyStream, hipStreamNonBlocking));
    if (!A.specStream) {
        HIPCK(c, hipStreamCreateWithFlags(&A.specStream, hipStreamNonBlocking));
        HIPCK(c, A.dSpecLists.reserve_exact(MAPLE_SPEC_CAP)); HIPCK(c, A.dSpecLeaf.reserve_exact(MAPLE_SPEC_CAP));
        A.capSpecPatch = (size_t)MAPLE_SPEC_CAP * (sizeof(double) + 1) + 64;
        HIPCK(c, hipHostMalloc((void **)&A.hSpecPatch, A.capSpecPatch, hipHostMallocDefault));
        void *dps = nullptr;
        HIPCK(c, hipHostGetDevicePointer(&dps, A.hSpecPatch, 0));
        A.dSpecPatch = (double *)dps;
    }
    A.K = K; A.next = 0; A.pp = *pp;
    A.rowInBuf[0] = A.rowInBuf[1] = A.rowInBuf[2] = -1;
    A.q.assign(qLists, qLists + K);
    A.dirtyCols.clear(); A.dirtyLeaves.clear(); A.rootDirty = false;
    TRY(h2d(c, A.dQ, A.q.data(), (size_t)K));
    HIPCK(c, A.dCols.reserve(4096)); HIPCK(c, A.dLists.reserve(4096));
    const int32_t rootVect = M.rootVect;
    HIPCK(c, hipMemcpyAsync(M.d_candList.p + nC, &rootVect, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    const bool dbgA = c->tuning.verbose > 1;
    const auto tA0 = std::chrono::steady_clock::now();
    auto msSince = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t).count() * 1e-3; };
    bool expanded = false;
    if (c->tuning.noAheadExpansion != 1) TRY(ahead_expand(c, pp, &expanded));
    A.sparse = expanded;
    if (!expanded) {
        TRY(launch_append_queries(c, c->stream, K, A.dQ.p, nCols, M.d_candList.p, 1, pp->oneMutBLen, A.dTable.p, A.ld, nullptr, nullptr, nullptr,
                                  MAPLE_K_PLACE_SCORE, 0.0));
        // the root vector's score sits in the column behind the last branch -- a column the first new branch will take: to the end of the row
        HIPCK(c, hipMemcpy2DAsync(A.dTable.p + (A.ld - 1), (size_t)A.ld * sizeof(double), A.dTable.p + nC, (size_t)A.ld * sizeof(double), sizeof(double),
                                  (size_t)K, hipMemcpyDeviceToDevice, c->stream));
    }
    if (dbgA) { HIPCK(c, hipStreamSynchronize(c->stream)); fprintf(stderr, "[maple] rows ahead: %d x %d scores %s in %.1f ms\n", K, nCols, expanded ? "(by expansion)" : "(every branch)", msSince(tA0)); }
    const auto tA1 = std::chrono::steady_clock::now();
    if (nL > 0) {
        hipLaunchKernelGGL(k_place_minor, dim3(grid_for((int)std::min<long long>((long long)K * nL, 1 << 30))), dim3(MAPLE_BLOCK), 0,
                           c->stream, c->lRef, view(c), K, 1, A.dQ.p, nL, M.d_leafList.p, (const int32_t *)nullptr,
                           pp->onlyFindIdentical, A.dMinor, (long long)A.ldL, (const int32_t *)nullptr);
        HIPCK(c, hipGetLastError());
    }
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (dbgA) fprintf(stderr, "[maple] rows ahead: %d x %d minor-sequence tests in %.1f ms\n", K, nL, msSince(tA1));
    A.active = true;
    *nTaken = K;
    return MAPLE_OK;
}

extern "C" int maple_placement_ahead_stats(maple_ctx *c, int64_t *out5)
{
    if (!c || !out5) return MAPLE_ERR_ARG;
    for (int i = 0; i < 5; i++) out5[i] = 0;
    if (!c->ahead) return MAPLE_OK;
    const PlaceAhead &A = *c->ahead;
    out5[0] = A.searches; out5[1] = A.fallbacks; out5[2] = A.expanded; out5[3] = A.specUsed; out5[4] = A.specDropped;
    return MAPLE_OK;
}

// what the computePlacementSupportOnly=True exit of findBestParentForNewSample hands back (M:8101-8290), CSR over queries
struct SupportsOut {
    double thrOptTopo, minBranchSupport;
    int64_t cap;
    int64_t *off;                      // [nQ + 1]
    int32_t *node;                     // possiblePlacements: node, support, (top, bottom, appending)
    double *support, *blen3;
    int32_t *bestTotalLh;              // [nQ] list id of bestPlacementTotalLh (-1 = the reference's empty list)
};

static int placement_search_impl(maple_ctx *c, int32_t nQ, const int32_t *qLists, const maple_placement_params *pp,
                                 int32_t *bestNode, double *bestScore, double *blen3, int32_t *bestDiffs,
                                 int32_t *nAppend, int32_t *status, SupportsOut *sup);

extern "C" int maple_placement_search_batch(maple_ctx *c, int32_t nQ, const int32_t *qLists, const maple_placement_params *pp,
                                            int32_t *bestNode, double *bestScore, double *blen3, int32_t *bestDiffs,
                                            int32_t *nAppend, int32_t *status)
{
    if (!c || nQ < 0 || !qLists || !pp || !bestNode || !bestScore || !blen3 || !bestDiffs || !nAppend || !status)
        return MAPLE_ERR_ARG;
    return placement_search_impl(c, nQ, qLists, pp, bestNode, bestScore, blen3, bestDiffs, nAppend, status, nullptr);
}

extern "C" int maple_placement_supports_batch(maple_ctx *c, int32_t nQ, const int32_t *qLists, const maple_placement_params *pp,
                                              double thresholdLogLKoptimizationTopology, double minBranchSupport, int64_t cap,
                                              int64_t *outOff, int32_t *outNode, double *outSupport, double *outBlen3,
                                              int32_t *bestTotalLh, int32_t *status)
{
    if (!c || nQ < 0 || !qLists || !pp || cap < 0 || !outOff || !outNode || !outSupport || !outBlen3 || !bestTotalLh || !status)
        return MAPLE_ERR_ARG;
    SupportsOut sup{thresholdLogLKoptimizationTopology, minBranchSupport, cap, outOff, outNode, outSupport, outBlen3, bestTotalLh};
    outOff[0] = 0;
    std::vector<int32_t> bn(nQ), bd(nQ), na(nQ);
    std::vector<double> bs(nQ), bl(3 * (size_t)nQ);
    return placement_search_impl(c, nQ, qLists, pp, bn.data(), bs.data(), bl.data(), bd.data(), na.data(), status, &sup);
}

static int placement_search_impl(maple_ctx *c, int32_t nQ, const int32_t *qLists, const maple_placement_params *pp,
                                 int32_t *bestNode, double *bestScore, double *blen3, int32_t *bestDiffsOut,
                                 int32_t *nAppend, int32_t *status, SupportsOut *sup)
{
    if (nQ == 0) return MAPLE_OK;
    if (c->ahead) c->ahead->join();                                       // (a speculative traversal reads the tables this call may rebuild)
    const auto tEntry = std::chrono::steady_clock::now();
    int32_t *const bestDiffs = bestDiffsOut;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    if (!c->tree_set) return fail(c, MAPLE_ERR_STATE, "maple_tree_upload has not been called");
    TRY(check_ids(c, nQ, qLists, false, "qLists"));
    // a tree changed through maple_tree_patch: single queries (the serial placement phase) run on the patched columns and the
    // host's own tree; anything else rebuilds the tables first
    if (c->tree_stale && (nQ > 4 || !c->place->valid || c->place->effNon0 != pp->effectivelyNon0BLen))
        TRY(tree_rebuild_from_host(c));
    TRY(place_meta(c, pp->effectivelyNon0BLen));
    PlaceMeta &M = *c->place;
    const int32_t nF = M.nF, nC = (int32_t)M.cand.size(), nCols = nC + 1, nL = (int32_t)M.leaves.size();
    const int32_t root = c->dtree.root;
    const auto &mut = c->h_tree_mut;
    const bool dbg = c->tuning.verbose > 1;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tus = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return (long long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count();
    };
    auto t0 = tnow();
    // rootVector(probVect[root], False, False, tree, root), M:7958: does not depend on the query
    // (kept across calls while the arena is not released below it: call maple_placement_prepare before taking a mark)
    if (M.rootVect < 0) {
        const double zero = 0.0;
        const uint8_t nt = 0;
        const int64_t off[2] = {0, mut[root] >= 0 ? 1 : 0};
        const int32_t path[1] = {mut[root]};
        TRY(maple_root_vector_batch(c, 1, &c->h_tree_lower[root], &zero, &nt, off, path, &M.rootVect));
    }
    const int32_t rootVect = M.rootVect;
    HIPCK(c, hipMemcpyAsync(M.d_candList.p + nC, &rootVect, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    auto t1 = tnow();
    PlaceParams P;
    P.thrLK = pp->thresholdLogLK; P.thrOpt = pp->thresholdLogLKoptimization; P.thrConsec = pp->thresholdLogLKconsecutivePlacement;
    P.allowedFails = pp->allowedFails; P.strict = pp->strictStopRules;
    P.supportOnly = sup ? 1 : 0;
    P.thrFilter = sup ? std::max(pp->thresholdLogLKoptimization, sup->thrOptTopo) : pp->thresholdLogLKoptimization;
    // queries per chunk: the score matrix stays below 2 GiB and the per-frame query lists below 8 M arena lists
    const int64_t maxCells = (int64_t)1 << 28;
    const int64_t byFrames = std::max<int64_t>(1, ((int64_t)8 << 20) / std::max(nF, 1));
    int32_t chunk = (int32_t)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nQ, byFrames), maxCells / std::max(nCols, 1)));
    {   // ... and what a chunk adds to the arena -- every query re-expressed in every MAT frame, its shortened copies, the
        // refinement's upper lists -- has to fit in a third of what is free (the SPR wide path bounds itself the same way)
        int64_t maxEnt = 1, maxAux = 1;
        for (int q = 0; q < nQ; q++) { maxEnt = std::max<int64_t>(maxEnt, c->h_n_ent[qLists[q]]); maxAux = std::max<int64_t>(maxAux, c->h_n_aux[qLists[q]]); }
        const int64_t perQueryEnt = (int64_t)(nF + 2) * (maxEnt + 32), perQueryAux = (int64_t)(nF + 2) * (maxAux + 16);
        const int64_t byEnt = (c->cap_ent - c->used_ent) / 3 / perQueryEnt, byAux = (c->cap_aux - c->used_aux) / 3 / perQueryAux;
        const int64_t byLists = (c->cap_lists - (int64_t)c->h_n_ent.size()) / 3 / (2 * (int64_t)nF + 8);
        const int64_t byArena = std::max<int64_t>(1, std::min(std::min(byEnt, byAux), byLists));
        if (byArena < chunk) chunk = (int32_t)byArena;
        if (c->tuning.placementChunkMax > 0) chunk = std::max(1, std::min(chunk, (int)c->tuning.placementChunkMax));
    }
    const bool manyChunks = chunk < nQ;
    const int stackCap = M.maxDepth + 4, words = (nF + 31) / 32;
    for (int32_t q0 = 0; q0 < nQ; q0 += chunk) {
        const int32_t nq = std::min(chunk, nQ - q0);
        int64_t chunkMark = 0;
        if (manyChunks) TRY(maple_arena_mark(c, &chunkMark));       // a chunk's temporaries go when it is done (below)
        // ---- the query list in every reference frame, level by level (passGenomeListThroughBranch, M:8082-8092)
        std::vector<int32_t> U((size_t)nq * nF);
        for (int q = 0; q < nq; q++) U[(size_t)q * nF] = qLists[q0 + q];
        {
            int32_t a = 1;
            for (size_t l = 0; l < M.levelStart.size(); l++) {
                const int32_t b = M.levelStart[l];
                if (b > a) {
                    const size_t cnt = (size_t)nq * (b - a);
                    std::vector<int32_t> src(cnt), ml(cnt), out(cnt);
                    std::vector<uint8_t> dirUp(cnt, 0);
                    size_t k = 0;
                    for (int q = 0; q < nq; q++)
                        for (int f = a; f < b; f++, k++) { src[k] = U[(size_t)q * nF + M.frameParent[f]]; ml[k] = mut[M.frameNode[f]]; }
                    TRY(maple_pass_branch_batch(c, (int32_t)cnt, src.data(), ml.data(), dirUp.data(), out.data()));
                    k = 0;
                    for (int q = 0; q < nq; q++)
                        for (int f = a; f < b; f++, k++) U[(size_t)q * nF + f] = out[k];
                }
                a = b;
            }
        }
        auto t2 = tnow();
        // ---- scores, minor tests, traversal
        // (a sample whose rows were made ahead, maple_placement_ahead: no scoring launch -- the waiting rows are brought up to date
        // with the tree's changes, then the traversal reads this sample's row)
        PlaceAhead *const ah = (c->ahead && c->ahead->active && !sup && nQ == 1 && nF == 1 && c->ahead->next < c->ahead->K
                                && c->ahead->q[c->ahead->next] == qLists[0]
                                && memcmp(&c->ahead->pp, pp, sizeof(maple_placement_params)) == 0
                                && (int64_t)nCols < c->ahead->ld - 1 && (int64_t)nL <= c->ahead->ldL) ? c->ahead : nullptr;
        const double *aheadRow = nullptr;
        const bool specUse = ah && ahead_spec_usable(c);                   // (the traversal was made while the sample before was being placed)
        if (ah) TRY(ahead_refresh(c, &aheadRow, specUse));
        // (the NEXT announced sample's traversal sets off now, next to this search's own work: it needs the columns changed up to the
        // placement before this one -- known -- and its row, on its way since the search before)
        std::vector<int32_t> specHi; std::vector<double> specHf; std::vector<uint8_t> specHb;
        if (specUse) { specHi.swap(ah->spec.hi); specHf.swap(ah->spec.hf); specHb.swap(ah->spec.hb); }   // (this row's traversal, made ahead)
        if (ah) ahead_spec_kick(c, ah->next + 1, P, pp);
        if (dbg) fprintf(stderr, "[maple]   %s\n", ah ? "rows made ahead: brought up to date" : "scoring launch");
        DevBuf<int32_t> &dU = c->p_i32[0];
        if (!ah) TRY(h2d(c, dU, U.data(), U.size()));
        HIPCK(c, c->p_score.reserve((size_t)nq * nCols));
        HIPCK(c, c->p_minor.reserve((size_t)nq * std::max(nL, 1)));
        // a handful of queries: the traversal runs on the host (below), and the kernels write the scores it reads straight
        // into page-locked host memory -- they cross PCIe while the kernel is still producing them instead of in a copy after it
        const bool hostReplay = nq <= 4;
        const size_t nS = (size_t)nq * nCols, nM = (size_t)nq * std::max(nL, 1);
        double *hs = nullptr, *scoreOut = c->p_score.p;
        uint8_t *hm = nullptr, *minorOut = c->p_minor.p;
        bool zeroCopy = false;
        int rootCol = nC;
        if (ah) {
            hs = const_cast<double *>(aheadRow);
            hm = (uint8_t *)ah->hMinor + (size_t)ah->next * ah->ldL;
            rootCol = (int)(ah->ld - 1);
            zeroCopy = true;
        } else if (hostReplay) {
            HIPCK(c, c->pin_place.reserve(nS * sizeof(double) + nM));
            hs = (double *)c->pin_place.p;
            hm = (uint8_t *)(hs + nS);
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, hs, 0) == hipSuccess && dp) {
                zeroCopy = true;
                scoreOut = (double *)dp;
                minorOut = (uint8_t *)((double *)dp + nS);
            } else (void)hipGetLastError();
        }
        if (ah) { }
        else if (nF == 1)   // one reference frame: the plain batch kernel (query words staged in LDS) does the same job faster
            TRY(launch_append_queries(c, c->stream, nq, dU.p, nCols, M.d_candList.p, 1, pp->oneMutBLen, scoreOut, nCols,
                                      nullptr, nullptr, nullptr, MAPLE_K_PLACE_SCORE, 0.0));
        else
            TRY(launch_place_score(c, nq, nF, dU.p, nCols, M.d_candList.p, M.d_candFrame.p, 1, pp->oneMutBLen, scoreOut, nCols,
                                   nullptr, nullptr, nullptr));
        if (nL > 0 && !ah) {
            hipLaunchKernelGGL(k_place_minor, dim3(grid_for((int)std::min<long long>((long long)nq * nL, 1 << 30))), dim3(MAPLE_BLOCK), 0,
                               c->stream, c->lRef, view(c), nq, nF, dU.p, nL, M.d_leafList.p, M.d_leafFrame.p,
                               pp->onlyFindIdentical, minorOut, (long long)std::max(nL, 1), nullptr);
            HIPCK(c, hipGetLastError());
        }
        const size_t SL = MAPLE_PLACE_SHORTLIST;
        std::vector<int32_t> hi;
        std::vector<double> hf;
        std::vector<uint8_t> hb;
        std::vector<uint32_t> hFrom;                                       // per query: the "made from a shortened list" bit of every frame
        bool fromLaneMajor = false;                                        // (device replay: word i of query q at [i * nq + q])
        if (hostReplay) {
            // a handful of queries (the sequential placement loop hands over one at a time): one lane's ~1 us per visit
            // would dominate the call, so the scores come back (8 bytes per branch) and the SAME traversal function
            // runs on the host
            if (dbg) { HIPCK(c, hipStreamSynchronize(c->stream)); fprintf(stderr, "[maple]   scoring kernels done after %lld us\n", tus(t2, tnow())); }
            if (!zeroCopy) {
                HIPCK(c, hipMemcpyAsync(hs, c->p_score.p, nS * sizeof(double), hipMemcpyDeviceToHost, c->stream));
                if (nL > 0) HIPCK(c, hipMemcpyAsync(hm, c->p_minor.p, nM, hipMemcpyDeviceToHost, c->stream));
            }
            HIPCK(c, hipStreamSynchronize(c->stream));
            if (dbg) fprintf(stderr, "[maple]   scores on the host after %lld us\n", tus(t2, tnow()));
            hi.assign((size_t)nq * (6 + SL), 0);
            hf.assign((size_t)nq * (2 + SL), 0.0);
            hb.assign((size_t)nq * (1 + SL), 0);
            PlaceOut o;
            int32_t *ib = hi.data();
            o.status = ib; o.minorNode = ib + nq; o.bestNode = ib + 2 * (size_t)nq; o.nAppend = ib + 3 * (size_t)nq;
            o.missed = ib + 4 * (size_t)nq; o.nShort = ib + 5 * (size_t)nq; o.slNode = ib + 6 * (size_t)nq;
            o.bestLK = hf.data(); o.originalLK = hf.data() + nq; o.slLK = hf.data() + 2 * (size_t)nq;
            o.bestShort = hb.data(); o.slShort = hb.data() + nq;
            hFrom.assign((size_t)nq * words, 0u);
            o.fromBits = nF > 1 ? hFrom.data() : nullptr;
            std::vector<double> stL(stackCap);
            std::vector<int16_t> stF(stackCap);
            std::vector<uint32_t> bits(words);
            if (specUse) { hi.swap(specHi); hf.swap(specHf); hb.swap(specHb); }
            for (int q = 0; q < nq && !specUse; q++) {
                // per-query outputs are addressed [.. + q] inside, the work arrays as lane 0 of 1
                PlaceOut oq = o;
                oq.status += q; oq.minorNode += q; oq.bestNode += q; oq.nAppend += q; oq.missed += q; oq.nShort += q;
                oq.bestLK += q; oq.originalLK += q; oq.bestShort += q;
                oq.slNode += (size_t)q * SL; oq.slLK += (size_t)q * SL; oq.slShort += (size_t)q * SL;
                if (oq.fromBits) oq.fromBits += (size_t)q * words;
                place_replay_ptr(c, M, P, hs + (size_t)q * nCols, rootCol, hm + (size_t)q * std::max(nL, 1), nF, oq, ah && ah->sparse);
                if (ah && oq.status[0] == -7) {
                    // the traversal asked for a branch the expansion did not reach (the tree changed under the batch and led it
                    // elsewhere): every branch scored for this sample, as the plain search does, and once more
                    TRY(launch_append_queries(c, c->stream, 1, ah->dQ.p + ah->next, nC, M.d_candList.p, 1, pp->oneMutBLen,
                                              ah->dTable.p + (size_t)ah->next * ah->ld, ah->ld, nullptr, nullptr, nullptr, MAPLE_K_PLACE_SCORE, 0.0));
                    HIPCK(c, hipMemcpyAsync(hs, ah->dTable.p + (size_t)ah->next * ah->ld, (size_t)ah->ld * sizeof(double), hipMemcpyDeviceToHost, c->stream));
                    HIPCK(c, hipStreamSynchronize(c->stream));
                    ah->fallbacks++;
                    place_replay_ptr(c, M, P, hs, rootCol, hm, nF, oq, false);
                }
            }
            if (ah) { ah->searches++; ah->next++; if (ah->next >= ah->K) ah->active = false; }
        } else {
            HIPCK(c, c->p_f64[0].reserve((size_t)nq * stackCap));         // per-depth lastLK
            HIPCK(c, c->p_i16.reserve((size_t)nq * stackCap));            // per-depth fails
            HIPCK(c, c->p_i32[2].reserve((size_t)nq * words));            // frame bits
            HIPCK(c, c->p_i32[3].reserve((size_t)nq * (6 + SL)));         // status, minorNode, bestNode, nAppend, missed, nShort, slNode
            HIPCK(c, c->p_f64[1].reserve((size_t)nq * (2 + SL)));         // bestLK, originalLK, slLK
            HIPCK(c, c->p_u8.reserve((size_t)nq * (1 + SL)));             // bestShort, slShort
            PlaceOut o;
            int32_t *ib = c->p_i32[3].p;
            o.status = ib; o.minorNode = ib + nq; o.bestNode = ib + 2 * (size_t)nq; o.nAppend = ib + 3 * (size_t)nq;
            o.missed = ib + 4 * (size_t)nq; o.nShort = ib + 5 * (size_t)nq; o.slNode = ib + 6 * (size_t)nq;
            double *fb = c->p_f64[1].p;
            o.bestLK = fb; o.originalLK = fb + nq; o.slLK = fb + 2 * (size_t)nq;
            o.bestShort = c->p_u8.p; o.slShort = c->p_u8.p + nq;
            HIPCK(c, c->p_from.reserve((size_t)nq * words));
            o.fromBits = nF > 1 ? (uint32_t *)c->p_from.p : nullptr;
            fromLaneMajor = true;
            hipLaunchKernelGGL(k_place_replay, dim3((nq + 63) / 64), dim3(64), 0, c->stream, M.d_scan.p, (int)M.h_scan.size(), P, nq,
                               nCols, nC, c->p_score.p, std::max(nL, 1), c->p_minor.p, M.d_frameOf.p, nF, stackCap, c->p_f64[0].p,
                               c->p_i16.p, (uint32_t *)c->p_i32[2].p, o);
            HIPCK(c, hipGetLastError());
            TRY(d2h_vec(c, hi, ib, (size_t)nq * (6 + SL)));
            TRY(d2h_vec(c, hf, fb, (size_t)nq * (2 + SL)));
            TRY(d2h_vec(c, hb, c->p_u8.p, (size_t)nq * (1 + SL)));
            if (nF > 1) TRY(d2h_vec(c, hFrom, (const uint32_t *)c->p_from.p, (size_t)nq * words));
            HIPCK(c, hipStreamSynchronize(c->stream));
        }
        auto t3 = tnow();
        const int32_t *hStatus = hi.data(), *hMinor = hi.data() + nq, *hBest = hi.data() + 2 * (size_t)nq,
                      *hNApp = hi.data() + 3 * (size_t)nq, *hNShort = hi.data() + 5 * (size_t)nq,
                      *hSlNode = hi.data() + 6 * (size_t)nq;
        const double *hBestLK = hf.data(), *hOrig = hf.data() + nq;
        const uint8_t *hBestShort = hb.data(), *hSlShort = hb.data() + nq;
        // ---- shortened query lists that the outcome refers to (shorten, M:8066): distinct (query, frame) pairs
        std::vector<int32_t> S((size_t)nq * nF, -1), shSrc;
        std::vector<size_t> shKey;
        auto want_short = [&](int q, int f) {
            const size_t key = (size_t)q * nF + f;
            if (S[key] == -1) { S[key] = -2; shKey.push_back(key); shSrc.push_back(U[key]); }
        };
        for (int q = 0; q < nq; q++) {
            if (hStatus[q] < 0) continue;
            const int nb = hStatus[q] == 1 ? hMinor[q] : hBest[q];
            if (hBestShort[q]) want_short(q, M.frameOf[nb]);
            if (hStatus[q] == 0)
                for (int i = 0; i < hNShort[q]; i++)
                    if (hSlShort[(size_t)q * SL + i]) want_short(q, M.frameOf[hSlNode[(size_t)q * SL + i]]);
        }
        if (!shSrc.empty()) {
            std::vector<int32_t> out(shSrc.size());
            TRY(maple_shorten_batch(c, (int32_t)shSrc.size(), shSrc.data(), out.data()));
            for (size_t i = 0; i < shKey.size(); i++) S[shKey[i]] = out[i];
        }
        auto qlist = [&](int q, int node, bool shortened) {
            const size_t key = (size_t)q * nF + M.frameOf[node];
            return shortened ? S[key] : U[key];
        };
        // The list OBJECT the reference hands back as bestDiffs (M:8071 / 8186 / 8003).  A frame's query list is made once, when
        // the frame's node is pushed, from the parent frame's list as it is THEN: if that one had already been shortened in
        // place (M:8066), the child's list descends from the shortened form and differs -- in how reference runs are cut, not
        // in what it says -- from passing the original list down (U) and shortening that (S).  Rare; rebuilt along the chain
        // of frames when it happens.  *out = list id.
        auto from_bit = [&](int q, int f) -> bool {
            if (hFrom.empty()) return false;
            const uint32_t w = fromLaneMajor ? hFrom[(size_t)(f >> 5) * nq + q] : hFrom[(size_t)q * words + (f >> 5)];
            return (w >> (f & 31)) & 1u;
        };
        auto exact_qlist = [&](int q, int node, bool shortened, int32_t *outId) -> int {
            std::vector<int> chain;                                        // frames from the node's up to the top one
            for (int f = M.frameOf[node]; f > 0; f = M.frameParent[f]) chain.push_back(f);
            int first = -1;                                                // the outermost frame made from a shortened list
            for (int i = (int)chain.size() - 1; i >= 0; i--) if (from_bit(q, chain[i])) { first = i; break; }
            if (first < 0) { *outId = qlist(q, node, shortened); return MAPLE_OK; }
            int32_t cur = U[(size_t)q * nF + (first + 1 < (int)chain.size() ? chain[first + 1] : 0)];
            for (int i = first; i >= 0; i--) {
                const int f = chain[i];
                if (from_bit(q, f)) { int32_t sh; TRY(maple_shorten_batch(c, 1, &cur, &sh)); cur = sh; }
                const int32_t ml = c->h_tree_mut[M.frameNode[f]];
                const uint8_t down = 0;
                int32_t nx;
                TRY(maple_pass_branch_batch(c, 1, &cur, &ml, &down, &nx));
                cur = nx;
            }
            if (shortened) { int32_t sh; TRY(maple_shorten_batch(c, 1, &cur, &sh)); cur = sh; }
            *outId = cur;
            return MAPLE_OK;
        };
        auto t4 = tnow();
        // ---- short-list refinement, M:8101-8187: one batch over every (query, short-listed node)
        std::vector<int32_t> rq, rnode, ridx;
        for (int q = 0; q < nq; q++)
            if (hStatus[q] == 0)
                for (int i = 0; i < hNShort[q]; i++) { rq.push_back(q); rnode.push_back(hSlNode[(size_t)q * SL + i]); ridx.push_back(i); }
        const size_t nr = rq.size();
        std::vector<double> ev(4 * nr), comp(2 * nr);
        std::vector<int32_t> refinedUp;                                   // the upper list each refined branch was evaluated with
        if (nr) {
            // the upper list of each distinct node, expressed below the node's own mutations
            std::vector<int32_t> upOf(c->dtree.n, -2), needNode, needSrc, needMut;
            for (size_t i = 0; i < nr; i++) {
                const int32_t v = rnode[i];
                if (upOf[v] != -2) continue;
                const int32_t u = c->h_tree_up[v];
                const int32_t uid = c->h_tree_c0[u] == v ? c->h_tree_upRight[u] : c->h_tree_upLeft[u];
                if (uid < 0) return fail(c, MAPLE_ERR_STATE, "node %d has no upper genome list", u);
                upOf[v] = uid;
                if (mut[v] >= 0) { needNode.push_back(v); needSrc.push_back(uid); needMut.push_back(mut[v]); }
            }
            if (!needNode.empty()) {
                std::vector<int32_t> out(needNode.size());
                std::vector<uint8_t> dn(needNode.size(), 0);
                TRY(maple_pass_branch_batch(c, (int32_t)needNode.size(), needSrc.data(), needMut.data(), dn.data(), out.data()));
                for (size_t i = 0; i < needNode.size(); i++) upOf[needNode[i]] = out[i];
            }
            std::vector<int32_t> mid(nr), down(nr), upl(nr), ql(nr);
            std::vector<double> dist(nr);
            std::vector<uint8_t> remTip(nr, 1), tip(nr);
            for (size_t i = 0; i < nr; i++) {
                const int32_t v = rnode[i];
                mid[i] = c->h_tree_totUp[v]; down[i] = c->h_tree_lower[v]; upl[i] = upOf[v]; dist[i] = c->h_tree_dist[v];
                tip[i] = c->h_tree_tip[v];
                ql[i] = qlist(rq[i], v, hSlShort[(size_t)rq[i] * SL + ridx[i]] != 0);
            }
            // ... and, in the same launch, what the optimised placement is compared with (M:8101-8187): the node's lower list
            // appended to its upper list at the branch's own length and at the sum of the two optimised halves
            std::vector<double> c2(2 * nr);
            TRY(evaluate_placement_items(c, (int32_t)nr, mid.data(), down.data(), upl.data(), dist.data(), ql.data(), remTip.data(),
                                         tip.data(), ev.data(), c2.data()));
            refinedUp = upl;
            for (size_t i = 0; i < nr; i++) { comp[i] = c2[2 * i]; comp[nr + i] = c2[2 * i + 1]; }
        }
        auto t5 = tnow();
        if (dbg) fprintf(stderr, "[maple] placement batch of %d: tables %lld us, root vector %lld, frames %lld, score+minor+traversal %lld, shorten %lld, refine %lld\n",
                         nq, tus(tEntry, t0), tus(t0, t1), tus(t1, t2), tus(t2, t3), tus(t3, t4), tus(t4, t5));
        if (sup) {
            // ---- computePlacementSupportOnly=True, M:8101-8290: every refined branch is a possible placement; the mid-branch
            // vector of each (newMidVector, M:8131) is produced by one more merge batch from the optimised lengths
            const double eff = pp->effectivelyNon0BLen;
            std::vector<int32_t> midList(nr, -1);
            if (nr) {
                std::vector<int32_t> l1(nr), l2(nr);
                std::vector<double> b1(nr), b2(nr);
                std::vector<uint8_t> t1(nr, 0), t2(nr), ud(nr, 1);
                std::vector<int32_t> upOf2(nr);
                // (the same upper lists the refinement used)
                for (size_t i = 0; i < nr; i++) {
                    const int32_t v = rnode[i];
                    l2[i] = c->h_tree_lower[v]; b1[i] = ev[4 * i + 2]; b2[i] = ev[4 * i + 1]; t2[i] = c->h_tree_tip[v];
                }
                TRY(maple_merge_batch(c, (int32_t)nr, refinedUp.data(), b1.data(), t1.data(), l2.data(), b2.data(), t2.data(), ud.data(),
                                      nullptr, nullptr, midList.data(), nullptr));
            }
            const auto &up = c->h_tree_up;
            const auto &dist = c->h_tree_dist;
            size_t r2 = 0;
            for (int q = 0; q < nq; q++) {
                const int g = q0 + q;
                status[g] = hStatus[q];
                int64_t o = sup->off[g];
                sup->off[g + 1] = o;
                sup->bestTotalLh[g] = -1;
                if (hStatus[q] < 0) continue;
                int32_t bn = hBest[q];
                double bs = hBestLK[q];
                double bb[3] = {0.0, 0.0, pp->oneMutBLen};
                if (bn != root) { const double half = dist[bn] / 2; bb[0] = half; bb[1] = half / 2; }
                int32_t bmid = -1;
                std::vector<int32_t> pn, pm;
                std::vector<double> pc, pb;
                bool rootDone = false, haveRoot = false;
                int32_t rootNode = -1, rootMid = -1;
                double rootCost = 0.0, rootB[3] = {0, 0, 0};
                for (int i = 0; i < hNShort[q]; i++, r2++) {
                    const double sc = hf[2 * (size_t)nq + (size_t)q * SL + i];
                    if (!(sc >= hBestLK[q] - pp->thresholdLogLKoptimization || sc >= hBestLK[q] - sup->thrOptTopo)) continue;   // M:8109
                    const int32_t t1n = rnode[r2];
                    const double optimized = ev[4 * r2] + comp[nr + r2] - comp[r2];
                    const double top = ev[4 * r2 + 2], bottom = ev[4 * r2 + 1], app = ev[4 * r2 + 3];
                    if (optimized >= bs) { bn = t1n; bs = optimized; bb[0] = top; bb[1] = bottom; bb[2] = app; bmid = midList[r2]; }
                    bool different = true;                                 // M:8190-8201
                    if (top <= eff) different = false;
                    if (dist[t1n] <= eff && up[t1n] >= 0 && up[up[t1n]] >= 0) different = false;
                    if (!rootDone && top <= eff) {                         // M:8203-8212
                        int32_t tn = up[t1n];
                        while (dist[tn] <= eff && up[tn] >= 0) tn = up[tn];
                        if (up[tn] < 0) {
                            rootDone = true; haveRoot = true;
                            rootNode = tn; rootCost = optimized; rootB[0] = top; rootB[1] = bottom; rootB[2] = app; rootMid = midList[r2];
                        }
                    } else if (different) {
                        pn.push_back(t1n); pc.push_back(optimized); pm.push_back(midList[r2]);
                        pb.push_back(top); pb.push_back(bottom); pb.push_back(app);
                    }
                }
                if (bs == -INFINITY) bs = hOrig[q];
                if (haveRoot) {                                            // M:8219-8237
                    bool add = true;
                    if (c->h_tree_c0[root] >= 0)
                        for (int32_t v : pn) if (v == c->h_tree_c0[root] || v == c->h_tree_c1[root]) { add = false; break; }
                    if (add) {
                        pn.push_back(rootNode); pc.push_back(rootCost); pm.push_back(rootMid);
                        pb.push_back(rootB[0]); pb.push_back(rootB[1]); pb.push_back(rootB[2]);
                    }
                }
                if (pn.empty()) {                                          // M:8241-8245
                    pn.push_back(bn); pc.push_back(bs); pm.push_back(bmid);
                    pb.push_back(bb[0]); pb.push_back(bb[1]); pb.push_back(bb[2]);
                }
                for (size_t i = 0; i < pn.size(); i++) {                   // M:8248-8262
                    const double top = pb[3 * i];
                    if (top <= eff) {
                        int32_t tn = pn[i];
                        while (dist[tn] <= eff && up[tn] >= 0) tn = up[tn];
                        if (up[tn] >= 0) {
                            tn = up[tn];
                            while (dist[tn] <= eff && up[tn] >= 0) tn = up[tn];
                            pn[i] = tn;
                            pb[3 * i + 1] = top; pb[3 * i] = dist[tn];
                        }
                    }
                }
                double tot = 0.0;                                          // M:8265-8275
                for (auto &x : pc) { x = exp(x); tot += x; }
                for (auto &x : pc) x = tot ? x / tot : 0.0;
                double hi = 0.0;
                for (size_t i = 0; i < pn.size(); i++) {                   // M:8278-8287
                    if (pc[i] >= sup->minBranchSupport) {
                        if (o >= sup->cap) return fail(c, MAPLE_ERR_ARG, "supports output capacity %lld exhausted", (long long)sup->cap);
                        sup->node[o] = pn[i]; sup->support[o] = pc[i];
                        sup->blen3[3 * o] = pb[3 * i]; sup->blen3[3 * o + 1] = pb[3 * i + 1]; sup->blen3[3 * o + 2] = pb[3 * i + 2];
                        o++;
                    }
                    if (pc[i] > hi) { hi = pc[i]; sup->bestTotalLh[g] = pm[i]; }
                }
                sup->off[g + 1] = o;
            }
        } else {
        // ---- outcome per query
        size_t r = 0;
        for (int q = 0; q < nq; q++) {
            const int g = q0 + q;
            status[g] = hStatus[q];
            nAppend[g] = hNApp[q];
            blen3[3 * g] = blen3[3 * g + 1] = 0.0;
            blen3[3 * g + 2] = pp->oneMutBLen;
            if (hStatus[q] < 0) { bestNode[g] = -1; bestScore[g] = 0.0; bestDiffs[g] = -1; continue; }
            if (hStatus[q] == 1) {                                        // M:7986-8003: placed as a minor sequence
                bestNode[g] = hMinor[q]; bestScore[g] = 1.0;
                TRY(exact_qlist(q, hMinor[q], hBestShort[q] != 0, &bestDiffs[g]));
                continue;
            }
            int32_t bn = hBest[q];
            double bs = hBestLK[q];
            bool bshort = hBestShort[q] != 0;
            if (bn != root) {                                             // M:8072 (the reference halves the bottom length here)
                const double half = c->h_tree_dist[bn] / 2;
                blen3[3 * g] = half; blen3[3 * g + 1] = half / 2;
            }
            for (int i = 0; i < hNShort[q]; i++, r++) {
                const double optimized = ev[4 * r] + comp[nr + r] - comp[r];
                if (optimized >= bs) {                                    // M:8176
                    bn = rnode[r]; bs = optimized;
                    blen3[3 * g] = ev[4 * r + 2]; blen3[3 * g + 1] = ev[4 * r + 1]; blen3[3 * g + 2] = ev[4 * r + 3];
                    bshort = hSlShort[(size_t)q * SL + i] != 0;
                }
            }
            if (bs == -INFINITY) bs = hOrig[q];
            nAppend[g] += 3 * hNShort[q];
            bestNode[g] = bn; bestScore[g] = bs;
            TRY(exact_qlist(q, bn, bshort, &bestDiffs[g]));
        }
        }
        if (manyChunks) {
            int32_t *bestDiffs = sup ? sup->bestTotalLh : bestDiffsOut;   // the list ids this call hands to the caller
            // keep only what the caller was handed: the bestDiffs lists made by this chunk move to the bottom of the arena
            // (down to the chunk's mark), everything else the chunk allocated is released
            const int64_t listMark = chunkMark & (((int64_t)1 << 40) - 1);      // (a mark also carries the mutation-list count)
            std::vector<int32_t> keep;
            for (int q = 0; q < nq; q++) if (bestDiffs[q0 + q] >= listMark) keep.push_back(bestDiffs[q0 + q]);
            std::sort(keep.begin(), keep.end());
            keep.erase(std::unique(keep.begin(), keep.end()), keep.end());
            std::vector<int64_t> eo(keep.size() + 1, 0), ao(keep.size() + 1, 0);
            for (size_t i = 0; i < keep.size(); i++) { eo[i + 1] = eo[i] + c->h_n_ent[keep[i]]; ao[i + 1] = ao[i] + c->h_n_aux[keep[i]]; }
            std::vector<int32_t> pos((size_t)std::max<int64_t>(1, eo.back()));
            std::vector<uint32_t> meta((size_t)std::max<int64_t>(1, eo.back()));
            std::vector<double> aux((size_t)std::max<int64_t>(1, ao.back()));
            if (!keep.empty())
                TRY(maple_lists_download(c, (int32_t)keep.size(), keep.data(), eo.data(), pos.data(), meta.data(), ao.data(), aux.data()));
            TRY(maple_arena_release(c, chunkMark));
            int32_t first = 0;
            if (!keep.empty())
                TRY(maple_lists_upload(c, (int32_t)keep.size(), eo.data(), pos.data(), meta.data(), ao.data(), aux.data(), &first));
            for (int q = 0; q < nq; q++) {
                int32_t &b = bestDiffs[q0 + q];
                if (b >= listMark) b = first + (int32_t)(std::lower_bound(keep.begin(), keep.end(), b) - keep.begin());
            }
        }
    }
    if (dbg) fprintf(stderr, "[maple] placement call of %d: %lld us in all\n", nQ, tus(tEntry, tnow()));
    return MAPLE_OK;
}
