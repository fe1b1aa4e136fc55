#!/usr/bin/env python3
"""bench.py -- candidate SPR placements / second on MI355X (BASELINE.json metric).

A *step* is one pass of the hot path over one batch of synthetic input: every query genome list of
this rank is scored with appendProbNode (M:6505-6785) against every candidate branch of a
synthetic SARS-CoV-2-like tree (the mid-branch ``probVectTotUp`` lists, M:8050), followed by the
per-query arg-max and -- for N>1 -- one RCCL all-gather of the (score, branch) proposals, the
analogue of the reference's gather-and-sort of proposed moves (M:12306-12312).  Queries are sharded
round-robin over ranks exactly like ``assignCoreNumbers`` (M:12164-12195); the tree mirror is
replicated; per-GPU work is fixed (weak scaling).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Inputs (tree mirror, query lists, pair index arrays) are resident
in HBM before the timed region starts.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)

UNREST_Q = [[-0.5524, 0.0602, 0.3655, 0.1267],
            [0.1666, -2.6077, 0.0405, 2.4006],
            [0.8421, 0.1305, -2.4012, 1.4286],
            [0.0688, 0.4849, 0.0502, -0.6039]]


def model_kwargs(mode, l_ref):
    """Model tables of BASELINE.json's configs: [1] UNREST; [2] + per-site rates (--rateVariation); [3] + per-site
    error rates (--estimateSiteSpecificErrorRate).  Seeded as SURVEY.md section 8d prescribes."""
    kw = dict(Q=UNREST_Q)
    if mode in ("ratevar", "siteerr"):
        rng = np.random.default_rng(3)
        kw["siteRates"] = np.clip(rng.gamma(0.5, 2.0, size=l_ref), 0.001, 0.005 * l_ref)
    if mode == "siteerr":
        rng = np.random.default_rng(4)
        er = np.exp(rng.uniform(np.log(1e-10), np.log(1e-3), size=l_ref))
        kw.update(usingErrorRate=True, errorRates=er, errorRateGlobal=float(er.mean()))
    return kw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=10000, help="tips of the synthetic tree (BASELINE configs[1]: 10k)")
    ap.add_argument("--model", choices=["unrest", "ratevar", "siteerr"], default="unrest",
                    help="unrest = configs[1] (default, the headline); ratevar = configs[2]; siteerr = configs[3]")
    ap.add_argument("--queries", type=int, default=256, help="query genome lists per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="approximate host time spent on cpu_baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spr", action="store_true", help="skip the secondary SPR-search-round measurement")
    ap.add_argument("--spr-fast", action="store_true",
                    help="secondary SPR round with the reference's fast initial parameters (strict, 2 fails, 6 log lRef) "
                         "instead of the deep round; for very large trees")
    ap.add_argument("--no-local-refs", action="store_true", help="skip the SPR round on the tree with MAT local references")
    ap.add_argument("--pairs", action="store_true", help="experiment: explicit (parent, child) index arrays, untiled kernel")
    ap.add_argument("--no-sort", action="store_true",
                    help="experiment: leave the candidate branches in tree pre-order instead of ordering them by list length")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the placement path has no CPU fallback")
    # MAPLE_BENCH_BACKEND=gloo is a plumbing check of the N>1 path on a box with fewer GPUs than ranks (ranks then share
    # GPUs and the collectives go through host memory); the driver's runs use nccl (= RCCL over xGMI), one GPU per rank
    backend = os.environ.get("MAPLE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    distd = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        distd = dist
    coll = (lambda t: t) if backend == "nccl" else (lambda t: t.cpu())

    from maple_amd.host import reference_tables, tip_genome_list
    from maple_amd.runtime import Device
    from maple_amd.synth import make_dataset
    from maple_amd.tree_mirror import TreeMirror

    t_setup = time.time()
    data = make_dataset(n_samples=args.samples, l_ref=29903, seed=1, mean_diffs=30.0,
                        rate_variation=(args.model != "unrest"))
    ref_idx, root_freqs = reference_tables(data.ref)
    # genome-list arena: 4 GiB is plenty at 10 000 samples; bigger trees get more of the 288 GB (the per-frame removed
    # lists of the wide searches on trees with local references are the big temporary)
    dev = Device(ref_idx, root_freqs, device=local_rank, arena_bytes=min(128 << 30, max(4 << 30, args.samples * (640 << 10))))
    mkw = model_kwargs(args.model, len(ref_idx))
    dev.set_model(**mkw)
    tip_kw = dict(error_rates=mkw["errorRates"]) if args.model == "siteerr" else {}
    tip_lists = {int(v): tip_genome_list(dl, ref_idx, **tip_kw) for v, dl in zip(data.tip_node, data.diffs)}
    mirror = TreeMirror(dev, data.parent, data.blen, tip_lists).build()
    l_ref = dev.lRef
    cand_nodes = mirror.candidate_nodes(1.0 / (10 * l_ref)) if args.no_sort else mirror.candidates_by_length(1.0 / (10 * l_ref))
    cand_lists = mirror.tot_up[cand_nodes]
    # queries of this rank: samples rank, rank+world, ... (round-robin like coreNum, M:12164-12195)
    q_nodes = np.asarray(data.tip_node[rank::world][: args.queries], dtype=np.int64)
    q_lists = mirror.lower[q_nodes]
    Q, Cn = len(q_lists), len(cand_lists)
    n_pairs = Q * Cn
    parent_ids = np.tile(cand_lists.astype(np.int32), Q)
    child_ids = np.repeat(q_lists.astype(np.int32), Cn)
    # 8 B per entry word + 8 B per stored scalar (4 per O vector) + 8 B result per candidate;
    # the query list is counted once per query per launch
    q_ne, q_na = dev.sizes(q_lists)
    alg_bytes = dev.append_algorithmic_bytes(parent_ids) + 8 * int(q_ne.sum() + q_na.sum())
    cu = torch.device("cuda", local_rank)
    t_q = torch.from_numpy(q_lists.astype(np.int32)).to(cu)
    t_c = torch.from_numpy(cand_lists.astype(np.int32)).to(cu)
    t_out = torch.empty(n_pairs, dtype=torch.float64, device=cu)
    t_cand_nodes = torch.from_numpy(cand_nodes.astype(np.int64)).to(cu)
    stream = torch.cuda.current_stream().cuda_stream
    setup_s = time.time() - t_setup

    if args.pairs:
        t_parent = torch.from_numpy(parent_ids).to(cu)
        t_child = torch.from_numpy(child_ids).to(cu)
        t_tip = torch.ones(n_pairs, dtype=torch.uint8, device=cu)
        t_blen = torch.full((n_pairs,), 1.0 / l_ref, dtype=torch.float64, device=cu)

    def step():
        if args.pairs:
            dev.append_batch_dev(n_pairs, t_parent.data_ptr(), t_child.data_ptr(), t_tip.data_ptr(), t_blen.data_ptr(),
                                 t_out.data_ptr(), stream)
        else:
            dev.append_queries_dev(Q, t_q.data_ptr(), Cn, t_c.data_ptr(), True, 1.0 / l_ref, t_out.data_ptr(), stream)
        best_score, best_idx = t_out.view(Q, Cn).max(dim=1)
        rec = torch.stack([best_score, t_cand_nodes[best_idx].to(torch.float64)], dim=1)
        if distd is not None:
            rec = coll(rec)
            gathered = [torch.empty_like(rec) for _ in range(world)]
            distd.all_gather(gathered, rec)
            rec = torch.cat(gathered, dim=0)
        return rec

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distd is not None:
        distd.barrier()
    torch.cuda.synchronize()
    dev.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rec = step()
    torch.cuda.synchronize()
    if distd is not None:
        distd.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_launch, kernel_ms = dev.timing_read()
    if distd is not None:
        te = coll(torch.tensor([elapsed], dtype=torch.float64, device=cu))
        distd.all_reduce(te, op=distd.ReduceOp.MAX)
        elapsed = float(te.item())
        tp = coll(torch.tensor([float(n_pairs)], dtype=torch.float64, device=cu))
        distd.all_reduce(tp, op=distd.ReduceOp.SUM)
        total_pairs = float(tp.item())
    else:
        total_pairs = float(n_pairs)

    # ---- secondary measurement: one device-resident SPR search round (findBestParentTopology for every node
    # of this rank's shard, M:9580-9716), reported next to the headline, never mixed into it ----
    spr = None
    if not args.no_spr:
        import math
        log_lref = math.log(l_ref)
        nodes_all = np.arange(mirror.n_nodes)
        my_nodes = nodes_all[rank::world]
        dev.upload_tree(mirror.root, mirror.parent, mirror.children[:, 0], mirror.children[:, 1], mirror.dist,
                        mirror.is_tip, mirror.lower, mirror.up_right, mirror.up_left, mirror.tot_up,
                        -np.ones(mirror.n_nodes, dtype=np.int32))
        if args.spr_fast:
            kw = dict(strict=True, allowedFails=2, thresholdLogLKtopology=6.0 * log_lref)
        else:
            kw = dict(strict=False, allowedFails=4, thresholdLogLKtopology=14.0 * log_lref)
        kw.update(thresholdTopologyPlacement=-0.1,
                  thresholdLogLKoptimizationTopology=1.0 * log_lref, thresholdLogLKconsecutivePlacement=1.0,
                  effectivelyNon0BLen=1.0 / (10 * l_ref))
        from maple_amd.parallel import sharded_spr_round
        dev.spr_search_batch(my_nodes, **kw)                            # warm-up (also sizes the workspace)
        dev.timing_reset()
        if distd is not None:
            distd.barrier()
        t0 = time.perf_counter()
        # this rank's share of the round + ONE all-gather of the proposed moves, sorted on every rank (M:12306-12312)
        moves, res = sharded_spr_round(dev, nodes_all, kw, rank, world, device=(cu if backend == "nccl" else None))
        wall = time.perf_counter() - t0
        n_l, k_ms_spr = dev.timing_read()
        st = res["status"]
        spr = {"queries": int(len(my_nodes)), "searched": int((st == 0).sum()), "not_searched": int((st > 0).sum()),
               "failed_or_overflow": int((st < 0).sum()),
               "status_counts": {str(int(k)): int(v) for k, v in zip(*np.unique(st, return_counts=True))},
               "candidate_placements": int(res["nAppend"].sum()),
               "proposed_moves": int((res["placement"] >= 0).sum()), "proposed_moves_all_ranks": len(moves),
               "kernel_ms": k_ms_spr, "launches": n_l,
               "wall_ms": 1e3 * wall,
               "placements_per_s_kernel": float(res["nAppend"].sum() / (k_ms_spr * 1e-3)) if k_ms_spr else None,
               "placements_per_s_wall": float(res["nAppend"].sum() / wall),
               "params": ("fast round: strict, allowedFailsTopology 2, thresholdLogLKtopology 6 log(lRef)" if args.spr_fast else
                          "deep round: non-strict, allowedFailsTopology 4, thresholdLogLKtopology 14 log(lRef)")}
        # ---- and the batched placement search (findBestParentForNewSample for many samples on the frozen tree,
        # M:7912-8292 / 11190-11220): all-branch scoring + device-side traversal + short-list refinement ----
        pkw = dict(oneMutBLen=1.0 / l_ref, effectivelyNon0BLen=1.0 / (10 * l_ref), thresholdLogLK=18.0 * log_lref,
                   thresholdLogLKoptimization=1.0 * log_lref, thresholdLogLKconsecutivePlacement=1.0)
        from maple_amd.synth import perturb_diffs
        mark = dev.mark()
        prng = np.random.default_rng(11 + rank)
        new_samples = [tip_genome_list(perturb_diffs(data.diffs[i], data.ref, prng), ref_idx, **tip_kw)
                       for i in range(rank, len(data.diffs), world)][:Q]
        new_ids = dev.upload(new_samples)                          # samples NOT in the tree (2 extra substitutions each)
        dev.placement_search_batch(new_ids[:8], **pkw)
        t0 = time.perf_counter()
        pres = dev.placement_search_batch(new_ids, **pkw)
        pwall = time.perf_counter() - t0
        dev.release(mark)
        placement = {"queries": int(Q), "wall_ms": 1e3 * pwall, "queries_per_s": Q / pwall,
                     "reference_equivalent_placements_per_s": float(pres["nAppend"].sum() / pwall),
                     "branches_scored_per_s": float(Q * (Cn + 1) / pwall),
                     "minor_sequences": int((pres["status"] == 1).sum()), "failed": int((pres["status"] < 0).sum())}
        spr_mat = None
        if not args.no_local_refs:
            # ---- the same deep round on the same tree after giving it MAT local references (setUpMAT's rule, 50 descendants
            # per reference node, M:166 / 6152-6164): the form real MAPLE trees have; lists are shorter, every search crosses
            # reference frames ----
            from maple_amd.mat import add_local_references
            from maple_amd.tree_host import HostTree
            ht = HostTree.from_mirror(mirror)
            t0 = time.perf_counter()
            n_ref = add_local_references(dev, ht, 50)
            mat_s = time.perf_counter() - t0
            dev.upload_tree(ht.root, mirror.parent, mirror.children[:, 0], mirror.children[:, 1], mirror.dist, mirror.is_tip,
                            ht.id_lower, ht.id_upRight, ht.id_upLeft, ht.id_totUp, ht.id_mut)
            dev.spr_search_batch(my_nodes, **kw)
            dev.timing_reset()
            t0 = time.perf_counter()
            res_m = dev.spr_search_batch(my_nodes, **kw)
            wall_m = time.perf_counter() - t0
            n_lm, k_ms_m = dev.timing_read()
            spr_mat = {"reference_nodes": int(n_ref), "setup_s": round(mat_s, 2), "queries": int(len(my_nodes)),
                       "candidate_placements": int(res_m["nAppend"].sum()), "failed_or_overflow": int((res_m["status"] < 0).sum()),
                       "proposed_moves": int((res_m["placement"] >= 0).sum()), "kernel_ms": k_ms_m, "launches": n_lm,
                       "wall_ms": 1e3 * wall_m, "placements_per_s_kernel": float(res_m["nAppend"].sum() / (k_ms_m * 1e-3)),
                       "placements_per_s_wall": float(res_m["nAppend"].sum() / wall_m),
                       "same_moves_as_without_local_references": bool(np.array_equal(res_m["placement"], res["placement"])
                                                                      and np.array_equal(res_m["nAppend"], res["nAppend"]))}
            # back to the tree the rest of the run refers to
            dev.upload_tree(mirror.root, mirror.parent, mirror.children[:, 0], mirror.children[:, 1], mirror.dist,
                            mirror.is_tip, mirror.lower, mirror.up_right, mirror.up_left, mirror.tot_up,
                            -np.ones(mirror.n_nodes, dtype=np.int32))
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            spr["cpu_baseline"] = spr_cpu_baseline(dev, mirror, ref_idx, root_freqs, my_nodes, res, kw, args.cpu_seconds, mkw)

    # HBM-side bytes per launch come from separate rocprofv3 PMC passes over this same command (a running process
    # cannot read its own PMCs); they are recorded, with the FETCH_SIZE calibration for this access pattern, in
    # profiles/pmc_k_append_queries.json and only reported when the workload is the one they were measured on.
    traffic = None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_k_append_queries*.json"))):
        try:
            pmc = json.load(open(path))
            w = pmc["workload"]
            if ((w["samples"], w["queries_per_gpu"], w["candidate_branches"], w.get("model", "unrest"))
                    == (args.samples, Q, int(Cn), args.model) and not args.pairs):
                traffic = pmc["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_pairs * args.steps / elapsed
        k_ms = kernel_ms / max(1, n_launch)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "candidate SPR placements/sec", "value": value, "unit": "placements/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.samples} synthetic SARS-CoV-2 diff-lists (lRef 29903, ~30 diffs/sample), "
                                   f"{ {'unrest': 'UNREST', 'ratevar': 'UNREST + per-site rates', 'siteerr': 'UNREST + per-site rates + per-site error rates'}[args.model] }, "
                                   "appendProbNode over queries x all candidate branches",
                       "samples": args.samples, "queries_per_gpu": Q, "candidate_branches": int(Cn),
                       "pairs_per_step_per_gpu": int(n_pairs), "tree_nodes": int(mirror.n_nodes),
                       "parallelism": f"queries sharded round-robin over {world} GPU(s), tree mirror replicated",
                       "setup_s": round(setup_s, 1)},
            "roofline": {"bound": "hbm", "kernel": "k_append_queries", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "kernel_ms": k_ms, "launches_timed": n_launch},
        }
        if spr is not None:
            out["spr_search"] = spr
            if spr_mat is not None:
                out["spr_search_local_refs"] = spr_mat
            out["placement_batch"] = placement
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dev, mirror, cand_lists, q_lists, ref_idx, root_freqs,
                                               args.cpu_seconds, t_out, mkw)
        print(json.dumps(out), flush=True)
    if distd is not None:
        distd.destroy_process_group()
    dev.close()


def spr_cpu_baseline(dev, mirror, ref_idx, root_freqs, nodes, gpu_res, kw, cpu_seconds, mkw):
    """The C oracle's SPR search (a port of findBestParentTopology + the worker body, oracle/maple_oracle_search.c)
    on ONE host core over a bounded, evenly spread sample of the same pruned nodes; also cross-checks the GPU."""
    from oracle.oracle_py import Oracle, OracleTree
    orc = Oracle(ref_idx, root_freqs)
    orc.set_model(**mkw)
    n = mirror.n_nodes
    lists4 = []
    for ids in (mirror.lower, mirror.up_right, mirror.up_left, mirror.tot_up):
        have = np.nonzero(ids >= 0)[0]
        lists4.append((have, dev.download_packed(ids[have])))
    up = [None if p < 0 else int(p) for p in mirror.parent]
    children = [[] if mirror.children[v, 0] < 0 else [int(mirror.children[v, 0]), int(mirror.children[v, 1])] for v in range(n)]
    otree = OracleTree(orc, mirror.root, up, children, mirror.dist, [[] for _ in range(n)], [0] * n, lists4)
    # grow the sample until the time budget is used: every 997th, 499th, ... node
    done, t_used, placements, checked = 0, 0.0, 0, 0
    stride = 997
    sel_all = []
    while t_used < cpu_seconds and stride >= 1:
        sel = np.asarray([i for i in range(0, len(nodes), stride) if i not in set(sel_all)], dtype=np.int64)
        if len(sel) == 0:
            break
        t0 = time.perf_counter()
        o = orc.spr_worker(otree, nodes[sel], **kw)
        t_used += time.perf_counter() - t0
        placements += int(o["nAppend"].sum())
        for k in ("status", "bestNode", "placement", "nAppend"):
            if not np.array_equal(o[k], gpu_res[k][sel]):
                raise SystemExit(f"GPU SPR search disagrees with the oracle on {k}")
        checked += len(sel)
        sel_all.extend(sel.tolist())
        stride //= 2
    return {"value": placements / t_used, "unit": "placements/s", "cores": 1, "kind": "port",
            "sample": f"{checked} of the {len(nodes)} searches (evenly spread), {placements} candidate placements, "
                      f"C oracle search (oracle/maple_oracle_search.c), {t_used:.1f} s; node ids, moves and candidate "
                      "counts identical to the GPU's"}


def usable_host_threads():
    """Threads this process may really use: the scheduler affinity, capped by the cgroup CPU quota of the container."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(dev, mirror, cand_lists, q_lists, ref_idx, root_freqs, cpu_seconds, t_out, mkw):
    """The C oracle (a port of the reference's appendProbNode) timed on a bounded sample of the same (query, candidate)
    pairs, on one host core and on all host threads; also cross-checks the GPU scores."""
    from oracle.oracle_py import Oracle
    orc = Oracle(ref_idx, root_freqs)
    orc.set_model(**mkw)
    Cn = len(cand_lists)
    lists = dev.download(np.concatenate([cand_lists, q_lists]))
    packed = orc.pack_many(lists)
    # calibrate on one query, then time as many queries as fit the budget
    pl1 = np.arange(Cn, dtype=np.int32)
    t0 = time.perf_counter()
    orc.appendProbNode_batch(packed, pl1, np.full(Cn, Cn, dtype=np.int32), True, 1.0 / dev.lRef)
    per_query = max(1e-6, time.perf_counter() - t0)
    nq = int(max(1, min(len(q_lists), cpu_seconds / per_query)))
    reps = int(max(1, round(cpu_seconds / (per_query * nq))))
    pl = np.tile(pl1, nq)
    cl = np.repeat(np.arange(Cn, Cn + nq, dtype=np.int32), Cn)
    t0 = time.perf_counter()
    for _ in range(reps):
        ref = orc.appendProbNode_batch(packed, pl, cl, True, 1.0 / dev.lRef)
    dt = time.perf_counter() - t0
    gpu = t_out[: nq * Cn].cpu().numpy()
    both_inf = np.isinf(ref) & np.isinf(gpu)
    err = np.abs(ref - gpu) / np.maximum(1.0, np.abs(ref))
    err[both_inf] = 0.0
    single = reps * nq * Cn / dt
    # the same loop on every host thread (OpenMP over the independent pairs): all queries, repeated for ~2 s of wall time
    threads = usable_host_threads()
    pl_all = np.tile(pl1, len(q_lists))
    cl_all = np.repeat(np.arange(Cn, Cn + len(q_lists), dtype=np.int32), Cn)
    orc.appendProbNode_batch(packed, pl_all, cl_all, True, 1.0 / dev.lRef, threads=threads)          # spin the pool up
    reps_mt = int(max(1, round(2.0 * single * threads / len(pl_all))))
    t0 = time.perf_counter()
    for _ in range(reps_mt):
        ref_mt = orc.appendProbNode_batch(packed, pl_all, cl_all, True, 1.0 / dev.lRef, threads=threads)
    dt_mt = time.perf_counter() - t0
    same = bool(np.array_equal(ref_mt[: len(ref)], ref))
    return {"value": reps_mt * len(pl_all) / dt_mt, "unit": "placements/s", "cores": threads, "kind": "port",
            "sample": f"{reps_mt} pass(es) over all {len(q_lists)} queries x {Cn} candidate branches of the same workload, "
                      f"C oracle (oracle/maple_oracle.c) with OpenMP over the pairs on {threads} host threads (scheduler affinity capped by the container's CPU quota), {dt_mt:.1f} s wall"
                      f" ({dt_mt * threads:.0f} thread-seconds); identical to the scalar run: {same}",
            "single_core": {"value": single, "cores": 1,
                            "sample": f"{reps} pass(es) over {nq} queries = {reps * nq * Cn} pairs, {dt:.1f} s"},
            "max_rel_diff_vs_gpu": float(err.max())}


if __name__ == "__main__":
    main()
