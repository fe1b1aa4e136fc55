#!/usr/bin/env python3
"""bench.py -- candidate SPR placements / second on MI355X (BASELINE.json metric).

Workload (default = BASELINE.json configs[2], the largest single-GPU configuration): a synthetic tree of 100 000
SARS-CoV-2-like samples (lRef 29 903, ~30 diffs/sample), UNREST + per-site rates (--rateVariation), tree mirror
resident in HBM.  A *step* is one pass of the hot path over one batch of pruned nodes: the worker body of
startTopologyUpdatesParallel (M:9580-9716) -- findBestParentTopology (M:6817-7724) for every node of the batch with the
deep-round parameters of the reference's SPR rounds (non-strict, allowedFailsTopology 4, 14 log lRef) -- followed by
the combine step of M:12306-12312 (one all-gather of the proposed moves when N > 1, sorted by improvement).  By default
the batch is EVERY node of the tree (one whole search round, as the reference runs it with all nodes dirty); with
`--batch B` it is the next B nodes of the tree's pre-order.  Either way the nodes are dealt round-robin over the ranks
in pre-order exactly like assignCoreNumbers (M:12164-12195): total work per step is fixed (strong scaling), the tree
mirror is replicated.

`value` = candidate placements (appendProbNode evaluations the reference's search issues, M:7011 / 7223, counted by the
search itself) of all ranks in the K timed steps / wall time (max over ranks).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Everything the searches read is resident in HBM before the timed region starts.

Next to the headline the default run has (rank 0, N = 1): a second leg on BASELINE configs[3]'s tree (`config_1M`: 1 000 000
samples, full model, 131 072 searches per step), a third on configs[4] (`config_5`: 50 000 samples added to that tree one after the
other -- the serial placement loop with its samples announced, maple_placement_ahead -- then a round), a whole round on a
changing tree at the headline's size (`changing_tree`: search, every proposed move re-searched and applied, the next round),
`tree_log_lk` against the oracle and `cpu_baseline`; everything else goes to bench_detail.json.
"""
import argparse
import glob
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0     # ... the measured copy rate, the denominator SURVEY 8d names: reported next to the spec peak (frac_of_copy)

UNREST_Q = [[-0.5524, 0.0602, 0.3655, 0.1267],
            [0.1666, -2.6077, 0.0405, 2.4006],
            [0.8421, 0.1305, -2.4012, 1.4286],
            [0.0688, 0.4849, 0.0502, -0.6039]]
MODEL_TEXT = {"unrest": "UNREST", "ratevar": "UNREST + per-site rates (--rateVariation)",
              "siteerr": "UNREST + per-site rates + per-site error rates (--rateVariation --estimateSiteSpecificErrorRate)"}


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


def search_kwargs(l_ref, fast=False):
    """Parameters of the reference's SPR rounds (M:53-57, 3609-3614): the deep round, or the fast initial round."""
    ll = math.log(l_ref)
    kw = (dict(strict=True, allowedFails=2, thresholdLogLKtopology=6.0 * ll) if fast else
          dict(strict=False, allowedFails=4, thresholdLogLKtopology=14.0 * ll))
    kw.update(thresholdTopologyPlacement=-0.1, thresholdLogLKoptimizationTopology=1.0 * ll,
              thresholdLogLKconsecutivePlacement=1.0, effectivelyNon0BLen=1.0 / (10 * l_ref))
    return kw


def preorder_nodes(mirror):
    """Pre-order with child 0 first: the order assignCoreNumbers numbers the nodes in (M:12164-12195)."""
    order, stack = [], [mirror.root]
    ch = mirror.children
    while stack:
        v = stack.pop()
        order.append(v)
        if ch[v, 0] >= 0:
            stack.append(int(ch[v, 1]))
            stack.append(int(ch[v, 0]))
    return np.asarray(order, dtype=np.int64)

def optimise_branch_lengths(dev, mirror, tip_ids, mark_tree, eff_non0, max_passes=8):
    """The branch-length optimisation MAPLE runs before its SPR rounds (M:11899-11906: traverseTreeToOptimizeBranchLengths
    until nothing moves) on the bench tree, in the form of the reference's fastPass=True (M:8727-8893): every branch
    re-estimated from the same frozen lists (estimateBranchLengthWithDerivative(upper vector, lower list), ONE launch for
    the whole tree), a length replaced when it moves by more than 1 % (M:8873), then all genome lists rebuilt on the GPU;
    repeated until a pass moves nothing.  (The two branches below the root keep their lengths: the reference splits their
    sum by a grid search, M:8743-8810.)  Returns what it did; mirror.dist and the lists are updated in place."""
    nodes = np.nonzero(mirror.parent >= 0)[0]
    nodes = nodes[mirror.parent[nodes] != mirror.root]
    out = {"form": "fastPass (every branch from the same frozen lists) + full rebuild of the genome lists, until no length moves by > 1 %",
           "branches": int(len(nodes)), "updates_per_pass": [], "estimate_ms_per_pass": [], "rebuild_ms_per_pass": []}

    def impossible():
        p = mirror.parent[nodes]
        upv = np.where(mirror.children[p, 0] == nodes, mirror.up_right[p], mirror.up_left[p]).astype(np.int32)
        cur = dev.append_batch(upv, mirror.lower[nodes], mirror.is_tip[nodes], mirror.dist[nodes])
        return upv, int(np.isinf(cur).sum())
    upv, out["current_placement_impossible_before"] = impossible()
    out["zero_length_branches_before"] = int((mirror.dist[nodes] == 0).sum())
    for _ in range(max_passes):
        t0 = time.perf_counter()
        t, isf = dev.blen_batch(upv, mirror.lower[nodes], mirror.is_tip[nodes])
        out["estimate_ms_per_pass"].append(round(1e3 * (time.perf_counter() - t0), 2))
        best = np.where(isf.astype(bool), 0.0, t)
        d = mirror.dist[nodes]
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = d / best
        upd = ~((best == 0) & (d == 0)) & ((best == 0) | (d == 0) | (ratio > 1.01) | (ratio < 0.99))          # M:8873
        out["updates_per_pass"].append(int(upd.sum()))
        if not upd.any():
            break
        mirror.dist[nodes[upd]] = best[upd]
        t0 = time.perf_counter()
        dev.release(mark_tree)
        mirror.lower = tip_ids.copy()
        mirror.up_right[:] = -1
        mirror.up_left[:] = -1
        mirror.tot_up[:] = -1
        mirror.build()
        out["rebuild_ms_per_pass"].append(round(1e3 * (time.perf_counter() - t0), 1))
        upv, _ = impossible()
    _, out["current_placement_impossible_after"] = impossible()
    out["zero_length_branches_after"] = int((mirror.dist[nodes] == 0).sum())
    out["passes"] = len(out["updates_per_pass"])
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=100000,
                    help="tips of the synthetic tree (default: BASELINE configs[2]; 10000 = configs[1], 1000000 = configs[3])")
    ap.add_argument("--model", choices=["unrest", "ratevar", "siteerr"], default="ratevar",
                    help="ratevar = configs[2] (default); unrest = configs[1]; siteerr = configs[3]")
    ap.add_argument("--batch", type=int, default=0,
                    help="pruned nodes searched per step, over all GPUs (default 0 = every node of the tree: one whole SPR "
                         "search round per step, what the reference does per round, M:12283-12316)")
    ap.add_argument("--spr-fast", action="store_true",
                    help="the reference's fast initial round (strict, 2 fails, 6 log lRef) instead of the deep round")
    ap.add_argument("--tree", choices=["optimised", "truth"], default="optimised",
                    help="optimised (default): the simulated tree after the branch-length passes MAPLE runs before its SPR rounds "
                         "(traverseTreeToOptimizeBranchLengths, M:11899-11906), until no length moves; truth: the simulated tree "
                         "with branch lengths = mutations / lRef (rounds 1-2)")
    ap.add_argument("--refs", choices=["auto", "local", "none"], default="auto",
                    help="local: the tree carries MAT local references (setUpMAT's rule, a reference node per 50 descendants, M:166 / "
                         "6152-6164 -- the form MAPLE's own trees have, M:8296-8354), the searches re-express their lists at every "
                         "reference branch they cross; none: every list in the root's frame (rounds 1-3); auto = local")
    ap.add_argument("--synth", choices=["auto", "v1", "v2"], default="auto",
                    help="generator of the synthetic input: v1 = maple_amd.synth.make_dataset (numpy stream; the 10 000 / 100 000-sample "
                         "trees of rounds 1-4: 21 s at 100 000 samples), v2 = the same model from csrc/synth_gen.c (1 s at 1 000 000 "
                         "samples); auto = v2")
    ap.add_argument("--queries", type=int, default=256, help="query lists of the all-pairs scoring sub-block")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="approximate host time spent on cpu_baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the sub-blocks (all-pairs kernel, batched placement, local references)")
    ap.add_argument("--local-refs", action="store_true",
                    help="also run the steps on the same tree after giving it MAT local references (default when samples <= 200000)")
    ap.add_argument("--no-1m", action="store_true",
                    help="skip the second leg of the default run (BASELINE configs[3]: 1 000 000 samples, full model)")
    ap.add_argument("--steps-1m", type=int, default=3, help="timed steps of the 1 000 000-sample leg (131 072 searches each)")
    ap.add_argument("--online-add", type=int, default=50000,
                    help="BASELINE configs[4] on the 1 000 000-sample leg's tree (one GPU): this many new samples added one after the "
                         "other (placement search, tree edit, maple_update_partials, maple_tree_patch: M:11692-11752), then one round "
                         "of searches on the grown tree; 0 = skip")
    return ap.parse_args(argv)


class Env:
    """What every leg of a run shares: torch, the process group, this rank's GPU."""


def init_env(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # started as plain `python bench.py --gpus N`: one process per GPU under torch.distributed.run (RCCL needs one rank per
        # GPU), same arguments
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
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
    env = Env()
    env.torch, env.rank, env.world, env.local_rank, env.backend, env.distd = torch, rank, world, local_rank, backend, distd
    env.cu = torch.device("cuda", local_rank)
    env.coll_dev = env.cu if backend == "nccl" else None
    return env


LINE_LIMIT = 4096          # bytes: the final line must stay small enough for the driver to parse (round 4's 25 KB line was not)


def _short(text, n):
    text = str(text)
    return text if len(text) <= n else text[: n - 1].rstrip() + "~"


def _num(x, digits=6):
    """Floats of the line to `digits` significant digits (the detail file keeps full precision)."""
    if isinstance(x, float) and math.isfinite(x):
        return float(f"{x:.{digits}g}")
    return x


def compact_line(full):
    """The ONE line rank 0 prints: the contract's fields, `roofline` and `cpu_baseline` as flat objects, the tree log-LK
    check and a summary of the 1 000 000-tip leg -- numbers only, no prose beyond the workload sentence.  Everything else
    run_leg measured (notes, per-kernel blocks, PMC counters, sub-blocks) goes to bench_detail.json (write_detail)."""
    cfg = full.get("config", {})

    def roof(r):
        if not r:
            return None
        src = r.get("traffic_source")
        return {"bound": r["bound"], "kernel": str(r["kernel"]).split()[0], "achieved": _num(r["achieved"]), "peak": r["peak"],
                "unit": r["unit"], "frac": _num(r["frac"]), "peak_copy": r.get("peak_copy"), "frac_of_copy": _num(r.get("frac_of_copy")),
                "traffic": _num(r.get("traffic")),
                "traffic_over_algorithmic": _num(r["traffic"] / r["algorithmic_bytes_per_launch"], 4)
                if r.get("traffic") and r.get("algorithmic_bytes_per_launch") else None,
                "traffic_source": src.get("file") if isinstance(src, dict) else src,
                "algorithmic_bytes_per_launch": _num(r.get("algorithmic_bytes_per_launch")), "kernel_ms": _num(r.get("kernel_ms")),
                "launches_timed": r.get("launches_timed"), "kernel_ms_per_step": _num(r.get("kernel_ms_per_step"))}
    line = {k: _num(full.get(k)) for k in ("metric", "value", "value_walked", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                           "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": _short(cfg.get("workload", ""), 260), "samples": cfg.get("samples"), "model": cfg.get("model"),
                      "tree": cfg.get("tree"), "local_references": (cfg.get("local_references") or {}).get("form"),
                      "tree_nodes": cfg.get("tree_nodes"), "searches_per_step": cfg.get("searches_per_step"), "searches_timed": cfg.get("searches_timed"),
                      "candidate_placements_timed": cfg.get("candidate_placements_timed"),
                      "parallelism": _short(cfg.get("parallelism", ""), 120), "setup_s": cfg.get("setup_s")}
    line["roofline"] = roof(full.get("roofline"))
    line["roofline_other_kernels"] = [
        {"kernel": str(r["kernel"]).split()[0], "bound": r["bound"], "frac": _num(r["frac"], 4), "kernel_ms_per_step": _num(r["kernel_ms_per_step"], 4)}
        for r in full.get("roofline_by_kernel", []) if r is not full.get("roofline") and r["kernel"] != (full.get("roofline") or {}).get("kernel")]
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _num(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": _short(cb.get("sample_short") or cb["sample"], 120),
                                "one_core": _num((cb.get("one_core") or {}).get("value"))}
    if full.get("tree_log_lk"):
        line["tree_log_lk"] = {k: full["tree_log_lk"].get(k) for k in ("gpu", "oracle", "rel_delta", "tolerance")}
    line["first_call_ms"] = _num(full.get("first_call_ms"), 5)
    line["first_step_after_upload_ms"] = _num(full.get("first_step_after_upload_ms"), 5)
    leg = full.get("config_1M_full_model")
    if leg:
        lc = leg.get("config", {})
        line["config_1M"] = {"samples": lc.get("samples"), "model": lc.get("model"), "searches_per_step": lc.get("searches_per_step"),
                             "value": _num(leg.get("value")), "value_walked": _num(leg.get("value_walked")),
                             "ms_per_step": _num(leg.get("ms_per_step")), "steps": leg.get("steps"),
                             "first_call_ms": _num(leg.get("first_call_ms"), 5),
                             "first_step_after_upload_ms": _num(leg.get("first_step_after_upload_ms"), 5),
                             "roofline": {k: v for k, v in (roof(leg.get("roofline")) or {}).items()
                                          if k in ("bound", "kernel", "achieved", "peak", "frac", "kernel_ms", "kernel_ms_per_step")}}
    rc = full.get("round_on_a_changing_tree")
    if rc and "error" not in rc:
        line["changing_tree"] = {"moves": rc["proposed_moves"], "applied": rc["moves_applied"], "apply_ms_per_move": _num(rc["apply_ms_per_proposed_move"], 4),
                                 "round_s": _num(rc["round_s"], 4), "next_first_ms": _num(rc["next_round_first_search_ms"], 5),
                                 "next_ms": _num(rc["next_round_search_ms"], 5)}
    c5 = full.get("config_5")
    if c5 and "error" in c5:
        line["config_5"] = {"error": _short(c5["error"], 120)}
    elif c5:
        line["config_5"] = {"samples_added": c5["samples_added"], "ms_per_sample": _num(c5["ms_per_sample"], 4), "total_s": _num(c5["total_s"], 4),
                            "round_after_ms": _num(c5["round_after"]["round_after_ms"], 5), "round_after_searches": c5["round_after"]["searches"]}
    line["detail"] = full.get("detail_file")
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:                            # (cannot happen with the bounded fields above; never print a long line)
        for k in ("roofline_other_kernels", "config_1M", "detail"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) <= LINE_LIMIT:
                break
    return line, text


def write_detail(full):
    """Everything a run measured, with its notes: bench_detail.json next to bench.py (and under gpurun_out/ when that exists, so
    that a gpurun call brings it back).  Returns the path written (relative to the repository)."""
    name = "bench_detail.json"
    full["detail_file"] = name
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, name), "w") as f:
                    json.dump(full, f, indent=1)
            except OSError:
                pass
    return name


def main():
    args = parse_args()
    env = init_env(args)
    out = run_leg(args, env)
    # The default run has a second leg: BASELINE configs[3] -- the tree the metric's target is quoted on (1 000 000 samples, full
    # model) -- on this one GPU, 131 072 searches per step (rotating through the tree's pre-order), its own roofline block.
    default_line = (args.samples, args.model, args.batch, args.spr_fast, args.tree) == (100000, "ratevar", 0, False, "optimised")
    if default_line and not args.no_1m:
        a2 = argparse.Namespace(**vars(args))
        # (three warm-up calls: the frontier tier's pools are sized by what the calls before asked for, and a call whose pools ran over
        # under-reports -- at this size they stop growing after the third call)
        a2.samples, a2.model, a2.batch, a2.steps, a2.warmup = 1000000, "siteerr", 131072, args.steps_1m, 3
        a2.no_extras, a2.no_cpu_baseline, a2.synth = True, True, "v2"
        a2.online_leg = args.online_add if env.world == 1 else 0   # (configs[4]: the serial loop of one process; see online_update_leg)
        leg2 = run_leg(a2, env)
        if env.rank == 0:
            keep = ("metric", "value", "value_walked", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype", "data",
                    "config", "roofline", "roofline_by_kernel", "per_rank", "spr_search", "first_call_ms", "first_step_after_upload_ms", "tree_reupload_ms")
            out["config_1M_full_model"] = {k: leg2[k] for k in keep if k in leg2}
            if leg2.get("config_5"):
                out["config_5"] = leg2["config_5"]
    if env.rank == 0:
        write_detail(out)
        _, text = compact_line(out)
        print(text, flush=True)
    if env.distd is not None:
        env.distd.barrier()
        env.distd.destroy_process_group()


class BenchTree:
    """What build_bench_tree leaves behind: the device, the tree and what it took to make it."""


def build_bench_tree(samples, model, *, device=0, tree="optimised", refs="auto", synth="auto", big_arena=False, debug_library=False):
    """The tree the bench steps search (tests/test_hip_configs.py builds the same one): synthetic samples, genome
    lists built on the GPU, branch lengths optimised as MAPLE does before its SPR rounds (tree="optimised"), MAT local
    references added (refs="local" / "auto"), uploaded for the searches."""
    from maple_amd.host import reference_tables, tip_genome_list, tip_lists_packed
    from maple_amd.runtime import Device
    from maple_amd.synth import make_dataset, make_dataset_native
    from maple_amd.tree_mirror import TreeMirror
    bt = BenchTree()
    bt.t_setup = time.time()
    synth = synth if synth != "auto" else "v2"
    gen = make_dataset if synth == "v1" else make_dataset_native
    data = gen(n_samples=samples, l_ref=29903, seed=1, mean_diffs=30.0, rate_variation=(model != "unrest"))
    bt.gen_s = time.time() - bt.t_setup
    ref_idx, root_freqs = reference_tables(data.ref)
    # genome-list arena: bigger trees get more of the 288 GB (the per-frame removed lists of the wide searches on trees
    # with local references are the big temporary)
    # (the tree's own lists take ~10 KB per sample; the sub-block with local references needs the large arena)
    per_sample = (320 << 10) if big_arena else (64 << 10)
    # (debug_library: libmaple_hip_debug.so, for tools that read the level profile -- never the bench itself)
    dev = Device(ref_idx, root_freqs, device=device, arena_bytes=min(128 << 30, max(4 << 30, samples * per_sample)), debug=debug_library)
    mkw = model_kwargs(model, len(ref_idx))
    dev.set_model(**mkw)
    tip_kw = dict(error_rates=mkw["errorRates"]) if model == "siteerr" else {}
    t_tips = time.time()
    if synth == "v1":
        tip_lists = {int(v): tip_genome_list(dl, ref_idx, **tip_kw) for v, dl in zip(data.tip_node, data.diffs)}
        mirror = TreeMirror(dev, data.parent, data.blen, tip_lists)
    else:                                                # (no Python object per entry: 30 M MAPLE entries at 1 000 000 samples)
        dc = data.diffs
        mirror = TreeMirror(dev, data.parent, data.blen, tip_packed=(data.tip_node, tip_lists_packed(dc.off, dc.code, dc.pos, dc.length,
                                                                                                      ref_idx, **tip_kw)))
    bt.tips_s = time.time() - t_tips
    tip_ids = mirror.lower.copy()
    mark_tree = dev.mark()
    t_b = time.perf_counter()
    mirror.build()
    build_ms = 1e3 * (time.perf_counter() - t_b)
    l_ref = dev.lRef
    blen_opt = None
    if tree == "optimised":
        blen_opt = optimise_branch_lengths(dev, mirror, tip_ids, mark_tree, 1.0 / (10 * l_ref))
        blen_opt["tree_build_ms"] = round(build_ms, 1)
    no_mut = -np.ones(mirror.n_nodes, dtype=np.int32)

    def upload_plain_tree():
        dev.upload_tree(mirror.root, mirror.parent, mirror.children[:, 0], mirror.children[:, 1], mirror.dist, mirror.is_tip,
                        mirror.lower, mirror.up_right, mirror.up_left, mirror.tot_up, no_mut)
    refs = refs if refs != "auto" else "local"
    ht, n_ref, refs_s, ht_dist = None, 0, 0.0, None
    if refs == "local":
        # the tree as MAPLE itself keeps it: MAT local references (maple_amd/mat.py: the reference nodes chosen by setUpMAT's rule, every
        # list of a clade written against its reference node's genome, all four lists of every node rebuilt on the GPU)
        from maple_amd.mat import add_local_references
        from maple_amd.tree_host import HostTree
        t_r = time.perf_counter()
        ht = HostTree.from_mirror(mirror)
        n_ref = add_local_references(dev, ht, 50)
        ht_dist = np.asarray([float(x or 0.0) for x in ht.dist])
        refs_s = time.perf_counter() - t_r

    def upload_headline_tree():
        if ht is None:
            upload_plain_tree()
        else:
            dev.upload_tree(ht.root, mirror.parent, mirror.children[:, 0], mirror.children[:, 1], ht_dist, mirror.is_tip, ht.id_lower,
                            ht.id_upRight, ht.id_upLeft, ht.id_totUp, ht.id_mut)
    upload_headline_tree()
    bt.dev, bt.mirror, bt.ht, bt.ht_dist, bt.data, bt.ref_idx, bt.root_freqs, bt.mkw, bt.tip_kw = (dev, mirror, ht, ht_dist, data, ref_idx,
                                                                                                   root_freqs, mkw, tip_kw)
    bt.tip_ids, bt.mark_tree, bt.blen_opt, bt.n_ref, bt.refs, bt.refs_s, bt.synth = tip_ids, mark_tree, blen_opt, n_ref, refs, refs_s, synth
    bt.upload_plain_tree, bt.upload_headline_tree = upload_plain_tree, upload_headline_tree
    return bt


def run_leg(args, env):
    """One workload: build the tree on this rank's GPU, warm up, time the steps; returns the JSON object on rank 0 (None elsewhere)."""
    torch, rank, world, local_rank, backend, distd, cu, coll_dev = (env.torch, env.rank, env.world, env.local_rank, env.backend,
                                                                     env.distd, env.cu, env.coll_dev)
    from maple_amd.parallel import gather_proposals, pack_proposals
    from maple_amd.runtime import Device

    bt = build_bench_tree(args.samples, args.model, device=local_rank, tree=args.tree, refs=args.refs, synth=args.synth,
                          big_arena=(args.local_refs or args.samples <= 200000) and not args.no_extras)
    dev, mirror, ht, data, ref_idx, root_freqs, mkw, tip_kw, tip_ids = (bt.dev, bt.mirror, bt.ht, bt.data, bt.ref_idx, bt.root_freqs,
                                                                        bt.mkw, bt.tip_kw, bt.tip_ids)
    t_setup, gen_s, tips_s, refs_s, n_ref, refs, blen_opt = bt.t_setup, bt.gen_s, bt.tips_s, bt.refs_s, bt.n_ref, bt.refs, bt.blen_opt
    upload_plain_tree, upload_headline_tree = bt.upload_plain_tree, bt.upload_headline_tree
    l_ref = dev.lRef
    t_up = time.perf_counter()
    upload_headline_tree()                               # (timed once more, warm: what a caller pays per change of the tree)
    tree_upload_ms = 1e3 * (time.perf_counter() - t_up)
    st_res = dev.stats()
    n_lists_res, n_ent_res, n_aux_res = st_res["n_lists"], st_res["n_entries"], st_res["n_aux"]
    kw = search_kwargs(l_ref, args.spr_fast)
    order = preorder_nodes(mirror)
    B = min(args.batch, len(order)) if args.batch > 0 else len(order)
    setup_s = time.time() - t_setup

    def batch_of(i):
        """This rank's share of step i: the next B nodes of the pre-order (wrapping), dealt round-robin (coreNum)."""
        sel = np.arange(i * B, (i + 1) * B) % len(order)
        return order[sel][rank::world]

    dbg = os.environ.get("MAPLE_DEBUG") is not None
    if os.environ.get("MAPLE_VERBOSE"):                   # (the library's own account of a call on stderr: tools/timeline_step.py's companion)
        dev.set_tuning(verbose=int(os.environ["MAPLE_VERBOSE"]))

    def step(i):
        ta = time.perf_counter()
        mine = batch_of(i)
        tb = time.perf_counter()
        res = dev.spr_search_batch(mine, **kw)
        if dbg:
            print(f"[bench] step {i}: batch_of {1e3 * (tb - ta):.1f} ms, spr_search_batch {1e3 * (time.perf_counter() - tb):.1f} ms",
                  file=sys.stderr, flush=True)
        bad = res["status"][(res["status"] < -1)]
        if len(bad):
            raise SystemExit(f"SPR search failed: status {sorted(set(bad.tolist()))} (workspace / pool capacity)")
        rec = pack_proposals(mine, res["placement"], res["improvement"])
        moves = gather_proposals(rec, device=coll_dev, cap=-(-B // world)) if distd is not None else rec   # (cap: the largest shard)
        return res, moves

    first_call_ms = None
    for i in range(args.warmup):
        t_w = time.perf_counter()
        step(args.steps + i)                              # (other batches than the timed ones; also sizes every buffer)
        if first_call_ms is None:
            first_call_ms = 1e3 * (time.perf_counter() - t_w)
    torch.cuda.synchronize()
    if distd is not None:
        distd.barrier()
    torch.cuda.synchronize()
    dev.timing_reset()
    placements = 0
    status_counts = {}
    n_moves = 0
    kept = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        res, moves = step(i)
        placements += int(res["nAppend"][res["status"] >= -1].sum())
        kept.append(res)
    torch.cuda.synchronize()
    if distd is not None:
        distd.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    K = {name: dev.timing_read_kind(getattr(Device, "KIND_" + name)) for name in
         ("SPR_SCORE", "SPR_SEARCH", "SPR_REPLAY", "FR_UPDATING", "FR_CACHED", "FR_REPLAY", "FR_WIDE")}     # (launches, ms, units, bytes)
    # what a caller pays for the first round after the tree changed: the tree uploaded again (every per-tree table of the library is
    # dropped), then one step -- the pools are there, the tables are not
    t_u = time.perf_counter()
    upload_headline_tree()
    reupload_ms = 1e3 * (time.perf_counter() - t_u)
    t_u = time.perf_counter()
    step(args.steps + args.warmup)
    first_after_upload_ms = 1e3 * (time.perf_counter() - t_u)
    for res in kept:
        for k, v in zip(*np.unique(res["status"], return_counts=True)):
            status_counts[str(int(k))] = status_counts.get(str(int(k)), 0) + int(v)
        n_moves += int((res["placement"] >= 0).sum())
    searches = sum(len(r["status"]) for r in kept)
    # candidate placements this rank really WALKED (a candidate list against the removed list, entry by entry): those of the
    # searches the frontier / lane tiers finished, plus the (search, branch) pairs the witness filter or the dense kernel scored
    # for the whole-tree searches -- the rest of `value` are placements of whole-tree searches that are proved -inf and counted
    # (the dense kernel scores EVERY branch for a search that went over its budget, the search then visits some of them: of
    # those pairs, the ones that count are the placements the searches replayed)
    walked_local = float(K["SPR_SEARCH"][2] + min(K["SPR_SCORE"][2], K["SPR_REPLAY"][2] + K["FR_WIDE"][2]))
    if distd is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cu)
        t = t if backend == "nccl" else t.cpu()
        distd.all_reduce(t, op=distd.ReduceOp.MAX)
        elapsed = float(t.item())
        t = torch.tensor([float(placements), float(searches), walked_local], dtype=torch.float64, device=cu)
        t = t if backend == "nccl" else t.cpu()
        distd.all_reduce(t, op=distd.ReduceOp.SUM)
        total_placements, total_searches, total_walked = float(t[0].item()), float(t[1].item()), float(t[2].item())
    else:
        total_placements, total_searches, total_walked = float(placements), float(searches), walked_local

    # per-rank kernel times of the timed steps (what a scaling run is read with: the ranks search disjoint shares of a step)
    mine_ms = {"rank": rank, "wall_ms_per_step": 1e3 * elapsed_local / args.steps,
               "frontier_tier": K["SPR_SEARCH"][1] / args.steps, "witness_filter_and_scoring": K["SPR_SCORE"][1] / args.steps,
               "replay_of_whole_tree_searches": (K["SPR_REPLAY"][1] + K["FR_WIDE"][1]) / args.steps,
               "searches_per_step": len(batch_of(0)), "placements_per_step": placements / args.steps}
    per_rank = [mine_ms]
    if distd is not None:
        per_rank = [None] * world
        distd.all_gather_object(per_rank, mine_ms)
    extras = {}
    if not args.no_extras and rank == 0:
        extras = sub_blocks(args, dev, mirror, data, ref_idx, tip_kw, kw, order, B, upload_plain_tree, torch, cu,
                            first_step=kept[0] if kept else None, first_nodes=batch_of(0), headline_refs=(ht is not None))

    # HBM-side bytes per launch come from separate rocprofv3 PMC passes over this same command (a running process cannot
    # read its own PMCs); they are recorded under profiles/ and only reported for the workload they were measured on, with the
    # file they come from and the commit that file was made at (`traffic_source`; null: no PMC pass of this workload on record).
    traffic, pmc_issue, traffic_source = {}, {}, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_spr_*.json"))):
        try:
            pmc = json.load(open(path))
            w = pmc["workload"]
            if (w["samples"], w["model"], w["batch"], w["n_gpus"], w.get("tree", "truth"), w.get("refs", "none"), w.get("synth", "v1")) == \
                    (args.samples, args.model, B, world, args.tree, refs, bt.synth) \
                    and "traffic_bytes_per_step" in pmc:
                traffic = {k: v / max(1.0, pmc["launches_per_step"][k]) for k, v in pmc["traffic_bytes_per_step"].items()}
                pmc_issue = pmc.get("issue", {})
                traffic_source = {"file": os.path.relpath(path, ROOT), "library_commit": pmc.get("library_commit"),
                                  "note": "separate rocprofv3 --pmc passes of this command (tools/pmc_summary.py); not measured in this run"}
        except (OSError, KeyError, ValueError):
            pass

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_placements / elapsed
        steps = args.steps

        LDS_PEAK_GBS = 150000.0   # MI355X_MICROARCH.md, LDS: ~150 TB/s aggregate for ds_read_b64 / b128 with every CU streaming

        def roof(kernel, kind, what, bound="hbm"):
            """bound = "hbm": SURVEY 8d bytes against the HBM peak; "lds": a kernel that stages a tile of candidate lists in LDS once
            and walks it for hundreds of queries moves its algorithmic bytes out of LDS, not HBM -- priced against the LDS peak;
            "counted": placements the kernel accounts for without walking a list (no bandwidth statement: frac null)."""
            n, ms, units, bytes_ = K[kind]
            ach = (bytes_ / (ms * 1e-3) / 1e9) if ms else 0.0
            peak = {"hbm": HBM_PEAK_GBS, "lds": LDS_PEAK_GBS, "counted": None}[bound]
            return {"bound": bound, "kernel": kernel, "achieved": ach if peak else None, "peak": peak, "unit": "GB/s",
                    "frac": (ach / peak) if peak else None,
                    "peak_copy": HBM_COPY_GBS if bound == "hbm" else None, "frac_of_copy": (ach / HBM_COPY_GBS) if bound == "hbm" else None,
                    "traffic": traffic.get(kernel.split()[0]), "traffic_source": traffic_source,
                    "pmc_per_step": pmc_issue.get(kernel.split()[0]),
                    "algorithmic_bytes_per_launch": bytes_ / max(1, n), "kernel_ms": ms / max(1, n), "launches_timed": n,
                    "kernel_ms_per_step": ms / steps, "units_per_step": units / steps, "note": what}
        # every kernel of a step that matters, each timed by its own HIP events on the stream it is launched on
        roofs = [
            roof("k_fr_cached (frontier tier: one lane scores one (search, branch) item, appendProbNode of the branch's "
                 "probVectTotUp and the removed list)", "FR_CACHED",
                 "rank 0; units = placements scored, algorithmic bytes = SURVEY 8d (8E + 8A + 8 per candidate list, counted on the "
                 "device per scored item); one launch per level of the expansion"),
            roof("k_fr_updating (frontier tier: items that still update genome lists -- mergeVectors x 1-4, areVectorsDifferent, "
                 "appendProbNode per item)", "FR_UPDATING",
                 "rank 0; units = items one lane each walked (the few with very long lists go to k_fr_updating_wave, inside the same "
                 "events); algorithmic bytes, counted on the device = the two lists every mergeVectors of an item reads and the one it "
                 "writes (8E + 8A each).  A level lasts as long as its slowest wavefront (64 items, ~1000 dependent list steps): "
                 "latency-bound, see DESIGN.md section 3"),
            roof(("k_append_queries_lds (whole-tree searches with an error model: every (search, branch) pair walked, a tile of 64 "
                  "candidate lists staged in LDS per 512 queries)") if args.model == "siteerr" else
                 ("k_wit_score (whole-tree searches: the witness filter rules out the branches that score -inf by appendProbNode's own "
                  "rule, the pairs that are left are walked)"), "SPR_SCORE",
                 "rank 0; units = (search, branch) pairs walked, algorithmic bytes = SURVEY 8d for those pairs (8E + 8A + 8 per candidate "
                 "list, each removed list once); the time is the whole stage: witnesses, buckets, pair list, walks, bitmap prefix "
                 "(when the dense kernel k_append_queries_lds runs instead -- error model, local references, searches that were not "
                 "announced -- its launches are booked here too: with an error model it is the only one, and the stage is priced "
                 "against the LDS peak, since a tile of 64 candidate lists is staged once per 512 queries)",
                 bound="lds" if args.model == "siteerr" else "hbm"),
            roof("k_fr_replay_wide (exact replay of the whole-tree searches: a wavefront per search walks the search's expanded "
                 "items and scans the clades in the cached regime over the search's row of the dense score table)", "FR_WIDE",
                 "rank 0; algorithmic bytes = 8 B per placement replayed from the score table + the removed list once per search; the "
                 "rows are bitmaps of their finite scores, and a clade without one is counted instead of walked", bound="counted"),
            roof("k_spr_search (one wavefront per search from its first step: whole-tree searches the frontier tier handed back)",
                 "SPR_REPLAY",
                 "rank 0; algorithmic bytes = 8 B per placement replayed from the score table + the removed list once per search"),
        ]
        roofs = [r for r in roofs if r["launches_timed"]]
        # the headline roofline block: the kernel of the step that takes longest among those that move their bytes
        walking = [r for r in roofs if r["bound"] != "counted"]
        dominant = max(walking, key=lambda r: r["kernel_ms_per_step"]) if walking else None
        # ---- the two kinds of candidate placement of a step
        n_fr, ms_fr, u_fr, b_fr = K["SPR_SEARCH"]                       # the frontier tier as a whole (or the lane searches)
        # (k_fr_replay_wide runs inside the tier, next to k_fr_replay: its time is not taken out of the tier's)
        n_rp, ms_rp, u_rp, b_rp = (K["SPR_REPLAY"][i] + K["FR_WIDE"][i] for i in range(4))
        ms_dense = K["SPR_SCORE"][1]
        split = {
            "full_walk": {"what": "candidate placements of the searches that are not whole-tree searches: each scored by walking the "
                                  "candidate branch's genome list against the removed subtree's list (M:7011 / 7223); kernel time = "
                                  "the frontier tier as a whole (the updating steps of the whole-tree searches, ~10 per search, run "
                                  "in its level kernels, and their replay runs next to the other searches' replay: both are in it)",
                          "placements_per_step": u_fr / steps, "kernel_ms_per_step": ms_fr / steps,
                          "placements_per_s_of_its_kernels": (u_fr / (ms_fr * 1e-3)) if ms_fr else 0.0,
                          "algorithmic_GBps": (b_fr / (ms_fr * 1e-3) / 1e9) if ms_fr else 0.0,
                          "frac_of_hbm_peak": (b_fr / (ms_fr * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_fr else 0.0},
            "replayed": {"what": "candidate placements of whole-tree searches (searches from zero-length branches without an error "
                                 "model: a mismatch over zero length is impossible, M:6663, -inf never counts as a failed pass, the "
                                 "non-strict rule M:7095 descends everywhere): the branches that can score above -inf are found by "
                                 "the witness filter (witness.hip) and walked, the traversal is replayed over the search's row of "
                                 "scores; a clade without a finite score is counted, not walked -- so these placements are "
                                 "candidate placements the reference's search evaluates, not list walks this library performs",
                         "placements_per_step": u_rp / steps, "kernel_ms_per_step": (ms_rp + ms_dense) / steps,
                         "pairs_walked_per_step": K["SPR_SCORE"][2] / steps,
                         "placements_per_s_of_its_kernels": (u_rp / ((ms_rp + ms_dense) * 1e-3)) if (ms_rp + ms_dense) else 0.0},
        }
        out = {
            "metric": "candidate SPR placements/sec", "value": value, "unit": "placements/s",
            "value_walked": total_walked / elapsed,
            "value_walked_note": "of `value`, the candidate placements per second scored by a real walk of the candidate's genome list "
                                 "(searches finished by the frontier / lane tiers + the pairs the witness filter or the dense kernel "
                                 "walked for whole-tree searches); the rest are placements of whole-tree searches proved -inf and counted",
            "first_call_ms": first_call_ms,
            "first_call_note": "the first search call of the process (cold: every pool is allocated in it -- tens of GB of hipMalloc -- and "
                               "scan tables, witness buckets and root-frame copies are built); first_step_after_upload_ms: the tree "
                               "uploaded again after the timed steps (tree_reupload_ms), then one step: the per-tree tables only",
            "first_step_after_upload_ms": first_after_upload_ms, "tree_reupload_ms": reupload_ms,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.samples} synthetic SARS-CoV-2 diff-lists (lRef 29903, ~30 diffs/sample), "
                                   f"{MODEL_TEXT[args.model]}; SPR search ({'fast' if args.spr_fast else 'deep'}-round) of {B} pruned nodes per step; "
                                   f"tree: {'branch lengths optimised as MAPLE does before its SPR rounds' if args.tree == 'optimised' else 'simulated tree, lengths = mutations / lRef'}"
                                   f"{', with MAT local references (' + str(n_ref) + ' reference nodes)' if refs == 'local' else ', no local references'}",
                       "local_references": {"form": refs, "reference_nodes": int(n_ref), "setup_s": round(refs_s, 2),
                                            "note": "local: MAT local references as MAPLE's own trees carry them (a reference node per 50 "
                                                    "descendants; every list of a clade written against its reference node's genome); the plain "
                                                    "form of the same tree is the sub-block spr_search_plain_tree"},
                       "samples": args.samples, "model": args.model, "tree": args.tree, "tree_nodes": int(mirror.n_nodes),
                       "searches_per_step": int(B), "searches_timed": int(total_searches),
                       "candidate_placements_timed": int(total_placements),
                       "placements_split": split,
                       "branch_length_optimisation": blen_opt,
                       "parallelism": f"each step's {B} pruned nodes dealt round-robin in pre-order (coreNum) over {world} GPU(s), "
                                      "tree mirror replicated, one all-gather of proposed moves per step",
                       "setup_s": round(setup_s, 1),
                       "setup_breakdown_s": {"synthetic_input": round(gen_s, 1), "tip_lists_and_upload": round(tips_s, 1),
                                             "generator": data.meta.get("generator", "synth v1 (maple_amd.synth.make_dataset)"), "synth": bt.synth},
                       "resident_inputs": {"genome_lists": int(n_lists_res), "list_bytes": int(8 * n_ent_res + 8 * n_aux_res),
                                           "tree_upload_ms": round(tree_upload_ms, 1),
                                           "note": "lists and tree tables are in HBM when the timed region starts; a step's own "
                                                   "host traffic (node ids in, ~100 B of results per search out) is inside `value`"}},
            "roofline": dominant, "roofline_by_kernel": roofs, "per_rank": per_rank,
            "spr_search": {"status_counts": status_counts, "proposed_moves_rank0": n_moves,
                           "kernel_ms_rank0_per_step": {"frontier_tier": K["SPR_SEARCH"][1] / steps, "of_which_k_fr_updating": K["FR_UPDATING"][1] / steps,
                                                        "of_which_k_fr_cached": K["FR_CACHED"][1] / steps,
                                                        "of_which_replay_refine_finish": K["FR_REPLAY"][1] / steps,
                                                        "of_which_k_fr_replay_wide": K["FR_WIDE"][1] / steps,
                                                        "witness_filter_and_scoring": ms_dense / steps, "replay_outside_the_tier": K["SPR_REPLAY"][1] / steps},
                           "launches_rank0_per_step": {"frontier_levels": K["FR_CACHED"][0] / steps, "witness_filter_and_scoring": K["SPR_SCORE"][0] / steps,
                                                       "replay": n_rp / steps},
                           "params": ("fast round: strict, allowedFailsTopology 2, thresholdLogLKtopology 6 log(lRef)"
                                      if args.spr_fast else
                                      "deep round: non-strict, allowedFailsTopology 4, thresholdLogLKtopology 14 log(lRef)")},
        }
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            out["tree_log_lk"] = tree_log_lk_check(dev, mirror, ht, tip_ids, mkw, ref_idx, root_freqs)
            if not out["tree_log_lk"]["rel_delta"] <= 1e-6:
                raise SystemExit(f"tree log-LK: GPU and oracle differ by more than 1e-6 relative: {out['tree_log_lk']}")
            out["cpu_baseline"] = spr_cpu_baseline(dev, mirror, ht, ref_idx, root_freqs, batch_of, kept, kw, args.cpu_seconds, mkw,
                                                   args.steps)
    if getattr(args, "online_leg", 0) and rank == 0 and world == 1:
        try:                                             # (a secondary leg must never take the headline's line with it)
            out["config_5"] = online_update_leg(dev, mirror, data, ref_idx, tip_kw, kw, args.online_leg, B)
        except (Exception, SystemExit) as e:
            out["config_5"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            print(f"[bench] config 5 leg failed: {out['config_5']['error']}", file=sys.stderr, flush=True)
    if distd is not None:
        distd.barrier()
    dev.close()
    return out if rank == 0 else None


def online_update_leg(dev, mirror, data, ref_idx, tip_kw, kw, n_add, round_nodes, ahead=512):
    """BASELINE configs[4]: ``n_add`` new samples added to the leg's tree one after the other -- the reference's loop M:11692-11752:
    findBestParentForNewSample, placeSampleOnTree, updatePartials; here serial_phase (single-query placement search, the
    stand-in tree edit, maple_update_partials, maple_tree_patch) -- then one round of searches on the grown tree
    (--largeUpdate: the rounds run with every node dirty, M:12143-12159; timed here: ``round_nodes`` of them, the nodes the
    additions touched first).  On the plain form of the tree (every list in the root's frame), as the config-5 test
    (tests/test_hip_configs.py) runs it; the new samples are tips of the tree with one to three changes each
    (maple_amd.synth.perturb_diffs).  Wall times through the Python binding, the sample lists packed beforehand."""
    from maple_amd.host import tip_genome_list
    from maple_amd.synth import perturb_diffs
    l_ref = dev.lRef
    ll = math.log(l_ref)
    pkw = dict(oneMutBLen=1.0 / l_ref, effectivelyNon0BLen=1.0 / (10 * l_ref), thresholdLogLK=18.0 * ll,
               thresholdLogLKoptimization=1.0 * ll, thresholdLogLKconsecutivePlacement=1.0)
    prng = np.random.default_rng(21)
    t0 = time.perf_counter()
    src = prng.choice(len(data.diffs), size=n_add, replace=False)
    new_lists = [tip_genome_list(perturb_diffs(data.diffs[int(i)], data.ref, prng), ref_idx, **tip_kw) for i in src]
    prep_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    sp = serial_phase(dev, mirror, new_lists, pkw, ahead=ahead)
    total_s = time.perf_counter() - t0
    cols = sp["cols"]
    n = cols["n"]
    per = {k: (1e3 * float(np.mean(v)) if len(v) else float("nan")) for k, v in sp["times"].items()}
    third = max(1, len(sp["times"]["search"]) // 3)
    # ---- the round after the update, on the grown tree (the library's copy is current through the patches)
    first = sp["touched_nodes"]
    rest = np.setdiff1d(np.arange(n), first)
    nodes = np.concatenate([first[:: max(1, -(-len(first) // (round_nodes // 2)))], rest[:: max(1, len(rest) // (round_nodes // 2))]])[:round_nodes]
    t0 = time.perf_counter()
    r = dev.spr_search_batch(nodes.astype(np.int64), **kw)
    round_first_ms = 1e3 * (time.perf_counter() - t0)
    t0 = time.perf_counter()
    r = dev.spr_search_batch(nodes.astype(np.int64), **kw)
    round_ms = 1e3 * (time.perf_counter() - t0)
    bad = int((r["status"] < -1).sum())
    if bad:
        raise SystemExit(f"config 5: {bad} searches of the round after the update failed")
    return {"samples_added": int(sp["placed"]), "samples_offered": int(n_add), "skipped": int(sp["skipped"]),
            "ms_per_sample": 1e3 * total_s / max(1, n_add), "total_s": total_s,
            "loop_ms_per_sample_mean": {"upload_of_the_sample": 1e3 * float(np.sum(sp["times"]["upload"])) / max(1, n_add),
                                        "rows_made_ahead": 1e3 * float(np.sum(sp["times"]["ahead"])) / max(1, n_add),
                                        "placement_search": per["search"], "update_partials": per["update"], "tree_patch": per["patch"]},
            "samples_announced_at_a_time": int(ahead),
            "ahead_note": "maple_placement_ahead: the score rows of the next samples made ahead, by an expansion of all of them down the "
                          "tree under permissive rules (every branch the reference's traversal can visit); maple_tree_patch notes the "
                          "columns whose list changed, and a search scores those for its own sample before it reads its row; the searches' "
                          "results are those of the plain loop (tests/test_hip_scale.py)",
            "placement_search_ms_first_third_last_third": [1e3 * float(np.mean(sp["times"]["search"][:third])),
                                                           1e3 * float(np.mean(sp["times"]["search"][-third:]))],
            "nodes_patched_per_sample_mean": float(np.mean(sp["patched"])) if sp["patched"] else 0.0, "nodes_touched": int(len(first)),
            "tree_nodes_before_after": [int(mirror.n_nodes), int(n)],
            "sample_lists_prepared_s": prep_s,
            "round_after": {"searches": int(len(nodes)), "first_call_ms": round_first_ms, "round_after_ms": round_ms,
                            "candidate_placements": int(r["nAppend"][r["status"] >= -1].sum()),
                            "proposed_moves": int((r["placement"] >= 0).sum())},
            "note": "one process, one GPU: the loop is serial by construction (every sample is placed on the tree the one before it "
                    "left); tree edit = bench.serial_phase's stand-in for placeSampleOnTree (host code of the reference, out of scope)"}


DEPTH_STEP = 1 << 12        # (the least a level is worth in serial_phase's depths, see tree_depths)


def tree_depths(root, c0, c1, n, cap):
    """Depths for maple_update_partials (which only compares them) with room in between: a level is worth the largest power of two
    that keeps the deepest tip below 2^30, so that a node put on a branch gets a depth of its own between its neighbours'
    (place_depths) -- 20-odd times over on one branch before the tree has to be numbered again.  Level by level, vectorised."""
    level = np.zeros(cap, dtype=np.int64)
    front = np.asarray([root], dtype=np.int64)
    d = 0
    while len(front):
        level[front] = d
        kids = np.concatenate([c0[front], c1[front]])
        front = kids[kids >= 0].astype(np.int64)
        d += 1
    step = max(DEPTH_STEP, 1 << int(math.floor(math.log2((1 << 30) / (d + 2)))))
    depth = np.zeros(cap, dtype=np.int32)
    depth[:n] = (level[:n] * step).astype(np.int32)
    return depth, step


def place_depths(depth, step, g, p, b, s, root, c0, c1, n):
    """Node p goes on the branch g -> b and gets the new tip s: depths for p and s; the tree is numbered again (returns the new
    array and step) when the branch has no depth left in between."""
    if depth[b] - depth[g] < 2:
        depth, step = tree_depths(root, c0, c1, n, len(depth))             # (p and s are not linked in yet: numbered below)
    depth[p] = (int(depth[g]) + int(depth[b])) // 2
    depth[s] = min(int(depth[p]) + step, (1 << 31) - 2)                    # (a tip: anything below p; room for a later node above it)
    if not (depth[g] < depth[p] < depth[b]):
        raise SystemExit("serial_phase: no depth left between two nodes right after numbering the tree")
    return depth, step


def serial_phase(dev, m, new_lists, pkw, ahead=0):
    """The serial placement phase on the tree of TreeMirror ``m``: the samples ``new_lists`` one after the other --
    single-query placement search, tree edit, maple_update_partials around the new nodes, maple_tree_patch of the touched
    records.  The tree edit is a STAND-IN for MAPLE's placeSampleOnTree (M:8300-8722), which stays host code of the
    reference: a new internal node on the branch above the best node with the three branch lengths the search returned (a
    sample the search calls a minor sequence, or a placement at the root, is skipped) -- good for timing (the edits have
    the shape and the locality of the reference's), not for parity
    (tests/test_hip_search.py::test_online_sample_additions_through_tree_patch applies the reference's own recorded edits).
    The tree lives in plain numpy columns with room to grow; nothing in the loop touches all nodes.  Returns the per-step
    times (s) and the columns of the final tree (which is also the tree uploaded to ``dev`` when it returns).
    ``ahead`` > 0: the samples are announced that many at a time (maple_placement_ahead: their score rows made in one launch of
    the batch kernel and kept current under the patches; the searches' results are the same)."""
    n_add = len(new_lists)
    n0, cap = m.n_nodes, m.n_nodes + 2 * n_add

    def grown(a, fill, dtype):
        out = np.full(cap, fill, dtype=dtype)
        out[:n0] = a
        return out
    up = grown(m.parent, -1, np.int32)
    c0, c1 = grown(m.children[:, 0], -1, np.int32), grown(m.children[:, 1], -1, np.int32)
    tip = grown(m.is_tip, 0, np.uint8)
    dist = grown(m.dist, 0.0, np.float64)
    mut = np.full(cap, -1, dtype=np.int32)
    lower, up_right = grown(m.lower, -1, np.int32), grown(m.up_right, -1, np.int32)
    up_left, tot_up = grown(m.up_left, -1, np.int32), grown(m.tot_up, -1, np.int32)
    depth, dstep = tree_depths(m.root, c0, c1, n0, cap)   # (maple_update_partials only compares depths)
    n = n0
    dev.upload_tree(m.root, up[:n], c0[:n], c1[:n], dist[:n], tip[:n], lower[:n], up_right[:n], up_left[:n], tot_up[:n], mut[:n])
    t = dict(upload=[], search=[], update=[], patch=[], ahead=[])
    placed, skipped, patched, touched_nodes, results = 0, 0, [], [], []
    waiting, rest = [], []                               # list ids of the samples announced and not yet searched / uploaded and not yet announced
    for k, lst in enumerate(new_lists):
        dev.placement_prepare(**pkw)
        t0 = time.perf_counter()
        if ahead > 0:
            if not waiting:
                if not rest:                             # the next samples' lists, uploaded together
                    rest = [int(x) for x in dev.upload(new_lists[k: k + ahead])]
                    t["upload"].append(time.perf_counter() - t0)
                t0 = time.perf_counter()
                got = dev.placement_ahead(np.asarray(rest, dtype=np.int32), **pkw)
                t["ahead"].append(time.perf_counter() - t0)
                # (fewer rows than samples: the others are announced when these are done; no rows at all: the plain search)
                got = got if got > 0 else len(rest)
                waiting, rest = rest[:got], rest[got:]
            qid = waiting.pop(0)
        else:
            qid = int(dev.upload([lst])[0])              # the sample's list stays: it becomes the new tip's lower list
            t["upload"].append(time.perf_counter() - t0)
        mark = dev.mark()
        t0 = time.perf_counter()
        out = dev.placement_search_batch(np.asarray([qid], dtype=np.int32), **pkw)
        t["search"].append(time.perf_counter() - t0)
        dev.release(mark)
        b = int(out["bestNode"][0])
        results.append((int(out["status"][0]), b, float(out["bestScore"][0]), tuple(float(x) for x in out["blen"][0]), int(out["nAppend"][0])))
        if out["status"][0] != 0 or up[b] < 0:
            skipped += 1
            continue
        top, bottom, app = (float(x) for x in out["blen"][0])
        g, p, s = int(up[b]), n, n + 1
        # ---- the stand-in tree edit: p on the branch above b, the sample s as p's other child
        depth, dstep = place_depths(depth, dstep, g, p, b, s, m.root, c0, c1, n)   # (before p is linked in: the tree as it was)
        if c0[g] == b:
            c0[g] = p
        else:
            c1[g] = p
        up[p], c0[p], c1[p], dist[p], tip[p] = g, b, s, top, 0
        up[b], dist[b] = p, bottom
        up[s], dist[s], tip[s], lower[s] = p, app, 1, qid
        n += 2
        # ---- updatePartials around the new nodes, inside the library, on these very columns
        t0 = time.perf_counter()
        dev.update_partials(m.root, up[:n], c0[:n], c1[:n], tip[:n], mut[:n], depth[:n], dist[:n], lower[:n], up_right[:n],
                            up_left[:n], tot_up[:n], [b, s, p])
        t["update"].append(time.perf_counter() - t0)
        # ---- the library's copy of the tree: only the nodes that changed
        t0 = time.perf_counter()
        touched = np.unique(np.concatenate([dev.update_partials_touched(), [g, b, p, s]])).astype(np.int32)
        dev.tree_patch(n, touched, up[touched], c0[touched], c1[touched], dist[touched], tip[touched], lower[touched],
                       up_right[touched], up_left[touched], tot_up[touched])
        t["patch"].append(time.perf_counter() - t0)
        patched.append(len(touched))
        touched_nodes.append(touched)
        placed += 1
    cols = dict(n=n, root=m.root, up=up, c0=c0, c1=c1, dist=dist, tip=tip, lower=lower, up_right=up_right, up_left=up_left,
                tot_up=tot_up, mut=mut)
    return dict(times=t, placed=placed, skipped=skipped, patched=patched, cols=cols, results=results,
                touched_nodes=np.unique(np.concatenate(touched_nodes)) if touched_nodes else np.zeros(0, dtype=np.int32))


def sub_blocks(args, dev, mirror, data, ref_idx, tip_kw, kw, order, B, upload_plain_tree, torch, cu, first_step=None, first_nodes=None,
               headline_refs=False):
    """Secondary measurements next to the headline, never mixed into it (rank 0 only).  They run on the PLAIN form of the tree
    (every list in the root's frame), which is what they were measured on in rounds 1-3."""
    from maple_amd.host import tip_genome_list
    from maple_amd.runtime import Device
    from maple_amd.synth import perturb_diffs
    l_ref = dev.lRef
    out = {}
    upload_plain_tree()
    if headline_refs:
        # ---- the headline's steps once more on the plain form of the same tree (what rounds 1-3 timed) ----
        nsteps = max(1, min(args.steps, 3))
        dev.spr_search_batch(order[np.arange(nsteps * B, (nsteps + 1) * B) % len(order)], **kw)
        t0 = time.perf_counter()
        pl = fails = 0
        for i in range(nsteps):
            r = dev.spr_search_batch(order[np.arange(i * B, (i + 1) * B) % len(order)], **kw)
            pl += int(r["nAppend"][r["status"] >= -1].sum())
            fails += int((r["status"] < -1).sum())
        wall = time.perf_counter() - t0
        out["spr_search_plain_tree"] = {"what": "the same steps on the same tree WITHOUT local references (every list in the root's frame: the "
                                                "tree of rounds 1-3)", "steps": nsteps, "candidate_placements": pl, "failed_or_overflow": fails,
                                        "ms_per_step": 1e3 * wall / nsteps, "placements_per_s": pl / wall}
    # ---- the all-pairs scoring kernel on its own: Q query lists x every candidate branch (appendProbNode, M:8050) ----
    cand_nodes = mirror.candidates_by_length(1.0 / (10 * l_ref))
    cand_lists = mirror.tot_up[cand_nodes]
    q_nodes = np.asarray(data.tip_node[: args.queries], dtype=np.int64)
    q_lists = mirror.lower[q_nodes]
    Q, Cn = len(q_lists), len(cand_lists)
    t_q = torch.from_numpy(q_lists.astype(np.int32)).to(cu)
    t_c = torch.from_numpy(cand_lists.astype(np.int32)).to(cu)
    t_out = torch.empty(Q * Cn, dtype=torch.float64, device=cu)
    stream = torch.cuda.current_stream().cuda_stream
    q_ne, q_na = dev.sizes(q_lists)
    alg = dev.append_algorithmic_bytes(np.tile(cand_lists.astype(np.int32), Q)) + 8 * int(q_ne.sum() + q_na.sum())
    for _ in range(2):
        dev.append_queries_dev(Q, t_q.data_ptr(), Cn, t_c.data_ptr(), True, 1.0 / l_ref, t_out.data_ptr(), stream)
    torch.cuda.synchronize()
    dev.timing_reset()
    for _ in range(5):
        dev.append_queries_dev(Q, t_q.data_ptr(), Cn, t_c.data_ptr(), True, 1.0 / l_ref, t_out.data_ptr(), stream)
    torch.cuda.synchronize()
    n_l, ms, _, _ = dev.timing_read_kind(Device.KIND_APPEND_QUERIES)
    # the same pairs with the arg-max fused into the kernel (wavefront reduction per 64-candidate tile, no score matrix)
    t_best = torch.empty(Q, dtype=torch.float64, device=cu)
    t_idx = torch.empty(Q, dtype=torch.int32, device=cu)
    dev.append_queries_argmax_dev(Q, t_q.data_ptr(), Cn, t_c.data_ptr(), 0, True, 1.0 / l_ref, t_best.data_ptr(), t_idx.data_ptr(), stream)
    torch.cuda.synchronize()
    dev.timing_reset()
    for _ in range(5):
        dev.append_queries_argmax_dev(Q, t_q.data_ptr(), Cn, t_c.data_ptr(), 0, True, 1.0 / l_ref, t_best.data_ptr(), t_idx.data_ptr(), stream)
    torch.cuda.synchronize()
    n_f, ms_f, _, _ = dev.timing_read_kind(Device.KIND_APPEND_QUERIES)
    m_best, m_idx = t_out.view(Q, Cn).max(dim=1)
    fused_ok = bool(torch.equal(m_best, t_best))
    out["all_pairs_kernel"] = {"queries": int(Q), "candidate_branches": int(Cn), "kernel_ms": ms / max(1, n_l),
                               "pairs_per_s": Q * Cn / (ms / max(1, n_l) * 1e-3),
                               "algorithmic_GBps": alg / (ms / max(1, n_l) * 1e-3) / 1e9,
                               "frac_of_hbm_peak": alg / (ms / max(1, n_l) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "fused_argmax_kernel_ms": ms_f / max(1, n_f), "fused_argmax_equals_matrix_max": fused_ok}
    # ---- batched placement search (findBestParentForNewSample for many samples on the frozen tree, M:7912-8292 /
    # 11190-11220): all-branch scoring + device-side traversal + short-list refinement ----
    ll = math.log(l_ref)
    pkw = dict(oneMutBLen=1.0 / l_ref, effectivelyNon0BLen=1.0 / (10 * l_ref), thresholdLogLK=18.0 * ll,
               thresholdLogLKoptimization=1.0 * ll, thresholdLogLKconsecutivePlacement=1.0)
    mark = dev.mark()
    prng = np.random.default_rng(11)
    new_samples = [tip_genome_list(perturb_diffs(data.diffs[i], data.ref, prng), ref_idx, **tip_kw) for i in range(Q)]
    new_ids = dev.upload(new_samples)                              # samples NOT in the tree (2 extra substitutions each)
    dev.placement_search_batch(new_ids[:8], **pkw)
    t0 = time.perf_counter()
    pres = dev.placement_search_batch(new_ids, **pkw)
    pwall = time.perf_counter() - t0
    ts = []
    for qid in new_ids[:32]:                                       # ... and one query at a time (the serial placement phase)
        t0 = time.perf_counter()
        dev.placement_search_batch(np.asarray([qid], dtype=np.int32), **pkw)
        ts.append(time.perf_counter() - t0)
    dev.release(mark)
    out["placement_batch"] = {"queries": int(Q), "wall_ms": 1e3 * pwall, "queries_per_s": Q / pwall,
                              "reference_equivalent_placements_per_s": float(pres["nAppend"].sum() / pwall),
                              "branches_scored_per_s": float(Q * (Cn + 1) / pwall),
                              "minor_sequences": int((pres["status"] == 1).sum()), "failed": int((pres["status"] < 0).sum())}
    # ---- the serial path of the placement phase (one sample, then one repair of the lists, M:11744-11752): one query at a
    # time through the same entry point, and updatePartials for one changed branch through maple_update_partials
    from maple_amd.tree_host import HostTree, update_genome_lists
    mark = dev.mark()
    ht1 = HostTree.from_mirror(mirror)
    rng1 = np.random.default_rng(5)
    cand1 = np.nonzero((mirror.parent >= 0) & (mirror.dist > 1e-5))[0]
    tu, rep = [], 0
    for v in rng1.choice(cand1, size=33, replace=False):
        ht1.dist[v] = ht1.dist[v] * 1.5
        t0 = time.perf_counter()
        rep += update_genome_lists(dev, ht1, [int(v)])
        tu.append(time.perf_counter() - t0)
    dev.release(mark)
    # ... and the loop itself: 64 new samples placed one after the other (serial_phase above: search, stand-in tree edit,
    # maple_update_partials, maple_tree_patch)
    mark = dev.mark()
    sp = serial_phase(dev, mirror, new_samples[:64], pkw)

    def med_ms(x):
        return 1e3 * float(np.median(x[len(x) // 8:])) if len(x) else float("nan")
    per = {k: med_ms(v) for k, v in sp["times"].items()}
    upload_plain_tree()                                  # (the original tree again; the loop's lists go with the mark)
    dev.release(mark)
    out["serial_path"] = {"single_query_placement_ms_median": 1e3 * float(np.median(ts[1:])),
                          "single_change_update_partials_ms_median": 1e3 * float(np.median(tu[1:])),
                          "lists_replaced_per_change": rep / len(tu),
                          "loop_ms_per_sample": {"upload_of_the_sample": per["upload"], "placement_search": per["search"],
                                                 "update_partials": per["update"], "tree_patch": per["patch"],
                                                 "total": sum(per.values())},
                          "loop_samples": int(sp["placed"]), "nodes_patched_per_sample_median": float(np.median(sp["patched"])) if sp["patched"] else 0.0,
                          "note": "wall times through the Python binding; the loop's tree edit is a stand-in for MAPLE's "
                                  "placeSampleOnTree (bench.serial_phase); the reference's CPython updatePartials takes ~0.4 ms"}
    # ---- the apply phase of the round: the proposed moves of the first timed step, best first, re-searched one at a time on the
    # current tree and applied (bench.apply_phase)
    if first_step is not None:
        from maple_amd.spr_apply import SprApplier
        res0 = first_step
        prop = np.nonzero(res0["placement"] >= 0)[0]
        prop = prop[np.argsort(-res0["improvement"][prop], kind="stable")]
        moves = first_nodes[prop][:96]

        def med(x):
            return 1e3 * float(np.median(x)) if len(x) else float("nan")
        rep = {}
        for mode in ("sequential", "batched"):
            mark = dev.mark()
            ap = SprApplier.from_mirror(dev, mirror)
            t0 = time.perf_counter()
            (ap.apply_sequential if mode == "sequential" else ap.apply_batched)(moves, kw)
            wall = time.perf_counter() - t0
            rep[mode] = ap
            out_ap = {"moves": int(len(moves)), "applied": len(ap.applied), "no_longer_proposed": int(ap.no_longer_proposed),
                      "skipped_by_the_stand_in_edit": int(ap.skipped), "wall_ms_per_move": 1e3 * wall / max(1, len(moves)),
                      "ms_median": {"search_call": med(ap.times["search"]), "update_partials": med(ap.times["update"]),
                                    "tree_patch": med(ap.times["patch"])},
                      "search_calls": len(ap.times["search"]),
                      "nodes_patched_per_move_median": float(np.median(ap.patched)) if ap.patched else 0.0}
            if mode == "batched":
                out_ap["batches_searched_kept"] = [list(x) for x in ap.batches]
            else:
                out_ap["whole_tree_re_searches"] = int(ap.whole_tree_searches)
                out_ap["search_call_ms_mean"] = 1e3 * float(np.mean(ap.times["search"])) if ap.times["search"] else float("nan")
                out_ap["search_call_ms_max"] = 1e3 * float(np.max(ap.times["search"])) if ap.times["search"] else float("nan")
            out["serial_path"].setdefault("apply_phase", {})[mode] = out_ap
            upload_plain_tree()
            dev.release(mark)
        out["serial_path"]["apply_phase"]["same_applied_sequence"] = rep["sequential"].applied == rep["batched"].applied
        out["serial_path"]["apply_phase"]["note"] = (
            "applySPRMovesParallel (M:9470-9484) over the proposed moves of the first timed step, best first, with a stand-in "
            "tree edit (maple_amd/spr_apply.py); sequential = one re-search per move (frontier tier on the patched node records; a "
            "re-search from a zero-length branch -- a whole-tree search -- after the tree's tables were brought up to date); "
            "batched = 32 moves re-searched per call, a speculative result kept while nothing its search may have read was "
            "touched by the moves applied before it")
    if first_step is not None and len(first_nodes) == len(order):
        # ---- a WHOLE round on a tree that changes (M:12283-12316 search, 12306-12312 gather + sort, applySPRMovesParallel
        # M:9470-9484, then the next round's search): every node searched (the timed steps' round), EVERY proposed move
        # re-searched and applied best first, the next round searched on the tree the moves left -- its first call (the
        # per-tree tables follow the patches or are rebuilt) against the call after it
        from maple_amd.spr_apply import SprApplier
        try:
            mark = dev.mark()
            ap = SprApplier.from_mirror(dev, mirror)
            all_nodes = np.asarray(order)
            t0 = time.perf_counter()
            r1 = dev.spr_search_batch(all_nodes, **kw)
            search1_ms = 1e3 * (time.perf_counter() - t0)
            prop = np.nonzero(r1["placement"] >= 0)[0]
            prop = prop[np.argsort(-r1["improvement"][prop], kind="stable")]
            t0 = time.perf_counter()
            ap.apply_batched(all_nodes[prop], kw)
            apply_s = time.perf_counter() - t0
            t0 = time.perf_counter()
            r2 = dev.spr_search_batch(all_nodes, **kw)
            search2_first_ms = 1e3 * (time.perf_counter() - t0)
            t0 = time.perf_counter()
            r3 = dev.spr_search_batch(all_nodes, **kw)
            search2_ms = 1e3 * (time.perf_counter() - t0)
            out["round_on_a_changing_tree"] = {
                "searches_per_round": int(len(all_nodes)), "search_ms": search1_ms, "proposed_moves": int(len(prop)),
                "moves_applied": len(ap.applied), "no_longer_proposed": int(ap.no_longer_proposed),
                "skipped_by_the_stand_in_edit": int(ap.skipped), "apply_phase_s": apply_s,
                "apply_ms_per_proposed_move": 1e3 * apply_s / max(1, len(prop)), "search_calls_of_the_apply_phase": len(ap.times["search"]),
                "whole_tree_re_searches": int(ap.whole_tree_searches),
                "apply_ms_split": {"search_calls": 1e3 * float(np.sum(ap.times["search"])), "update_partials": 1e3 * float(np.sum(ap.times["update"])),
                                   "tree_patch": 1e3 * float(np.sum(ap.times["patch"]))},
                "round_s": (search1_ms * 1e-3) + apply_s,
                "next_round_first_search_ms": search2_first_ms, "next_round_search_ms": search2_ms,
                "first_over_steady": search2_first_ms / max(1e-9, search2_ms),
                "next_round_proposed_moves": int((r3["placement"] >= 0).sum()),
                "next_round_same_twice": bool(np.array_equal(r2["placement"], r3["placement"]) and np.array_equal(r2["bestScore"], r3["bestScore"])),
                "note": "plain form of the headline tree; the tree edit is maple_amd/spr_apply.py's stand-in for cutAndPasteNode (host "
                        "code of the reference, out of scope); apply_batched: 32 moves re-searched per call, a speculative result kept "
                        "while nothing its search read was touched by the moves applied before it (the applied sequence is the "
                        "sequential driver's, tests/test_hip_scale.py)"}
            upload_plain_tree()
            dev.release(mark)
        except (Exception, SystemExit) as e:
            out["round_on_a_changing_tree"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            upload_plain_tree()
    if (args.local_refs or args.samples <= 200000) and not headline_refs:
        # ---- the same steps on the same tree after giving it MAT local references (setUpMAT's rule, 50 descendants per
        # reference node, M:166 / 6152-6164): the form real MAPLE trees have; lists are shorter, searches cross frames ----
        from maple_amd.mat import add_local_references
        from maple_amd.tree_host import HostTree
        ht = HostTree.from_mirror(mirror)
        t0 = time.perf_counter()
        n_ref = add_local_references(dev, ht, 50)
        mat_s = time.perf_counter() - t0
        dev.upload_tree(ht.root, mirror.parent, mirror.children[:, 0], mirror.children[:, 1], mirror.dist, mirror.is_tip,
                        ht.id_lower, ht.id_upRight, ht.id_upLeft, ht.id_totUp, ht.id_mut)
        nsteps = max(1, min(args.steps, 4 if args.samples <= 20000 else 1))
        dev.spr_search_batch(order[np.arange(nsteps * B, (nsteps + 1) * B) % len(order)], **kw)
        dev.timing_reset()
        t0 = time.perf_counter()
        pl = fails = 0
        for i in range(nsteps):
            r = dev.spr_search_batch(order[np.arange(i * B, (i + 1) * B) % len(order)], **kw)
            pl += int(r["nAppend"][r["status"] >= -1].sum())
            fails += int((r["status"] < -1).sum())
        wall = time.perf_counter() - t0
        out["spr_search_local_refs"] = {"reference_nodes": int(n_ref), "setup_s": round(mat_s, 2), "steps": nsteps,
                                        "candidate_placements": pl, "failed_or_overflow": fails, "wall_ms": 1e3 * wall,
                                        "placements_per_s": pl / wall}
        upload_plain_tree()
    return out


def spr_cpu_baseline(dev, mirror, ht, ref_idx, root_freqs, batch_of, gpu_results, kw, cpu_seconds, mkw, steps):
    """The C oracle's SPR search (a port of findBestParentTopology + the worker body, oracle/maple_oracle_search.c,
    pinned to the reference's recorded searches) on one host core and on all of them (OpenMP over searches) over bounded,
    evenly spread samples of the timed searches; also cross-checks the GPU's node ids, moves and candidate counts."""
    from oracle.oracle_py import Oracle, OracleTree
    orc = Oracle(ref_idx, root_freqs)
    orc.set_model(**mkw)
    n = mirror.n_nodes
    lists4 = []
    # (ht: the timed tree carried MAT local references -- its lists, branch lengths and mutation lists)
    cols = (mirror.lower, mirror.up_right, mirror.up_left, mirror.tot_up) if ht is None else (ht.id_lower, ht.id_upRight, ht.id_upLeft, ht.id_totUp)
    for ids in cols:
        ids = np.asarray(ids)
        have = np.nonzero(ids >= 0)[0]
        have = have[np.argsort(ids[have], kind="stable")]              # arena order: the download moves whole runs
        lists4.append((have, dev.download_packed(ids[have])))
    dist = mirror.dist if ht is None else np.asarray([float(x or 0.0) for x in ht.dist])
    otree = OracleTree(orc, mirror.root, mirror.parent.astype(np.int32), mirror.children, dist, None if ht is None else ht.mutations,
                       np.zeros(n, dtype=np.int32), lists4)
    nodes = np.concatenate([batch_of(i) for i in range(steps)])
    gpu = {k: np.concatenate([r[k] for r in gpu_results]) for k in ("status", "bestNode", "placement", "nAppend", "bestScore")}
    ties = [0]
    cores = usable_host_threads()

    def timed_sample(threads, budget_s, stride0, seen):
        """Grow an evenly spread sample of the timed searches (every stride0-th, then every (stride0 / 2)-th, ...) until the
        time budget is used; every result is checked against the GPU's."""
        t_used, placements, checked = 0.0, 0, 0
        stride = stride0
        while t_used < budget_s and stride >= 1:
            sel = np.asarray([i for i in range(0, len(nodes), stride) if i not in seen], dtype=np.int64)
            if len(sel) == 0:
                break
            # bound one call by what is left of the budget (a whole-tree search costs ~10 ms on one core)
            est = max(1e-9, t_used / max(1, placements)) if placements else 6e-8 / threads
            budget_pl = max(1.0, (budget_s - t_used) / est)
            cum = np.cumsum(gpu["nAppend"][sel].astype(np.float64))
            sel = sel[: max(1, int(np.searchsorted(cum, budget_pl)) + 1)]
            t0 = time.perf_counter()
            o = orc.spr_worker(otree, nodes[sel], threads=threads, **kw)
            t_used += time.perf_counter() - t0
            placements += int(o["nAppend"].sum())
            # node ids, moves and candidate counts are compared exactly.  The one tolerated difference: the final selection
            # (M:7635) takes the later of two equal optimised scores, and the device's log() and the host's differ in the last
            # bit, so when two BRANCHES tie to 1e-11 the search may end on the other one -- bestNode differs, both report the same
            # score, and the move follows the node.  Nothing else is excused: the same bestNode with another placement, another
            # candidate count or status is an error, and so is more than a handful of ties (counted and reported).
            tie = (np.abs(o["bestScore"] - gpu["bestScore"][sel]) <= 1e-11 * np.maximum(1.0, np.abs(o["bestScore"]))) \
                & (o["bestNode"] != gpu["bestNode"][sel]) & (o["bestNode"] >= 0) & (gpu["bestNode"][sel] >= 0)
            ties[0] += int(tie.sum())
            for k in ("status", "bestNode", "placement", "nAppend"):
                diff = o[k] != gpu[k][sel]
                if k in ("bestNode", "placement"):
                    diff &= ~tie
                if diff.any():
                    j = int(np.nonzero(diff)[0][0])
                    raise SystemExit(f"GPU SPR search disagrees with the oracle on {k}: timed search {int(sel[j])} (node {int(nodes[sel[j]])}, "
                                     f"{threads} oracle thread(s)): oracle " + str({q: o[q][j].tolist() for q in ("status", "bestNode", "placement", "nAppend", "bestScore")})
                                     + " GPU " + str({q: gpu[q][sel[j]].tolist() for q in ("status", "bestNode", "placement", "nAppend", "bestScore")}))
            if ties[0] > max(3, (checked + len(sel)) // 500):
                raise SystemExit(f"{ties[0]} of {checked + len(sel)} checked searches end on another branch with an equal score: too many to be ties")
            checked += len(sel)
            seen.update(sel.tolist())
            stride //= 2
        return t_used, placements, checked
    seen = set()
    t1, p1, c1 = timed_sample(1, cpu_seconds / 3.0, 997, seen)
    tn, pn, cn = timed_sample(cores, 2.0 * cpu_seconds / 3.0, 499, seen) if cores > 1 else (t1, p1, c1)
    full = float(gpu["nAppend"][gpu["status"] >= -1].sum())
    return {"value": pn / tn, "unit": "placements/s", "cores": cores, "kind": "port",
            "sample_short": f"{cn} of {len(nodes)} timed searches, {pn} placements, C oracle (OpenMP) on {cores} threads, {tn:.1f} s; results = GPU's",
            "sample": f"{cn} of the {len(nodes)} timed searches (evenly spread), {pn} candidate placements, C oracle search "
                      f"(oracle/maple_oracle_search.c, OpenMP over searches like the reference's Pool.map over --numCores, M:12283-12293) "
                      f"on {cores} host threads, {tn:.1f} s; node ids, moves and candidate counts identical to the GPU's"
                      + (f" ({ties[0]} searches whose two best branches tie to 1e-11 end on the other one)" if ties[0] else ""),
            "one_core": {"value": p1 / t1, "unit": "placements/s", "cores": 1,
                         "sample": f"{c1} searches, {p1} candidate placements, {t1:.1f} s"},
            "whole_workload_estimate_s": {"one_core": full / (p1 / t1), f"{cores}_cores": full / (pn / tn)}}


def lk_post_order(root, children):
    """Internal nodes in the order calculateTreeLikelihood adds their contributions (M:9721-9779: post-order, child 0 first)."""
    order, stack = [], [(int(root), False)]
    while stack:
        v, done = stack.pop()
        if children[v][0] < 0:
            continue
        if done:
            order.append(v)
        else:
            stack.append((v, True))
            stack.append((int(children[v][1]), False))
            stack.append((int(children[v][0]), False))
    return np.asarray(order, dtype=np.int64)


def oracle_tree_log_lk(cpu, root, parent, children, dist, tip_nodes, tip_packed, mutations=None):
    """calculateTreeLikelihood (M:9721-9779) by the C oracle alone, from the TIPS' lists and the branch lengths: the lower
    lists bottom-up (mergeVectors + shorten per internal node, reCalculateAllGenomeLists pass 1, M:6031-6200), each merge
    with returnLK (M:9756), the contributions summed in the reference's post-order, + findProbRoot (M:4865-4912) of the
    root's list.  ``mutations`` (per node, the MAT mutation list of the branch above a reference node, M:8296-8354): the tips'
    lists -- handed in against the root's reference -- are first taken down into their frames (passGenomeListThroughBranch
    through every reference branch above them, outermost first, then shorten), and a reference node's lower list goes up
    through its branch before it is merged (M:9749-9754); the value then is the reference's for THAT form of the tree (its
    whole-genome term, M:4487, always uses the root's reference, so the two forms differ in the fifth digit).
    ``cpu`` is a Device over oracle/libmaple_cpu.so (the CPU twin: oracle/maple_oracle.c behind the same C ABI); nothing of the
    GPU's enters.  Returns (log-LK, root term)."""
    n = len(parent)
    children = np.asarray(children)
    parent = np.asarray(parent)
    is_tip = children[:, 0] < 0
    depth = np.zeros(n, dtype=np.int64)
    levels = []
    level = np.asarray([root], dtype=np.int64)
    while len(level):
        depth[level] = len(levels)
        levels.append(level)
        ch = children[level].reshape(-1)
        level = ch[ch >= 0]
    tip_nodes = np.asarray(tip_nodes, dtype=np.int64)
    cur = cpu.upload_packed(tip_packed)
    mut_id = -np.ones(n, dtype=np.int32)
    has = [v for v in range(n) if mutations[v]] if mutations is not None else []
    if has:
        mut_id[has] = cpu.upload_mutations([mutations[v] for v in has])
        frame = -np.ones(n, dtype=np.int64)                  # the innermost reference node at or above each node
        for lev in levels[1:]:
            frame[lev] = np.where(mut_id[lev] >= 0, lev, frame[parent[lev]])
        uniq, inv = np.unique(frame[tip_nodes], return_inverse=True)
        chains = []
        for f in uniq.tolist():                              # the reference nodes above a tip, outermost first
            ch = []
            while f >= 0:
                ch.append(f)
                f = int(frame[parent[f]])
            chains.append(ch[::-1])
        for k in range(max(len(c) for c in chains)):
            fk = np.asarray([c[k] if len(c) > k else -1 for c in chains], dtype=np.int64)[inv]
            need = np.nonzero(fk >= 0)[0]
            cur[need] = cpu.pass_branch_batch(cur[need], mut_id[fk[need]], False)
        cur = cpu.shorten_batch(cur)
    lower = -np.ones(n, dtype=np.int32)
    lower[tip_nodes] = cur
    lk_node = np.zeros(n)
    for lev in reversed(levels):
        nodes = lev[~is_tip[lev]]
        if len(nodes) == 0:
            continue
        c0, c1 = children[nodes, 0], children[nodes, 1]
        l0, l1 = lower[c0].copy(), lower[c1].copy()
        for arr, ch in ((l0, c0), (l1, c1)):
            need = np.nonzero(mut_id[ch] >= 0)[0]
            if len(need):
                arr[need] = cpu.pass_branch_batch(arr[need], mut_id[ch[need]], True)
        out, lk = cpu.merge_batch(l0, dist[c0], is_tip[c0], l1, dist[c1], is_tip[c1], False, returnLK=True)
        if (out < 0).any():
            raise SystemExit("oracle_tree_log_lk: mergeVectors returned None (inconsistent zero-length branch, M:9761)")
        lower[nodes] = cpu.shorten_batch(out)
        lk_node[nodes] = lk
    total = 0.0
    for x in lk_node[lk_post_order(root, children)].tolist():
        total += x
    root_lk = float(cpu.root_prob_batch([lower[root]])[0])
    return total + root_lk, root_lk


def tree_log_lk_check(dev, mirror, ht, tip_ids, mkw, ref_idx, root_freqs):
    """The metric's second half ("tree log-LK delta vs ref"): calculateTreeLikelihood of the bench tree by the library
    (maple_merge_batch(returnLK) over every internal node's two stored lower lists + maple_root_prob_batch,
    tree_host.tree_log_likelihood -- on the tree as the timed steps searched it, local references included) against the C
    oracle's own value from the tips' lists (oracle_tree_log_lk: no list of the GPU's enters).  North star: <= 1e-6 relative."""
    import ctypes
    from maple_amd.runtime import Device
    from maple_amd.tree_host import HostTree, tree_log_likelihood
    from oracle.oracle_py import build as build_oracle
    t0 = time.perf_counter()
    tree = ht if ht is not None else HostTree.from_mirror(mirror)
    gpu_lk, gpu_root = tree_log_likelihood(dev, tree)
    gpu_s = time.perf_counter() - t0
    build_oracle()
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libmaple_cpu.so"))
    lib.maple_last_error.restype = ctypes.c_char_p
    cpu = Device(ref_idx, root_freqs, lib=lib)
    cpu.set_model(**mkw)
    tips = np.nonzero(mirror.is_tip)[0]
    t0 = time.perf_counter()
    dist = np.asarray([float(x or 0.0) for x in tree.dist]) if ht is not None else mirror.dist
    orc_lk, orc_root = oracle_tree_log_lk(cpu, mirror.root, mirror.parent, mirror.children, dist, tips, dev.download_packed(tip_ids[tips]),
                                          None if ht is None else ht.mutations)
    orc_s = time.perf_counter() - t0
    cpu.close()
    return {"gpu": gpu_lk, "oracle": orc_lk, "rel_delta": abs(gpu_lk - orc_lk) / abs(orc_lk), "tolerance": 1e-6,
            "gpu_root_term": gpu_root, "oracle_root_term": orc_root, "gpu_s": round(gpu_s, 2), "oracle_s": round(orc_s, 2),
            "what": "calculateTreeLikelihood (M:9721-9779) of the bench tree in the form the timed steps searched (with its local "
                    "references): the library over its stored lists vs oracle/libmaple_cpu.so from the tips' lists alone"}


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


if __name__ == "__main__":
    main()
