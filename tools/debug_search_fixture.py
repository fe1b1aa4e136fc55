"""Which tier of the SPR search disagrees with a recorded fixture?  usage: debug_search_fixture.py <fixture> [round]"""
import sys, os, gzip, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import GOLDEN, model_args, ref_indices
from maple_amd.runtime import Device
from maple_amd.tree_host import HostTree

name = sys.argv[1]
f = json.load(gzip.open(os.path.join(GOLDEN, f"search_{name}.json.gz"), "rt"))
ctx, t = f["context"], f["tree"]
dev = Device(ref_indices(ctx), ctx["rootFreqs"], thresholdProb=ctx["thresholdProb"], minBLenSensitivity=ctx["minBLenSensitivity"],
             thresholdDiffForUpdate=ctx["thresholdDiffForUpdate"], thresholdFoldChangeUpdate=ctx["thresholdFoldChangeUpdate"],
             defaultBLen=ctx["defaultBLen"], arena_bytes=256 << 20)
dev.set_model(**model_args(f["model"]))
tree = HostTree(t["root"], t["up"], t["children"], t["dist"], t["mutations"], t["nMinor"], t["probVect"], t["probVectUpRight"],
                t["probVectUpLeft"], t["probVectTotUp"]).upload(dev)
for ri, rnd in enumerate(f["spr"]):
    ps, calls = rnd["params"], rnd["calls"]
    nodes = [tree.children[c["node"]][c["child"]] for c in calls]
    kw = dict(strict=ps["strict"], allowedFails=ps["fails"], thresholdLogLKtopology=ps["thr"], thresholdTopologyPlacement=ps["place"],
              thresholdLogLKoptimizationTopology=ctx["thresholdLogLKoptimizationTopology"],
              thresholdLogLKconsecutivePlacement=ctx["thresholdLogLKconsecutivePlacement"], effectivelyNon0BLen=ctx["effectivelyNon0BLen"])
    for label, extra in (("auto", {}), ("lane-only kernel, never dense", dict(search_tier=1, wide_search_budget=-1)),
                         ("lane kernel + dense", dict(search_tier=1)), ("frontier, never dense", dict(wide_search_budget=-1)),
                         ("budget 8", dict(wide_search_budget=8)), ("lane kernel budget 8", dict(search_tier=1, wide_search_budget=8))):
        out = dev.spr_search_batch(nodes, **kw, **extra)
        bad = []
        for k, c in enumerate(calls):
            w = c["ret"]
            if w is None:
                continue
            if out["status"][k] != 0 or int(out["bestNode"][k]) != w["bestNode"] or int(out["nAppend"][k]) != c["n_append"] \
                    or abs(out["bestScore"][k] - w["bestScore"]) > 1e-8 * max(1, abs(w["bestScore"])):
                bad.append((k, nodes[k], int(out["status"][k]), int(out["bestNode"][k]), w["bestNode"], int(out["nAppend"][k]), c["n_append"],
                            float(out["bestScore"][k]), w["bestScore"]))
        print(f"round {ri} [{label}]: {len(bad)} of {len(calls)} differ; first: {bad[:4]}", flush=True)
