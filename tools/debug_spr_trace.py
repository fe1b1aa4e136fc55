#!/usr/bin/env python3
"""Debug helper (GPU box): print the device search's visit trace for one recorded query."""
import gzip, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from golden_util import GOLDEN, model_args, ref_indices
from maple_amd.runtime import Device
from maple_amd.tree_host import HostTree
name, rnd_i, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = json.load(gzip.open(os.path.join(GOLDEN, f"search_{name}.json.gz"), "rt"))
ctx, t = f["context"], f["tree"]
dev = Device(ref_indices(ctx), ctx["rootFreqs"], thresholdProb=ctx["thresholdProb"], minBLenSensitivity=ctx["minBLenSensitivity"],
             thresholdDiffForUpdate=ctx["thresholdDiffForUpdate"], thresholdFoldChangeUpdate=ctx["thresholdFoldChangeUpdate"],
             defaultBLen=ctx["defaultBLen"], arena_bytes=256 << 20, debug=True)
dev.set_model(**model_args(f["model"]))
tree = HostTree(t["root"], t["up"], t["children"], t["dist"], t["mutations"], t["nMinor"], t["probVect"],
                t["probVectUpRight"], t["probVectUpLeft"], t["probVectTotUp"]).upload(dev)
rnd = f["spr"][rnd_i]
ps, c = rnd["params"], rnd["calls"][k]
node = tree.children[c["node"]][c["child"]]
dev.debug_trace_query(0)
out = dev.spr_search_batch([node], strict=ps["strict"], allowedFails=ps["fails"], thresholdLogLKtopology=ps["thr"],
                           thresholdTopologyPlacement=ps["place"],
                           thresholdLogLKoptimizationTopology=ctx["thresholdLogLKoptimizationTopology"],
                           thresholdLogLKconsecutivePlacement=ctx["thresholdLogLKconsecutivePlacement"],
                           effectivelyNon0BLen=ctx["effectivelyNon0BLen"])
it, va = dev.debug_trace_read()
print("query", c["node"], c["child"], "pruned", node, "nAppend", out["nAppend"][0], "ref", c["n_append"], "trace", c.get("trace"))
for a, b in zip(it, va):
    print(list(a), list(b))
