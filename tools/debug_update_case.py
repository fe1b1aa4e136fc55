import sys, os, gzip, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import GOLDEN, model_args, ref_indices, tup
from maple_amd.runtime import Device
from maple_amd.tree_host import HostTree, update_genome_lists, tree_log_likelihood
name, idx = sys.argv[1], int(sys.argv[2])
f = json.load(gzip.open(os.path.join(GOLDEN, f"search_{name}.json.gz"), "rt"))
u = json.load(gzip.open(os.path.join(GOLDEN, f"update_{name}.json.gz"), "rt"))
ctx, t = f["context"], f["tree"]
dev = Device(ref_indices(ctx), ctx["rootFreqs"], thresholdProb=ctx["thresholdProb"], minBLenSensitivity=ctx["minBLenSensitivity"],
             thresholdDiffForUpdate=ctx["thresholdDiffForUpdate"], thresholdFoldChangeUpdate=ctx["thresholdFoldChangeUpdate"],
             defaultBLen=ctx["defaultBLen"], arena_bytes=256 << 20)
dev.set_model(**model_args(f["model"]))
case = u["cases"][idx]
tree = HostTree(t["root"], t["up"], t["children"], t["dist"], t["mutations"], t["nMinor"], t["probVect"], t["probVectUpRight"],
                t["probVectUpLeft"], t["probVectTotUp"]).upload(dev)
ch = case["change"]; v = ch["node"]
print("change", ch["kind"], v, "ref dist changes", case["dist"], "ref lists touched", {k: list(x.keys()) for k, x in case["lists"].items()})
d0 = tree.dist.copy()
pre = {a: getattr(tree, a).copy() for a in ("id_lower", "id_upRight", "id_upLeft", "id_totUp")}
if ch["kind"] == "dist":
    tree.dist[v] = ch["dist"]
else:
    tree.id_lower[v] = dev.upload([tup(ch["probVect"])])[0]
try:
    rep = update_genome_lists(dev, tree, [v])
    print("replaced", rep)
except Exception as e:
    print("ERROR", e)
print("dist changed:", {int(i): (d0[i], tree.dist[i]) for i in np.nonzero(d0 != tree.dist)[0]})
for a in pre:
    print(a, "replaced at", np.nonzero(getattr(tree, a) != pre[a])[0].tolist())
got, _ = tree_log_likelihood(dev, tree)
print("LK", got, "want", case["treeLK"])
