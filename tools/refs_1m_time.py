"""How long do local references take at 1 000 000 tips?  (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.host import reference_tables, tip_lists_packed
from maple_amd.mat import add_local_references
from maple_amd.runtime import Device
from maple_amd.synth import make_dataset_native
from maple_amd.tree_host import HostTree
from maple_amd.tree_mirror import TreeMirror
samples = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
model = "siteerr"
data = make_dataset_native(n_samples=samples, l_ref=29903, seed=1, mean_diffs=30.0, rate_variation=True)
ref_idx, root_freqs = reference_tables(data.ref)
dev = Device(ref_idx, root_freqs, arena_bytes=min(128 << 30, samples * (96 << 10)))
mkw = bench.model_kwargs(model, len(ref_idx))
dev.set_model(**mkw)
c = data.diffs
m = TreeMirror(dev, data.parent, data.blen, tip_packed=(data.tip_node, tip_lists_packed(c.off, c.code, c.pos, c.length, ref_idx, error_rates=mkw["errorRates"]))).build()
t0 = time.perf_counter(); ht = HostTree.from_mirror(m); t1 = time.perf_counter()
n_ref = add_local_references(dev, ht, 50); t2 = time.perf_counter()
print(f"HostTree.from_mirror {t1 - t0:.1f} s, add_local_references {t2 - t1:.1f} s, {n_ref} reference nodes", flush=True)
dist = np.asarray([float(x or 0.0) for x in ht.dist])
t0 = time.perf_counter()
dev.upload_tree(ht.root, m.parent, m.children[:, 0], m.children[:, 1], dist, m.is_tip, ht.id_lower, ht.id_upRight, ht.id_upLeft, ht.id_totUp, ht.id_mut)
print(f"upload {time.perf_counter() - t0:.1f} s", flush=True)
kw = bench.search_kwargs(dev.lRef)
order = bench.preorder_nodes(m)
for i in range(3):
    t0 = time.perf_counter()
    r = dev.spr_search_batch(order[i * 131072:(i + 1) * 131072], **kw)
    print(f"step {i}: {1e3 * (time.perf_counter() - t0):.0f} ms, placements {int(r['nAppend'][r['status'] >= -1].sum()):.3e}, failed {(r['status'] < -1).sum()}", flush=True)
