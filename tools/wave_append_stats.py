#!/usr/bin/env python3
"""Latency of ONE appendProbNode: one lane's walk against the wavefront-wide walk (wave_dev.h) (GPU box).

    python tools/wave_append_stats.py [tips]

A single pair per launch measures latency (what a search that needs this one score waits for); many pairs per launch the
throughput of either form.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from maple_amd.host import reference_tables, tip_genome_list  # noqa: E402
from maple_amd.runtime import Device  # noqa: E402
from maple_amd.synth import make_dataset  # noqa: E402
from maple_amd.tree_mirror import TreeMirror  # noqa: E402


def main():
    n_tips = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    model = sys.argv[2] if len(sys.argv) > 2 else "ratevar"
    data = make_dataset(n_samples=n_tips, l_ref=29903, seed=1, mean_diffs=30.0, rate_variation=(model != "unrest"))
    ref_idx, rf = reference_tables(data.ref)
    dev = Device(ref_idx, rf, arena_bytes=4 << 30, debug=True)
    dev.set_model(**bench.model_kwargs(model, len(ref_idx)))
    tips = {int(v): tip_genome_list(dl, ref_idx) for v, dl in zip(data.tip_node, data.diffs)}
    m = TreeMirror(dev, data.parent, data.blen, tips).build()
    rng = np.random.default_rng(3)
    cand = np.nonzero(m.tot_up >= 0)[0]
    tip_nodes = np.nonzero(m.is_tip.astype(bool))[0]
    for n in (1, 64, 4096, 262144):
        pl = m.tot_up[rng.choice(cand, size=n)]
        cl = m.lower[rng.choice(tip_nodes, size=n)]
        bl = 1.0 / dev.lRef
        a = dev.append_batch(pl, cl, True, bl)
        w, _ = dev.debug_wave_append_batch(pl, cl, True, bl)
        assert np.array_equal(a, w)
        t_one, t_wave = [], []
        for _ in range(20):
            dev.timing_reset()
            t0 = time.perf_counter()
            dev.append_batch(pl, cl, True, bl)
            t_one.append(time.perf_counter() - t0)
            _, ms = dev.debug_wave_append_batch(pl, cl, True, bl)
            t_wave.append(ms)
        ne, _ = dev.sizes(pl)
        print(f"{n} pairs (parent lists {np.median(ne):.0f} entries): one-lane call {1e3 * np.median(t_one):.3f} ms wall; "
              f"wavefront-wide kernel {np.median(t_wave):.4f} ms ({1e3 * np.median(t_wave) / n:.2f} us per pair)")


if __name__ == "__main__":
    main()
