#!/usr/bin/env python3
"""Kernel timeline of one step (default: the last) of a bench run traced with rocprofv3 --kernel-trace:
timeline_step.py <kernel_trace.csv> <steps in the trace> [which step, counted from the end: 1 = last (default), 2 = the one before ...]
(warm-up + timed steps; a step may hold several frontier passes: the k_fr_begin launches are split evenly over the steps).  Prints the
step's span, how long no kernel ran, per kernel launches / busy time / first start / last end, and the idle gaps over 2 ms."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
def nm(r):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("maple::", "")
    return k.split("(")[0].split("<")[0]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r)) for r in rows), key=lambda t: t[0])
begins = [i for i, e in enumerate(ev) if e[2] == "k_fr_begin"]
per = len(begins) // steps
which = int(sys.argv[3]) if len(sys.argv) > 3 else 1
i0 = begins[-per * which]
prev_end = max(e[1] for e in ev[:i0] if e[2] in ("k_fr_finish", "k_spr_search", "k_spr_search_assisted"))
first = next(i for i, e in enumerate(ev) if e[0] >= prev_end)
last = len(ev)
if which > 1:
    i1 = begins[-per * (which - 1)]
    nxt_prev_end = max(e[1] for e in ev[:i1] if e[2] in ("k_fr_finish", "k_spr_search", "k_spr_search_assisted"))
    last = next(i for i, e in enumerate(ev) if e[0] >= nxt_prev_end)
rnd = ev[first:last]
t0 = rnd[0][0]
span = max(e[1] for e in rnd) - t0
busy, cur_s, cur_e = 0, None, None
for s, e, _ in rnd:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"step: span {span/1e6:.1f} ms, some kernel running {busy/1e6:.1f} ms, idle {(span-busy)/1e6:.1f} ms, {len(rnd)} launches, {per} frontier passes")
agg = collections.OrderedDict()
for s, e, n in rnd:
    a = agg.setdefault(n, [0, 0, s, e])
    a[0] += 1; a[1] += e - s; a[3] = max(a[3], e)
for n, (c, tot, s, e) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if tot > 0.5e6:
        print(f"  {n:36s} x{c:5d}  busy {tot/1e6:8.2f} ms   first start {(s-t0)/1e6:8.2f}   last end {(e-t0)/1e6:8.2f}")
cur_e = None
for s, e, n in rnd:
    if cur_e is not None and s - cur_e > 2e6: print(f"  gap {(s-cur_e)/1e6:.2f} ms before {n} at {(s-t0)/1e6:.2f}")
    cur_e = e if cur_e is None else max(cur_e, e)
for i in (begins[-per * which:] if which == 1 else begins[-per * which:-per * (which - 1)]):
    print(f"  frontier pass starts at {(ev[i][0]-t0)/1e6:.2f} ms")
