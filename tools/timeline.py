#!/usr/bin/env python3
"""Kernel timeline of the LAST search round in a rocprofv3 --kernel-trace CSV: timeline.py <kernel_trace.csv>
Prints the span of the round, how long the GPU ran no kernel at all inside it, and per kernel: launches, busy time, first start and
last end relative to the round's start."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def nm(r):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("maple::", "")
    return k.split("(")[0].split("<")[0]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r)) for r in rows), key=lambda t: t[0])
begins = [i for i, e in enumerate(ev) if e[2] == "k_fr_begin"]
i0 = begins[-1]
# the round starts a little before k_fr_begin (prologue kernels): take everything after the previous round's last kernel
prev_end = max((e[1] for e in ev[:i0] if e[2] in ("k_fr_finish", "k_spr_search")), default=ev[0][0])
first = next(i for i, e in enumerate(ev) if e[0] >= prev_end)
rnd = ev[first:]
t0 = rnd[0][0]
span = max(e[1] for e in rnd) - t0
busy, cur_s, cur_e = 0, None, None
for s, e, _ in rnd:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"round: span {span/1e6:.1f} ms, some kernel running {busy/1e6:.1f} ms, idle {(span-busy)/1e6:.1f} ms, {len(rnd)} launches")
agg = collections.OrderedDict()
for s, e, n in rnd:
    a = agg.setdefault(n, [0, 0, s, e])
    a[0] += 1; a[1] += e - s; a[3] = max(a[3], e)
for n, (c, tot, s, e) in agg.items():
    print(f"  {n:32s} x{c:4d}  busy {tot/1e6:8.2f} ms   first start {(s-t0)/1e6:8.2f}   last end {(e-t0)/1e6:8.2f}")
# the idle gaps longer than 1 ms
cur_e = None
for s, e, n in rnd:
    if cur_e is not None and s - cur_e > 1e6: print(f"  gap {(s-cur_e)/1e6:.2f} ms before {n} at {(s-t0)/1e6:.2f}")
    cur_e = e if cur_e is None else max(cur_e, e)
# optional: the kernels of ONE level of the expansion (between the level's k_fr_snap and the next), with their start / end in ms
if len(sys.argv) > 2:
    snaps = [i for i, e in enumerate(rnd) if e[2] == "k_fr_snap"]
    for lvl in (int(x) for x in sys.argv[2:]):
        if lvl + 1 >= len(snaps):
            continue
        a, b = snaps[lvl], snaps[lvl + 1]
        tl = rnd[a][0]
        print(f"level {lvl}:")
        for s, e, n in rnd[a:b + 1]:
            print(f"    {n:28s} start {(s - tl)/1e6:8.3f}  end {(e - tl)/1e6:8.3f}  ({(e - s)/1e6:.3f} ms)")
