#!/usr/bin/env python3
"""rocprofv3 CSVs of tools/profile_round2.sh -> one markdown summary + the PMC json bench.py reads for roofline.traffic.

    python tools/summarize_round2.py NAME OUT_DIR "extra bench args"
"""
import collections
import csv
import glob
import json
import os
import sys

name, out = sys.argv[1], sys.argv[2]
extra = sys.argv[3] if len(sys.argv) > 3 else ""


def one(pattern):
    g = glob.glob(pattern)
    return g[0] if g else None


def short(k):
    return k.split("(")[0].replace("void ", "")[:60]


lines = [f"# profiles/{name}\n\n"]
lines.append("Commands (MI355X box, from /tmp with TMPDIR=/tmp; counters in their own passes, never with a trace):\n\n")
lines.append(f"    python bench.py {extra}                                              # the bench line below\n")
lines.append(f"    rocprofv3 --kernel-trace --stats -f csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras {extra}\n")
lines.append("    rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_* | TCC_* -f csv -- (the same command)\n\n")
try:
    bench = json.loads([ln for ln in open(os.path.join(out, "bench.json")) if ln.startswith("{")][-1])
except Exception as e:                                                     # noqa: BLE001
    bench = None
    lines.append(f"(no bench line: {e})\n")
st = one(out + "/stats/*/*_kernel_stats.csv")
kstats = {}
if st:
    lines.append("## kernel_stats.csv of the traced run (1 warm-up + 2 timed steps, plus tree build)\n\n"
                 "| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|\n")
    for i, r in enumerate(csv.DictReader(open(st))):
        kstats[short(r["Name"])] = r
        if i < 12:
            lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | "
                         f"{float(r['AverageNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |\n")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(out + "/*/*/*_counter_collection.csv"):
    if "/calib/" in f or "/calibw/" in f:
        continue
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"))
calib = None
f = one(out + "/calib/*/*_counter_collection.csv")
if f:
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_calib_walk" in r["Kernel_Name"]]
    if v:
        calib = (sum(v) / len(v)) * 1024 / (2 << 30)
wcal = {}
f = one(out + "/calibw/*/*_counter_collection.csv")
if f:
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_calib_write" in r["Kernel_Name"]]
    if len(v) >= 2:
        wcal = {"stream": v[0] * 1024 / (2 << 30), "per_line_store": v[1] * 1024 / ((2 << 30) / 8)}
lines.append("\n## PMC, average per dispatch (kernels of the search)\n\n")
lines.append("| kernel | dispatches | FETCH_SIZE MB | WRITE_SIZE MB | VALU / SALU / VMEM_RD / LDS / BRANCH wave-insts (M) | "
             "WAVE_CYCLES / WAIT_ANY / ACTIVE_VALU (M) | TCC hit / miss (M) | VGPR SGPR LDS scratch |\n|---|---|---|---|---|---|---|---|\n")


def avg(d, c):
    v = d.get(c)
    return sum(v) / len(v) if v else float("nan")


for k, d in agg.items():
    if not any(t in k for t in ("k_append", "k_spr", "k_place", "k_merge")):
        continue
    n = max(len(v) for v in d.values())
    lines.append(f"| `{k}` | {n} | {avg(d,'FETCH_SIZE')/1024:.1f} | {avg(d,'WRITE_SIZE')/1024:.1f} | "
                 f"{avg(d,'SQ_INSTS_VALU')/1e6:.1f} / {avg(d,'SQ_INSTS_SALU')/1e6:.1f} / {avg(d,'SQ_INSTS_VMEM_RD')/1e6:.1f} / "
                 f"{avg(d,'SQ_INSTS_LDS')/1e6:.1f} / {avg(d,'SQ_INSTS_BRANCH')/1e6:.1f} | "
                 f"{avg(d,'SQ_WAVE_CYCLES')/1e6:.0f} / {avg(d,'SQ_WAIT_ANY')/1e6:.0f} / {avg(d,'SQ_ACTIVE_INST_VALU')/1e6:.0f} | "
                 f"{avg(d,'TCC_HIT_sum')/1e6:.1f} / {avg(d,'TCC_MISS_sum')/1e6:.1f} | {' '.join(str(x) for x in meta[k])} |\n")
if wcal:
    lines.append(f"\nWRITE_SIZE calibration (k_calib_write over a 2 GiB buffer): a coalesced 8-byte stream is reported as "
                 f"{wcal['stream']:.4f} of the bytes written; one 8-byte store into every 64-byte line (the score-matrix pattern) as "
                 f"{wcal['per_line_store']:.2f} x the useful bytes.\n")
if calib:
    lines.append(f"\nFETCH_SIZE calibration for this library's access pattern (k_calib_walk: 2 GiB read by dependent 8-byte "
                 f"per-lane walks): the counter reports {calib:.4f} of the bytes read.\n")
# per kernel: HBM-side traffic per launch (calibrated), and for the scoring kernel the per-pair instruction counts
traffic = {}
for key in ("k_append_queries_lds", "k_append_queries", "k_spr_search"):
    # (k_spr_search: the plain kernel -- replay launches -- and k_spr_search_assisted -- the lane searches -- pooled: bench.py's
    # roofline block averages over both kinds of launch)
    ks = [x for x in agg if x.startswith(key + "<") or x == key or (key == "k_spr_search" and x.startswith("k_spr_search_assisted"))]
    if not ks:
        continue
    d = collections.defaultdict(list)
    for k in ks:
        for cn, vals in agg[k].items():
            d[cn].extend(vals)
    fetch_b = avg(d, "FETCH_SIZE") * 1024 / (calib or 1.0)
    write_b = avg(d, "WRITE_SIZE") * 1024 / (wcal.get("stream") or 1.0)      # HBM-side bytes (partial-line stores included)
    traffic[key] = fetch_b + write_b
    lines.append(f"\n* `{key}`: HBM traffic per launch FETCH {fetch_b/1e9:.2f} GB (calibrated) + WRITE {write_b/1e9:.2f} GB = "
                 f"{traffic[key]/1e9:.2f} GB\n")
if bench:
    roofs = [bench.get("roofline"), bench.get("roofline_second_kernel")]
    sc = next((r for r in roofs if r and r["kernel"].startswith("k_append")), None)
    sk = next((k for k in agg if "k_append_queries" in k), None)
    if sc and sk:
        d = agg[sk]
        pairs, alg = sc["units_per_launch"], sc["algorithmic_bytes_per_launch"]
        t = traffic.get("k_append_queries_lds") or traffic.get("k_append_queries") or float("nan")
        lines.append(f"\n## the dense scoring kernel inside the search, per (query, branch) pair\n\n"
                     f"* pairs per launch (bench run) {pairs:.4g}, algorithmic bytes per launch {alg:.4g} ({alg/pairs:.0f} B / pair); "
                     f"HBM traffic per launch {t/1e9:.2f} GB = {t/alg:.3f} of the algorithmic bytes\n"
                     f"* wave-instructions per 64 pairs: VALU {avg(d,'SQ_INSTS_VALU')/pairs*64:.0f}, SALU {avg(d,'SQ_INSTS_SALU')/pairs*64:.0f}, "
                     f"VMEM_RD {avg(d,'SQ_INSTS_VMEM_RD')/pairs*64:.0f}, LDS {avg(d,'SQ_INSTS_LDS')/pairs*64:.0f}, "
                     f"BRANCH {avg(d,'SQ_INSTS_BRANCH')/pairs*64:.0f}\n"
                     f"* wave cycles: waiting {avg(d,'SQ_WAIT_ANY')/avg(d,'SQ_WAVE_CYCLES'):.2f} of a wave's life, issue-stalled "
                     f"{avg(d,'SQ_WAIT_INST_ANY')/avg(d,'SQ_WAVE_CYCLES'):.2f}, VALU active {avg(d,'SQ_ACTIVE_INST_VALU')/avg(d,'SQ_WAVE_CYCLES'):.2f}\n")
    w = bench["config"]
    pj = {"traffic_bytes_per_launch": traffic, "calibration": {"fetch_counted_over_read": calib, "write": wcal},
          "workload": {"samples": w["samples"], "model": w["model"], "batch": w["searches_per_step"], "n_gpus": bench["n_gpus"]},
          "source": f"profiles/{name}.md"}
    json.dump(pj, open(os.path.join(out, "pmc_spr.json"), "w"), indent=1)
if bench:
    lines.append("\n## bench.py line of the same build (un-profiled run)\n\n```json\n" + json.dumps(bench) + "\n```\n")
sys.stdout.write("".join(lines))
