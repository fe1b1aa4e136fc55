#!/usr/bin/env python3
"""rocprofv3 CSVs of tools/profile_round6.sh -> one markdown summary + the PMC json bench.py reads for `traffic`.

    python tools/summarize_round6.py NAME OUT_DIR "extra bench args"
"""
import collections
import csv
import glob
import json
import os
import sys

name, out = sys.argv[1], sys.argv[2]
extra = sys.argv[3] if len(sys.argv) > 3 else ""
commit = sys.argv[4] if len(sys.argv) > 4 else None
KERNELS = ("k_fr_pass", "k_fr_wide_frames", "k_fr_cached", "k_fr_replay_wide", "k_fr_layout_sizes", "k_fr_layout_place", "k_fr_sort_level", "k_wit_score", "k_wit_pairs", "k_wit_hist", "k_fr_updating_wave_s", "k_fr_updating_wave", "k_fr_updating", "k_fr_replay", "k_fr_refine", "k_fr_begin", "k_append_queries_lds",
           "k_append_queries", "k_spr_search_assisted", "k_spr_search", "k_finite_prefix")


def one(pattern):
    g = glob.glob(pattern)
    return g[0] if g else None


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").replace("maple::", "")
    return k.split("(")[0][:60]


def base(k):
    k = short(k).split("<")[0]
    return k


lines = [f"# profiles/{name}\n\n"]
lines.append("Commands (MI355X box, from /tmp with TMPDIR=/tmp; counters in their own passes, never with a trace):\n\n")
lines.append(f"    python bench.py {extra}                                              # the bench line below\n")
lines.append(f"    rocprofv3 --kernel-trace --stats -f csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-1m {extra}\n")
lines.append("    rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_* | SQ_LDS_* GRBM_GUI_ACTIVE -f csv -- (the same command)\n\n")
try:
    line = [ln for ln in open(os.path.join(out, "bench.json")) if ln.startswith("{")][-1]
    bench = json.load(open(os.path.join(out, "bench_detail.json")))          # (everything the run measured; bench.json is the printed line)
except Exception as e:                                                     # noqa: BLE001
    bench = None
    lines.append(f"(no bench line: {e})\n")
st = one(out + "/stats/*/*_kernel_stats.csv")
if st:
    lines.append("## kernel_stats.csv of the traced run (1 warm-up + 2 timed steps + 1 step after a re-upload of the tree, plus tree build and branch-length passes)\n\n"
                 "| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|\n")
    for i, r in enumerate(csv.DictReader(open(st))):
        if i < 16:
            lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | "
                         f"{float(r['AverageNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |\n")
# per kernel and STEP: summed over the launches of the two timed steps (the traced command runs 1 warm-up + 2 timed steps: / 3)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(out + "/*/*/*_counter_collection.csv"):
    if "/calib/" in f or "/calibw/" in f:
        continue
    for r in csv.DictReader(open(f)):
        k = base(r["Kernel_Name"])
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
STEPS = 4.0   # (1 warm-up + 2 timed steps + the step after the re-upload)


def tot(d, c):
    v = d.get(c)
    return sum(v) / STEPS if v else float("nan")


lines.append("\n## PMC, summed over a kernel's launches, per STEP of the bench (a step is one deep SPR round)\n\n")
lines.append("| kernel | launches / step | FETCH_SIZE GB (calibrated) | WRITE_SIZE GB | VALU / SALU / VMEM_RD / LDS / BRANCH wave-insts (M) | "
             "WAVE_CYCLES / WAIT_ANY / WAIT_INST_ANY / ACTIVE_VALU / ACTIVE_LDS (M quad-cycles) | LDS_IDX_ACTIVE / BANK_CONFLICT (M) | VGPR SGPR LDS scratch |\n"
             "|---|---|---|---|---|---|---|---|\n")
traffic = {}
issue = {}
for k in KERNELS:
    d = agg.get(k)
    if not d:
        continue
    n = max(len(v) for v in d.values()) / STEPS
    fetch_b = tot(d, "FETCH_SIZE") * 1024 / (calib or 1.0)
    write_b = tot(d, "WRITE_SIZE") * 1024 / (wcal.get("stream") or 1.0)
    traffic[k] = {"fetch_bytes_per_step": fetch_b, "write_bytes_per_step": write_b, "launches_per_step": n}
    insts = sum(tot(d, c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM")
                if tot(d, c) == tot(d, c))
    issue[k] = {"wave_insts_per_step": insts, "wave_quad_cycles_per_step": tot(d, "SQ_WAVE_CYCLES"), "wait_any": tot(d, "SQ_WAIT_ANY"),
                "wait_inst_any": tot(d, "SQ_WAIT_INST_ANY"), "active_valu": tot(d, "SQ_ACTIVE_INST_VALU"), "active_lds": tot(d, "SQ_ACTIVE_INST_LDS"),
                "lds_idx_active": tot(d, "SQ_LDS_IDX_ACTIVE"), "lds_bank_conflict": tot(d, "SQ_LDS_BANK_CONFLICT"),
                "busy_cycles": tot(d, "SQ_BUSY_CYCLES")}
    lines.append(f"| `{k}` | {n:.1f} | {fetch_b/1e9:.2f} | {write_b/1e9:.2f} | "
                 f"{tot(d,'SQ_INSTS_VALU')/1e6:.0f} / {tot(d,'SQ_INSTS_SALU')/1e6:.0f} / {tot(d,'SQ_INSTS_VMEM_RD')/1e6:.0f} / "
                 f"{tot(d,'SQ_INSTS_LDS')/1e6:.0f} / {tot(d,'SQ_INSTS_BRANCH')/1e6:.0f} | "
                 f"{tot(d,'SQ_WAVE_CYCLES')/1e6:.0f} / {tot(d,'SQ_WAIT_ANY')/1e6:.0f} / {tot(d,'SQ_WAIT_INST_ANY')/1e6:.0f} / {tot(d,'SQ_ACTIVE_INST_VALU')/1e6:.0f} / {tot(d,'SQ_ACTIVE_INST_LDS')/1e6:.0f} | "
                 f"{tot(d,'SQ_LDS_IDX_ACTIVE')/1e6:.0f} / {tot(d,'SQ_LDS_BANK_CONFLICT')/1e6:.0f} | {' '.join(str(x) for x in meta[k])} |\n")
if wcal:
    lines.append(f"\nWRITE_SIZE calibration (k_calib_write over a 2 GiB buffer): a coalesced 8-byte stream is reported as "
                 f"{wcal['stream']:.4f} of the bytes written; one 8-byte store into every 64-byte line as {wcal['per_line_store']:.2f} x the useful bytes.\n")
if calib:
    lines.append(f"\nFETCH_SIZE calibration for this library's access pattern (k_calib_walk: 2 GiB read by dependent 8-byte "
                 f"per-lane walks): the counter reports {calib:.4f} of the bytes read.\n")
if bench:
    lines.append("\n## against the bench line's own numbers (HIP events, un-profiled run)\n\n"
                 "| kernel | ms / step | algorithmic GB / step (SURVEY 8d) | HBM traffic GB / step (PMC) | traffic / algorithmic | "
                 "issue: wave-insts / (CUs x 4 SIMDs x cycles at 2.4 GHz) | waves waiting |\n|---|---|---|---|---|---|---|\n")
    for r in bench.get("roofline_by_kernel", []):
        k = r["kernel"].split()[0]
        t = traffic.get(k)
        if not t:
            continue
        ms = r["kernel_ms_per_step"]
        alg = r["algorithmic_bytes_per_launch"] * r["launches_timed"] / bench["steps"]
        tr = t["fetch_bytes_per_step"] + t["write_bytes_per_step"]
        iss = issue[k]
        # one VALU wave-instruction occupies a SIMD for 4 cycles (wave64 on 16 lanes): peak = 256 CUs x 4 SIMDs / 4 per cycle
        util = iss["wave_insts_per_step"] * 4.0 / (256 * 4 * 2.4e9 * ms * 1e-3) if ms else float("nan")
        lines.append(f"| `{k}` | {ms:.1f} | {alg/1e9:.2f} | {tr/1e9:.2f} | {tr/alg if alg else float('nan'):.2f} | {util:.3f} | "
                     f"{iss['wait_any']/iss['wave_quad_cycles_per_step'] if iss['wave_quad_cycles_per_step'] else float('nan'):.2f} |\n")
    w = bench["config"]
    pj = {"traffic_bytes_per_step": {k: v["fetch_bytes_per_step"] + v["write_bytes_per_step"] for k, v in traffic.items()},
          "launches_per_step": {k: v["launches_per_step"] for k, v in traffic.items()},
          "issue": issue,
          "calibration": {"fetch_counted_over_read": calib, "write": wcal},
          "workload": {"samples": w["samples"], "model": w["model"], "batch": w["searches_per_step"], "n_gpus": bench["n_gpus"],
                       "tree": w.get("tree"), "refs": (w.get("local_references") or {}).get("form", "none"),
                       "synth": (w.get("setup_breakdown_s") or {}).get("synth", "v1")},
          "library_commit": commit,
          "source": f"profiles/{name}.md"}
    json.dump(pj, open(os.path.join(out, "pmc_spr.json"), "w"), indent=1)
    lines.append("\n## bench.py line of the same build (un-profiled run; its bench_detail.json is next to this file)\n\n```json\n" + line.strip() + "\n```\n")
sys.stdout.write("".join(lines))
