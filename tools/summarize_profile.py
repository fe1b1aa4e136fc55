#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs merged back under gpurun_out/ into the committed summary under profiles/.

    python tools/summarize_profile.py NAME STATS_DIR FETCH_DIR WRITE_DIR BENCH_JSON "free text"
"""
import csv
import glob
import json
import sys


def one(pattern):
    g = glob.glob(pattern)
    if not g:
        raise SystemExit(f"nothing matches {pattern}")
    return g[0]


def main():
    name, stats_dir, fetch_dir, write_dir, bench_json = sys.argv[1:6]
    note = sys.argv[6] if len(sys.argv) > 6 else ""
    out = [f"# profiles/{name}\n\n"]
    out.append("rocprofv3 passes (run from /tmp with TMPDIR=/tmp on the MI355X box; counters in their own passes):\n\n")
    out.append("    rocprofv3 --kernel-trace --stats -f csv -d <dir> -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline\n")
    out.append("    rocprofv3 --pmc FETCH_SIZE -f csv -d <dir> -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline\n")
    out.append("    rocprofv3 --pmc WRITE_SIZE -f csv -d <dir> -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline\n\n")
    out.append("## kernel_stats.csv (top rows)\n\n| kernel | calls | total ns | avg ns | % | min ns | max ns |\n|---|---|---|---|---|---|---|\n")
    for i, r in enumerate(csv.DictReader(open(one(stats_dir + "/*/*_kernel_stats.csv")))):
        if i >= 8:
            break
        out.append(f"| `{r['Name'][:72]}` | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | "
                   f"{r['Percentage']} | {r['MinNs']} | {r['MaxNs']} |\n")
    agg = {}
    for cname, d in (("FETCH", fetch_dir), ("WRITE", write_dir)):
        for r in csv.DictReader(open(one(d + "/*/*_counter_collection.csv"))):
            k = r["Kernel_Name"][:72]
            a = agg.setdefault(k, {"FETCH": [], "WRITE": [],
                                   "info": (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])})
            a[cname].append(float(r["Counter_Value"]))
    out.append("\n## PMC, average per dispatch\n\n| kernel | dispatches | FETCH_SIZE (KB) | WRITE_SIZE (KB) | VGPR | SGPR | LDS | scratch |\n|---|---|---|---|---|---|---|---|\n")
    for k, a in agg.items():
        if "k_" in k[:12]:
            f = sum(a["FETCH"]) / max(1, len(a["FETCH"]))
            w = sum(a["WRITE"]) / max(1, len(a["WRITE"]))
            out.append(f"| `{k}` | {len(a['FETCH'])} | {f:.1f} | {w:.1f} | {a['info'][0]} | {a['info'][1]} | {a['info'][2]} | {a['info'][3]} |\n")
    b = json.load(open(bench_json))
    out.append("\n## bench.py line of the same build (un-profiled run)\n\n```json\n" + json.dumps(b) + "\n```\n")
    if note:
        out.append("\n" + note + "\n")
    open(f"profiles/{name}.md", "w").write("".join(out))
    print("".join(out))


if __name__ == "__main__":
    main()
