#!/bin/bash
# usage (GPU box): tools/profile_config5.sh -- rocprofv3 kernel stats of BASELINE config 5's loop (tools/online_probe.py: 3 072 samples added
# to the 1 000 000-tip full-model tree, announced 512 at a time, then a round)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r06_config5_prof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -- python $R/tools/online_probe.py 1000000 3072 512 > $OUT/run.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv
python - <<PY > $OUT/summary.md
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
print("# profiles/r06_config5 -- rocprofv3 --kernel-trace --stats of tools/online_probe.py 1000000 3072 512\n")
print("(tree build and the round after the loop included; the loop's own kernels: k_pe_level / k_pe_* = rows by expansion, k_place_minor,\nk_ahead_cols = changed columns for a traversal made ahead, k_append_queries = the same for a search, k_update_items* / k_commit /\nk_pass = maple_update_partials, k_evalplace* / k_shorten* = refinement, k_patch_* = maple_tree_patch)\n")
print("| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|")
for r in rows[:28]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("maple::", "").split("(")[0][:60]
    print(f"| \`{n}\` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e6:.4f} | {float(r['Percentage']):.2f} |")
PY
grep -A9 loop_ms $OUT/run.log >> $OUT/summary.md
rm -rf $OUT/stats
cat $OUT/summary.md
