#!/bin/bash
# usage (GPU box): tools/pmc_pairs.sh NAME  -- PMC passes over tools/pair_bench.py (k_append over unrelated pairs, the dense kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}; NAME=$1; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python $R/tools/pair_bench.py"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH -f csv -d $OUT/sq1 -- $B > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -f csv -d $OUT/sq2 -- $B > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE -f csv -d $OUT/lds -- $B > $OUT/lds.log 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_append" in k:
            agg[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    for k, d in agg.items():
        o.write(k + "\n")
        for cn in sorted(d):
            v = d[cn]
            o.write(f"   {cn:24s} {sum(v)/len(v):18.1f}  (n={len(v)})\n")
print(open("$OUT/summary.txt").read())
PY
rm -rf $OUT/sq1 $OUT/sq2 $OUT/lds
