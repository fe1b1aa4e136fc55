#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/spr_single; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM -f csv -d $OUT/pmc -- python $R/tools/spr_stats.py 10000 > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, collections
rows=collections.defaultdict(dict)
for f in glob.glob("$OUT/pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_spr_search" in r["Kernel_Name"]:
            rows[(r["Dispatch_Id"], r["Grid_Size"])][r["Counter_Name"]]=float(r["Counter_Value"])
for k in sorted(rows, key=lambda x:int(x[0])):
    print(k, {a:int(b) for a,b in rows[k].items()})
PY
tail -4 $OUT/log.txt
