#!/bin/bash
# usage (GPU box): tools/timeline_10k.sh -- kernel timeline of one round of BASELINE config 2 (10 000 samples, UNREST)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r06_10k_trace; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -- python $R/bench.py --samples 10000 --model unrest --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
python $R/tools/timeline_step.py $OUT/stats/*/*_kernel_trace.csv 4 2 | sed 's/busy  *\([0-9.]*\) ms/busy \1 ms/' > $OUT/timeline.txt 2>&1
python - <<PY >> $OUT/timeline.txt
import csv,glob,collections
rows=list(csv.DictReader(open(glob.glob("$OUT/stats/*/*_kernel_trace.csv")[0])))
PY
cat $OUT/timeline.txt
rm -rf $OUT/stats
