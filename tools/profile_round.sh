#!/bin/bash
# usage (GPU box): tools/profile_round.sh NAME ["extra bench.py args"]  -- bench line + rocprofv3 stats + FETCH/WRITE passes + FETCH calibration
R=${GRAFT_REPO_ROOT:-/root/repo}; NAME=$1; EXTRA=$2
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
python $R/bench.py $EXTRA > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline $EXTRA > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/fetch -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/write -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/write.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/calib -- python $R/tools/calib_fetch.py > $OUT/calib.log 2>&1
cat $OUT/calib.log | tail -2
python - <<PY
import csv, glob
for f in glob.glob("$OUT/calib/*/*_counter_collection.csv"):
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_calib_walk" in r["Kernel_Name"]]
    print("k_calib_walk FETCH_SIZE KB per launch:", v, "-> bytes counted / bytes read =", [x*1024/(2<<30) for x in v])
PY
cat $OUT/bench.json
