#!/bin/bash
# usage (GPU box): tools/profile_round2.sh NAME ["extra bench.py args"]
# The default bench.py command under rocprofv3: kernel trace + stats, then the PMC passes (each in its own run, never
# together with a trace), then the FETCH_SIZE calibration.  tools/summarize_round2.py turns the CSVs into profiles/*.
R=${GRAFT_REPO_ROOT:-/root/repo}; NAME=$1; EXTRA=$2
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $EXTRA"
python $R/bench.py $EXTRA > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/write -- $B > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH -f csv -d $OUT/sq1 -- $B > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -f csv -d $OUT/sq2 -- $B > $OUT/sq2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -f csv -d $OUT/tcc -- $B > $OUT/tcc.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/calib -- python $R/tools/calib_fetch.py > $OUT/calib.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/calibw -- python $R/tools/calib_fetch.py > $OUT/calibw.log 2>&1
python $R/tools/summarize_round2.py $NAME $OUT "$EXTRA" > $OUT/summary.md 2> $OUT/summary.err
cp $OUT/summary.md $OUT/../${NAME}_summary.md 2>/dev/null
tail -40 $OUT/summary.md
