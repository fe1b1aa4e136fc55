#!/bin/bash
# usage (GPU box): tools/profile_round6.sh NAME ["extra bench.py args"]
# The default bench.py command under rocprofv3: kernel trace + stats, then the PMC passes (each in its own run, never
# together with a trace), then the FETCH_SIZE / WRITE_SIZE calibrations.  tools/summarize_round6.py turns the CSVs into
# profiles/<NAME>.md, profiles/<NAME>_kernel_stats.csv and the PMC json bench.py reads for `traffic`.
R=${GRAFT_REPO_ROOT:-/root/repo}; NAME=$1; EXTRA=$2; COMMIT=$3
# (the traced and counted runs leave out the 1 000 000-sample leg of the default line; the un-profiled bench line has it)
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-1m $EXTRA"
python $R/bench.py $EXTRA > $OUT/bench.json 2> $OUT/bench.err; cp $R/bench_detail.json $OUT/bench_detail.json
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/write -- $B > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH -f csv -d $OUT/sq1 -- $B > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -f csv -d $OUT/sq2 -- $B > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -f csv -d $OUT/lds -- $B > $OUT/lds.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/calib -- python $R/tools/calib_fetch.py > $OUT/calib.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/calibw -- python $R/tools/calib_fetch.py > $OUT/calibw.log 2>&1
python $R/tools/summarize_round6.py $NAME $OUT "$EXTRA" "$COMMIT" > $OUT/summary.md 2> $OUT/summary.err
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
# (the traced command runs 1 warm-up + 2 timed steps + the step after the re-upload: the timeline of the second timed step)
python $R/tools/timeline_step.py $OUT/stats/*/*_kernel_trace.csv 4 2 > $OUT/timeline_step.txt 2>&1
# (the raw per-dispatch CSVs are hundreds of MB: only the summaries travel back)
du -sh $OUT/* 2>/dev/null | sort -h | tail -12 > $OUT/sizes.txt
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2 $OUT/lds $OUT/calib $OUT/calibw
tail -40 $OUT/summary.md; tail -3 $OUT/summary.err
