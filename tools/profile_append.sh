#!/bin/bash
# usage (on the GPU box): tools/profile_append.sh NAME "tuning string" -- PMC passes of the headline kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; NAME=$1; TUNE=$2
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-spr $TUNE"
python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-spr $TUNE > $OUT/bench.json 2>/dev/null
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_BRANCH -f csv -d $OUT/pmc1 -- $B > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -f csv -d $OUT/pmc2 -- $B > $OUT/pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -f csv -d $OUT/pmc3 -- $B > $OUT/pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc4 -- $B > $OUT/pmc4.log 2>&1
python $R/tools/pmc_summary.py $OUT
