#!/bin/bash
# usage (GPU box): tools/time_variants2.sh lib1.so lib2.so ...   -- one short default-bench run per build of the library
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  MAPLE_HIP_LIB=$R/$lib python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=d['spr_search']['kernel_ms_rank0']
print('lib [$lib] value %.4g  ms/step %.1f  scoring ms/launch %.1f  frac %.3f  lane %.0f replay %.0f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], k['budgeted_lane_searches']/d['steps'], k['replay_and_refinement']/d['steps']))"
done
