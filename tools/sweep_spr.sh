#!/bin/bash
# usage (GPU box): tools/sweep_spr.sh "samples model ENV=val ENV=val" ...   -- one short bench run per configuration
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out/sweep
for cfg in "$@"; do set -- $cfg; S=$1; M=$2; shift 2
  env "$@" timeout 900 python $R/bench.py --samples $S --model $M --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/sweep/s.json 2> $R/gpurun_out/sweep/s.err
  python -c "
import json;d=json.loads(open('$R/gpurun_out/sweep/s.json').read());k=d['spr_search']['kernel_ms_rank0'];st=d['steps'];print('$cfg', '%.3g'%d['value'],'ms/step %.1f'%d['ms_per_step'],'lane %.1f dense %.1f replay %.1f'%(k['budgeted_lane_searches']/st,k['dense_scoring']/st,k['replay_and_refinement']/st), d['spr_search']['launches_rank0'])"
done
