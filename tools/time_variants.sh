#!/bin/bash
# usage (GPU box): tools/time_variants.sh "args..." "args..."   -- kernel ms of the headline kernel per bench argument string
R=${GRAFT_REPO_ROOT:-/root/repo}
for t in "$@"; do
  python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-spr $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('args [$t] kernel_ms %.4f  value %.4g  frac %.4f' % (d['roofline']['kernel_ms'], d['value'], d['roofline']['frac']))"
done
