#!/bin/bash
ATTR=$1; shift
for rep in 1 2; do for v in "$@"; do
  python -c "
import sys, runpy
import starcop_amd.network as n
setattr(n.HyperStarcopUNet, '$ATTR', $v)
sys.argv = ['bench.py', '--steps', '40', '--warmup', '8', '--no-cpu-baseline', '--no-extras']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$ATTR=$v', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"
done; done
