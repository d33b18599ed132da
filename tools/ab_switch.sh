#!/bin/bash
# same-box A/B of a class switch of network.HyperStarcopUNet: bash tools/ab_switch.sh <attribute> [bench args]
# prints tiles/s, ms/step and the families of the serial pass that moved, switch on / off / on / off
ATTR=$1; shift
for f in True False True False; do
  python -c "
import sys, runpy
import starcop_amd.network as n
setattr(n.HyperStarcopUNet, '$ATTR', $f)
sys.argv = ['bench.py', '--steps', '40', '--warmup', '8', '--no-cpu-baseline', '--no-extras'] + '$*'.split()
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); f = d['roofline']['families_ms_per_step']
print('$ATTR=$f', d['value'], d['ms_per_step'], {k.split(' ')[0]: v for k, v in f.items()})"
done
