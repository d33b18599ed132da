#!/bin/bash
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the thin / weight-gradient kernels in the real step (one pmc pass)   usage: bash tools/pmc_lds_thin.sh <tag> [lib]
set -u
TAG=${1:-r06c}; LIB=${2:-}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/plt
[ -n "$LIB" ] && export STARCOP_HIP_LIB=$LIB
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/plt -o run -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --overlap 0 > $OUT/pmc_lds.log 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/plt -name "*.db" | head -1) "thin_h|k_wgrad3_bx3" | python -c "
import sys
name=None; d={}
for l in sys.stdin:
    if not l.startswith(' '): name=l.strip()[:90]; d[name]={}
    else:
        p=l.split(); d[name][p[0]]=float(p[1])
for k,v in d.items():
    if 'SQ_LDS_IDX_ACTIVE' in v and v['SQ_LDS_IDX_ACTIVE']>0:
        print(f\"{k.replace('void (anonymous namespace)::','')[:60]:60s} conflict/active {v.get('SQ_LDS_BANK_CONFLICT',0)/v['SQ_LDS_IDX_ACTIVE']:.3f}  active {v['SQ_LDS_IDX_ACTIVE']:.3e}\")
"
