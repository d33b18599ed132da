#!/bin/bash
# Elimination-experiment builds of conv_bx3.hip (the same source with ONE ingredient removed; results are
# wrong by construction, timings valid).  The experiment blocks are NOT in the shipped source: they live in
# tools/experiments/conv_bx3_experiments.patch, applied here to a temporary copy.
#   tools/build_exp.sh 1 2 3 -> starcop_amd/libstarcop_hip_exp{1,2,3}.so   (use with STARCOP_HIP_LIB=...)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/starcop_amd/csrc"
make -s
cp conv_bx3.hip /tmp/conv_bx3_exp.hip
patch -s /tmp/conv_bx3_exp.hip "$ROOT/tools/experiments/conv_bx3_experiments.patch"
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -fno-slp-vectorize -DSC_EXP=$n -c /tmp/conv_bx3_exp.hip -o /tmp/conv_bx3_exp$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC conv_mfma.o /tmp/conv_bx3_exp$n.o conv_pw3.o conv_valu.o elementwise.o mag1c.o features.o validation.o host_io.o -o ../libstarcop_hip_exp$n.so
done
ls -la ../libstarcop_hip*.so
