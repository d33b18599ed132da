#!/bin/bash
# experiment builds of conv_bx3.hip: tools/build_exp.sh 1 2 3 -> starcop_amd/libstarcop_hip_exp{1,2,3}.so (use with STARCOP_HIP_LIB=...)
set -e
cd "$(dirname "$0")/../starcop_amd/csrc"
make -s
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -fno-slp-vectorize -DSC_EXP=$n -c conv_bx3.hip -o /tmp/conv_bx3_exp$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC conv_mfma.o /tmp/conv_bx3_exp$n.o conv_valu.o elementwise.o mag1c.o features.o validation.o host_io.o -o ../libstarcop_hip_exp$n.so
done
ls -la ../libstarcop_hip*.so
