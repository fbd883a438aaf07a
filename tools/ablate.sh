#!/bin/bash
# Diagnostics: build variants of libdad3d_hip.so with pieces of the fused decode kernel compiled out
# (-DDAD3D_ABLATE=bits; results are WRONG, only the timing is meaningful) into tools/ablate/lib_<bits>.so.
# Bits read by flame_decode.hip today: 64 extra phase stamps + a second, instruction-cache-warm pass of the pose role;
# 128 no pose role and no hand-off; 256 the pose role runs but nobody polls or fetches; 512 the pose role only arrives
# (no compute, no stores) and the decode role polls and fetches as usual (what each half of the hand-off costs:
# profiles/r03_kernel_log.md section 3).
# Use with:  DAD3D_LIB_PATH=$PWD/tools/ablate/lib_<bits>.so python tools/trace_decode.py 64
set -e
cd "$(dirname "$0")/../dad-3dheads_amd/csrc"
make -s
mkdir -p ../../tools/ablate
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDAD3D_ABLATE=$n -c flame_decode.hip -o /tmp/fd_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/lib_$n.so /tmp/fd_$n.o sim3dr_kernels.o capi.o sim3dr_compat.o
done
