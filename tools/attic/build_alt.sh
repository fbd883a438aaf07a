#!/bin/bash
# Builds an ALTERNATE libdad3d_hip from one of the decode-kernel variants kept here, next to (never instead of) the
# product library:   tools/attic/build_alt.sh tools/attic/flame_decode_feeder_share.hip.txt [out.so]
# Use it through DAD3D_LIB_PATH=<out.so> (dad_3dheads_amd/_lib.py), always under a short `timeout`:
#   DAD3D_LIB_PATH=$PWD/tools/attic/alt.so timeout 25 python tools/attic/probe_mw8.py   (edit its LIB_PATH line away)
set -e
root="$(cd "$(dirname "$0")/../.." && pwd)"; S="$root/dad-3dheads_amd/csrc"
src="$1"; out="${2:-$root/tools/attic/alt.so}"
H=/opt/rocm/bin/hipcc; C="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I$S -I$root/include"
tmp="$(mktemp -d)"; cp "$src" "$S/_alt_variant.hip"; trap 'rm -f "$S/_alt_variant.hip"; rm -rf "$tmp"' EXIT
(cd "$S" && make -s)   # the other objects are the product's
$H $C -c "$S/_alt_variant.hip" -o "$tmp/fd.o"
$H $C -DDAD3D_DIAG_SPIN_ENV -x hip -c "$S/capi.cpp" -o "$tmp/capi.o"   # honours DAD3D_SPIN_LIMIT
$H --offload-arch=gfx950 -shared -fPIC -o "$out" "$tmp/fd.o" "$tmp/capi.o" "$S/flame_backward.o" "$S/sim3dr_kernels.o" "$S/projection.o" "$S/preprocess.o" "$S/sim3dr_compat.o"
echo "built $out"
