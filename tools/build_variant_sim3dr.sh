#!/bin/bash
# Variant of libdad3d_hip.so with sim3dr_kernels.hip rebuilt with extra flags / another source:
#   tools/build_variant_sim3dr.sh <name> "<flags>" [source]
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"; S="$root/dad-3dheads_amd/csrc"
name="$1"; flags="$2"; src="${3:-$S/sim3dr_kernels.hip}"
H=/opt/rocm/bin/hipcc; C="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -I$S -I$root/include"
mkdir -p "$root/tools/_variants"; tmp="$(mktemp -d)"; trap 'rm -rf "$tmp"' EXIT
(cd "$S" && make -s)
$H $C $flags -x hip -c "$src" -o "$tmp/s3.o"
$H --offload-arch=gfx950 -shared -fPIC -o "$root/tools/_variants/lib_$name.so" "$tmp/s3.o" "$S/flame_decode.o" "$S/flame_decode_pipe.o" "$S/capi.o" "$S/flame_backward.o" "$S/projection.o" "$S/preprocess.o" "$S/mesh_losses.o" "$S/cnn_glue.o" "$S/sim3dr_compat.o"
echo "built tools/_variants/lib_$name.so"
