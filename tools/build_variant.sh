#!/bin/bash
# Builds a VARIANT of libdad3d_hip.so next to (never instead of) the product library, for A/B timing on the GPU box:
#   tools/build_variant.sh <name> "<extra hipcc flags for flame_decode>" [decode source]
# -> tools/_variants/lib_<name>.so ; use through DAD3D_LIB_PATH (dad_3dheads_amd/_lib.py). tools/_variants/ is git-ignored
# but travels with the gpurun snapshot.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"; S="$root/dad-3dheads_amd/csrc"
name="$1"; flags="$2"; src="${3:-$S/flame_decode.hip}"
H=/opt/rocm/bin/hipcc; C="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I$S -I$root/include"
mkdir -p "$root/tools/_variants"; tmp="$(mktemp -d)"; trap 'rm -rf "$tmp"' EXIT
(cd "$S" && make -s)
$H $C $flags -x hip -c "$src" -o "$tmp/fd.o"
$H $C $flags -x hip -c "${PIPE_SRC:-$S/flame_decode_pipe.hip}" -o "$tmp/fdp.o"  # PIPE_SRC: another revision of the pipelined kernel (e.g. `git show HEAD:...` into a file)
$H $C $flags -fno-slp-vectorize -x hip -c "${SPLIT_SRC:-$S/flame_decode_split.hip}" -o "$tmp/fds.o"  # SPLIT_SRC: another revision of the bf16x3 split kernel
$H $C $flags -DDAD3D_DIAG_SPIN_ENV -x hip -c "$S/capi.cpp" -o "$tmp/capi.o"
$H --offload-arch=gfx950 -shared -fPIC -o "$root/tools/_variants/lib_$name.so" "$tmp/fd.o" "$tmp/fdp.o" "$tmp/fds.o" "$tmp/capi.o" "$S/flame_backward.o" "$S/sim3dr_kernels.o" "$S/projection.o" "$S/preprocess.o" "$S/mesh_losses.o" "$S/cnn_glue.o" "$S/sim3dr_compat.o"
echo "built tools/_variants/lib_$name.so"
