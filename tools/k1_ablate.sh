#!/bin/bash
# Diagnostics: build variants of libdad3d_hip.so with parts of tri_geometry_kernel compiled out (-DDAD3D_K1_ABLATE=bits:
# 1 no LDS binning atomics, 2 no list writes, 4 no record writes; the tile queue stays empty, results are WRONG) and
# print the kernel's rocprofv3 average for each.  Usage (on the GPU box): bash tools/k1_ablate.sh build|run 0 1 2 3 4 7
set -e
mode=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
if [ "$mode" = build ]; then
  cd "$root/dad-3dheads_amd/csrc" && make -s && mkdir -p "$root/tools/_ablate"
  for n in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DDAD3D_K1_ABLATE=$n -c sim3dr_kernels.hip -o /tmp/sk_$n.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/_ablate/lib_k1_$n.so" flame_decode.o /tmp/sk_$n.o capi.o sim3dr_compat.o
  done
else
  export TMPDIR=/tmp; cd /tmp
  for n in "$@"; do
    rm -rf /tmp/rp_$n
    DAD3D_LIB_PATH="$root/tools/_ablate/lib_k1_$n.so" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$n -- python "$root/tools/raster_probe.py" 64 > /dev/null 2>&1 || true
    f=$(find /tmp/rp_$n -name "*kernel_stats.csv" | head -1)
    echo "ablate=$n $(grep tri_geometry $f | awk -F'",' '{print $2}' | cut -d, -f1-3)"
  done
fi
