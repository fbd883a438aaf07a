#!/bin/bash
# raster_blend_kernel variants, same call (round 5): product | vb = product with variable-trip channel loops | vc = round 4's kernel
# with fixed-bound channel loops | vd = round 4's kernel (variant sources: patches of sim3dr_kernels.hip built with tools/build_variant_sim3dr.sh;
# see profiles/r05_raster_spills_ab.txt for what each one changes)
root="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$root"; mkdir -p gpurun_out/r05_raster
for round in 1 2; do
  python tools/ab_sim3dr.py blend_product 2>/dev/null | grep AB3D
  for v in vb vc vd; do DAD3D_LIB_PATH="$root/tools/_variants/lib_blend_$v.so" python tools/ab_sim3dr.py blend_$v 2>/dev/null | grep AB3D; done
done | tee gpurun_out/r05_raster/blend_ab.txt
