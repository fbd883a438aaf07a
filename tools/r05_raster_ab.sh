#!/bin/bash
# Round 5: raster kernels with / without the scalar-register spills. Same call, same box:
#   product                = this tree (raster_kernel<0> 0 spills / 100 VGPR, raster_blend_kernel 0 spills)
#   tools/_variants/lib_r04raster.so = round 4's sim3dr_kernels.hip (61 / 23 / 110 spills) rebuilt by tools/build_variant_sim3dr.sh
# Build the comparison library first (authoring container):
#   git show 94943aa:dad-3dheads_amd/csrc/sim3dr_kernels.hip > /tmp/sim_old.hip && tools/build_variant_sim3dr.sh r04raster "" /tmp/sim_old.hip
# Output: gpurun_out/r05_raster/{ab.txt, pmc_*.txt, stats_*.csv}
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r05_raster"; mkdir -p "$out"
cd "$root"
old="$root/tools/_variants/lib_r04raster.so"
: > "$out/ab.txt"
for round in 1 2 3; do
  python tools/ab_sim3dr.py product_r05 2>/dev/null | grep AB3D >> "$out/ab.txt"
  DAD3D_LIB_PATH="$old" python tools/ab_sim3dr.py r04_raster 2>/dev/null | grep AB3D >> "$out/ab.txt"
done
cat "$out/ab.txt"
cd /tmp
for tag in product r04; do
  lib=""; [ "$tag" = r04 ] && lib="$old"
  rm -rf /tmp/pmc_$tag /tmp/st_$tag
  DAD3D_LIB_PATH="$lib" timeout 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python $root/tools/raster_probe.py 64 > /dev/null 2>/tmp/pmc_err_$tag
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$tag" <<'PY' | tee "$out/pmc_$tag.txt"
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    key = "raster_kernel" if "raster_kernel" in n else "tri_geometry" if "tri_geometry" in n else None
    if key:
        acc[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    v = v[2:] if len(v) > 3 else v
    print(sys.argv[2], k, c, "launches", len(v), "mean per launch", sum(v) / len(v))
PY
  else echo "pmc $tag: no output"; tail -3 /tmp/pmc_err_$tag; fi
  DAD3D_LIB_PATH="$lib" timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$tag -- python $root/tools/ab_sim3dr.py prof_$tag > /dev/null 2>&1
  f=$(find /tmp/st_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/stats_$tag.csv" && head -8 "$f"
done
