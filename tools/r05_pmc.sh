#!/bin/bash
# Round 5 PMC passes (same method as tools/r03_pmc.sh, refreshed after the raster kernels lost their spills) (rocprofv3 --pmc, one counter per pass, --kernel-trace only):
#   1. HBM-side traffic of every kernel of the Sim3DR chain (tools/ab_sim3dr.py: normals, normals + Phong, rasterize, render
#      of 64 heads) -> gpurun_out/r05/pmc_raster.json   (FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM": the counter tallies
#      128-byte requests at 64 bytes; both counters are in KB)
#   2. the decode kernel of the contract benchmark (tools/pmc_decode.sh) -> gpurun_out/pmc_decode/summary.json
export TMPDIR=/tmp; root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r05"; mkdir -p "$out"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcr_$c
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcr_$c -- python $root/tools/ab_sim3dr.py pmc > /dev/null 2>/tmp/pmcr_err_$c)
  f=$(find /tmp/pmcr_$c -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "pmc pass $c: no output"; tail -3 /tmp/pmcr_err_$c; continue; }
  cp "$f" "$out/pmc_raster_$c.csv"
done
python3 - "$out" <<'PY'
import collections, csv, json, re, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    try:
        rows = list(csv.DictReader(open(f"{out}/pmc_raster_{c}.csv")))
    except Exception as e:
        print("missing", c, e); continue
    for r in rows:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if "dad3d" not in n or "flame_decode" in n: continue
        m = re.search(r"(\w+(<[^(]*>)?)\(", n)
        k = m.group(1) if m else n
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
B = 64
alg = {"raster_kernel<0>": None, "tri_geometry_kernel<true, 0>": None}
res = {"method": "tools/r05_pmc.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over tools/ab_sim3dr.py, "
                 "B = 64 heads, 9976 triangles, 256x256x3; mean per launch; counters in KB; fetch_bytes = 2 x FETCH_SIZE x 1024 "
                 "(gfx950 tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md), write_bytes = WRITE_SIZE x 1024", "kernels": {}}
for k, cs in acc.items():
    f = sum(cs.get("FETCH_SIZE", [0])) / max(len(cs.get("FETCH_SIZE", [1])), 1)
    w = sum(cs.get("WRITE_SIZE", [0])) / max(len(cs.get("WRITE_SIZE", [1])), 1)
    res["kernels"][k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "fetch_bytes": 2 * f * 1024, "write_bytes": w * 1024,
                         "traffic_bytes": 2 * f * 1024 + w * 1024, "kernel_us_under_counters": sum(dur[k]) / max(len(dur[k]), 1),
                         "launches": len(cs.get("FETCH_SIZE", []))}
chain = [k for k in res["kernels"] if k.startswith("raster_kernel<0>") or k.startswith("tri_geometry_kernel<true, 0>")]
tot = sum(res["kernels"][k]["traffic_bytes"] for k in chain)
res["rasterize chain (tri_geometry_kernel<true, 0> + raster_kernel<0>)"] = {
    "traffic_bytes_per_64_images": tot, "algorithmic_bytes_per_64_images": B * 513768, "ratio": tot / (B * 513768) if tot else None}
nk = [k for k in res["kernels"] if "ver_normal" in k]
if nk:
    t = res["kernels"][nk[0]]["traffic_bytes"]
    res["get_normal (" + nk[0] + ")"] = {"traffic_bytes_per_64_images": t, "algorithmic_bytes_per_64_images": B * 120552 + 119712,
                                       "ratio": t / (B * 120552 + 119712)}
json.dump(res, open(out + "/pmc_raster.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
PMC_OUT=$out/pmc_decode bash $root/tools/pmc_decode.sh > $out/pmc_decode.log 2>&1; tail -30 $out/pmc_decode.log
