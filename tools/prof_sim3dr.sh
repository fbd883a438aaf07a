#!/bin/bash
# rocprofv3 kernel durations of the Sim3DR chain for variant libraries (GPU time, not host-paced timing):
#   bash tools/prof_sim3dr.sh product na1 ...
export TMPDIR=/tmp; root="${GRAFT_REPO_ROOT:-/root/repo}"
for v in "$@"; do
  if [ "$v" = product ]; then unset DAD3D_LIB_PATH; else export DAD3D_LIB_PATH="$root/tools/_variants/lib_$v.so"; fi
  rm -rf /tmp/ps_$v; (cd /tmp && timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$v -- python $root/tools/ab_sim3dr.py $v > /tmp/ps_$v.log 2>&1)
  f=$(find /tmp/ps_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(grep AB3D /tmp/ps_$v.log | sed 's/.*normals_exact/normals_exact/')"
  python3 - "$f" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "dad3d" in n and "flame_decode" not in n:
        m = re.search(r"(\w+(<[^(]*>)?)\(", n.replace("(anonymous namespace)::", ""))
        short = (m.group(1) if m else n)[:60]
        print(f"   {short:62s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:7.2f} us")
PY
done
