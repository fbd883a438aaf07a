#!/bin/bash
# rocprofv3 PMC passes (--kernel-trace only, one counter set per pass) over the Sim3DR kernels of tools/ab_sim3dr.py:
#   bash tools/pmc_sim3dr.sh <kernel name substring> "<set 1>" "<set 2>" ...      honours DAD3D_LIB_PATH
# Prints the mean per launch of every counter for kernels whose name contains the substring; csv under gpurun_out/pmc_sim3dr/
export TMPDIR=/tmp; root="${GRAFT_REPO_ROOT:-/root/repo}"; pat="$1"; shift
out="$root/gpurun_out/pmc_sim3dr"; mkdir -p "$out"; i=0
for set in "$@"; do
  i=$((i + 1)); rm -rf /tmp/pmc3_$i
  (cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc3_$i -- python $root/tools/ab_sim3dr.py pmc > /dev/null 2>/tmp/pmc3_err_$i)
  f=$(find /tmp/pmc3_$i -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "pmc pass $i: no output"; tail -3 /tmp/pmc3_err_$i; continue; }
  cp "$f" "$out/pass$i.csv"
  python3 - "$out/pass$i.csv" "$pat" <<'PY'
import collections, csv, sys
acc, dur = collections.defaultdict(list), []
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
print("PMC3", sys.argv[2], {k: round(sum(v) / max(len(v), 1), 1) for k, v in acc.items()}, "launches", len(dur) // max(len(acc), 1), "kernel_us", round(sum(dur) / max(len(dur), 1), 2))
PY
done
