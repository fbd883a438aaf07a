#!/bin/bash
# One rocprofv3 PMC pass (--kernel-trace only) over bench.py's decode kernel: tools/pmc_one.sh <tag> "<counters>"
# honours DAD3D_LIB_PATH. Prints mean per launch per counter; csv kept under gpurun_out/pmc_one/<tag>.csv
export TMPDIR=/tmp; root="${GRAFT_REPO_ROOT:-/root/repo}"; tag="$1"; set="$2"; cd /tmp
out="$root/gpurun_out/pmc_one"; mkdir -p "$out"; rm -rf /tmp/pmc1_$tag
timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc1_$tag -- \
    python $root/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > /dev/null 2>/tmp/pmc1_err_$tag
f=$(find /tmp/pmc1_$tag -name "*counter_collection.csv" | head -1)
[ -z "$f" ] && { echo "pmc $tag: no output"; tail -3 /tmp/pmc1_err_$tag; exit 0; }
cp "$f" "$out/$tag.csv"
python3 - "$out/$tag.csv" "$tag" <<'PY'
import collections, csv, sys
acc, dur = collections.defaultdict(list), []
for r in csv.DictReader(open(sys.argv[1])):
    if "flame_decode_kernel" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]].append(float(r["Counter_Value"])); dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
print("PMC", sys.argv[2], {k: round(sum(v[20:]) / max(len(v[20:]), 1), 1) for k, v in acc.items()}, "kernel_us", round(sum(dur) / max(len(dur), 1), 2))
PY
