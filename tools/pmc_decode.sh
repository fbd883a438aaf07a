#!/bin/bash
# PMC passes over the contract benchmark's dominant kernel (bench.py, batch 64): one rocprofv3 run per counter set with
# --kernel-trace only (never combined with the sys / hip / hsa trace domains). Writes the per-dispatch csv files and a
# per-launch summary json under gpurun_out/pmc_decode/; the json is what profiles/r01_pmc_*.json are built from.
export TMPDIR=/tmp; cd /tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"
out="${PMC_OUT:-$root/gpurun_out/pmc_decode}"; mkdir -p "$out"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1)); rm -rf /tmp/pmcd_$i
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcd_$i -- \
      python $root/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > /dev/null 2>/tmp/pmcd_err_$i
  f=$(find /tmp/pmcd_$i -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "set $i: no output"; tail -3 /tmp/pmcd_err_$i; continue; }
  cp "$f" "$out/set$i.csv"
done
python3 - "$out" <<'PY'
import collections, csv, glob, json, sys
acc, dur, name = collections.defaultdict(list), [], None
for path in sorted(glob.glob(sys.argv[1] + "/set*.csv")):
    for r in csv.DictReader(open(path)):
        if "flame_decode" not in r["Kernel_Name"] or "readjust" in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"]
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
summary = {"kernel": name, "launches_per_counter": {k: len(v) for k, v in acc.items()},
           "mean_per_launch": {k: sum(v[20:]) / max(len(v[20:]), 1) for k, v in acc.items()},
           "kernel_us_under_counter_collection": sum(dur) / max(len(dur), 1)}
json.dump(summary, open(sys.argv[1] + "/summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
