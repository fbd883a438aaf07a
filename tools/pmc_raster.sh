#!/bin/bash
# PMC passes over the raster chain (tools/raster_probe.py): per-kernel means of SQ / TA / TCP counters, one pass per set.
export TMPDIR=/tmp; cd /tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_TA_BUSY_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1)); rm -rf /tmp/pmcr_$i
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcr_$i -- python $root/tools/raster_probe.py 64 > /dev/null 2>/tmp/pmcr_err_$i
  f=$(find /tmp/pmcr_$i -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "set $i: no output"; tail -3 /tmp/pmcr_err_$i; continue; }
  cp $f $root/gpurun_out/pmc_raster_$i.csv
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    key = "raster" if "raster_kernel" in n else "geometry" if "tri_geometry" in n else "queue" if "raster_queue" in n else None
    if key:
        acc[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    v = v[2:] if len(v) > 3 else v
    print(k, c, len(v), sum(v) / len(v))
PY
done
