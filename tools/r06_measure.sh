#!/bin/bash
# Round 6 evidence in one gpurun call: the driver's bench command plain and under rocprofv3 (kernel trace + stats), PMC passes over the
# split kernel and the fp32 kernel at B = 256 (separate --pmc runs, --kernel-trace only), the headline kernel's PMC refresh.
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r06"; mkdir -p "$out"; cd "$root"
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > "$out/bench_driver_command.json" 2> "$out/bench_driver_command.err"
tail -3 "$out/bench_driver_command.err"
cd /tmp; rm -rf /tmp/prof_bench
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python "$root/bench.py" --gpus 1 --steps 20 --warmup 5 > "$out/bench_profiled.json" 2> "$out/bench_profiled.err"
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/bench_kernel_stats.csv"
python "$root/tools/r06_kernel_summary.py" /tmp/prof_bench "$out/bench_profiled.json" > "$out/bench_kernel_summary.md"; cat "$out/bench_kernel_summary.md"
# PMC: the two decode kernels at B = 256, one counter set per pass
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for k in split split_f16 fp32; do
    rm -rf /tmp/pmcs_${k}_$i
    (cd /tmp && timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcs_${k}_$i -- python $root/tools/split_pmc_driver.py $k 256 120 > /dev/null 2>/tmp/pmcs_err)
    f=$(find /tmp/pmcs_${k}_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" "/tmp/pmcs_${k}_set$i.csv" || { echo "pmc $k set $i: no output"; tail -2 /tmp/pmcs_err; }
  done
done
python3 - "$out" <<'PY'
import collections, csv, glob, json, sys
res = {"method": "tools/r06_measure.sh: rocprofv3 --pmc <one set per pass> --kernel-trace over tools/split_pmc_driver.py (B = 256, every output, 120 launches; "
                 "mean per launch over the last 100); FETCH_SIZE / WRITE_SIZE in KB, fetch bytes = 2 x FETCH_SIZE x 1024 on gfx950 (MI355X_MICROARCH.md)", "kernels": {}}
for k in ("split", "split_f16", "fp32"):
    acc, dur = collections.defaultdict(lambda: collections.defaultdict(list)), collections.defaultdict(list)
    for path in sorted(glob.glob(f"/tmp/pmcs_{k}_set*.csv")):
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"]
            if "flame_decode" not in n and "split_params" not in n:
                continue
            short = n[:n.rfind("(")].replace("void dad3d::", "").replace("dad3d::", "").replace("(anonymous namespace)::", "")
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    for short, cs in acc.items():
        m = {c: sum(v[-100:]) / max(len(v[-100:]), 1) for c, v in cs.items()}
        e = {"mean_per_launch": m, "kernel_us_under_counters": sum(dur[short][-400:]) / max(len(dur[short][-400:]), 1)}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"]:
            e["mfma_busy_fraction_of_kernel_time_all_1024_simds"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["GRBM_GUI_ACTIVE"])
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e["traffic_bytes_per_launch"] = 2 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024
        res["kernels"][short] = e
json.dump(res, open(sys.argv[1] + "/pmc_split_b256.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
PMC_OUT=$out/pmc_decode bash $root/tools/pmc_decode.sh > $out/pmc_decode.log 2>&1; tail -25 $out/pmc_decode.log
