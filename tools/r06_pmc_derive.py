#!/usr/bin/env python3
"""Derived fields of profiles/r06/pmc_split_b256.json (tools/r06_measure.sh writes the raw means):

    python tools/r06_pmc_derive.py gpurun_out/r06/pmc_split_b256.json <bf16x3 tile us> <fp16x2 tile us> <fp32 us>

the three times = the kernels' averages over the timed pass of the SAME box's rocprofv3 --kernel-trace run of the driver command
(bench_kernel_summary.md): counters slow a launch down, the busy fraction is taken against the kernel's time without them."""
import json
import sys

path = sys.argv[1]
plain = dict(zip(("flame_decode_split_kernel<Bf16x3, false, false>", "flame_decode_split_kernel<F16x2, false, false>", "flame_decode_pipe_kernel<false, false>"),
                 map(float, sys.argv[2:5])))
d = json.load(open(path))
for k, v in d["kernels"].items():
    m = v["mean_per_launch"]
    v.pop("mfma_busy_fraction_of_kernel_time_all_1024_simds", None)
    if k in plain and m.get("SQ_INSTS_MFMA"):
        v["mfma_busy_cycles_per_simd"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
        v["valu_instructions_per_mfma"] = round(m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"], 3)
        v["kernel_us_without_counters"] = plain[k]
        v["mfma_busy_fraction_of_kernel_time"] = round(v["mfma_busy_cycles_per_simd"] / (plain[k] * 2400), 3)
    print(k, {a: b for a, b in v.items() if a != "mean_per_launch"})
d["note"] = ("derived fields (tools/r06_pmc_derive.py): mfma_busy_cycles_per_simd = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs; mfma_busy_fraction_of_kernel_time = that / "
             "(kernel_us_without_counters x 2400 MHz nominal), kernel_us_without_counters = the kernel's average over the timed pass of the same box's rocprofv3 "
             "--kernel-trace run of the driver command (bench_kernel_summary.md); valu_instructions_per_mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA. SQ_INSTS_MFMA: bf16x3 "
             "1 257 984 = 252 x 4 x 78 x 16, fp16x2 628 992 = 252 x 4 x 39 x 16. The split kernels run below the nominal clock on real data at large batches "
             "(r06_kernel_log.md section 8): their busy fraction in TIME is higher than the figure here by the clock ratio.")
json.dump(d, open(path, "w"), indent=1)
