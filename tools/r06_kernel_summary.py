#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run of the driver's bench command, with the one split --stats cannot make:
`flame_decode_pipe_kernel<false, false>` is launched by two legs of that command (secondary.decode_b256: 256 images; secondary.render_b64:
64 images) -- the same kernel name and grid (252 workgroups), told apart by LAUNCH ORDER: the B = 256 leg comes first and issues exactly
50 warm-up launches + (settle_passes + 1) x steps, both printed in the bench line (durations would not do: in the render leg's two-stream
part kernels overlap and a 64-image launch can take as long as a 256-image one).

    python tools/r06_kernel_summary.py <dir with *_kernel_trace.csv> [bench.json]   -> markdown on stdout"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    if "flame_decode_split_kernel" in name or "split_params_kernel" in name:  # templates over the split form (Bf16x3 | F16x2), TO2D, WB
        form = "F16x2" if "F16x2" in name else "Bf16x3"
        if "split_params" in name:
            return f"split_params_kernel<{form}>"
        flags = name[name.index(form) + len(form):].split(">")[0]  # ", false, false"
        return f"flame_decode_split_kernel<{form}{flags}>"
    for key in ("flame_decode_pipe_kernel<true, false>", "flame_decode_pipe_kernel<false, false>", "flame_decode_pipe_kernel<true, true>",
                "flame_decode_kernel", "raster_kernel<0>", "raster_blend_kernel",
                "tri_geometry_kernel<true, 2>", "tri_geometry_kernel<true, 0>", "readjust_kernel", "ncclDevKernel", "copyBuffer", "fillBuffer"):
        if key in name:
            return key
    return name.split("(")[0][-60:]


PAIRS = defaultdict(list)


def main():
    files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *_kernel_trace.csv under " + sys.argv[1])
    bench = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]) if len(sys.argv) > 2 else None
    n_b256 = None
    if bench and "secondary" in bench:
        leg = bench["secondary"]["decode_b256"]
        n_b256 = 50 + (int(leg["settle_passes"]) + 1) * int(leg["steps"])
    dur, seen_false = defaultdict(list), 0
    last_pre = {}
    for r in sorted(csv.DictReader(open(files[0])), key=lambda r: int(r["Start_Timestamp"])):  # launch order
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = short(r["Kernel_Name"])
        if k.startswith("split_params_kernel"):
            last_pre[k[len("split_params_kernel"):]] = d  # "<Form>"
        elif k.startswith("flame_decode_split_kernel"):
            form = k[len("flame_decode_split_kernel"):].split(",")[0] + ">"
            PAIRS[k].append((last_pre.get(form, 0.0), d))  # a step = the pre-pass in front of it + the tile kernel
        if k == "flame_decode_pipe_kernel<false, false>":
            seen_false += 1
            first_leg = (seen_false <= n_b256) if n_b256 is not None else d > 25.0
            k += " B=256 (secondary.decode_b256)" if first_leg else " B=64 (secondary.render_b64, one and two streams)"
        dur[k].append(d)
    print("| kernel | launches | average us | min us | max us |\n|---|---|---|---|---|")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"| {k} | {len(v)} | {sum(v) / len(v):.3f} | {min(v):.2f} | {max(v):.2f} |")
    if bench:
        d = bench
        b256 = dur.get("flame_decode_pipe_kernel<false, false> B=256 (secondary.decode_b256)", [])
        b64 = dur.get("flame_decode_pipe_kernel<true, false>", [])
        print()
        if b64:
            a = sum(b64) / len(b64)
            print(f"headline kernel: rocprofv3 average {a:.3f} us over {len(b64)} launches; the same run printed roofline.kernel_us {d['roofline']['kernel_us']:.3f} "
                  f"({(a / d['roofline']['kernel_us'] - 1) * 100:+.1f} %), long_region {d['long_region']['ms_per_step'] * 1e3:.3f}")
        if b256:
            leg = d["secondary"]["decode_b256"]
            p, steps = leg["ms_per_step"] * 1e3, int(leg["steps"])
            a_all, timed = sum(b256) / len(b256), b256[-steps:]  # the leg's LAST pass is the timed one (untimed settle passes precede it)
            a = sum(timed) / len(timed)
            print(f"B = 256 kernel: rocprofv3 average over the timed pass (the last {len(timed)} of its {len(b256)} launches; {leg.get('settle_passes')} untimed "
                  f"settle passes of the same length and 50 warm-up launches precede it, clock still ramping: average over all {a_all:.3f} us) {a:.3f} us; "
                  f"the same run printed secondary.decode_b256.ms_per_step {p:.3f} us ({(a / p - 1) * 100:+.1f} %)")
        split_lines(dur, d)


def split_lines(dur, bench):
    for form, legname, what in (("Bf16x3", "decode_b256_split", "bf16x3 split"), ("F16x2", "decode_b256_split_f16", "fp16x2 split")):
        pairs = PAIRS.get(f"flame_decode_split_kernel<{form}, false, false>", [])  # the leg's 3-component launches (its sub-legs are 2-D / landmark-only)
        if pairs and bench and legname in bench.get("secondary", {}):
            leg = bench["secondary"][legname]
            steps = int(leg["steps"])
            timed = pairs[-steps:]
            a, b = sum(t for _, t in timed) / len(timed), sum(p for p, _ in timed) / len(timed)
            print(f"{what}, B = 256: tile kernel {a:.3f} us + pre-pass {b:.3f} us = {a + b:.3f} us per step over the timed pass (the last {len(timed)} of the leg's "
                  f"{len(pairs)} 3-component launches); the same run printed secondary.{legname}.ms_per_step {leg['ms_per_step'] * 1e3:.3f} us (the difference is the "
                  f"gap between the two launches)")
    lm = dur.get("flame_decode_pipe_kernel<true, true>", [])
    if lm:
        print(f"landmark sub-model (chunked grid): {len(lm)} launches, B = 256 and B = 2048 legs together; min {min(lm):.2f} us, max {max(lm):.2f} us")


if __name__ == "__main__":
    main()
