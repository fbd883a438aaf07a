#!/bin/bash
export TMPDIR=/tmp; root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r05"; mkdir -p "$out"; cd /tmp; rm -rf /tmp/chain
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/chain -- python $root/tools/r05_chain_probe.py 200 > /dev/null 2>/tmp/chain_err
f=$(find /tmp/chain -name "*kernel_trace.csv" | head -1)
[ -z "$f" ] && { tail -5 /tmp/chain_err; exit 0; }
python3 - "$f" <<'PY' | tee "$out/chain_probe.txt"
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
dec = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "flame_decode_pipe" in r["Kernel_Name"]]
N = 200
dec = dec[50:]  # warm-up render steps
names = ["A decode<false> proj-only alone", "B decode<false> inside the render step", "C decode<false> + 192 MB memset between launches",
         "D decode<true> headline outputs alone", "E decode<false> verts3d + proj3 alone"]
for i, n in enumerate(names):
    d = [x[1] for x in dec[i * N:(i + 1) * N]][20:]
    if d:
        d.sort()
        print(f"{n:55s} n={len(d)} mean {sum(d)/len(d):6.2f} us  median {d[len(d)//2]:6.2f}  min {d[0]:6.2f}")
PY
