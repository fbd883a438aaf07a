#!/bin/bash
# Steady-state kernel mix of the DAD-3DNet forward (tools/prof_cnn.py): rocprofv3 kernel trace, the LAST 6000 kernel
# records only (the front of the run is MIOpen's find phase, which tries -- and times -- naive reference kernels).
export TMPDIR=/tmp; root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r03"; mkdir -p "$out"
rm -rf /tmp/pc; (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pc -- python $root/tools/prof_cnn.py ${1:-64} > $out/prof_cnn.log 2>&1)
grep "CNN ms" $out/prof_cnn.log
f=$(find /tmp/pc -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $out/cnn_steady_state.txt
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-6000:]
span = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) * 1e-6
acc = collections.defaultdict(lambda: [0, 0.0])
for r in tail:
    k = r["Kernel_Name"][:110]
    acc[k][0] += 1
    acc[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
busy = sum(v[1] for v in acc.values())
print(f"steady-state window: {len(tail)} kernels, {span:.1f} ms wall, {busy:.1f} ms of kernel time ({busy / span:.0%} busy)")
for k, (n, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{ms:8.2f} ms {ms / busy:6.1%} {n:6d} x {ms / n * 1e3:8.1f} us  {k}")
PY
