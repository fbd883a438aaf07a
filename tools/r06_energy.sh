#!/bin/bash
# Round 6, section 8 of the kernel log in one call: data dependence of the three kernels, the shader clock of the split kernels with parts of
# the work removed (variants tools/_variants/lib_clk*.so: tools/build_variant.sh clk_aN "-DDAD3D_SPLIT_CLOCKS -DDAD3D_SPLIT_ABLATE=N"),
# the pure-MFMA clock probe.
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r06"; mkdir -p "$out"; cd "$root"
{ python tools/data_dependence_ab.py 64 256 2048; DAD3D_DECODE_KERNEL=split python tools/data_dependence_ab.py 64 256 2048
  DAD3D_DECODE_KERNEL=split_f16 python tools/data_dependence_ab.py 64 256 2048; } 2>&1 | grep DATA | tee "$out/data_dependence_ab.txt"
SPLIT_FORM=split bash tools/r06_clock_ablate.sh clk clk_a1 clk_a2 clk_a4 clk_a256 clk_a512; cp "$out/clock_ablate.txt" "$out/clock_ablate_bf16.txt"
SPLIT_FORM=split_f16 bash tools/r06_clock_ablate.sh clk clk_a1 clk_a2 clk_a4 clk_a256 clk_a512; cp "$out/clock_ablate.txt" "$out/clock_ablate_f16.txt"
[ -x tools/_variants/clock_probe ] && timeout 250 tools/_variants/clock_probe 30 | tee "$out/clock_probe_mfma.txt"
