"""CPU: the kernels the metric rests on stay spill-free across compiler releases. Compiles the decode and Sim3DR translation units for
gfx950 with `-Rpass-analysis=kernel-resource-usage` (hipcc cross-compiles without a GPU) and asserts, for EVERY kernel in them: no scratch
(private segment 0 bytes per lane), no VGPR spills, and SGPR spills only where csrc documents them -- the pipelined decode kernel keeps
the exec mask of its role split in two lanes of a VGPR (one v_writelane pair at the top of the launch, one v_readlane pair at its end:
flame_decode_pipe.hip), the blend kernel of the alpha != 1 path its loop masks. The one exception is not on any BASELINE config: the
two-role kernel's FULL-POSE instantiations (K = 448: configs that drive neck and eyeballs, `flame_decode_kernel<28, false, ..>`) keep 20
bytes per lane on the stack. VERDICT r5 #5."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dad-3dheads_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"]
# (source, extra flags as in csrc/Makefile, {kernel-name substring: SGPR spills allowed}, VGPR ceiling of the unit's hot kernel)
UNITS = [
    ("flame_decode_pipe.hip", [], {"flame_decode_pipe_kernel": 3}, 256),
    ("flame_decode_split.hip", ["-fno-slp-vectorize"], {}, 256),
    ("flame_decode.hip", [], {"flame_decode_kernelILi26": 40, "flame_decode_kernelILi28": 96}, 256),  # (fallback + training forward; SCRATCH_OK below)
    ("sim3dr_kernels.hip", ["-ffp-contract=off"], {"raster_blend_kernel": 24, "raster_kernel": 4, "tri_geometry_kernel": 8}, 128),
]


SCRATCH_OK = {"flame_decode_kernelILi28ELb0": 32}  # bytes per lane: full-pose instantiations of the two-role kernel only


def _resource_usage(source, flags, tmp_path):
    out = subprocess.run([HIPCC, *COMMON, *flags, "-c", os.path.join(CSRC, source), "-o", str(tmp_path / (source + ".o"))],
                         capture_output=True, text=True, cwd=CSRC)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"remark: .*?Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark: .*?\s{2,}([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return kernels


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("source,flags,sgpr_ok,vgpr_cap", UNITS, ids=[u[0] for u in UNITS])
def test_no_scratch_no_vgpr_spills(source, flags, sgpr_ok, vgpr_cap, tmp_path):
    kernels = _resource_usage(source, flags, tmp_path)
    assert kernels, "no kernel-resource-usage remarks: did the flag change?"
    for name, k in kernels.items():
        assert int(k["ScratchSize"]) <= max((n for key, n in SCRATCH_OK.items() if key in name), default=0), (name, k)
        assert int(k["VGPRs Spill"]) == 0, (name, k)
        allowed = max((n for key, n in sgpr_ok.items() if key in name), default=0)
        assert int(k["SGPRs Spill"]) <= allowed, (name, k, allowed)
        assert int(k["VGPRs"]) + int(k.get("AGPRs", 0)) <= 256, (name, k)  # two waves per SIMD: 512 registers / 2
    hot = max(int(k["VGPRs"]) for k in kernels.values())
    assert hot <= vgpr_cap, (source, hot)
    print(source, {n[-60:]: (k["VGPRs"], k["SGPRs Spill"]) for n, k in kernels.items()})


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_split_kernels_register_budgets_and_no_packed_f32(tmp_path):
    """The fp16x2 form launches TEN waves per workgroup (three per SIMD on two of them): 512 / 3 -> at most 168 registers, without spills; the
    bf16x3 form eight (256). Neither may contain a packed-f32 VALU instruction: `v_pk_*_f32` in a wave beside a 16-bit MFMA stream returned wrong
    low halves (profiles/r06_kernel_log.md section 4) -- the file is built with -fno-slp-vectorize and scales its vectors element by element."""
    kernels = _resource_usage("flame_decode_split.hip", ["-fno-slp-vectorize"], tmp_path)
    f16 = {n: k for n, k in kernels.items() if "flame_decode_split_kernel" in n and "F16x2" in n}
    bf16 = {n: k for n, k in kernels.items() if "flame_decode_split_kernel" in n and "Bf16x3" in n}
    assert len(f16) == 4 and len(bf16) == 4, list(kernels)  # TO2D x write-back / write-through stores
    for name, k in f16.items():
        assert int(k["VGPRs"]) + int(k.get("AGPRs", 0)) <= 168 and int(k["Occupancy"]) >= 3, (name, k)
    for name, k in bf16.items():
        assert int(k["VGPRs"]) + int(k.get("AGPRs", 0)) <= 256 and int(k["Occupancy"]) >= 2, (name, k)
    asm = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-fno-slp-vectorize", "-S",
                          os.path.join(CSRC, "flame_decode_split.hip"), "-o", "-"], capture_output=True, text=True, cwd=CSRC)
    assert asm.returncode == 0, asm.stderr[-2000:]
    packed = sorted(set(re.findall(r"\bv_pk_\w*f32\b", asm.stdout)))
    assert not packed, packed
    assert asm.stdout.count("v_mfma_f32_16x16x32_f16") > 0 and asm.stdout.count("v_mfma_f32_16x16x32_bf16") > 0
