"""Build liborbhip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

-ffp-contract=off and correctly rounded fp32 divide/sqrt are part of the numerical contract ("Bit-exactness").
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liborbhip.so")
SOURCES = ["orbhip.hip", "orb_kernels.h", "orb_match_kernels.h", "orb_match_mfma.h", "orb_mfma_util.h", "orb_common.h", "orb_sincos.h", "orb_project_kernels.h", "orb_atan2f.h"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function",
         # MFMA results in VGPRs (gfx950's register file is unified): k_match_scan_mfma feeds every accumulator to the VALU, from AGPRs each
         # would cost a v_accvgpr_read first
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "..", "include", "orbhip.h"),
                                                        os.path.join(HERE, "..", "include", "orb_pattern_data.inc")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, "orbhip.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
