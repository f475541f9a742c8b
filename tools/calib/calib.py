"""Build and load tools/calib/liborbcalib.so (measurement aids: HBM-counter calibration copy, vector-issue ceilings per opcode class).
Not part of the product: nothing under 3_orb_slam3_selfnote_amd/ imports this."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liborbcalib.so")
SOURCES = ["orbcalib.hip", "orbcalib.h", "orb_calib.h"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def build(force=False, verbose=False):
    if not force and os.path.exists(LIB) and all(os.path.getmtime(os.path.join(HERE, s)) <= os.path.getmtime(LIB) for s in SOURCES):
        return LIB
    cmd = [HIPCC] + FLAGS + [os.path.join(HERE, "orbcalib.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def load(build_if_needed=True):
    if build_if_needed and os.path.exists(HIPCC):
        build()
    L = C.CDLL(LIB)
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    L.orbx_calibration_copy.argtypes = [vp, vp, sz, vp]
    L.orbx_calibration_valu_ops.argtypes = []
    L.orbx_calibration_valu_name.restype = C.c_char_p
    L.orbx_calibration_valu_name.argtypes = [i32]
    L.orbx_calibration_valu.argtypes = [i32, i32, i32, i32, vp, vp, vp]
    return L


if __name__ == "__main__":
    print(build(force=True, verbose=True))
