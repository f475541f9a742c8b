"""CPU: the libm replicas compiled into the kernels (csrc/orb_sincos.h, csrc/orb_atan2f.h - evaluated here on the HOST from the
same source, through orbx_ref_*) equal this host's libm bit for bit on the domains the device uses them on:
  cosf / sinf   psi = atan2f(y, x) in [-pi, pi] for KannalaBrandt8::project (KannalaBrandt8.cpp:42-43) and the keypoint angle in
                [0, 2 pi] for the descriptor (ORBextractor.cc:111)
  atanf         every float;  atan2f: random pairs of all magnitudes plus camera-like coordinates plus the special cases.
The sweeps run natively inside the oracle library (function pointers to the product's host evaluations).  Default: strided
samples (a few seconds); ORB_EXHAUSTIVE=1: every float (verified while authoring, glibc 2.35 x86-64: 0 mismatches)."""
import ctypes as C
import os

import numpy as np

EXH = os.environ.get("ORB_EXHAUSTIVE") == "1"


def _fn(L, name):
    return C.cast(getattr(L, name), C.c_void_p)


def _bits(x):
    return int(np.array([x], np.float32).view(np.uint32)[0])


def test_sincos_on_minus_pi_to_two_pi(pkg, oracle):
    L, O = pkg.load(), oracle.lib()
    O.orc_sweep_unary.restype = C.c_long
    O.orc_sweep_unary.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    step = 1 if EXH else 251
    for which, name in ((0, "orbx_ref_cosf"), (1, "orbx_ref_sinf")):
        assert O.orc_sweep_unary(_fn(L, name), which, 0, _bits(3.2), step, 1) == 0          # [-3.2, 3.2]
        assert O.orc_sweep_unary(_fn(L, name), which, _bits(3.2), _bits(6.3), step, 0) == 0   # (3.2, 6.3]


def test_atanf_every_float(pkg, oracle):
    L, O = pkg.load(), oracle.lib()
    O.orc_sweep_unary.restype = C.c_long
    O.orc_sweep_unary.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    assert O.orc_sweep_unary(_fn(L, "orbx_ref_atanf"), 2, 0, 0x7f800000, 1 if EXH else 173, 1) == 0
    # every breakpoint of the argument reduction, +- 4096 ulps
    for b in (0x31000000, 0x3ee00000, 0x3f300000, 0x3f980000, 0x401c0000, 0x4c000000, 0x3f800000):
        assert O.orc_sweep_unary(_fn(L, "orbx_ref_atanf"), 2, b - 4096, b + 4096, 1, 1) == 0


def test_atan2f_pairs_and_special_cases(pkg, oracle):
    L, O = pkg.load(), oracle.lib()
    O.orc_sweep_atan2f.restype = C.c_long
    O.orc_sweep_atan2f.argtypes = [C.c_void_p, C.c_uint64, C.c_long, C.c_float]
    O.orc_libm_atan2f.restype = C.c_float
    O.orc_libm_atan2f.argtypes = [C.c_float, C.c_float]
    n = 2_000_000_000 if EXH else 20_000_000
    for seed, scale in ((1, 10.0), (2, 0.5), (3, 1000.0)):
        assert O.orc_sweep_atan2f(_fn(L, "orbx_ref_atan2f"), seed, n // 3, C.c_float(scale)) == 0
    sp = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 1e-40, -1e-40, 3.4e38, -3.4e38, 1e-30, 2.5, 2.0 ** 61, -2.0 ** -61]
    for y in sp:
        for x in sp:
            a = np.float32(O.orc_libm_atan2f(C.c_float(y), C.c_float(x)))
            b = np.float32(L.orbx_ref_atan2f(C.c_float(y), C.c_float(x)))
            assert a.view(np.uint32) == b.view(np.uint32), (y, x)


def test_project_equals_oracle(pkg, oracle):
    """orbm_project (host evaluation of what k_lastframe_project computes) vs the oracle's libm-based restatement of
    Pinhole::project / KannalaBrandt8::project, on points all around the fisheye field of view."""
    rng = np.random.default_rng(4)
    kb8 = np.array([190.978477, 190.973307, 254.931706, 256.897442, 0.003482389402, 0.000715034845, -0.002053236141, 0.000202936736], np.float32)
    pin = np.array([458.654, 457.296, 367.215, 248.375], np.float32)
    for _ in range(20000):
        X, Y = rng.uniform(-5, 5, 2)
        Z = rng.uniform(-0.5, 8)
        for t, p in ((1, kb8), (0, pin)):
            if t == 0 and abs(Z) < 1e-3:
                continue
            assert pkg.project(t, p, X, Y, Z) == oracle.project(t, p, X, Y, Z)
