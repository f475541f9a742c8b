#!/usr/bin/env python3
"""Where the drop-in (host-pointer) path spends a frame's time: orbx_extract and orbm_search_by_projection per call, wall clock
around the C call and the kernels' own time from the library's HIP events.  GPU box only."""
import ctypes as C, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("3_orb_slam3_selfnote_amd")
synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
frames, offs = synth.make_stream(1000, 2)
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
_, k0, d0 = ex(frames[0]); _, k1, d1 = ex(frames[1])
sf = ex.GetScaleFactors()
m = pkg.ORBmatcher(0.9, True)
F = pkg.FrameView(k1, d1, (0.0, 752.0, 0.0, 480.0))
lvl = k0["octave"].astype(np.int32)
args = dict(qdesc=d0, u=k0["x"], v=k0["y"], radius=(15.0 * sf[lvl]).astype(np.float32), min_level=lvl - 1, max_level=lvl + 1)
def search():
    F.slot[:] = -1; F.slot_obs[:] = 0
    return m.search_window(F, args["qdesc"], args["u"], args["v"], args["radius"], args["min_level"], args["max_level"], nnratio=0.9, th_dist=100, use_second=False)
for _ in range(10): search(); ex(frames[1])
R = 300
t0 = time.perf_counter()
for _ in range(R): n = search()[0]
t_search = (time.perf_counter() - t0) / R
t0 = time.perf_counter()
for _ in range(R): ex(frames[1], None, (0, 0))
t_ext = (time.perf_counter() - t0) / R
m.set_profiling(True); ex.set_profiling(True)
for _ in range(20): search(); ex(frames[1], None, (0, 0))
print("search (python call, tracking window): %.3f ms; kernels: %s" % (t_search * 1e3, {k: round(v, 4) for k, v in m.stage_ms().items()}))
print("extract (python call): %.3f ms; kernels: %s  sum %.4f" % (t_ext * 1e3, {k: round(v, 4) for k, v in ex.stage_ms().items()}, sum(ex.stage_ms().values())))
ex.set_profiling(False)
acc = np.zeros(4)
for _ in range(100):
    ex(frames[1], None, (0, 0))
    us = np.zeros(4, np.float32); ex.L.orbx_get_host_us(ex.h, us.ctypes.data_as(C.c_void_p), 4); acc += us
print("extract host side, us: staging copy %.1f, submission %.1f, wait %.1f, copy-out %.1f" % tuple(acc / 100))
# raw C call without the Python marshalling of search_window
a = lambda x, t: np.ascontiguousarray(x, dtype=t)
q = [a(args["qdesc"], np.uint8), a(args["u"], np.float32), a(args["v"], np.float32), a(args["radius"], np.float32), a(args["min_level"], np.int32), a(args["max_level"], np.int32)]
qs = pkg.QueryStruct(len(q[1]), *[x.ctypes.data for x in q], None, None)
fs = F.struct()
moq = np.zeros(len(q[1]), np.int32)
m.set_profiling(False)
t0 = time.perf_counter()
for _ in range(R):
    F.slot[:] = -1; F.slot_obs[:] = 0
    m.L.orbm_search_by_projection(m.m, C.byref(fs), C.byref(qs), C.c_float(0.9), 100, 0, F.slot.ctypes.data_as(C.c_void_p), F.slot_obs.ctypes.data_as(C.c_void_p), moq.ctypes.data_as(C.c_void_p), None)
print("search (raw C call): %.3f ms" % ((time.perf_counter() - t0) / R * 1e3))
