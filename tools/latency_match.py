#!/usr/bin/env python3
"""PCIe-inclusive latency of the host-pointer projection search (orbm_search_by_projection = ORBmatcher::SearchByProjection as the
adapter calls it): 1000 map points against a 1000-keypoint frame, window radius as in tracking.  GPU box only."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("3_orb_slam3_selfnote_amd")
synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
frames, offs = synth.make_stream(1000, 2)
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
_, k0, d0 = ex(frames[0]); _, k1, d1 = ex(frames[1])
sf = ex.GetScaleFactors()
m = pkg.ORBmatcher(0.8, True)
F = pkg.FrameView(k1, d1, (0.0, 752.0, 0.0, 480.0))
px = (k0["x"] + np.float32(offs[0][0] - offs[1][0])).astype(np.float32); py = (k0["y"] + np.float32(offs[0][1] - offs[1][1])).astype(np.float32)
vc = np.ones(len(k0), np.float32); lvl = k0["octave"].astype(np.int32); iv = np.ones(len(k0), np.uint8)
for _ in range(5):
    F.slot[:] = -1; F.slot_obs[:] = 0
    n, _, _ = m.SearchByProjection(F, iv, d0, px, py, vc, lvl, sf, th=3.0)
t0 = time.perf_counter(); R = 200
for _ in range(R):
    F.slot[:] = -1; F.slot_obs[:] = 0
    m.SearchByProjection(F, iv, d0, px, py, vc, lvl, sf, th=3.0)
dt = (time.perf_counter() - t0) / R
print("orbm_search_by_projection host API: %.3f ms/call (%d map points, %d keypoints, %d matches; PCIe + sync inclusive)" % (dt * 1e3, len(k0), len(k1), n))
