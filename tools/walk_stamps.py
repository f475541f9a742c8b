#!/usr/bin/env python3
"""Diagnostic (never timed, never shipped): cycles per phase of k_match_walk, thread 0 of every workgroup.
Needs a stamp build:  hipcc ... -DWALK_STAMPS csrc/orbhip.hip -o build/liborbhip_walk.so.  GPU box only."""
import os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ORBHIP_LIB"] = os.path.join(ROOT, "build", "liborbhip_walk.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("3_orb_slam3_selfnote_amd")
synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
frames, offs = synth.make_stream(1000, 2)
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
_, k0, d0 = ex(frames[0]); _, k1, d1 = ex(frames[1])
sf = ex.GetScaleFactors()
m = pkg.ORBmatcher(0.9, True)
F = pkg.FrameView(k1, d1, (0.0, 752.0, 0.0, 480.0))
lvl = k0["octave"].astype(np.int32)
dbg = torch.zeros((64, 8), dtype=torch.int64, device="cuda")
os.environ["ORBHIP_DBG_PTR"] = str(dbg.data_ptr())
for it in range(3):
    dbg.zero_()
    F.slot[:] = -1; F.slot_obs[:] = 0
    n = m.search_window(F, d0, k0["x"], k0["y"], (15.0 * sf[lvl]).astype(np.float32), lvl - 1, lvl + 1, nnratio=0.9, th_dist=100, use_second=False)[0]
    torch.cuda.synchronize()
d = dbg.cpu().numpy()
names = ["entry+vote", "load+histogram", "prefix sums", "scatter", "walk", "store"]
for wg in np.nonzero(d[:, 0])[0]:
    print("workgroup %2d: " % wg + "  ".join("%s %d" % (a, b) for a, b in zip(names, d[wg, :6])))
print("matches", n)
