#!/usr/bin/env python3
"""Diagnostic (never timed, never shipped): cycles per phase of k_octree for ONE frame (the single-frame path, 1024-thread workgroups),
thread 0 of each level's workgroup.  Needs a stamp build:  hipcc ... -DOCT_STAMPS csrc/orbhip.hip -o build/liborbhip_oct.so.  GPU box only."""
import ctypes as C, os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ORBHIP_LIB"] = os.path.join(ROOT, "build", "liborbhip_oct.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("3_orb_slam3_selfnote_amd")
synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
frames, _ = synth.make_stream(1000, 2)
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
L = pkg.load()
buf = torch.zeros((64, 8), dtype=torch.int32, device="cuda")
L.orbx_debug_fast_stamps(C.c_void_p(buf.data_ptr()))
for it in range(3):
    buf.zero_()
    ex(frames[1], None, (0, 0))
    torch.cuda.synchronize()
v = buf.cpu().numpy()
names = ["A gather", "B roots", "C count", "C rank+scans", "C build", "C relabel", "D output"]
for lvl in range(8):
    print("level %d: " % lvl + "  ".join("%s %d" % (a, b) for a, b in zip(names, v[lvl, :7])) + "  total %d" % v[lvl, :7].sum())
