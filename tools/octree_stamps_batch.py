#!/usr/bin/env python3
"""Diagnostic (never timed, never shipped): cycles per phase of k_octree in BATCH form (256-thread workgroups, one per frame and level),
thread 0 of every workgroup, averaged over the frames of a level.  Needs a stamp build:
  hipcc <build.py FLAGS> -DOCT_STAMPS csrc/orbhip.hip -o build/liborbhip_oct.so.   GPU box only."""
import ctypes as C, os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ORBHIP_LIB"] = os.path.join(ROOT, "build", "liborbhip_oct.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("3_orb_slam3_selfnote_amd")
synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
B, H, W = 64, 480, 752
frames, _ = synth.make_stream(1000, B)
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
cap = ex.configure(H, W, B)
L = pkg.load()
dev = "cuda"
d_img = torch.from_numpy(frames).to(dev)
d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
d_cnt = torch.zeros((B, 2), dtype=torch.int32, device=dev)
buf = torch.zeros((8 * B, 8), dtype=torch.int32, device=dev)
L.orbx_debug_fast_stamps(C.c_void_p(buf.data_ptr()))
for it in range(3):
    buf.zero_()
    ex.extract_batch_device(d_img.data_ptr(), H, W, W, H * W, B, d_kps.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap, (0, 1000), stream=0)
    torch.cuda.synchronize()
v = buf.cpu().numpy().reshape(8, B, 8).astype(np.float64)
names = ["A gather", "B roots", "C count", "C rank+scans", "C build", "C relabel", "D output"]
for lvl in range(8):
    m = v[lvl].mean(axis=0)
    print("level %d: " % lvl + "  ".join("%s %d" % (a, b) for a, b in zip(names, m[:7])) + "  total %d" % m[:7].sum())
