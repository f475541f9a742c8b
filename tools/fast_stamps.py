#!/usr/bin/env python3
"""Cycle shares of k_fast's sections (diagnostic build -DFAST_STAMPS, ORBHIP_LIB=build/liborbhip_fast.so).  GPU box only."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

pkg = importlib.import_module("3_orb_slam3_selfnote_amd")
synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
B, H, W = 256, 480, 752
frames, _ = synth.make_stream(1000, B)
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
cap = ex.configure(H, W, B)
d_img = torch.from_numpy(frames).to("cuda")
d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
d_cnt = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
L = pkg.load()
nwg = 982 * B
buf = torch.zeros((nwg, 8), dtype=torch.int32, device="cuda")
L.orbx_debug_fast_stamps(C.c_void_p(buf.data_ptr()))
for it in range(3):
    buf.zero_()
    ex.extract_batch_device(d_img.data_ptr(), H, W, W, H * W, B, d_kps.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap, (0, 1000), stream=0)
    torch.cuda.synchronize()
v = buf.cpu().numpy().astype(np.float64)
live = v[:, 0] > 0
names = ["prologue+tile load", "pass1 compass+list", "pass2 score", "pass3 NMS + emit", "(unused)", "(unused)", "(unused)", "cell count"]
m = v[live].mean(axis=0)
for n, x in zip(names, m):
    print("%-22s %6.1f %%   %.0f cycles/workgroup" % (n, 100 * x / m.sum(), x))
print("workgroups with work: %d of %d; total %.0f cycles/workgroup" % (live.sum(), nwg, m.sum()))
