#!/usr/bin/env python3
"""Diagnostic (never timed, never shipped): k_match_scan_mfma's cycles per phase (staging / MFMA / selection / barrier), thread 0 of every workgroup, BASELINE's 1000x1000 setting on 256 frame pairs.
Needs a stamp build:  hipcc ... -DMF_STAMPS csrc/orbhip.hip -o build/liborbhip_mf.so.  GPU box only."""
import os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ORBHIP_LIB"] = os.path.join(ROOT, "build", "liborbhip_mf.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("3_orb_slam3_selfnote_amd")
synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
C = pkg.C
B, H, W = 256, 480, 752
frames, offs = synth.make_stream(1000, B + 1)
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7); mt = pkg.ORBmatcher(0.8, True)
cap = ex.configure(H, W, B + 1)
dev = "cuda"
d_img = torch.from_numpy(frames).to(dev)
d_kps = torch.zeros((B + 1, cap, 7), dtype=torch.float32, device=dev)
d_desc = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device=dev)
d_cnt = torch.zeros((B + 1, 2), dtype=torch.int32, device=dev)
ex.extract_batch_device(d_img.data_ptr(), H, W, W, H * W, B + 1, d_kps.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap, (0, 1000), stream=0)
torch.cuda.synchronize()
u = d_kps[:B, :, 0].contiguous(); v = d_kps[:B, :, 1].contiguous()
rad = torch.full((B, cap), 1.0e4, dtype=torch.float32, device=dev); lvl = torch.full((B, cap), -1, dtype=torch.int32, device=dev)
slot = torch.full((B, cap), -1, dtype=torch.int32, device=dev); sobs = torch.zeros((B, cap), dtype=torch.uint8, device=dev)
moq = torch.empty((B, cap), dtype=torch.int32, device=dev); nm = torch.zeros((B,), dtype=torch.int32, device=dev)
nwg = 8 * ((cap + 255) // 256) * ((B + 7) // 8)
dbg = torch.zeros((nwg, 8), dtype=torch.int64, device=dev)
os.environ["ORBHIP_DBG_PTR"] = str(dbg.data_ptr())
fs = pkg.FrameStruct(cap, d_kps[1:].data_ptr(), d_desc[1:].data_ptr(), None, 0.0, float(W), 0.0, float(H))
qs = pkg.QueryStruct(cap, d_desc.data_ptr(), u.data_ptr(), v.data_ptr(), rad.data_ptr(), lvl.data_ptr(), lvl.data_ptr(), None, None)
for it in range(2):
    slot.fill_(-1); sobs.zero_(); dbg.zero_()
    rc = mt.L.orbm_search_by_projection_batch_device(mt.m, C.byref(fs), cap, C.c_void_p(d_cnt[1:].data_ptr()), 2, C.byref(qs), cap, C.c_void_p(d_cnt.data_ptr()), 2, B,
                                                     C.c_float(0.8), 100, 1, C.c_void_p(slot.data_ptr()), C.c_void_p(sobs.data_ptr()), C.c_void_p(moq.data_ptr()), None,
                                                     C.c_void_p(nm.data_ptr()), None)
    torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(np.float64)
live = d[:, 4] > 0
print("workgroups %d; per workgroup: staging %.0f, mfma %.0f, selection %.0f, barrier %.0f, total %.0f cycles" % (live.sum(), d[live, 0].mean(), d[live, 1].mean(), d[live, 2].mean(), d[live, 3].mean(), d[live, 4].mean()))
t0 = d[live, 5]
print("kernel span (first start to last end): %.0f cycles" % ((t0 + d[live, 4]).max() - t0.min()))
