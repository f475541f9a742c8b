#!/usr/bin/env python3
"""Diagnostic (never timed, never shipped): where does the WIDE form of k_match_resolve (fused list build included) spend its cycles?
Needs a stamp build:  hipcc ... -DRESOLVE_STAMPS csrc/orbhip.hip -o build/liborbhip_wide.so   (see DESIGN.md)."""
import os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ORBHIP_LIB"] = os.path.join(ROOT, "build", "liborbhip_wide.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("3_orb_slam3_selfnote_amd")
synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
C = pkg.C
B, H, W = 64, 480, 752
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
shift = np.array([[offs[p][0] - offs[p + 1][0], offs[p][1] - offs[p + 1][1]] for p in range(B)], dtype=np.float32)
d_shift = torch.from_numpy(shift).to(dev)
u = (d_kps[:B, :, 0] + d_shift[:, 0:1]).contiguous(); v = (d_kps[:B, :, 1] + d_shift[:, 1:2]).contiguous()
rad = torch.full((B, cap), 1.0e4, dtype=torch.float32, device=dev); lvl = torch.full((B, cap), -1, dtype=torch.int32, device=dev)
lvl_hi = lvl
if "--tracking" in sys.argv:   # windows of a tracking search: radius 15 * scale factor of the query's level, levels +-1
    sf = torch.tensor(ex.GetScaleFactors(), dtype=torch.float32, device=dev)
    octv = d_kps[:B, :, 5].contiguous().view(torch.int32).clamp(0, 7)
    rad = (15.0 * sf[octv.long()]).contiguous(); lvl = (octv - 1).contiguous(); lvl_hi = (octv + 1).contiguous()
slot = torch.full((B, cap), -1, dtype=torch.int32, device=dev); sobs = torch.zeros((B, cap), dtype=torch.uint8, device=dev)
moq = torch.empty((B, cap), dtype=torch.int32, device=dev); nm = torch.zeros((B,), dtype=torch.int32, device=dev)
dbg = torch.zeros((B, 16), dtype=torch.int64, device=dev)
os.environ["ORBHIP_DBG_PTR"] = str(dbg.data_ptr())
fs = pkg.FrameStruct(cap, d_kps[1:].data_ptr(), d_desc[1:].data_ptr(), None, 0.0, float(W), 0.0, float(H))
qs = pkg.QueryStruct(cap, d_desc.data_ptr(), u.data_ptr(), v.data_ptr(), rad.data_ptr(), lvl.data_ptr(), lvl_hi.data_ptr(), None, None)
rc = mt.L.orbm_search_by_projection_batch_device(mt.m, C.byref(fs), cap, C.c_void_p(d_cnt[1:].data_ptr()), 2, C.byref(qs), cap, C.c_void_p(d_cnt.data_ptr()), 2, B,
                                                 C.c_float(0.8), 100, 1, C.c_void_p(slot.data_ptr()), C.c_void_p(sobs.data_ptr()), C.c_void_p(moq.data_ptr()), None,
                                                 C.c_void_p(nm.data_ptr()), None)
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(np.float64)
names = ["list build (+ chunk set-up)", "rounds + commits", "refresh passes", "rounds", "refresh passes (count)", "prefix cuts", "total", "nq"]
m = d.mean(axis=0)
for n, x in zip(names, m): print("%-28s %12.0f" % (n, x))
