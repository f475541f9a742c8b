#!/usr/bin/env python3
"""Diagnostic (GPU box): the candidate scan of BASELINE's 1000x1000 setting alone on the chip, 256 frame pairs per launch,
matrix-pipe engine against the vector-ALU scan.  Prints the HIP-event stage times (scan = walk vote + rank + scan kernels,
resolve) per engine and checks that both engines leave the same matches.   python tools/scan_bench.py [--reps 20] [--engines 1,0]
Under rocprofv3 --kernel-trace --stats --output-format csv the per-kernel durations are the ones to read."""
import argparse, os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--engines", default="2,1,0")
ap.add_argument("--batch", type=int, default=256)
args = ap.parse_args()
pkg = importlib.import_module("3_orb_slam3_selfnote_amd")
synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
C = pkg.C
B, H, W = args.batch, 480, 752
frames, offs = synth.make_stream(1000, B + 1)
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
cap = ex.configure(H, W, B + 1)
dev = "cuda"
d_img = torch.from_numpy(frames).to(dev)
d_kps = torch.zeros((B + 1, cap, 7), dtype=torch.float32, device=dev)
d_desc = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device=dev)
d_cnt = torch.zeros((B + 1, 2), dtype=torch.int32, device=dev)
ex.extract_batch_device(d_img.data_ptr(), H, W, W, H * W, B + 1, d_kps.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap, (0, 1000), stream=0)
torch.cuda.synchronize()
sh = torch.from_numpy(np.array([(offs[p][0] - offs[p + 1][0], offs[p][1] - offs[p + 1][1]) for p in range(B)], np.float32)).to(dev)
u = (d_kps[:B, :, 0] + sh[:, 0:1]).contiguous(); v = (d_kps[:B, :, 1] + sh[:, 1:2]).contiguous()
rad = torch.full((B, cap), 1.0e4, dtype=torch.float32, device=dev); lvl = torch.full((B, cap), -1, dtype=torch.int32, device=dev)
slot = torch.full((B, cap), -1, dtype=torch.int32, device=dev); sobs = torch.zeros((B, cap), dtype=torch.uint8, device=dev)
moq = torch.empty((B, cap), dtype=torch.int32, device=dev); nm = torch.zeros((B,), dtype=torch.int32, device=dev)
fs = pkg.FrameStruct(cap, d_kps[1:].data_ptr(), d_desc[1:].data_ptr(), None, 0.0, float(W), 0.0, float(H))
qs = pkg.QueryStruct(cap, d_desc.data_ptr(), u.data_ptr(), v.data_ptr(), rad.data_ptr(), lvl.data_ptr(), lvl.data_ptr(), None, None)
res = {}
for eng in [int(e) for e in args.engines.split(",")]:
    mt = pkg.ORBmatcher(0.8, True)
    mt.set_hamming_engine(eng)
    for it in range(args.reps + 2):
        if it == 2:
            mt.set_profiling(True)
        slot.fill_(-1); sobs.zero_()
        rc = mt.L.orbm_search_by_projection_batch_device(mt.m, C.byref(fs), cap, C.c_void_p(d_cnt[1:].data_ptr()), 2, C.byref(qs), cap, C.c_void_p(d_cnt.data_ptr()), 2, B,
                                                         C.c_float(0.8), 100, 1, C.c_void_p(slot.data_ptr()), C.c_void_p(sobs.data_ptr()), C.c_void_p(moq.data_ptr()), None,
                                                         C.c_void_p(nm.data_ptr()), None)
        assert rc == 0
        torch.cuda.synchronize()
    st = mt.stage_ms()
    res[eng] = (moq.cpu().numpy().copy(), nm.cpu().numpy().copy())
    print("engine %d: scan %.4f ms  resolve %.4f ms  (mean matches %.1f)" % (eng, st["match_scan"], st["match_resolve"], res[eng][1].mean()), flush=True)
    mt.close()
if len(res) >= 2:
    vals = list(res.values())
    cnt = d_cnt.cpu().numpy()
    same = all(all(np.array_equal(vals[0][0][p, :cnt[p, 0]], b[0][p, :cnt[p, 0]]) for p in range(B)) and np.array_equal(vals[0][1], b[1]) for b in vals[1:])
    print("engines agree:", same)
    sys.exit(0 if same else 1)
