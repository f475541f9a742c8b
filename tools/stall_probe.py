#!/usr/bin/env python3
"""Diagnostic (GPU box): the single-frame matcher call repeated on the same data, C-call time per iteration.
Round 2 saw one call in ~100-250 take 36-45 ms through the Python mirror.  Findings with this probe (profiles/README.md):
  * no HIP API call of that length in `rocprofv3 --hip-trace`, no buffer growth (ORBHIP_TRACE_ALLOC=1: every re-allocation < 0.02 ms);
  * gone with --no-torch (fewer Python objects) and gone with --gc-off: it was CPython's full (generation-2) garbage collection,
    started by the ctypes argument objects that the mirror allocated INSIDE its timed region; with PyTorch imported a full
    collection walks a few hundred thousand objects (~40 ms).  The C ABI itself never stalls (tools/adapter_harness.c: max 0.27 ms
    over 380 frames); the mirror now makes its arguments before it starts the clock.
python tools/stall_probe.py [--calls 300] [--no-torch] [--gc-off]"""
import argparse, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--calls", type=int, default=300)
ap.add_argument("--no-torch", action="store_true")
ap.add_argument("--gc-off", action="store_true")
args = ap.parse_args()
if args.no_torch:
    sys.modules["torch"] = None          # the package then loads the system HIP runtime instead of the wheel's
pkg = importlib.import_module("3_orb_slam3_selfnote_amd")
synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
frames, offs = synth.make_stream(77, 2)
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
(_, k0, d0), (_, k1, d1) = ex(frames[0]), ex(frames[1])
sf = np.asarray(ex.GetScaleFactors(), np.float32)
mt = pkg.ORBmatcher(0.9, True)
F = pkg.FrameView(k1, d1, (0.0, 752.0, 0.0, 480.0))
lvl = k0["octave"].astype(np.int32)
u = (k0["x"] + np.float32(offs[0][0] - offs[1][0])).astype(np.float32); v = (k0["y"] + np.float32(offs[0][1] - offs[1][1])).astype(np.float32)
if args.gc_off:
    import gc
    gc.disable()
ts = []
for i in range(args.calls):
    F.slot[:] = -1; F.slot_obs[:] = 0
    mt.search_window(F, d0, u, v, (15.0 * sf[lvl]).astype(np.float32), lvl - 1, lvl + 1, nnratio=0.9, th_dist=100, use_second=False)
    ts.append(mt.last_call_s * 1e3)
ts = np.array(ts)
print("calls %d: median %.3f ms, p99 %.3f, max %.3f at call %d; calls > 2 ms: %s" % (len(ts), np.median(ts), np.percentile(ts, 99), ts.max(), ts.argmax(), np.nonzero(ts > 2.0)[0].tolist()))
