#!/usr/bin/env python3
"""PCIe-inclusive latency/throughput of the drop-in single-frame host entry point (orbx_extract = ORBextractor::operator()):
361 KB H2D + kernels + 60 KB D2H + one stream sync per call.  Not bench.py's `value`."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("3_orb_slam3_selfnote_amd")
synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
frames, _ = synth.make_stream(1000, 32)
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
for f in frames[:8]:
    ex(f)
t0 = time.perf_counter(); n = 0
for rep in range(10):
    for f in frames:
        ex(f); n += 1
dt = time.perf_counter() - t0
print("orbx_extract host API: %.3f ms/frame, %.0f frames/s (PCIe + sync inclusive, 1 frame in flight)" % (dt / n * 1e3, n / dt))
