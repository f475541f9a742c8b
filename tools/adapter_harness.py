#!/usr/bin/env python3
"""Builds and runs tools/adapter_harness.c ON THE GPU BOX: writes the synthetic 752x480 stream as a raw file, compiles the harness
with gcc against liborbhip.so (plain C: also the proof that the ABI links from C) and prints its JSON lines (pyramid fill off / on).
    python tools/adapter_harness.py [--frames 320]"""
import argparse, importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=320)
args = ap.parse_args()
synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
importlib.import_module("3_orb_slam3_selfnote_amd").load()      # builds liborbhip.so when the sources are newer
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
frames, offs = synth.make_stream(9000, args.frames)
frames.tofile(os.path.join(out, "harness_frames.raw"))
np.asarray(offs, np.int32).tofile(os.path.join(out, "harness_shifts.raw"))
libdir = os.path.join(ROOT, "3_orb_slam3_selfnote_amd")
exe = os.path.join(ROOT, "build", "adapter_harness")
subprocess.check_call(["gcc", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "adapter_harness.c"), "-o", exe,
                       "-L", libdir, "-lorbhip", "-Wl,-rpath," + libdir, "-lm"])
env = dict(os.environ)
torch_lib = None
try:
    import torch
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
except Exception:
    pass
env["LD_LIBRARY_PATH"] = ":".join(x for x in ("/opt/rocm/lib", torch_lib, env.get("LD_LIBRARY_PATH", "")) if x)
for extra in ([], ["--fill-pyramid"]):
    p = subprocess.run([exe, os.path.join(out, "harness_frames.raw"), str(args.frames), "--shifts", os.path.join(out, "harness_shifts.raw")] + extra,
                       env=env, capture_output=True, text=True)
    sys.stdout.write(p.stdout)
    if p.returncode:
        sys.stderr.write(p.stderr)
        sys.exit(p.returncode)
os.remove(os.path.join(out, "harness_frames.raw"))
