#!/usr/bin/env python3
"""Collect HBM traffic per kernel launch with rocprofv3 PMC counters (run ON THE GPU BOX, from the repo root).

Follows /opt/skills/guides/MI355X_MICROARCH.md "HBM" + "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one
pass (TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2), so they are collected in SEPARATE runs with --pmc only
(no --sys-trace / runtime trace).  Units: the counters are in KiB.  On gfx950 FETCH_SIZE under-reports wide coalesced
reads by 2x and other widths are uncalibrated, so a known-traffic kernel in the product kernels' 4-byte-per-lane access
pattern (tools/calib, orbx_calibration_copy: 256 MiB read + 256 MiB written) runs in the same profile; its measured/known ratio is
the correction applied to every kernel (reported alongside the raw numbers).

Writes gpurun_out/traffic.json:  {kernel: {"launches", "fetch_raw_B", "write_raw_B", "fetch_B", "write_B", "hbm_B"}}
(per-launch averages, batch 256) plus the calibration factors.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
CALIB_BYTES = 256 << 20

DRIVER = r'''
import importlib, sys, ctypes as C
sys.path.insert(0, %r)
import torch
import importlib.util, os
spec = importlib.util.spec_from_file_location("orbcalib", os.path.join(%r, "tools", "calib", "calib.py"))
calib = importlib.util.module_from_spec(spec)
spec.loader.exec_module(calib)
L = calib.load()
a = torch.randint(0, 255, (%d,), dtype=torch.uint8, device="cuda")
b = torch.empty_like(a)
torch.cuda.synchronize()
for _ in range(3):
    L.orbx_calibration_copy(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_size_t(a.numel()), None)
torch.cuda.synchronize()
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--batch", "256", "--streams", "1", "--batches-per-step", "4", "--groups", "2", "--no-cpu-baseline", "--no-host-fed"]
exec(open(%r).read())
''' % (ROOT, ROOT, CALIB_BYTES, os.path.join(ROOT, "bench.py"))


def run_pass(counter):
    d = os.path.join(OUT, "pmc_" + counter.lower())
    drv = os.path.join(OUT, "traffic_driver.py")
    open(drv, "w").write(DRIVER)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.check_call(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, drv],
                          cwd="/tmp", env=env, stdout=open(os.path.join(OUT, "pmc_%s.log" % counter.lower()), "w"), stderr=subprocess.STDOUT)
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    os.makedirs(OUT, exist_ok=True)
    fetch = run_pass("FETCH_SIZE")
    write = run_pass("WRITE_SIZE")
    def avg(d, k):
        v = d.get(k, [])
        return sum(v) / len(v) * 1024.0 if v else 0.0   # KiB -> bytes
    calib_name = [k for k in fetch if "k_calib_copy" in k][0]
    cf = CALIB_BYTES / max(avg(fetch, calib_name), 1.0)
    cw = CALIB_BYTES / max(avg(write, calib_name), 1.0)
    out = {"_calibration": {"kernel": calib_name, "known_read_B": CALIB_BYTES, "known_write_B": CALIB_BYTES,
                            "fetch_raw_B": avg(fetch, calib_name), "write_raw_B": avg(write, calib_name),
                            "fetch_factor": cf, "write_factor": cw}}
    for k in sorted(set(fetch) | set(write)):
        if not (k.startswith("k_") or "k_match" in k or "k_octree" in k):
            continue
        fr, wr = avg(fetch, k), avg(write, k)
        out[k.split("(")[0].replace("void ", "")] = {"launches": len(fetch.get(k, [])), "fetch_raw_B": fr, "write_raw_B": wr,
                                                      "fetch_B": fr * cf, "write_B": wr * cw, "hbm_B": fr * cf + wr * cw}
    json.dump(out, open(os.path.join(OUT, "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
