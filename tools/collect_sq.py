#!/usr/bin/env python3
"""Per-kernel SQ counters with rocprofv3 (run ON THE GPU BOX, from the repo root):  python tools/collect_sq.py [pass ...]

Each pass is a comma-separated counter list that fits the 8 SQ slots; passes run as separate profiles (--pmc with
--kernel-trace only).  Prints per-launch averages per kernel and writes gpurun_out/sq_counters.json.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md).
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
DEFAULT = ["SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_SCA,SQ_ACTIVE_INST_LDS",
           "SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_WAIT_INST_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_ACTIVE_INST_ANY,SQ_INSTS_SMEM,SQ_WAVES"]

DRIVER = r'''
import sys
sys.path.insert(0, %r)
import torch
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--batch", "256", "--streams", "1", "--batches-per-step", "4", "--groups", "2", "--no-cpu-baseline", "--no-host-fed"] + %r
exec(open(%r).read())
'''


def run_pass(i, counters, extra):
    d = os.path.join(OUT, "sq_pass%d" % i)
    shutil.rmtree(d, ignore_errors=True)          # a stale run directory would be picked up by the glob below
    drv = os.path.join(OUT, "sq_driver.py")
    open(drv, "w").write(DRIVER % (ROOT, extra, os.path.join(ROOT, "bench.py")))
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.check_call(["rocprofv3", "--kernel-trace", "--pmc"] + counters.split(",") + ["--output-format", "csv", "-d", d, "--", sys.executable, drv],
                          cwd="/tmp", env=env, stdout=open(os.path.join(OUT, "sq_pass%d.log" % i), "w"), stderr=subprocess.STDOUT)
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    os.makedirs(OUT, exist_ok=True)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    extra = [a for a in sys.argv[1:] if a.startswith("--")]
    passes = args or DEFAULT
    result = collections.defaultdict(dict)
    for i, c in enumerate(passes):
        for k, d in run_pass(i, c, extra).items():
            for name, vals in d.items():
                result[k][name] = sum(vals) / len(vals)
                result[k]["launches"] = len(vals)
    json.dump(result, open(os.path.join(OUT, "sq_counters.json"), "w"), indent=1, sort_keys=True)
    for k in sorted(result):
        print(k)
        for name in sorted(result[k]):
            print("   %-24s %16.0f" % (name, result[k][name]))


if __name__ == "__main__":
    main()
