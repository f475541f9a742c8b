#!/usr/bin/env python3
"""What do SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU say about an instruction's issue cost?  (run ON THE GPU BOX, from the repo root)

Runs the calibration kernels of a full-rate class (v_add_u32: 2 cycles per wave64 instruction when several waves share a SIMD)
and of half-rate classes (v_pk_min_i16 / v_pk_max_i16, v_bcnt_u32_b32, v_cndmask_b32: 4 cycles) under rocprofv3 --pmc and prints
SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU per kernel.  If the quotient tracks the issue cost, the same quotient of the product's
kernels (profiles/sq_counters.json) gives each kernel's dynamic average issue cost, i.e. the vector-issue ceiling of ITS OWN
instruction mix; if it is 1.0 for every class the counter only counts issue events.  Writes gpurun_out/valu_counter_check.json."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
OPS = "0,3,5,15,22,39"   # v_add_u32, v_pk_min/max_i16, v_bcnt, v_cndmask(vcc), k_fast mix, v_cndmask (SGPR pair)


def main():
    d = os.path.join(OUT, "valu_counter_check")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
           "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "tools", "collect_valu_calib.py"), "--ops", OPS, "--waves", "4",
           "--out", os.path.join(OUT, "valu_calib_under_pmc.json")]
    subprocess.check_call(cmd, cwd="/tmp", env=env, stdout=open(os.path.join(OUT, "valu_counter_check.log"), "w"), stderr=subprocess.STDOUT)
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, c in sorted(acc.items()):
        m = {n: sum(v) / len(v) for n, v in c.items()}
        if m.get("SQ_INSTS_VALU"):
            m["quad_cycles_per_valu_instr"] = m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"]
        res[k] = m
        print("%-44s INSTS_VALU %14.0f  ACTIVE_INST_VALU %14.0f  ratio %.3f  WAIT_INST_ANY/WAVE_CYCLES %.3f" % (
            k[:44], m.get("SQ_INSTS_VALU", 0), m.get("SQ_ACTIVE_INST_VALU", 0), m.get("quad_cycles_per_valu_instr", 0),
            m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
    json.dump(res, open(os.path.join(OUT, "valu_counter_check.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
