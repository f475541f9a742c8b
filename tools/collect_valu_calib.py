#!/usr/bin/env python3
"""Measure the chip's vector-issue ceiling per opcode class (orbx_calibration_valu, tools/calib/orb_calib.h) and write
profiles/valu_calib.json -- the `peak` of bench.py's `valu_issue` object comes from this file, not from an assumed
cycles-per-instruction figure.

    python tools/collect_valu_calib.py [--out profiles/valu_calib.json] [--trips 2000]

Per class and per residency (1 / 2 / 4 / 8 wavefronts per SIMD on every CU): wave-instructions per second of the whole
chip (HIP events around the launch), shader cycles one wave-instruction occupies its SIMD (s_memtime in the kernel, median
over workgroups) and the shader clock held during the run."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "valu_calib.json"))
    ap.add_argument("--trips", type=int, default=2000)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--ops", default="", help="comma-separated opcode-class indices (default: all)")
    ap.add_argument("--waves", default="1,2,4,8", help="resident wavefronts per SIMD to sweep")
    args = ap.parse_args()
    import importlib.util
    spec = importlib.util.spec_from_file_location("orbcalib", os.path.join(ROOT, "tools", "calib", "calib.py"))
    calib = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(calib)
    L = calib.load()
    nops = L.orbx_calibration_valu_ops()
    out = {"unit": "G wave-instructions/s (whole chip)", "trips": args.trips, "instructions_per_trip": 128,
           "note": "independent instruction streams, 16 chains per lane; residency fixed by LDS; see tools/calib/orb_calib.h", "ops": {}}
    ops = [int(x) for x in args.ops.split(",")] if args.ops else list(range(nops))
    for op in ops:
        name = L.orbx_calibration_valu_name(op).decode()
        rows = {}
        for w in [int(x) for x in args.waves.split(",")]:
            rate, cyc, clk = C.c_double(), C.c_double(), C.c_double()
            rc = L.orbx_calibration_valu(args.device, op, w, args.trips, C.byref(rate), C.byref(cyc), C.byref(clk))
            if rc != 0:
                raise SystemExit("orbx_calibration_valu(op=%d, w=%d) rc=%d" % (op, w, rc))
            rows[str(w)] = {"G_wave_instr_per_s": round(rate.value / 1e9, 2), "cycles_per_wave_instr_per_simd": round(cyc.value, 3),
                            "clock_GHz": round(clk.value, 3)}
        best = max(rows.values(), key=lambda r: r["G_wave_instr_per_s"])
        out["ops"][name] = {"waves_per_simd": rows, "ceiling_G_wave_instr_per_s": best["G_wave_instr_per_s"],
                            "cycles_at_ceiling": best["cycles_per_wave_instr_per_simd"]}
        print("%-52s" % name, "  ".join("w%s %7.1f G/s %5.2f cyc %4.2f GHz" % (w, r["G_wave_instr_per_s"], r["cycles_per_wave_instr_per_simd"],
                                                                               r["clock_GHz"]) for w, r in rows.items()), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
