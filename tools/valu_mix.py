#!/usr/bin/env python3
"""Vector-issue ceiling of each product kernel for ITS OWN opcode mix -> profiles/valu_mix.json (runs on the CPU: hipcc -S).

The per-class issue rates are measured (profiles/valu_calib.json, tools/calib/orb_calib.h): on gfx950 a handful of simple opcodes
(v_add/sub_u32, v_and/or/xor_b32, v_lshrrev_b32, v_mov_b32, v_add/mul/fma_f32, v_min/max_u16, v_bitop3_b32) issue a wave64
instruction every 2 cycles per SIMD - unless one of their sources is a scalar register, then every 4 -, nearly everything else (min/max, compares, selects, packed 16-bit, dot, perm, bcnt, mul24/mad, three-operand
integer forms, shifts left, conversions) every 4 - and the hardware counters do not tell the two apart (SQ_ACTIVE_INST_VALU ==
SQ_INSTS_VALU for both, tools/check_valu_counters.py).  So the mix is taken from the code object: every VALU instruction of a
kernel is weighted by 10^(loop depth of its basic block) (LLVM's loop annotations in the assembly) and priced with the measured
rate of its opcode class; opcodes without a measurement are priced at the half rate and their share is reported.  The kernel's
ceiling is the weighted harmonic mean.  This is an estimate of the dynamic mix, not a measurement; both bounds are printed."""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
         "-mllvm", "-amdgpu-mfma-vgpr-form", "--cuda-device-only", "-S"]
KERNELS = {"k_resize": r"^_Z8k_resize", "k_fast": r"^_Z6k_fast", "k_octree": r"^_Z8k_octreeILi256ELb1E", "k_blur": r"^_Z6k_blur",
           "k_describe": r"^_Z10k_describe", "k_match_scan": r"^_Z12k_match_scanI5Key32Li0E", "k_match_walk": r"^_Z12k_match_walkI5Key32Li0E", "k_match_resolve": r"^_Z15k_match_resolveI5Key32Lb1ELb1EE", "k_match_scan_mfma": r"^_Z17k_match_scan_mfma", "k_match_rank": r"^_Z12k_match_rank",
           "k_lastframe_project": r"^_Z19k_lastframe_project", "k_rot_prune": r"^_Z11k_rot_prune"}
# opcode (suffix-stripped) -> name of the calibration class that measured it
CLASS_OF = {
    "v_add_u32": "v_add_u32", "v_sub_u32": "v_sub_u32", "v_subrev_u32": "v_sub_u32", "v_xor_b32": "v_xor_b32", "v_and_b32": "v_and_b32",
    "v_or_b32": "v_or_b32", "v_lshrrev_b32": "v_lshrrev_b32", "v_lshlrev_b32": "v_lshlrev_b32", "v_mov_b32": "v_mov_b32", "v_fma_f32": "v_fma_f32",
    "v_fmac_f32": "v_fma_f32", "v_add_f32": "v_add_f32", "v_sub_f32": "v_sub_f32", "v_subrev_f32": "v_sub_f32", "v_mul_f32": "v_mul_f32",
    "v_bitop3_b32": "v_bitop3_b32", "v_min_u32": "v_min_u32/v_max_u32", "v_max_u32": "v_min_u32/v_max_u32", "v_min_i32": "v_min_i32/v_max_i32",
    "v_max_i32": "v_min_i32/v_max_i32", "v_pk_min_i16": "v_pk_min_i16/v_pk_max_i16", "v_pk_max_i16": "v_pk_min_i16/v_pk_max_i16",
    "v_pk_mad_u16": "v_pk_mad_u16", "v_pk_add_u16": "v_pk_add_u16", "v_pk_sub_i16": "v_pk_sub_i16", "v_bcnt_u32_b32": "v_bcnt_u32_b32",
    "v_dot4_u32_u8": "v_dot4_u32_u8", "v_dot2_u32_u16": "v_dot2_u32_u16", "v_perm_b32": "v_perm_b32", "v_alignbyte_b32": "v_alignbyte_b32",
    "v_alignbit_b32": "v_alignbyte_b32", "v_mul_u32_u24": "v_mul_u32_u24", "v_mul_i32_i24": "v_mul_u32_u24", "v_mad_u32_u24": "v_mad_u32_u24",
    "v_mad_i32_i24": "v_mad_u32_u24", "v_mul_lo_u32": "v_mul_lo_u32", "v_lshl_add_u32": "v_lshl_add_u32", "v_lshl_add_u64": "v_lshl_add_u64",
    "v_cndmask_b32": "v_cndmask_b32 (SGPR-pair mask)", "v_add3_u32": "v_add3_u32", "v_and_or_b32": "v_and_or_b32", "v_lshl_or_b32": "v_lshl_or_b32",
    "v_bfe_u32": "v_bfe_u32", "v_bfe_i32": "v_bfe_u32", "v_min3_u32": "v_min3_u32", "v_med3_i32": "v_med3_i32", "v_sad_u8": "v_sad_u8",
    "v_cvt_f32_u32": "v_cvt_f32_u32", "v_cvt_f32_i32": "v_cvt_f32_u32", "v_add_co_u32": "v_add_co_u32", "v_fma_f64": "v_fma_f64",
    "v_fmac_f64": "v_fma_f64", "v_mul_f64": "v_fma_f64", "v_add_f64": "v_fma_f64", "v_min_u16": "v_min_u16/v_max_u16", "v_max_u16": "v_min_u16/v_max_u16",
}
CMP = re.compile(r"^v_cmpx?_")
SUFFIX = re.compile(r"(_e32|_e64|_sdwa|_dpp|_e64_dpp)$")
SGPR_SRC = re.compile(r"(?<![\w.])(s\d+|s\[\d+:\d+\]|vcc(_lo|_hi)?|exec(_lo|_hi)?|m0)(?![\w])")


def assembly():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "orbhip.s")
        subprocess.check_call([HIPCC] + FLAGS + ["-o", out, os.path.join(ROOT, "3_orb_slam3_selfnote_amd", "csrc", "orbhip.hip")],
                              stderr=subprocess.DEVNULL)
        return open(out).read().splitlines()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calib", default=os.path.join(ROOT, "profiles", "valu_calib.json"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "valu_mix.json"))
    args = ap.parse_args()
    calib = json.load(open(args.calib))["ops"]
    rate = {k: v["ceiling_G_wave_instr_per_s"] for k, v in calib.items()}
    half = rate["v_pk_min_i16/v_pk_max_i16"]
    full = rate["v_add_u32"]
    lines = assembly()
    res = {"weighting": "10^(loop depth) per static VALU instruction", "half_rate_G": half, "full_rate_G": full, "kernels": {}}
    label = re.compile(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)")
    for kname, pat in KERNELS.items():
        start = next((i for i, l in enumerate(lines) if re.match(pat, l) and l.rstrip().split(";")[0].strip().endswith(":")), None)
        if start is None:
            continue
        depth, w_by_class, n_static = 0, collections.Counter(), 0
        unknown = collections.Counter()
        for l in lines[start + 1:]:
            if "s_endpgm" in l:
                break
            if label.match(l):
                m = re.search(r"Depth[= ](\d+)", l)
                depth = int(m.group(1)) if m else 0
                continue
            t = l.strip().split()
            if not t or not t[0].startswith("v_"):
                continue
            op = t[0]
            base = op
            while SUFFIX.search(base):
                base = SUFFIX.sub("", base)
            w = 10.0 ** depth
            n_static += 1
            if CMP.match(base):
                cls = "v_cmp_lt_u32"
            elif base == "v_cndmask_b32" and op.endswith("_e32"):
                cls = "v_cndmask_b32"            # reads VCC: the slow form (measured 16 cycles)
            else:
                cls = CLASS_OF.get(base)
            # a full-rate opcode that reads a scalar register issues at the half rate (measured: "v_xor_b32 (SGPR source)")
            if cls in rate and rate[cls] > 1.3 * half and SGPR_SRC.search(" ".join(t[2:]).split(";")[0]):
                cls = "v_xor_b32 (SGPR source)" if "v_xor_b32 (SGPR source)" in rate else "v_pk_min_i16/v_pk_max_i16"
            if cls is None or cls not in rate:
                unknown[base] += w
                cls = None
            w_by_class[cls] += w
        tot = sum(w_by_class.values())
        t_issue = sum(w / (rate[c] if c else half) for c, w in w_by_class.items())
        shares = {(c or "unmeasured (priced at the half rate)"): round(w / tot, 4) for c, w in w_by_class.most_common(12)}
        res["kernels"][kname] = {"static_valu_instructions": n_static, "ceiling_G_wave_instr_per_s": round(tot / t_issue, 1),
                                 "weighted_share_by_class": shares,
                                 "unmeasured_opcodes": {k: round(v / tot, 4) for k, v in unknown.most_common(8)}}
        print("%-20s %5d static VALU, ceiling %7.1f G wave-instr/s (half rate %.0f, full rate %.0f); unmeasured share %.3f" % (
            kname, n_static, tot / t_issue, half, full, sum(unknown.values()) / tot))
    json.dump(res, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
