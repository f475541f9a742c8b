#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the CPU oracle.

The reference ships no golden vectors and cannot be built here (DESIGN.md "Oracle"), so these fixtures pin the
ORACLE itself (parity unpinned against the real reference): any later change to oracle/ or to the synthetic
generator that alters a result shows up as a fixture mismatch.  Inputs are re-generated from the seed; only
expected outputs are stored (per-level CRC32s, candidate counts, final keypoints + descriptors, match indices).
Run from the repo root:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402

synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
OUT = os.path.dirname(os.path.abspath(__file__))

CASES = [  # name, seed, H, W, nfeatures
    ("euroc_1000", 1000, 480, 752, 1000),
    ("euroc_1001", 1001, 480, 752, 1000),
    ("euroc_1002", 1002, 480, 752, 1000),
    ("tumvi_2000", 2000, 512, 512, 1500),
]


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def bow_nodes(desc, nodes=128):
    """Stand-in for DBoW2's FeatureVector (the vocabulary file is not part of the reference mount): node id = hash of descriptor bits."""
    ids = (desc[:, 0].astype(np.int64) >> 2) * 2 + (desc[:, 7].astype(np.int64) >> 7)
    fv = {}
    for i, n in enumerate(ids % nodes):
        fv.setdefault(int(n) * 7 + 3, []).append(i)
    return fv


def main():
    for name, seed, H, W, nf in CASES:
        img = synth.make_frame(seed, H, W)
        e = O.OracleExtractor(nf, 1.2, 8, 20, 7)
        pyr = e.pyramid(img)
        d = {"seed": seed, "H": H, "W": W, "nfeatures": nf, "image_crc": crc(img)}
        d["pyr_crc"] = np.array([crc(p) for p in pyr], dtype=np.uint32)
        d["blur_crc"] = np.array([crc(O.gaussian_blur7(p)) for p in pyr], dtype=np.uint32)
        cands = [e.level_candidates(p) for p in pyr]
        d["cand_count"] = np.array([len(c) for c in cands], dtype=np.int32)
        d["cand_crc"] = np.array([crc(c) for c in cands], dtype=np.uint32)
        d["level7"] = pyr[7]
        for tag, lap in (("lap1000", (0, 1000)), ("lap0", (0, 0))):
            mono, kps, desc = e.extract(img, lap)
            d["mono_" + tag] = mono
            d["kps_" + tag] = kps
            d["desc_" + tag] = desc
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, "keypoints", len(kps), "candidates", d["cand_count"].tolist())
    # matcher: M2 on a shifted pair, stress 1000x1000, and a claim/tie scenario
    frames, offs = synth.make_stream(3000, 2)
    e = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    _, k0, d0 = e.extract(frames[0])
    _, k1, d1 = e.extract(frames[1])
    sf = e.scale_factors
    u = (k0["x"] + np.float32(offs[0][0] - offs[1][0])).astype(np.float32)
    v = (k0["y"] + np.float32(offs[0][1] - offs[1][1])).astype(np.float32)
    bounds = (0.0, 752.0, 0.0, 480.0)
    F = O.OracleFrame(k1["x"], k1["y"], k1["octave"], k1["angle"], d1, bounds, sf)
    nq = len(k0)
    n_m2, moq_m2 = F.search_by_projection_mp(np.ones(nq, np.uint8), d0, u, v, np.ones(nq, np.float32), k0["octave"], 3.0, 0.8)
    F2 = O.OracleFrame(k1["x"], k1["y"], k1["octave"], k1["angle"], d1, bounds, sf)
    m1 = np.full(nq, -1, np.int32)
    n_st, moq_st, bd_st = F2.search_by_projection_win(d0, u, v, np.full(nq, 1.0e4, np.float32), m1, m1, 0.8, 100, True)
    np.savez_compressed(os.path.join(OUT, "match_3000.npz"), seed=3000, n_m2=n_m2, moq_m2=moq_m2, slot_m2=F.slot,
                        n_stress=n_st, moq_stress=moq_st, bd_stress=bd_st, slot_stress=F2.slot,
                        kps0_crc=crc(k0), kps1_crc=crc(k1))
    print("match: m2", n_m2, "stress", n_st)
    # wider rows on the same pair: SearchForInitialization, SearchByBoW x2, ComputeDistinctiveDescriptors,
    # and ComputeStereoMatches on a 20-px stereo pair
    prev = np.stack([k0["x"], k0["y"]], axis=1).astype(np.float32).copy()
    n_init, m12_init = O.search_for_initialization(k0, d0, O.OracleFrame(k1["x"], k1["y"], k1["octave"], k1["angle"], d1, bounds, sf), prev, 100, 0.9, True)
    sigma2 = (sf * sf).astype(np.float32)
    mp0 = (np.arange(len(k0)) % 5 != 0).astype(np.uint8)
    mp1 = (np.arange(len(k1)) % 4 != 0).astype(np.uint8)
    fv0, fv1 = bow_nodes(d0), bow_nodes(d1)
    K0 = O.OracleKeyFrame(k0, d0, fv0, sf, sigma2, has_mp=mp0)
    K1 = O.OracleKeyFrame(k1, d1, fv1, sf, sigma2, has_mp=mp1)
    n_bow, m_bow = O.search_by_bow(K0, K1, 0.7, True)
    n_bowkk, m_bowkk = O.search_by_bow_keyframes(K0, K1, 0.8, True)
    groups = [d0[i:i + 3 + (i % 9)] for i in range(0, 400, 13)]
    best = np.array([O.distinctive_descriptor(g) for g in groups], np.int32)
    big = synth.make_frame(4120, H=480, W=752 + 64)
    imgL, imgR = np.ascontiguousarray(big[:, 0:752]), np.ascontiguousarray(big[:, 20:20 + 752])
    _, kL, dL = e.extract(imgL, (0, 0))
    _, kR, dR = e.extract(imgR, (0, 0))
    uR, depth = e.compute_stereo_matches(imgL, imgR, kL, dL, kR, dR, 0.11, 47.9)
    np.savez_compressed(os.path.join(OUT, "wider_3000.npz"), n_init=n_init, m12_init=m12_init, prev_crc=crc(prev), n_bow=n_bow, m_bow=m_bow,
                        n_bowkk=n_bowkk, m_bowkk=m_bowkk, best=best, uR=uR, depth=depth, kL_crc=crc(kL), kR_crc=crc(kR))
    print("wider: init", n_init, "bow", n_bow, n_bowkk, "stereo", int((uR >= 0).sum()))


if __name__ == "__main__":
    main()
