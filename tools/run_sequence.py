#!/usr/bin/env python3
"""Sequence driver modelled on Examples/Monocular/mono_euroc.cc (timestamp list -> image -> front end -> timing statistics),
restricted to the path this repository implements: per frame ORBextractor::operator() and a constant-velocity
SearchByProjection against the previous frame (what Tracking::TrackWithMotionModel does with the extractor's output).

    python tools/run_sequence.py                       # synthetic stream (seeded), 200 frames
    python tools/run_sequence.py --euroc DIR --times Examples/Monocular/EuRoC_TimeStamps/MH03.txt
                                                       # DIR/mav0/cam0/data/<timestamp>.png, as mono_euroc.cc:66 loads them

Prints what mono_euroc.cc prints at the end (median / mean tracking time, :183-190) for the front-end part and writes a CSV
`frame, n_keypoints, n_matches, ORB_Ext(ms), Match(ms), ORB_Ext_C(ms), Match_C(ms)` (the last two: the C call alone, as the
reference's own `ORB_Ext` timer sees its C++ call, Frame.cc:333-343).  Everything runs through the host-pointer C ABI (the drop-in
path of the adapter: one image in, keypoints + descriptors out), so the numbers are PCIe- and sync-inclusive.
GPU box only; PNG decoding needs Pillow (datasets are not part of this repository)."""
import argparse
import csv
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_frames(args, synth):
    if args.euroc:
        from PIL import Image  # noqa: PLC0415
        with open(args.times) as f:
            stamps = [ln.strip() for ln in f if ln.strip()]
        for ts in stamps[: args.frames]:
            path = os.path.join(args.euroc, "mav0", "cam0", "data", ts + ".png")
            im = np.asarray(Image.open(path))
            yield ts, im
    else:
        frames, _ = synth.make_stream(args.seed, args.frames)
        for i, im in enumerate(frames):
            yield str(i), im


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--euroc", default=os.environ.get("ORB_EUROC_DIR"))
    ap.add_argument("--times", default=None)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--clahe", action="store_true", help="apply createCLAHE(3.0, Size(8, 8)) to every image first, as mono_tum_vi.cc:101-109 does")
    ap.add_argument("--csv", default="gpurun_out/sequence.csv")
    args = ap.parse_args()
    if args.euroc and not args.times:
        ap.error("--euroc needs --times (a timestamp list as in Examples/Monocular/EuRoC_TimeStamps)")
    pkg = importlib.import_module("3_orb_slam3_selfnote_amd")
    synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)          # EuRoC.yaml ORBextractor.* (Examples/Monocular/EuRoC.yaml:34-47)
    mt = pkg.ORBmatcher(0.9, True)                      # Tracking.cc:2872
    sf = ex.GetScaleFactors()
    rows, prev = [], None
    for name, im in load_frames(args, synth):
        if im.ndim == 3:
            im = ex.cvtColorGray(im, rgb=True)          # Tracking.cc:1122-1135
        if args.clahe:
            im = ex.CLAHE(im, 3.0, (8, 8))              # mono_tum_vi.cc:101-109
        t0 = time.perf_counter()
        _, kps, desc = ex(im, None, (0, 0))
        t1 = time.perf_counter()
        c_ext, c_mat = ex.last_call_s, 0.0
        nmatch = 0
        if prev is not None and len(kps) and len(prev[0]):
            pk, pd = prev
            H, W = im.shape
            F = pkg.FrameView(kps, desc, (0.0, float(W), 0.0, float(H)))
            lvl = pk["octave"].astype(np.int32)
            # zero-motion prediction: last frame's keypoints projected where they were; window th = 15 px * scale (Tracking.cc:2898)
            nmatch, _, _ = mt.search_window(F, pd, pk["x"], pk["y"], (15.0 * sf[lvl]).astype(np.float32), lvl - 1, lvl + 1,
                                            nnratio=0.9, th_dist=100, use_second=False)
            c_mat = mt.last_call_s
        t2 = time.perf_counter()
        rows.append((name, len(kps), int(nmatch), (t1 - t0) * 1e3, (t2 - t1) * 1e3, c_ext * 1e3, c_mat * 1e3))
        prev = (kps, desc)
    os.makedirs(os.path.dirname(args.csv) or ".", exist_ok=True)
    with open(args.csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["frame", "n_keypoints", "n_matches", "ORB_Ext(ms)", "Match(ms)", "ORB_Ext_C(ms)", "Match_C(ms)"])
        w.writerows(rows)
    ext = np.array([r[3] for r in rows[1:]]); mat = np.array([r[4] for r in rows[1:]])
    cext = np.array([r[5] for r in rows[1:]]); cmat = np.array([r[6] for r in rows[1:]])
    print("frames: %d   keypoints/frame: %.0f   matches/frame: %.0f" % (len(rows), np.mean([r[1] for r in rows]), np.mean([r[2] for r in rows[1:]])))
    # the reference times the C++ call (Frame.cc:333-343): the C ABI calls alone, then the same with this Python mirror's marshalling
    print("C ABI calls (orbx_extract + orbm_search_by_projection): median %.3f ms   (ORB_Ext %.3f ms, match %.3f ms), 99th percentile %.3f ms, mean %.3f ms" % (
        np.median(cext + cmat), np.median(cext), np.median(cmat), np.percentile(cext + cmat, 99), np.mean(cext + cmat)))
    print("median front-end time: %.3f ms   (ORB_Ext %.3f ms, match %.3f ms)   [incl. the Python mirror's array handling]" % (np.median(ext + mat), np.median(ext), np.median(mat)))
    print("mean front-end time:   %.3f ms" % np.mean(ext + mat))
    ex.close(); mt.close()


if __name__ == "__main__":
    main()
