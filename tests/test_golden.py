"""Golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from the oracle).
CPU: the oracle reproduces them (regression pin of the oracle and of the synthetic generator).
GPU: the HIP path reproduces them without the oracle in the loop."""
import os
import zlib

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["euroc_1000", "euroc_1001", "euroc_1002", "tumvi_2000"]


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(oracle, synth, name):
    g = load(name)
    img = synth.make_frame(int(g["seed"]), int(g["H"]), int(g["W"]))
    assert crc(img) == int(g["image_crc"])
    e = oracle.OracleExtractor(int(g["nfeatures"]), 1.2, 8, 20, 7)
    pyr = e.pyramid(img)
    assert [crc(p) for p in pyr] == g["pyr_crc"].tolist()
    assert np.array_equal(pyr[7], g["level7"])
    assert [crc(oracle.gaussian_blur7(p)) for p in pyr] == g["blur_crc"].tolist()
    cands = [e.level_candidates(p) for p in pyr]
    assert [len(c) for c in cands] == g["cand_count"].tolist()
    assert [crc(c) for c in cands] == g["cand_crc"].tolist()
    for tag, lap in (("lap1000", (0, 1000)), ("lap0", (0, 0))):
        mono, kps, desc = e.extract(img, lap)
        assert mono == int(g["mono_" + tag])
        assert kps.tobytes() == g["kps_" + tag].tobytes()
        assert np.array_equal(desc, g["desc_" + tag])


def test_golden_output_order(oracle):
    """with vLappingArea {0,1000} every keypoint takes the 'stereo' branch, so the
    output is the {0,0} output reversed, and operator() returns 0."""
    g = load("euroc_1000")
    assert int(g["mono_lap1000"]) == 0 and int(g["mono_lap0"]) == len(g["kps_lap0"])
    assert g["kps_lap1000"].tobytes() == g["kps_lap0"][::-1].tobytes()
    assert np.array_equal(g["desc_lap1000"], g["desc_lap0"][::-1])
    assert g["kps_lap0"]["octave"].tolist() == sorted(g["kps_lap0"]["octave"].tolist())   # level-major walk


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_reproduces_golden(pkg, synth, name):
    g = load(name)
    img = synth.make_frame(int(g["seed"]), int(g["H"]), int(g["W"]))
    e = pkg.ORBextractor(int(g["nfeatures"]), 1.2, 8, 20, 7)
    for tag, lap in (("lap1000", (0, 1000)), ("lap0", (0, 0))):
        mono, kps, desc = e(img, None, lap)
        assert mono == int(g["mono_" + tag])
        assert kps.tobytes() == g["kps_" + tag].tobytes()
        assert np.array_equal(desc, g["desc_" + tag])
    assert [crc(e.image_pyramid_level(l)) for l in range(8)] == g["pyr_crc"].tolist()
    assert [crc(e.blurred_level(l)) for l in range(8)] == g["blur_crc"].tolist()
    assert [crc(e.level_candidates(l)) for l in range(8)] == g["cand_crc"].tolist()
    e.close()


@pytest.mark.gpu
def test_hip_reproduces_match_golden(pkg, synth):
    g = load("match_3000")
    frames, offs = synth.make_stream(3000, 2)
    e = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    _, k0, d0 = e(frames[0])
    _, k1, d1 = e(frames[1])
    assert crc(k0) == int(g["kps0_crc"]) and crc(k1) == int(g["kps1_crc"])
    sf = e.GetScaleFactors()
    u = (k0["x"] + np.float32(offs[0][0] - offs[1][0])).astype(np.float32)
    v = (k0["y"] + np.float32(offs[0][1] - offs[1][1])).astype(np.float32)
    nq = len(k0)
    m = pkg.ORBmatcher(0.8, True)
    F = pkg.FrameView(k1, d1, (0.0, 752.0, 0.0, 480.0))
    n, moq, _ = m.SearchByProjection(F, np.ones(nq, np.uint8), d0, u, v, np.ones(nq, np.float32), k0["octave"], sf, th=3.0)
    assert n == int(g["n_m2"]) and np.array_equal(moq, g["moq_m2"]) and np.array_equal(F.slot, g["slot_m2"])
    F2 = pkg.FrameView(k1, d1, (0.0, 752.0, 0.0, 480.0))
    m1 = np.full(nq, -1, np.int32)
    n, moq, bd = m.search_window(F2, d0, u, v, np.full(nq, 1.0e4, np.float32), m1, m1, nnratio=0.8, th_dist=100, use_second=True)
    assert n == int(g["n_stress"]) and np.array_equal(moq, g["moq_stress"]) and np.array_equal(bd, g["bd_stress"])
    assert np.array_equal(F2.slot, g["slot_stress"])
    e.close(); m.close()


def _wider_inputs(synth, extract):
    """Inputs of wider_3000.npz re-made from the seeds; `extract(img, lap)` -> (mono, kps, desc)."""
    from golden.make_golden import bow_nodes
    frames, offs = synth.make_stream(3000, 2)
    _, k0, d0 = extract(frames[0], (0, 1000))
    _, k1, d1 = extract(frames[1], (0, 1000))
    big = synth.make_frame(4120, H=480, W=752 + 64)
    imgL, imgR = np.ascontiguousarray(big[:, 0:752]), np.ascontiguousarray(big[:, 20:20 + 752])
    mp0 = (np.arange(len(k0)) % 5 != 0).astype(np.uint8)
    mp1 = (np.arange(len(k1)) % 4 != 0).astype(np.uint8)
    groups = [d0[i:i + 3 + (i % 9)] for i in range(0, 400, 13)]
    return k0, d0, k1, d1, bow_nodes(d0), bow_nodes(d1), mp0, mp1, groups, imgL, imgR


def test_oracle_reproduces_wider_golden(oracle, synth):
    g = load("wider_3000")
    e = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
    k0, d0, k1, d1, fv0, fv1, mp0, mp1, groups, imgL, imgR = _wider_inputs(synth, lambda im, lap: e.extract(im, lap))
    sf = e.scale_factors
    sigma2 = (sf * sf).astype(np.float32)
    bounds = (0.0, 752.0, 0.0, 480.0)
    prev = np.stack([k0["x"], k0["y"]], axis=1).astype(np.float32).copy()
    n, m12 = oracle.search_for_initialization(k0, d0, oracle.OracleFrame(k1["x"], k1["y"], k1["octave"], k1["angle"], d1, bounds, sf), prev, 100, 0.9, True)
    assert n == int(g["n_init"]) and np.array_equal(m12, g["m12_init"]) and crc(prev) == int(g["prev_crc"])
    K0 = oracle.OracleKeyFrame(k0, d0, fv0, sf, sigma2, has_mp=mp0)
    K1 = oracle.OracleKeyFrame(k1, d1, fv1, sf, sigma2, has_mp=mp1)
    n, m = oracle.search_by_bow(K0, K1, 0.7, True)
    assert n == int(g["n_bow"]) and np.array_equal(m, g["m_bow"])
    n, m = oracle.search_by_bow_keyframes(K0, K1, 0.8, True)
    assert n == int(g["n_bowkk"]) and np.array_equal(m, g["m_bowkk"])
    assert np.array_equal(np.array([oracle.distinctive_descriptor(x) for x in groups], np.int32), g["best"])
    _, kL, dL = e.extract(imgL, (0, 0))
    _, kR, dR = e.extract(imgR, (0, 0))
    assert crc(kL) == int(g["kL_crc"]) and crc(kR) == int(g["kR_crc"])
    uR, depth = e.compute_stereo_matches(imgL, imgR, kL, dL, kR, dR, 0.11, 47.9)
    assert np.array_equal(uR.view(np.uint32), g["uR"].view(np.uint32)) and np.array_equal(depth.view(np.uint32), g["depth"].view(np.uint32))


@pytest.mark.gpu
def test_hip_reproduces_wider_golden(pkg, synth):
    """The wider rows through the C ABI against the frozen fixture - no oracle in the loop."""
    g = load("wider_3000")
    ex, exR = pkg.ORBextractor(1000, 1.2, 8, 20, 7), pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    k0, d0, k1, d1, fv0, fv1, mp0, mp1, groups, imgL, imgR = _wider_inputs(synth, lambda im, lap: ex(im, None, lap))
    sf = ex.GetScaleFactors()
    sigma2 = (sf * sf).astype(np.float32)
    bounds = (0.0, 752.0, 0.0, 480.0)
    m9, m7, m8 = pkg.ORBmatcher(0.9, True), pkg.ORBmatcher(0.7, True), pkg.ORBmatcher(0.8, True)
    try:
        prev = np.stack([k0["x"], k0["y"]], axis=1).astype(np.float32).copy()
        n, m12 = m9.SearchForInitialization(pkg.FrameView(k0, d0, bounds), pkg.FrameView(k1, d1, bounds), prev, 100)
        assert n == int(g["n_init"]) and np.array_equal(m12, g["m12_init"]) and crc(prev) == int(g["prev_crc"])
        K0 = pkg.KeyFrameView(k0, d0, fv0, sf, sigma2, has_mappoint=mp0)
        K1 = pkg.KeyFrameView(k1, d1, fv1, sf, sigma2, has_mappoint=mp1)
        n, m = m7.SearchByBoW(K0, K1)
        assert n == int(g["n_bow"]) and np.array_equal(m, g["m_bow"])
        n, m = m8.SearchByBoWKeyFrames(K0, K1)
        assert n == int(g["n_bowkk"]) and np.array_equal(m, g["m_bowkk"])
        assert np.array_equal(m8.ComputeDistinctiveDescriptors(groups), g["best"])
        _, kL, dL = ex(imgL, None, (0, 0))
        _, kR, dR = exR(imgR, None, (0, 0))
        assert crc(kL) == int(g["kL_crc"]) and crc(kR) == int(g["kR_crc"])
        uR, depth = ex.ComputeStereoMatches(exR, kL, dL, kR, dR, 0.11, 47.9)
        assert np.array_equal(uR.view(np.uint32), g["uR"].view(np.uint32)) and np.array_equal(depth.view(np.uint32), g["depth"].view(np.uint32))
    finally:
        for x in (m9, m7, m8, ex, exR):
            x.close()

