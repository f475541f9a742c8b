"""GPU parity tests of SearchForTriangulation for camera models whose epipolarConstrain is not Pinhole's (BASELINE config 5's
KannalaBrandt8 keyframes reach ORBmatcher.cc:1148 from LocalMapping.cc:583; two-camera rigs likewise).

The device delivers the candidate lists in front of the predicate (orbm_triangulation_candidates), the library walks them with the
caller's predicate (orbm_search_for_triangulation_pred) - what the adapter does with the reference's own
pCamera1->epipolarConstrain.  Checked against the oracle's restatement of the member around the SAME injected predicate:
  * the lists themselves (CSR, order = distance ascending, node position descending) against the oracle's lists;
  * Pinhole's predicate injected on both sides, which must also reproduce the all-device Pinhole member;
  * an arbitrary pure predicate (a hash of the pair) that rejects most first choices, so the walk goes deep into the lists and
    the "last minimum among the accepted" rule decides (duplicated descriptors give equal distances);
  * bCoarse, bOnlyStereo, the epipole gate off (rigs), orientation check on and off."""
import numpy as np
import pytest

from conftest import EUROC
from test_gpu_match import _bow

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(pkg, oracle, synth):
    frames, offs = synth.make_stream(6100, 2)
    o = oracle.OracleExtractor(**EUROC)
    (_, k0, d0), (_, k1, d1) = o.extract(frames[0]), o.extract(frames[1])
    sf = np.asarray(o.scale_factors, np.float32)
    rng = np.random.default_rng(61)
    d1 = d1.copy()
    dup = rng.permutation(len(k1))[:150]                     # duplicated descriptors in KF2: equal distances inside a node
    d1[dup] = d1[rng.integers(0, len(k1), 150)]
    sigma2 = (sf * sf).astype(np.float32)
    cam = np.array([458.654, 457.296, 367.215, 248.375], np.float32)
    z = np.float32(5.0)
    dx, dy = offs[0][0] - offs[1][0], offs[0][1] - offs[1][1]
    R1w, t1w = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    ang = 0.001
    R2w = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    t2w = np.array([dx * z / cam[0], dy * z / cam[1], 0.02], np.float32)
    Cw1 = np.zeros(3, np.float32)
    ur0 = np.where(rng.random(len(k0)) < 0.4, k0["x"] - np.float32(9.0), np.float32(-1)).astype(np.float32)
    ur1 = np.where(rng.random(len(k1)) < 0.4, k1["x"] - np.float32(9.0), np.float32(-1)).astype(np.float32)
    mp0 = (rng.random(len(k0)) < 0.2).astype(np.uint8)
    mp1 = (rng.random(len(k1)) < 0.2).astype(np.uint8)
    return dict(k0=k0, d0=d0, k1=k1, d1=d1, sf=sf, sigma2=sigma2, cam=cam, R1w=R1w, t1w=t1w, R2w=R2w, t2w=t2w, Cw1=Cw1, ur0=ur0, ur1=ur1, mp0=mp0, mp1=mp1)


def views(pkg, oracle, S, nodes, rig=False):
    fv0, fv1 = _bow(S["d0"], nodes), _bow(S["d1"], nodes)
    ur0 = np.full(len(S["k0"]), -1, np.float32) if rig else S["ur0"]     # rigs: bStereo is false for every keypoint (:1059, :1086)
    ur1 = np.full(len(S["k1"]), -1, np.float32) if rig else S["ur1"]
    KF1 = pkg.KeyFrameView(S["k0"], S["d0"], fv0, S["sf"], S["sigma2"], u_right=ur0, has_mappoint=S["mp0"])
    KF2 = pkg.KeyFrameView(S["k1"], S["d1"], fv1, S["sf"], S["sigma2"], u_right=ur1, has_mappoint=S["mp1"])
    O1 = oracle.OracleKeyFrame(S["k0"], S["d0"], fv0, S["sf"], S["sigma2"], u_right=ur0, has_mp=S["mp0"])
    O2 = oracle.OracleKeyFrame(S["k1"], S["d1"], fv1, S["sf"], S["sigma2"], u_right=ur1, has_mp=S["mp1"])
    return KF1, KF2, O1, O2


@pytest.mark.parametrize("nodes,gate,only_stereo", [(128, True, False), (16, True, False), (16, False, False), (128, True, True), (4, False, False)])
def test_candidate_lists_equal_the_oracle(pkg, oracle, scene, nodes, gate, only_stereo):
    S = scene
    KF1, KF2, O1, O2 = views(pkg, oracle, S, nodes, rig=not gate)
    ep, _ = oracle.pinhole_pair_geometry(S["R1w"], S["t1w"], S["R2w"], S["t2w"], S["Cw1"], S["cam"], S["cam"])
    m = pkg.ORBmatcher(0.6, False)
    try:
        st, i2, di = m.TriangulationCandidates(KF1, KF2, ep, gate, only_stereo)
    finally:
        m.close()
    st_r, i2_r, di_r = oracle.triangulation_candidates(O1, O2, ep, gate, only_stereo)
    assert np.array_equal(st, st_r) and np.array_equal(i2, i2_r) and np.array_equal(di, di_r)
    assert len(i2_r) > (100 if not only_stereo else 20)
    cnt = np.diff(st_r)
    assert cnt.max() >= (2 if only_stereo else 3)                # some keypoints have several candidates: the order matters
    for a, b in zip(st_r[:-1], st_r[1:]):
        assert np.all(np.diff(di_r[a:b]) >= 0)


@pytest.mark.parametrize("coarse,only_stereo,check_ori", [(False, False, False), (False, False, True), (True, False, True), (False, True, False)])
def test_pinhole_predicate_injected_on_both_sides(pkg, oracle, scene, coarse, only_stereo, check_ori):
    """The reference's Pinhole::epipolarConstrain as the injected predicate: product walk = oracle walk = the all-device Pinhole member."""
    S = scene
    KF1, KF2, O1, O2 = views(pkg, oracle, S, 128)
    ep, F12 = oracle.pinhole_pair_geometry(S["R1w"], S["t1w"], S["R2w"], S["t2w"], S["Cw1"], S["cam"], S["cam"])
    k0, k1, sigma2 = S["k0"], S["k1"], S["sigma2"]
    calls = []

    def pred(i1, i2):
        calls.append((i1, i2))
        return oracle.pinhole_epipolar_constrain(F12, k0["x"][i1], k0["y"][i1], k1["x"][i2], k1["y"][i2], sigma2[k1["octave"][i2]])

    m = pkg.ORBmatcher(0.6, check_ori)
    try:
        n_gpu, pairs_gpu = m.SearchForTriangulationPred(KF1, KF2, ep, True, pred, bOnlyStereo=only_stereo, bCoarse=coarse)
        n_dev, pairs_dev = m.SearchForTriangulation(KF1, KF2, S["R1w"], S["t1w"], S["R2w"], S["t2w"], S["Cw1"], S["cam"], S["cam"], bOnlyStereo=only_stereo, bCoarse=coarse)
    finally:
        m.close()
    n_ref, pairs_ref = oracle.search_for_triangulation_pred(O1, O2, ep, True, pred, only_stereo=only_stereo, coarse=coarse, check_ori=check_ori)
    assert n_gpu == n_ref and np.array_equal(pairs_gpu, pairs_ref)
    assert n_dev == n_ref and np.array_equal(pairs_dev, pairs_ref)
    assert n_ref > (20 if only_stereo else 60)


@pytest.mark.parametrize("nodes,gate,accept", [(128, True, 0.35), (16, False, 0.35), (16, True, 0.1), (4, False, 0.02)])
def test_arbitrary_predicate_walks_deep(pkg, oracle, scene, nodes, gate, accept):
    """A pure pseudo-random predicate that rejects most pairs: the answer is the first ACCEPTED entry of each list, often not the
    nearest one, and with equal distances the one later in the node."""
    S = scene
    KF1, KF2, O1, O2 = views(pkg, oracle, S, nodes, rig=not gate)
    ep, _ = oracle.pinhole_pair_geometry(S["R1w"], S["t1w"], S["R2w"], S["t2w"], S["Cw1"], S["cam"], S["cam"])
    thr = int(accept * 65536)

    def pred(i1, i2):
        return ((i1 * 40503 + i2 * 9973 + 12345) * 2654435761 >> 7) % 65536 < thr

    for check_ori in (False, True):
        m = pkg.ORBmatcher(0.6, check_ori)
        try:
            n_gpu, pairs_gpu = m.SearchForTriangulationPred(KF1, KF2, ep, gate, pred)
        finally:
            m.close()
        n_ref, pairs_ref = oracle.search_for_triangulation_pred(O1, O2, ep, gate, pred, check_ori=check_ori)
        assert n_gpu == n_ref and np.array_equal(pairs_gpu, pairs_ref)
    # the walk did go past first entries
    st, i2, di = oracle.triangulation_candidates(O1, O2, ep, gate)
    first = {i: i2[st[i]] for i in range(len(st) - 1) if st[i + 1] > st[i]}
    n_ref, pairs_ref = oracle.search_for_triangulation_pred(O1, O2, ep, gate, pred, check_ori=False)
    assert sum(1 for a, b in pairs_ref if first[a] != b) >= (5 if accept > 0.05 else 1)


def test_empty_and_degenerate(pkg, oracle, scene):
    S = scene
    KF1, KF2, O1, O2 = views(pkg, oracle, S, 128)
    m = pkg.ORBmatcher(0.6, True)
    try:
        # nothing accepted
        n, pairs = m.SearchForTriangulationPred(KF1, KF2, (0.0, 0.0), True, lambda a, b: False)
        assert n == 0 and len(pairs) == 0
        # no shared node
        fvA = {3: list(range(len(S["k0"])))}
        fvB = {5: list(range(len(S["k1"])))}
        A = pkg.KeyFrameView(S["k0"], S["d0"], fvA, S["sf"], S["sigma2"])
        B = pkg.KeyFrameView(S["k1"], S["d1"], fvB, S["sf"], S["sigma2"])
        st, i2, di = m.TriangulationCandidates(A, B, (0.0, 0.0), True)
        assert len(i2) == 0 and not st.any()
        n, pairs = m.SearchForTriangulationPred(A, B, (0.0, 0.0), True, lambda a, b: True)
        assert n == 0
        # one huge node (all keypoints of both keyframes): lists of dozens of candidates
        fv1 = {7: list(range(len(S["k0"])))}
        fv2 = {7: list(range(len(S["k1"])))}
        A = pkg.KeyFrameView(S["k0"], S["d0"], fv1, S["sf"], S["sigma2"], has_mappoint=S["mp0"])
        B = pkg.KeyFrameView(S["k1"], S["d1"], fv2, S["sf"], S["sigma2"], has_mappoint=S["mp1"])
        OA = oracle.OracleKeyFrame(S["k0"], S["d0"], fv1, S["sf"], S["sigma2"], has_mp=S["mp0"])
        OB = oracle.OracleKeyFrame(S["k1"], S["d1"], fv2, S["sf"], S["sigma2"], has_mp=S["mp1"])
        st, i2, di = m.TriangulationCandidates(A, B, (100.0, 100.0), True)
        st_r, i2_r, di_r = oracle.triangulation_candidates(OA, OB, (100.0, 100.0), True)
        assert np.array_equal(st, st_r) and np.array_equal(i2, i2_r) and np.array_equal(di, di_r) and len(i2_r) > 500
    finally:
        m.close()
