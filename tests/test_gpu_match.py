"""GPU parity tests of the ORBmatcher path: projection search with sequential claim semantics, best/second ratio
test, grid-walk tie order -- HIP (through the C ABI) vs the CPU oracle.  Index-exact."""
import numpy as np
import pytest

from conftest import EUROC

pytestmark = pytest.mark.gpu


def make_frame_pair(pkg, oracle, synth, seed, shift=(5, -3)):
    """Two synthetic frames related by an integer shift, extracted with the oracle (inputs for the matcher)."""
    frames, offs = synth.make_stream(seed, 2)
    o = oracle.OracleExtractor(**EUROC)
    _, k0, d0 = o.extract(frames[0])
    _, k1, d1 = o.extract(frames[1])
    return (k0, d0), (k1, d1), offs, o.scale_factors


def both_frames(pkg, oracle, k1, d1, sf, bounds=(0.0, 752.0, 0.0, 480.0)):
    F = pkg.FrameView(k1, d1, bounds)
    OF = oracle.OracleFrame(k1["x"], k1["y"], k1["octave"], k1["angle"], d1, bounds, sf)
    return F, OF


@pytest.fixture(scope="module", params=[0, 1, 2], ids=["auto", "scan", "walk"])
def matcher(pkg, request):
    """Every projection-search test runs three times: candidate enumeration decided on the device (default), forced to the
    all-keypoints scan (k_match_scan) and forced to the grid-window walk (k_match_walk, Frame::GetFeaturesInArea's own order).
    The oracle's answer is the same for all three."""
    m = pkg.ORBmatcher(0.8, True)
    m.set_scan_mode(request.param)
    yield m
    m.close()


def test_hamming_matrix(pkg, matcher, synth):
    q, c = synth.make_descriptor_sets(2000, n=1000)
    d = matcher.hamming_matrix(q, c)
    ref = np.unpackbits(q[:, None, :] ^ c[None, :, :], axis=2).sum(axis=2)
    assert np.array_equal(d, ref.astype(np.uint16))
    d2 = matcher.hamming_matrix(q[:7], c[:301])
    assert np.array_equal(d2, ref[:7, :301].astype(np.uint16))


@pytest.mark.parametrize("seed", [3000, 3001])
def test_search_by_projection_m2(pkg, oracle, synth, matcher, seed):
    """ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th): map points = keypoints of frame t-1."""
    (k0, d0), (k1, d1), offs, sf = make_frame_pair(pkg, oracle, synth, seed)
    F, OF = both_frames(pkg, oracle, k1, d1, sf)
    nq = len(k0)
    rng = np.random.default_rng(seed)
    projX = (k0["x"] + np.float32(offs[0][0] - offs[1][0])).astype(np.float32)
    projY = (k0["y"] + np.float32(offs[0][1] - offs[1][1])).astype(np.float32)
    viewCos = rng.choice(np.array([0.9, 0.9985, 1.0], dtype=np.float32), nq)
    level = k0["octave"].astype(np.int32)
    in_view = (rng.random(nq) < 0.9).astype(np.uint8)
    obs = (rng.random(nq) < 0.85).astype(np.uint8)
    for th in (1.0, 3.0):
        F.slot[:] = -1; F.slot_obs[:] = 0; OF.slot[:] = -1; OF.slot_obs[:] = 0
        n_gpu, moq_gpu, _ = matcher.SearchByProjection(F, in_view, d0, projX, projY, viewCos, level, sf, th=th, mp_obs=obs)
        n_ref, moq_ref = OF.search_by_projection_mp(in_view, d0, projX, projY, viewCos, level, th, 0.8, qobs=obs)
        assert n_gpu == n_ref and n_ref > 200
        assert np.array_equal(moq_gpu, moq_ref)
        assert np.array_equal(F.slot, OF.slot) and np.array_equal(F.slot_obs, OF.slot_obs)


def test_search_stress_1000x1000(pkg, oracle, synth, matcher):
    """BASELINE config 3 stress: window = whole image, levels open -> every query sees every keypoint; sequential
    claims on; nnratio 0.8; TH_HIGH 100."""
    (k0, d0), (k1, d1), offs, sf = make_frame_pair(pkg, oracle, synth, 3100)
    F, OF = both_frames(pkg, oracle, k1, d1, sf)
    nq = len(k0)
    u = (k0["x"] + np.float32(offs[0][0] - offs[1][0])).astype(np.float32)
    v = (k0["y"] + np.float32(offs[0][1] - offs[1][1])).astype(np.float32)
    radius = np.full(nq, 1.0e4, np.float32)
    ml = np.full(nq, -1, np.int32)
    n_gpu, moq_gpu, bd_gpu = matcher.search_window(F, d0, u, v, radius, ml, ml, nnratio=0.8, th_dist=100, use_second=True)
    n_ref, moq_ref, bd_ref = OF.search_by_projection_win(d0, u, v, radius, ml, ml, 0.8, 100, True)
    assert n_gpu == n_ref and n_ref > 300
    assert np.array_equal(moq_gpu, moq_ref) and np.array_equal(bd_gpu, bd_ref)
    assert np.array_equal(F.slot, OF.slot)


@pytest.mark.parametrize("N", [600, 4000, 15360])
def test_claims_and_ties(pkg, oracle, matcher, N):
    """Collisions: duplicated descriptors (distance ties decided by grid-walk order), duplicated queries (later
    queries lose a claimed keypoint only if the holder has observations), pre-occupied slots.
    N = 4000 exceeds the LDS-resident candidate copy of the resolve kernel (global-memory rescan path); 15360 is the
    largest frame the search kernels take (ORBM_MAX_KEYPOINTS), one keypoint more is refused."""
    rng = np.random.default_rng(11)
    kps = np.zeros(N, dtype=pkg.KP_DTYPE)
    kps["x"] = rng.uniform(5, 747, N).astype(np.float32)
    kps["y"] = rng.uniform(5, 475, N).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, N)
    kps["angle"] = rng.uniform(0, 360, N).astype(np.float32)
    base = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    desc = base[rng.integers(0, 40, N)].copy()            # heavy duplication -> ties everywhere
    flip = rng.random((N, 32)) < 0.02
    desc[flip] ^= 1
    sf = np.array([1.2 ** i for i in range(8)], dtype=np.float32)
    F, OF = both_frames(pkg, oracle, kps, desc, sf)
    occ = rng.random(N) < 0.2
    F.slot[occ] = 9999; OF.slot[occ] = 9999
    F.slot_obs[occ] = (rng.random(occ.sum()) < 0.5); OF.slot_obs[:] = F.slot_obs
    nq = 500
    qi = rng.integers(0, N, nq)
    qdesc = desc[qi].copy()
    u = (kps["x"][qi] + rng.uniform(-3, 3, nq)).astype(np.float32)
    v = (kps["y"][qi] + rng.uniform(-3, 3, nq)).astype(np.float32)
    radius = rng.choice(np.array([10.0, 40.0, 200.0], dtype=np.float32), nq)
    minl = rng.integers(-1, 4, nq).astype(np.int32)
    maxl = np.where(rng.random(nq) < 0.3, -1, minl + rng.integers(0, 5, nq)).astype(np.int32)
    obs = (rng.random(nq) < 0.6).astype(np.uint8)
    inv = (rng.random(nq) < 0.95).astype(np.uint8)
    flags = inv | (obs << 1)
    for use_second in (True, False):
        F.slot[:] = np.where(occ, 9999, -1); OF.slot[:] = F.slot
        F.slot_obs[:] = OF.slot_obs
        so0 = OF.slot_obs.copy()
        n_gpu, moq_gpu, bd_gpu = matcher.search_window(F, qdesc, u, v, radius, minl, maxl, flags=flags, nnratio=0.7, th_dist=60, use_second=use_second)
        n_ref, moq_ref, bd_ref = OF.search_by_projection_win(qdesc, u, v, radius, minl, maxl, 0.7, 60, use_second, qobs=obs, in_view=inv)
        assert n_gpu == n_ref and n_ref > 50
        assert np.array_equal(moq_gpu, moq_ref) and np.array_equal(bd_gpu, bd_ref)
        assert np.array_equal(F.slot, OF.slot) and np.array_equal(F.slot_obs, OF.slot_obs)
        OF.slot_obs[:] = so0; F.slot_obs[:] = so0
    if N == 15360:
        big = np.concatenate([kps, kps[:1]])
        F2 = pkg.FrameView(big, np.concatenate([desc, desc[:1]]), (0.0, 752.0, 0.0, 480.0))
        with pytest.raises((pkg.OrbError, ValueError)):
            matcher.search_window(F2, qdesc, u, v, radius, minl, maxl, flags=flags, nnratio=0.7, th_dist=60, use_second=True)


def test_empty_and_ragged(pkg, oracle, matcher):
    kps = np.zeros(3, dtype=pkg.KP_DTYPE)
    kps["x"] = [10, 700, -50]       # one keypoint outside the grid (PosInGrid false, Frame.cc:821-822)
    kps["y"] = [10, 400, 100]
    desc = np.zeros((3, 32), np.uint8)
    F = pkg.FrameView(kps, desc, (0.0, 752.0, 0.0, 480.0))
    n, moq, bd = matcher.search_window(F, np.zeros((0, 32), np.uint8), [], [], [], [], [])
    assert n == 0 and len(moq) == 0
    n, moq, bd = matcher.search_window(F, np.zeros((2, 32), np.uint8), [12.0, -48.0], [11.0, 100.0], [5.0, 5.0], [-1, -1], [-1, -1], th_dist=100, use_second=False)
    assert n == 1 and moq.tolist() == [0, -1]
    F0 = pkg.FrameView(kps[:0], desc[:0], (0.0, 752.0, 0.0, 480.0))
    n, moq, bd = matcher.search_window(F0, np.zeros((2, 32), np.uint8), [12.0, 5.0], [11.0, 5.0], [5.0, 5.0], [-1, -1], [-1, -1])
    assert n == 0 and moq.tolist() == [-1, -1]


def _m3_scene(pkg, oracle, synth, seed, cam, stereo):
    (k0, d0), (k1, d1), offs, sf = make_frame_pair(pkg, oracle, synth, seed)
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375            # Examples/Monocular/EuRoC.yaml:9-12
    n0 = len(k0)
    z = np.float32(5.0)
    Xw = np.stack([(k0["x"] - np.float32(cx)) / np.float32(fx) * z, (k0["y"] - np.float32(cy)) / np.float32(fy) * z,
                   np.full(n0, z, np.float32)], axis=1).astype(np.float32)
    dx, dy = offs[0][0] - offs[1][0], offs[0][1] - offs[1][1]
    if cam == 1:
        # KannalaBrandt8: the scene comes from UN-projection through the fisheye model (synth.make_last_frame_scene), so that the
        # reprojected map points fall on their keypoints all over the image and the search finds several hundred matches
        params = np.array([190.978477 * 2, 190.973307 * 2, 376.0, 240.0, 0.003482389402, 0.000715034845, -0.002053236141, 0.000202936736], np.float32)
        Xw, Tcw, Tlw = synth.make_last_frame_scene(1, params, k0["x"], k0["y"], (dx, dy), seed)
        Xw[rng.random(n0) < 0.03] *= np.float32(-1)                   # behind the camera -> invzc < 0
        has_mp = (rng.random(n0) < 0.8).astype(np.uint8)
        obs = (rng.random(n0) < 0.9).astype(np.uint8)
        return k0, d0, k1, d1, sf, Xw, Tcw, Tlw, has_mp, obs, params, None
    Xw[rng.random(n0) < 0.03, 2] = -1.0                               # behind the camera -> invzc < 0
    ang = 0.002
    Tcw = np.eye(4, dtype=np.float32)
    Tcw[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    Tcw[:3, 3] = [dx * z / fx, dy * z / fy, -0.3 if stereo else 0.0]
    Tlw = np.eye(4, dtype=np.float32)
    has_mp = (rng.random(n0) < 0.8).astype(np.uint8)
    obs = (rng.random(n0) < 0.9).astype(np.uint8)
    if cam == 0:
        params = np.array([fx, fy, cx, cy], np.float32)
    else:                                                             # Examples/Monocular/TUM_512.yaml:9-19, centred on this image
        params = np.array([190.978477 * 2, 190.973307 * 2, 376.0, 240.0, 0.003482389402, 0.000715034845, -0.002053236141, 0.000202936736], np.float32)
    u_right = None
    if stereo:
        u_right = np.where(rng.random(len(k1)) < 0.7, k1["x"] - np.float32(47.9) / z, np.float32(-1)).astype(np.float32)
    return k0, d0, k1, d1, sf, Xw, Tcw, Tlw, has_mp, obs, params, u_right


@pytest.mark.parametrize("cam,stereo", [(0, False), (0, True), (1, False)])
def test_search_by_projection_last_frame_m3(pkg, oracle, synth, matcher, cam, stereo):
    """ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) incl. camera projection (Pinhole /
    KannalaBrandt8), forward/backward level windows, stereo right-coordinate check and rotation-histogram pruning."""
    k0, d0, k1, d1, sf, Xw, Tcw, Tlw, has_mp, obs, params, u_right = _m3_scene(pkg, oracle, synth, 3200 + cam, cam, stereo)
    bounds = (0.0, 752.0, 0.0, 480.0)
    total = 0
    for th in (15.0, 30.0):
        F = pkg.FrameView(k1, d1, bounds, u_right=u_right)
        OF = oracle.OracleFrame(k1["x"], k1["y"], k1["octave"], k1["angle"], d1, bounds, sf, u_right=u_right)
        mb, mbf = (0.11, 47.9) if stereo else (0.0, 0.0)
        n_gpu = matcher.SearchByProjectionLastFrame(F, sf, has_mp, Xw, d0, k0, Tcw, Tlw, cam, params, th, bMono=not stereo, mb=mb, mbf=mbf, mp_obs=obs)
        n_ref = OF.search_by_projection_ff(has_mp, Xw, d0, k0["octave"], k0["angle"], Tcw, Tlw, cam, params, th, mono=not stereo,
                                           check_ori=True, mb=mb, mbf=mbf, qobs=obs)
        assert n_gpu == n_ref
        assert np.array_equal(F.slot, OF.slot) and np.array_equal(F.slot_obs, OF.slot_obs)
        total += n_ref
    assert total > (100 if cam == 0 else 600)


def _bow(desc, nodes=128):
    """Stand-in for DBoW2's FeatureVector (the ORB vocabulary file is missing from the reference mount,
    .MISSING_LARGE_BLOBS:5): node id = a hash of descriptor bits, so that similar descriptors mostly share a node."""
    ids = (desc[:, 0].astype(np.int64) >> 2) * 2 + (desc[:, 7].astype(np.int64) >> 7)
    fv = {}
    for i, n in enumerate(ids % nodes):
        fv.setdefault(int(n) * 7 + 3, []).append(i)      # sparse, unordered ids like real node ids
    return fv


@pytest.mark.parametrize("coarse,only_stereo,check_ori", [(False, False, False), (True, False, True), (False, True, False)])
def test_search_for_triangulation_m6(pkg, oracle, synth, coarse, only_stereo, check_ori):
    """ORBmatcher::SearchForTriangulation: vocabulary-node merge walk, all-pairs Hamming inside a node with the running
    `dist>bestDist` gate (last minimum wins), epipole-distance gate, Pinhole::epipolarConstrain, stereo-only mode."""
    (k0, d0), (k1, d1), offs, sf = make_frame_pair(pkg, oracle, synth, 3300)
    rng = np.random.default_rng(5)
    sigma2 = (sf * sf).astype(np.float32)
    cam = np.array([458.654, 457.296, 367.215, 248.375], np.float32)
    z = np.float32(5.0)
    dx, dy = offs[0][0] - offs[1][0], offs[0][1] - offs[1][1]
    R1w, t1w = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    ang = 0.001
    R2w = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    t2w = np.array([dx * z / cam[0], dy * z / cam[1], 0.02], np.float32)
    Cw1 = np.zeros(3, np.float32)
    ur0 = np.where(rng.random(len(k0)) < 0.5, k0["x"] - np.float32(9.0), np.float32(-1)).astype(np.float32)
    ur1 = np.where(rng.random(len(k1)) < 0.5, k1["x"] - np.float32(9.0), np.float32(-1)).astype(np.float32)
    mp0 = (rng.random(len(k0)) < 0.2).astype(np.uint8)
    mp1 = (rng.random(len(k1)) < 0.2).astype(np.uint8)
    fv0, fv1 = _bow(d0), _bow(d1)
    KF1 = pkg.KeyFrameView(k0, d0, fv0, sf, sigma2, u_right=ur0, has_mappoint=mp0)
    KF2 = pkg.KeyFrameView(k1, d1, fv1, sf, sigma2, u_right=ur1, has_mappoint=mp1)
    O1 = oracle.OracleKeyFrame(k0, d0, fv0, sf, sigma2, u_right=ur0, has_mp=mp0)
    O2 = oracle.OracleKeyFrame(k1, d1, fv1, sf, sigma2, u_right=ur1, has_mp=mp1)
    m = pkg.ORBmatcher(0.6, check_ori)
    n_gpu, pairs_gpu = m.SearchForTriangulation(KF1, KF2, R1w, t1w, R2w, t2w, Cw1, cam, cam, bOnlyStereo=only_stereo, bCoarse=coarse)
    n_ref, pairs_ref = oracle.search_for_triangulation(O1, O2, R1w, t1w, R2w, t2w, Cw1, cam, cam, only_stereo=only_stereo, coarse=coarse,
                                                       check_ori=check_ori)
    assert n_gpu == n_ref and n_ref > (20 if only_stereo else 60)
    assert np.array_equal(pairs_gpu, pairs_ref)
    m.close()


@pytest.mark.parametrize("cam", [0, 1])
def test_search_by_projection_keyframe_m4(pkg, oracle, synth, matcher, cam):
    """ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) -- relocalisation: depth-range gate,
    MapPoint::PredictScale, +-1 level window, every occupied slot blocks, rotation check."""
    k0, d0, k1, d1, sf, Xw, Tcw, Tlw, has_mp, obs, params, _ = _m3_scene(pkg, oracle, synth, 3400 + cam, cam, False)
    rng = np.random.default_rng(77)
    bounds = (0.0, 752.0, 0.0, 480.0)
    dist_last = np.sqrt((Xw.astype(np.float64) ** 2).sum(axis=1)).astype(np.float32)
    max_dist = (dist_last * sf[k0["octave"]]).astype(np.float32)            # MapPoint::UpdateNormalAndDepth
    max_dist[rng.random(len(k0)) < 0.05] *= np.float32(0.3)                  # some points outside the invariance range
    min_dist = (max_dist / sf[-1]).astype(np.float32)
    log_sf = float(np.log(np.float32(1.2)))
    total = 0
    for th, orb_dist in ((10.0, 100), (3.0, 64)):                            # Tracking.cc:3877, :3891
        F = pkg.FrameView(k1, d1, bounds)
        OF = oracle.OracleFrame(k1["x"], k1["y"], k1["octave"], k1["angle"], d1, bounds, sf)
        occ = rng.random(len(k1)) < 0.15
        F.slot[occ] = 1 << 30; F.slot_obs[occ] = 1; OF.slot[occ] = 1 << 30; OF.slot_obs[occ] = 1
        n_gpu = matcher.SearchByProjectionKeyFrame(F, sf, log_sf, has_mp, Xw, d0, k0["angle"], max_dist, min_dist, Tcw, cam, params, th, orb_dist)
        n_ref = oracle.search_by_projection_kf(OF, has_mp, Xw, d0, k0["angle"], max_dist, min_dist, Tcw, cam, params, log_sf, th, orb_dist, True)
        assert n_gpu == n_ref
        assert np.array_equal(F.slot, OF.slot) and np.array_equal(F.slot_obs, OF.slot_obs)
        total += n_ref
    assert total > (100 if cam == 0 else 300)


@pytest.mark.parametrize("cam", [0, 1])
@pytest.mark.parametrize("ratio", [1.0, 0.75])
def test_search_by_projection_sim3_m5(pkg, oracle, synth, matcher, ratio, cam):
    """ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th, ratioHamming) -- loop closing: Sim3
    decomposition, IsInImage (half-open), depth-range and viewing-angle gates, PredictScale, [lvl-1, lvl] window."""
    k0, d0, k1, d1, sf, Xw, Tcw, Tlw, has_mp, obs, params, _ = _m3_scene(pkg, oracle, synth, 3500 + cam, cam, False)   # cam 1: pKF->mpCamera is KannalaBrandt8
    rng = np.random.default_rng(31)
    bounds = (0.0, 752.0, 0.0, 480.0)
    scale = np.float32(1.37)
    Scw = Tcw.copy()
    Scw[:3, :] *= scale                                                        # s [R | t]
    dist_last = np.sqrt((Xw.astype(np.float64) ** 2).sum(axis=1)).astype(np.float32)
    max_dist = (dist_last * sf[k0["octave"]]).astype(np.float32)
    min_dist = (max_dist / sf[-1]).astype(np.float32)
    normal = (Xw / np.maximum(np.linalg.norm(Xw, axis=1, keepdims=True), 1e-6)).astype(np.float32)
    flip = rng.random(len(k0)) < 0.1
    normal[flip] *= np.float32(-1)                                             # seen from behind: fails the 60-degree gate
    log_sf = float(np.log(np.float32(1.2)))
    KF = pkg.FrameView(k1, d1, bounds)
    OKF = oracle.OracleFrame(k1["x"], k1["y"], k1["octave"], k1["angle"], d1, bounds, sf)
    occ = rng.random(len(k1)) < 0.2
    KF.slot[occ] = 1 << 30; KF.slot_obs[occ] = 1; OKF.slot[occ] = 1 << 30; OKF.slot_obs[occ] = 1
    n_gpu = matcher.SearchByProjectionSim3(KF, sf, log_sf, has_mp, Xw, normal, d0, max_dist, min_dist, Scw, params, 8, ratio, cam_type=cam)
    n_ref = oracle.search_by_projection_sim3(OKF, has_mp, Xw, normal, d0, max_dist, min_dist, Scw, params, log_sf, 8, ratio, cam_type=cam)
    assert n_gpu == n_ref and n_ref > 100
    assert np.array_equal(KF.slot, OKF.slot) and np.array_equal(KF.slot_obs, OKF.slot_obs)


def _fisheye_scene(pkg, oracle, synth, seed):
    """Map points = keypoints of frame 0; fisheye-stereo current frame: left image = frame 1, right image = frame 2."""
    frames, offs = synth.make_stream(seed, 3)
    o = oracle.OracleExtractor(**EUROC)
    (_, k0, d0), (_, kl, dl), (_, kr, dr) = o.extract(frames[0]), o.extract(frames[1]), o.extract(frames[2])
    sf = o.scale_factors
    rng = np.random.default_rng(seed)
    nl, nr = len(kl), len(kr)
    l2r = np.full(nl, -1, np.int32)
    r2l = np.full(nr, -1, np.int32)
    picks_l = rng.permutation(nl)[: int(0.45 * min(nl, nr))]
    picks_r = rng.permutation(nr)[: len(picks_l)]
    l2r[picks_l] = picks_r                                           # mvLeftToRightMatch / mvRightToLeftMatch
    r2l[picks_r] = picks_l
    bounds = (0.0, 752.0, 0.0, 480.0)
    keys = np.concatenate([kl, kr])
    desc = np.concatenate([dl, dr])
    F = pkg.FrameView(keys, desc, bounds)
    OFl = oracle.OracleFrame(kl["x"], kl["y"], kl["octave"], kl["angle"], dl, bounds, sf)
    OFr = oracle.OracleFrame(kr["x"], kr["y"], kr["octave"], kr["angle"], dr, bounds, sf)
    OF = oracle.OracleFisheyeFrame(OFl, OFr)
    return k0, d0, offs, sf, F, OF, l2r, r2l, nl, rng


@pytest.mark.parametrize("all_obs", [True, False])
def test_search_by_projection_m2_fisheye(pkg, oracle, synth, matcher, all_obs):
    """ORBmatcher.cc:44-214 complete for Nleft != -1: left + right halves, stereo-partner slot writes, the `continue` that
    drops the right half, and (all_obs=False) map points without observations whose partner writes release claims."""
    k0, d0, offs, sf, F, OF, l2r, r2l, nl, rng = _fisheye_scene(pkg, oracle, synth, 3500)
    nmp = len(k0)
    projX = (k0["x"] + np.float32(offs[0][0] - offs[1][0])).astype(np.float32)
    projY = (k0["y"] + np.float32(offs[0][1] - offs[1][1])).astype(np.float32)
    projXR = (k0["x"] + np.float32(offs[0][0] - offs[2][0])).astype(np.float32)
    projYR = (k0["y"] + np.float32(offs[0][1] - offs[2][1])).astype(np.float32)
    viewCos = rng.choice(np.array([0.9, 0.9985, 1.0], dtype=np.float32), nmp)
    viewCosR = rng.choice(np.array([0.9, 0.9985, 1.0], dtype=np.float32), nmp)
    level = k0["octave"].astype(np.int32)
    levelR = np.where(rng.random(nmp) < 0.05, -1, level).astype(np.int32)
    in_view = (rng.random(nmp) < 0.9).astype(np.uint8)
    in_view_r = (rng.random(nmp) < 0.85).astype(np.uint8)
    obs = np.ones(nmp, np.uint8) if all_obs else (rng.random(nmp) < 0.6).astype(np.uint8)
    pre = rng.permutation(F.N)[:60]                                   # keypoints already holding a map point
    pre_obs = (rng.random(60) < 0.5).astype(np.uint8)
    total = 0
    for th in (1.0, 4.0):
        F.slot[:] = -1; F.slot_obs[:] = 0; OF.slot[:] = -1; OF.slot_obs[:] = 0
        F.slot[pre] = 2 * (nmp + 7); OF.slot[pre] = nmp + 7          # device slots hold QUERY ids (2 per map point)
        F.slot_obs[pre] = pre_obs; OF.slot_obs[pre] = pre_obs
        n_gpu, ml_gpu, mr_gpu = matcher.SearchByProjectionFisheye(F, nl, l2r, r2l, d0, sf, th, in_view, projX, projY, viewCos, level,
                                                                  in_view_r, projXR, projYR, viewCosR, levelR, mp_obs=obs)
        n_ref, ml_ref, mr_ref = OF.search_by_projection_mp(l2r, r2l, in_view, in_view_r, d0, projX, projY, viewCos, level, projXR, projYR,
                                                           viewCosR, levelR, th, 0.8, qobs=obs)
        assert n_gpu == n_ref and n_ref > 200
        assert np.array_equal(ml_gpu, ml_ref) and np.array_equal(mr_gpu, mr_ref)
        holder = np.where(F.slot >= 0, F.slot >> 1, -1)
        assert np.array_equal(holder, OF.slot)
        assert np.array_equal(F.slot_obs[F.slot >= 0], OF.slot_obs[OF.slot >= 0])
        total += int((mr_ref >= 0).sum())
    assert total > 100


@pytest.mark.parametrize("cam,tz", [(1, 0.0), (1, -0.3), (0, 0.3)])
def test_search_by_projection_last_frame_m3_fisheye(pkg, oracle, synth, matcher, cam, tz):
    """ORBmatcher.cc:2027-2289 complete for a fisheye-stereo current frame: left search + right-camera pass through Trl,
    forward / backward / neutral level windows, rotation-histogram pruning over both images."""
    k0, d0, offs, sf, F, OF, l2r, r2l, nl, rng = _fisheye_scene(pkg, oracle, synth, 3600 + cam)
    fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
    n0 = len(k0)
    z = np.float32(5.0)
    Xw = np.stack([(k0["x"] - np.float32(cx)) / np.float32(fx) * z, (k0["y"] - np.float32(cy)) / np.float32(fy) * z,
                   np.full(n0, z, np.float32)], axis=1).astype(np.float32)
    Xw[rng.random(n0) < 0.03, 2] = -1.0
    dx, dy = offs[0][0] - offs[1][0], offs[0][1] - offs[1][1]
    Tcw = np.eye(4, dtype=np.float32)
    Tcw[:3, 3] = [dx * z / fx, dy * z / fy, tz]
    Tlw = np.eye(4, dtype=np.float32)
    Trl = np.eye(4, dtype=np.float32)                                 # right camera: shift that maps frame 1 onto frame 2
    Trl[:3, 3] = [(offs[1][0] - offs[2][0]) * z / fx, (offs[1][1] - offs[2][1]) * z / fy, 0.0]
    has_mp = (rng.random(n0) < 0.8).astype(np.uint8)
    obs = (rng.random(n0) < 0.9).astype(np.uint8)
    if cam == 0:
        params = np.array([fx, fy, cx, cy], np.float32)
    else:
        params = np.array([190.978477 * 2, 190.973307 * 2, 376.0, 240.0, 0.003482389402, 0.000715034845, -0.002053236141, 0.000202936736], np.float32)
    total = 0
    for th in (7.0, 14.0):
        F.slot[:] = -1; F.slot_obs[:] = 0; OF.slot[:] = -1; OF.slot_obs[:] = 0
        n_gpu = matcher.SearchByProjectionLastFrameFisheye(F, nl, sf, has_mp, Xw, d0, k0, Tcw, Tlw, Trl, cam, params, th, bMono=False, mb=0.11, mp_obs=obs)
        n_ref = OF.search_by_projection_ff(has_mp, Xw, d0, k0["octave"], k0["angle"], Tcw, Tlw, Trl, cam, params, th, mono=False,
                                           check_ori=True, mb=0.11, qobs=obs)
        assert n_gpu == n_ref
        assert np.array_equal(F.slot, OF.slot) and np.array_equal(F.slot_obs, OF.slot_obs)
        total += n_ref
    assert total > (100 if cam == 0 else 0)


@pytest.mark.parametrize("nfeatures,window", [(1000, 100), (5000, 100), (5000, 30), (10000, 100)])
def test_search_for_initialization_n2(pkg, oracle, synth, nfeatures, window):
    """ORBmatcher::SearchForInitialization (ORBmatcher.cc:722-837): level-0 window search with the vMatchedDistance rule,
    match stealing, rotation histogram and the vbPrevMatched update.  5000 features = the initialisation extractor
    (Tracking.cc:838-844), which takes the 64-bit-key path."""
    frames, offs = synth.make_stream(3700 + nfeatures, 2)
    cfg = dict(EUROC); cfg["nfeatures"] = nfeatures
    o = oracle.OracleExtractor(**cfg)
    (_, k0, d0), (_, k1, d1) = o.extract(frames[0]), o.extract(frames[1])
    bounds = (0.0, 752.0, 0.0, 480.0)
    m = pkg.ORBmatcher(0.9, True)                                     # Tracking.cc:2471: ORBmatcher matcher(0.9,true)
    try:
        prev_gpu = np.stack([k0["x"], k0["y"]], axis=1).astype(np.float32).copy()   # vbPrevMatched starts at F1's keypoints
        prev_ref = prev_gpu.copy()
        total = 0
        for rnd in range(2):                                          # second round starts from the updated vbPrevMatched
            F1, F2 = pkg.FrameView(k0, d0, bounds), pkg.FrameView(k1, d1, bounds)
            OF2 = oracle.OracleFrame(k1["x"], k1["y"], k1["octave"], k1["angle"], d1, bounds, o.scale_factors)
            n_gpu, m_gpu = m.SearchForInitialization(F1, F2, prev_gpu, window)
            n_ref, m_ref = oracle.search_for_initialization(k0, d0, OF2, prev_ref, window, 0.9, True)
            assert n_gpu == n_ref
            assert np.array_equal(m_gpu, m_ref)
            assert np.array_equal(prev_gpu, prev_ref)
            total += n_ref
        assert total > 50
    finally:
        m.close()


@pytest.mark.parametrize("nodes,check_ori", [(128, True), (16, True), (128, False)])
def test_search_by_bow_n3(pkg, oracle, synth, nodes, check_ori):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (ORBmatcher.cc:273-469): per vocabulary node, keyframe keypoints in
    node order take the best unclaimed frame keypoint (best/second ratio, TH_LOW); rotation-histogram pruning.
    nodes=16 gives nodes of more than 64 frame keypoints (several lane passes, many in-node claims)."""
    (k0, d0), (k1, d1), offs, sf = make_frame_pair(pkg, oracle, synth, 3800)
    rng = np.random.default_rng(nodes)
    sigma2 = (sf * sf).astype(np.float32)
    mp0 = (rng.random(len(k0)) < 0.8).astype(np.uint8)               # keyframe keypoints that have a (good) map point
    # descriptors that survive a frame step share a node: hash on bits of the KEYFRAME descriptor of the matching keypoint
    fv0, fv1 = _bow(d0, nodes), _bow(d1, nodes)
    KF = pkg.KeyFrameView(k0, d0, fv0, sf, sigma2, has_mappoint=mp0)
    F = pkg.KeyFrameView(k1, d1, fv1, sf, sigma2)
    OKF = oracle.OracleKeyFrame(k0, d0, fv0, sf, sigma2, has_mp=mp0)
    OF = oracle.OracleKeyFrame(k1, d1, fv1, sf, sigma2)
    m = pkg.ORBmatcher(0.7, check_ori)                                # Tracking.cc:2808: ORBmatcher matcher(0.7,true)
    try:
        n_gpu, m_gpu = m.SearchByBoW(KF, F)
        n_ref, m_ref = oracle.search_by_bow(OKF, OF, 0.7, check_ori)
        assert n_gpu == n_ref and n_ref > 100
        assert np.array_equal(m_gpu, m_ref)
    finally:
        m.close()


def test_edge_cases_of_the_wider_api(pkg, oracle, synth, matcher):
    """Empty / degenerate inputs of the members added after the core path: nothing to match must mean 0, not an error."""
    (k0, d0), (k1, d1), offs, sf = make_frame_pair(pkg, oracle, synth, 3900)
    bounds = (0.0, 752.0, 0.0, 480.0)
    sigma2 = (sf * sf).astype(np.float32)
    # SearchForInitialization against an empty frame, and from an empty frame
    F1, Fe = pkg.FrameView(k0, d0, bounds), pkg.FrameView(k0[:0], d0[:0], bounds)
    prev = np.stack([k0["x"], k0["y"]], axis=1).astype(np.float32).copy()
    n, m12 = matcher.SearchForInitialization(F1, Fe, prev, 100)
    assert n == 0 and (m12 == -1).all()
    n, m12 = matcher.SearchForInitialization(Fe, F1, np.zeros((0, 2), np.float32), 100)
    assert n == 0 and len(m12) == 0
    # SearchByBoW without a shared node / without map points
    fvA = {3: list(range(len(k0)))}
    fvB = {5: list(range(len(k1)))}
    KF = pkg.KeyFrameView(k0, d0, fvA, sf, sigma2, has_mappoint=np.ones(len(k0), np.uint8))
    Fr = pkg.KeyFrameView(k1, d1, fvB, sf, sigma2)
    n, mF = matcher.SearchByBoW(KF, Fr)
    assert n == 0 and (mF == -1).all()
    KF0 = pkg.KeyFrameView(k0, d0, _bow(d0), sf, sigma2, has_mappoint=np.zeros(len(k0), np.uint8))
    n, mF = matcher.SearchByBoW(KF0, pkg.KeyFrameView(k1, d1, _bow(d1), sf, sigma2))
    assert n == 0 and (mF == -1).all()
    # fisheye-stereo search with an empty right image and no partners
    F = pkg.FrameView(k1, d1, bounds)
    nmp = 50
    z = np.zeros(nmp, np.float32)
    lvl = k0["octave"][:nmp].astype(np.int32)
    n, ml, mr = matcher.SearchByProjectionFisheye(F, len(k1), None, None, d0[:nmp], sf, 3.0, np.ones(nmp, np.uint8), k0["x"][:nmp] + np.float32(offs[0][0] - offs[1][0]),
                                                  k0["y"][:nmp] + np.float32(offs[0][1] - offs[1][1]), np.ones(nmp, np.float32), lvl,
                                                  np.ones(nmp, np.uint8), z, z, np.ones(nmp, np.float32), lvl)
    assert (mr == -1).all() and n == int((ml >= 0).sum()) and n > 10


@pytest.mark.parametrize("nodes,check_ori", [(128, True), (16, False)])
def test_search_by_bow_keyframes_n3(pkg, oracle, synth, nodes, check_ori):
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...) (ORBmatcher.cc:839-979): as the KeyFrame/Frame overload, but the
    candidates need a good map point, the threshold is strict (< TH_LOW) and the result is indexed by the first keyframe."""
    (k0, d0), (k1, d1), offs, sf = make_frame_pair(pkg, oracle, synth, 3850)
    rng = np.random.default_rng(nodes + 1)
    sigma2 = (sf * sf).astype(np.float32)
    mp0 = (rng.random(len(k0)) < 0.8).astype(np.uint8)
    mp1 = (rng.random(len(k1)) < 0.7).astype(np.uint8)
    fv0, fv1 = _bow(d0, nodes), _bow(d1, nodes)
    K1 = pkg.KeyFrameView(k0, d0, fv0, sf, sigma2, has_mappoint=mp0)
    K2 = pkg.KeyFrameView(k1, d1, fv1, sf, sigma2, has_mappoint=mp1)
    O1 = oracle.OracleKeyFrame(k0, d0, fv0, sf, sigma2, has_mp=mp0)
    O2 = oracle.OracleKeyFrame(k1, d1, fv1, sf, sigma2, has_mp=mp1)
    m = pkg.ORBmatcher(0.8, check_ori)                                # LoopClosing.cc: ORBmatcher matcher(0.8,true)
    try:
        n_gpu, m_gpu = m.SearchByBoWKeyFrames(K1, K2)
        n_ref, m_ref = oracle.search_by_bow_keyframes(O1, O2, 0.8, check_ori)
        assert n_gpu == n_ref and n_ref > 80
        assert np.array_equal(m_gpu, m_ref)
    finally:
        m.close()


@pytest.mark.parametrize("cam,stereo", [(0, False), (0, True), (1, False)])
def test_fuse_n3(pkg, oracle, synth, matcher, cam, stereo):
    """ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) search part (ORBmatcher.cc:1425-1620): projection, depth-range and viewing
    gates, PredictScale, [lvl-1, lvl] window, chi-square gate on the reprojection error (5.99 mono / 7.8 with mvuRight)."""
    k0, d0, k1, d1, sf, Xw, Tcw, Tlw, has_mp, obs, params, u_right = _m3_scene(pkg, oracle, synth, 3950 + cam, cam, stereo)
    rng = np.random.default_rng(41)
    bounds = (0.0, 752.0, 0.0, 480.0)
    Ow = (-Tcw[:3, :3].T.astype(np.float64) @ Tcw[:3, 3].astype(np.float64)).astype(np.float32)
    dist_last = np.sqrt(((Xw - Ow) .astype(np.float64) ** 2).sum(axis=1)).astype(np.float32)
    max_dist = (dist_last * sf[k0["octave"]]).astype(np.float32)
    min_dist = (max_dist / sf[-1]).astype(np.float32)
    normal = ((Xw - Ow) / np.maximum(np.linalg.norm(Xw - Ow, axis=1, keepdims=True), 1e-6)).astype(np.float32)
    normal[rng.random(len(k0)) < 0.1] *= np.float32(-1)
    inv_sigma2 = (1.0 / (sf * sf)).astype(np.float32)
    log_sf = float(np.log(np.float32(1.2)))
    KF = pkg.FrameView(k1, d1, bounds, u_right=u_right)
    OKF = oracle.OracleFrame(k1["x"], k1["y"], k1["octave"], k1["angle"], d1, bounds, sf, u_right=u_right)
    total = 0
    for th in (3.0, 8.0):
        n_gpu, bi_gpu, bd_gpu = matcher.Fuse(KF, sf, inv_sigma2, log_sf, has_mp, Xw, normal, d0, max_dist, min_dist, Tcw, Ow, cam, params, 47.9, th)
        n_ref, bi_ref, bd_ref = oracle.fuse(OKF, has_mp, Xw, normal, d0, max_dist, min_dist, Tcw, Ow, cam, params, 47.9, inv_sigma2, log_sf, th)
        assert n_gpu == n_ref
        assert np.array_equal(bi_gpu, bi_ref) and np.array_equal(bd_gpu, bd_ref)
        total += n_ref
    assert total > (100 if (cam == 0 and not stereo) else 20 if cam == 0 else 0)   # the 3-term chi-square gate is tight in the stereo scene


@pytest.mark.parametrize("cam", [0, 1])
def test_fuse_sim3_n3(pkg, oracle, synth, matcher, cam):
    """ORBmatcher::Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) search part (ORBmatcher.cc:1660-1766)."""
    k0, d0, k1, d1, sf, Xw, Tcw, Tlw, has_mp, obs, params, _ = _m3_scene(pkg, oracle, synth, 3960 + cam, cam, False)
    rng = np.random.default_rng(43)
    bounds = (0.0, 752.0, 0.0, 480.0)
    Scw = Tcw.copy()
    Scw[:3, :] *= np.float32(0.83)
    dist_last = np.sqrt((Xw.astype(np.float64) ** 2).sum(axis=1)).astype(np.float32)
    max_dist = (dist_last * sf[k0["octave"]]).astype(np.float32)
    min_dist = (max_dist / sf[-1]).astype(np.float32)
    normal = (Xw / np.maximum(np.linalg.norm(Xw, axis=1, keepdims=True), 1e-6)).astype(np.float32)
    normal[rng.random(len(k0)) < 0.1] *= np.float32(-1)
    log_sf = float(np.log(np.float32(1.2)))
    KF = pkg.FrameView(k1, d1, bounds)
    OKF = oracle.OracleFrame(k1["x"], k1["y"], k1["octave"], k1["angle"], d1, bounds, sf)
    n_gpu, bi_gpu, bd_gpu = matcher.FuseSim3(KF, sf, log_sf, has_mp, Xw, normal, d0, max_dist, min_dist, Scw, params, 4.0, cam_type=cam)
    n_ref, bi_ref, bd_ref = oracle.fuse_sim3(OKF, has_mp, Xw, normal, d0, max_dist, min_dist, Scw, params, log_sf, 4.0, cam_type=cam)
    assert n_gpu == n_ref and n_ref > 100
    assert np.array_equal(bi_gpu, bi_ref) and np.array_equal(bd_gpu, bd_ref)


def test_search_by_sim3_n3(pkg, oracle, synth, matcher):
    """ORBmatcher::SearchBySim3 (ORBmatcher.cc:1788-2012): map points of KF1 searched in KF2 through the Sim3 and back, depth-range
    gate, PredictScale, [lvl-1, lvl] window, TH_HIGH, mutual-consistency pass."""
    (k0, d0), (k1, d1), offs, sf = make_frame_pair(pkg, oracle, synth, 3970)
    rng = np.random.default_rng(47)
    fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
    cam = np.array([fx, fy, cx, cy], np.float32)
    z = np.float32(5.0)
    bounds = (0.0, 752.0, 0.0, 480.0)
    dx, dy = offs[0][0] - offs[1][0], offs[0][1] - offs[1][1]
    # KF1 at the origin; KF2 displaced so that a point seen at (u, v) in KF1 lands at (u + dx, v + dy) in KF2; s12 = 1 + eps
    R1w, t1w = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    R2w, t2w = np.eye(3, dtype=np.float32), np.array([dx * z / fx, dy * z / fy, 0.0], np.float32)
    s12 = np.float32(1.02)
    R12 = np.eye(3, dtype=np.float32)
    t12 = (-t2w * s12).astype(np.float32)                             # p1 = s12 * R12 * p2 + t12  ~  inverse of p2 = p1 + t2w

    def backproject(k, tw):
        Xc = np.stack([(k["x"] - np.float32(cx)) / np.float32(fx) * z, (k["y"] - np.float32(cy)) / np.float32(fy) * z, np.full(len(k), z, np.float32)], axis=1)
        return (Xc - tw).astype(np.float32)                           # world point of the keypoint (R = I)

    def side(k, d, Rw, tw):
        Xw = backproject(k, tw)
        dist = np.sqrt((Xw.astype(np.float64) ** 2).sum(axis=1)).astype(np.float32)
        max_dist = (dist * sf[k["octave"]] * np.float32(1.1)).astype(np.float32)
        min_dist = (max_dist / sf[-1]).astype(np.float32)
        valid = (rng.random(len(k)) < 0.8).astype(np.uint8)           # pMP && !isBad && !alreadyMatched
        return dict(sf=sf, log_sf=float(np.log(np.float32(1.2))), valid=valid, Xw=Xw, desc=d, max_dist=max_dist, min_dist=min_dist, Rw=Rw, tw=tw)

    S1, S2 = side(k0, d0, R1w, t1w), side(k1, d1, R2w, t2w)
    KF1, KF2 = pkg.FrameView(k0, d0, bounds), pkg.FrameView(k1, d1, bounds)
    O1 = oracle.OracleFrame(k0["x"], k0["y"], k0["octave"], k0["angle"], d0, bounds, sf)
    O2 = oracle.OracleFrame(k1["x"], k1["y"], k1["octave"], k1["angle"], d1, bounds, sf)
    n_gpu, m_gpu = matcher.SearchBySim3(KF1, S1, KF2, S2, float(s12), R12, t12, cam, 7.5)
    n_ref, m_ref = oracle.search_by_sim3(O1, S1["log_sf"], S1["valid"], S1["Xw"], d0, S1["max_dist"], S1["min_dist"], R1w, t1w,
                                         O2, S2["log_sf"], S2["valid"], S2["Xw"], d1, S2["max_dist"], S2["min_dist"], R2w, t2w,
                                         float(s12), R12, t12, cam, 7.5)
    assert n_gpu == n_ref and n_ref > 100
    assert np.array_equal(m_gpu, m_ref)


def test_compute_distinctive_descriptors_n3(pkg, oracle, synth, matcher):
    """MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:350-436): least-median descriptor of each observation group,
    incl. N = 1, 2 (median index 0), even / odd N, ties (first wins), a group above 64 and one above 256 observations."""
    rng = np.random.default_rng(53)
    base, _ = synth.make_descriptor_sets(5300, n=400)
    groups = []
    for n in [1, 2, 3, 4, 7, 16, 33, 64, 65, 130, 300] + list(rng.integers(1, 40, size=60)):
        centre = base[rng.integers(0, len(base))]
        g = np.repeat(centre[None, :], n, axis=0).copy()
        flips = rng.random((n, 256)) < rng.uniform(0.02, 0.2)
        g ^= np.packbits(flips, axis=1, bitorder="little")
        if n > 3 and rng.random() < 0.5:
            g[n // 2] = g[0]                                          # exact duplicates -> tied medians
        groups.append(g)
    groups.append(np.zeros((0, 32), np.uint8))                        # a map point without usable observations
    got = matcher.ComputeDistinctiveDescriptors(groups)
    ref = np.array([oracle.distinctive_descriptor(g) if len(g) else -1 for g in groups], np.int32)
    assert np.array_equal(got, ref)


def test_knn_match2(pkg, matcher, synth):
    """BFMatcher(NORM_HAMMING).knnMatch(k=2) (Frame.cc:1246): checked against numpy on the full distance matrix."""
    q, c = synth.make_descriptor_sets(5400, n=700)
    c = np.concatenate([c, c[:5]])                                    # duplicates: tied distances, ordered by train index
    idx, dist = matcher.knnMatch2(q, c)
    D = np.unpackbits(q[:, None, :] ^ c[None, :, :], axis=2).sum(axis=2).astype(np.int64)
    order = np.argsort(D * (1 << 20) + np.arange(len(c))[None, :], axis=1)[:, :2]
    assert np.array_equal(idx, order.astype(np.int32))
    assert np.array_equal(dist, np.take_along_axis(D, order, axis=1).astype(np.int32))
    i1, d1 = matcher.knnMatch2(q[:3], c[:1])
    assert (i1[:, 0] == 0).all() and (i1[:, 1] == -1).all() and (d1[:, 1] == -1).all()


@pytest.mark.parametrize("nodes", [128, 16])
def test_search_by_bow_fisheye_n3(pkg, oracle, synth, nodes):
    """SearchByBoW(KeyFrame*, Frame&) with a fisheye-stereo frame (ORBmatcher.cc:338-363, :405-436): a keyframe keypoint takes the
    best left-image keypoint under the ratio test and, whenever its left best is within TH_LOW, also the best right-image one."""
    frames, offs = synth.make_stream(3880, 3)
    o = oracle.OracleExtractor(**EUROC)
    (_, k0, d0), (_, kl, dl), (_, kr, dr) = o.extract(frames[0]), o.extract(frames[1]), o.extract(frames[2])
    sf = o.scale_factors
    sigma2 = (sf * sf).astype(np.float32)
    rng = np.random.default_rng(nodes + 7)
    mp0 = (rng.random(len(k0)) < 0.85).astype(np.uint8)
    kf_keys, kf_desc = k0, d0
    f_keys, f_desc = np.concatenate([kl, kr]), np.concatenate([dl, dr])
    fv0, fv1 = _bow(kf_desc, nodes), _bow(f_desc, nodes)
    KF = pkg.KeyFrameView(kf_keys, kf_desc, fv0, sf, sigma2, has_mappoint=mp0)
    F = pkg.KeyFrameView(f_keys, f_desc, fv1, sf, sigma2)
    OKF = oracle.OracleKeyFrame(kf_keys, kf_desc, fv0, sf, sigma2, has_mp=mp0)
    OF = oracle.OracleKeyFrame(f_keys, f_desc, fv1, sf, sigma2)
    m = pkg.ORBmatcher(0.7, True)
    try:
        n_gpu, m_gpu = m.SearchByBoW(KF, F, n_left=len(kl))
        n_ref, m_ref = oracle.search_by_bow(OKF, OF, 0.7, True, n_left=len(kl))
        assert n_gpu == n_ref and n_ref > 150
        assert np.array_equal(m_gpu, m_ref)
        assert (m_ref[len(kl):] >= 0).sum() > 50 and (m_ref[:len(kl)] >= 0).sum() > 50     # both images receive matches
    finally:
        m.close()


@pytest.mark.parametrize("npairs", [2, 12])
def test_batch_pairs_vote_independently(pkg, oracle, synth, npairs):
    """orbm_search_by_projection_batch_device with frame pairs of BOTH kinds in one launch: even pairs search tracking-sized
    windows (the device vote sends them to k_match_walk and the wide resolve), odd pairs a window covering the frame
    (k_match_scan, chunked resolve).  Every pair must equal the oracle run on that pair alone: the vote is taken per pair and must
    come out the same in the walk, the scan, the list merge (2 pairs: the sliced scan of few-pair launches) and the resolve."""
    import ctypes as C
    import torch
    frames, offs = synth.make_stream(4200, npairs + 1)
    o = oracle.OracleExtractor(**EUROC)
    ext = [o.extract(f)[1:] for f in frames]
    sf = np.asarray(o.scale_factors, dtype=np.float32)
    cap = max(len(k) for k, _ in ext) + 5
    kp = np.zeros((npairs + 1, cap, 7), dtype=np.float32)
    de = np.zeros((npairs + 1, cap, 32), dtype=np.uint8)
    cnt = np.zeros((npairs + 1, 2), dtype=np.int32)
    for i, (k, d) in enumerate(ext):
        n = len(k)
        kp[i, :n] = np.ascontiguousarray(k).view(np.float32).reshape(n, 7)
        de[i, :n] = d
        cnt[i, 0] = n
    u = np.zeros((npairs, cap), np.float32); v = np.zeros((npairs, cap), np.float32); rad = np.zeros((npairs, cap), np.float32)
    lo = np.zeros((npairs, cap), np.int32); hi = np.zeros((npairs, cap), np.int32)
    for p in range(npairs):                  # queries = frame p's keypoints, searched in frame p + 1
        k = ext[p][0]; n = len(k)
        u[p, :n] = k["x"] + np.float32(offs[p][0] - offs[p + 1][0]); v[p, :n] = k["y"] + np.float32(offs[p][1] - offs[p + 1][1])
        lvl = k["octave"].astype(np.int32)
        if p % 2 == 0:
            rad[p, :n] = 15.0 * sf[lvl]; lo[p, :n] = lvl - 1; hi[p, :n] = lvl + 1
        else:
            rad[p, :n] = 1.0e4; lo[p, :n] = -1; hi[p, :n] = -1
    dev = "cuda"
    t = lambda a: torch.from_numpy(a).to(dev)
    d_kp, d_de, d_cnt, d_u, d_v, d_r, d_lo, d_hi = t(kp), t(de), t(cnt), t(u), t(v), t(rad), t(lo), t(hi)
    slot = torch.full((npairs, cap), -1, dtype=torch.int32, device=dev); sobs = torch.zeros((npairs, cap), dtype=torch.uint8, device=dev)
    moq = torch.full((npairs, cap), -7, dtype=torch.int32, device=dev); bd = torch.zeros((npairs, cap), dtype=torch.int32, device=dev)
    nm = torch.zeros((npairs,), dtype=torch.int32, device=dev)
    m = pkg.ORBmatcher(0.8, True)
    try:
        fs = pkg.FrameStruct(cap, d_kp[1:].data_ptr(), d_de[1:].data_ptr(), None, 0.0, 752.0, 0.0, 480.0)
        qs = pkg.QueryStruct(cap, d_de.data_ptr(), d_u.data_ptr(), d_v.data_ptr(), d_r.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(), None, None)
        rc = m.L.orbm_search_by_projection_batch_device(m.m, C.byref(fs), cap, C.c_void_p(d_cnt[1:].data_ptr()), 2, C.byref(qs), cap,
                                                        C.c_void_p(d_cnt.data_ptr()), 2, npairs, C.c_float(0.8), 100, 1,
                                                        C.c_void_p(slot.data_ptr()), C.c_void_p(sobs.data_ptr()), C.c_void_p(moq.data_ptr()),
                                                        C.c_void_p(bd.data_ptr()), C.c_void_p(nm.data_ptr()), None)
        assert rc == 0, m.L.orbm_last_error(m.m)
        torch.cuda.synchronize()
        moq_h, bd_h, nm_h, slot_h = moq.cpu().numpy(), bd.cpu().numpy(), nm.cpu().numpy(), slot.cpu().numpy()
        for p in range(npairs):
            (k0, d0), (k1, d1) = ext[p], ext[p + 1]
            n0, n1 = len(k0), len(k1)
            OF = oracle.OracleFrame(k1["x"], k1["y"], k1["octave"], k1["angle"], d1, (0.0, 752.0, 0.0, 480.0), o.scale_factors)
            n_ref, moq_ref, bd_ref = OF.search_by_projection_win(d0, u[p, :n0], v[p, :n0], rad[p, :n0], lo[p, :n0], hi[p, :n0], 0.8, 100, True)
            assert nm_h[p] == n_ref and n_ref > 200, "pair %d" % p
            assert np.array_equal(moq_h[p, :n0], moq_ref) and np.array_equal(bd_h[p, :n0], bd_ref), "pair %d" % p
            assert np.array_equal(slot_h[p, :n1], OF.slot), "pair %d" % p
    finally:
        m.close()
