"""GPU parity tests of the projection searches on the headline configuration's REAL frame geometry (G0 -> G1).

EuRoC is a pinhole camera WITH distortion (Examples/Monocular/EuRoC.yaml:14-17, k1 = -0.283): the Frame constructor
undistorts the keypoints (Frame::UndistortKeyPoints, Frame.cc:837-870), takes the image bounds from the undistorted
corners (Frame::ComputeImageBounds, :872-899: negative mnMinX / mnMinY, a non-integer cell width, :379-380) and files the
UNDISTORTED keypoints into the 64x48 grid (:815-825, some are rejected by PosInGrid).  Every other match test uses the
image rectangle as bounds; here the frames are built the way the reference builds them, on both sides:

    keys --orbm_undistort_keypoints--> keys_un,   orbm_image_bounds --> (min_x, max_x, min_y, max_y)

and the oracle gets its own undistortion / bounds (asserted bit-equal first).  All three candidate enumerations (device vote,
forced scan, forced grid-window walk), single-frame calls (four lanes per query in the walk) and batch calls (one lane)."""
import ctypes as C

import numpy as np
import pytest

from conftest import EUROC
from test_undistort import D as EUROC_D, K as EUROC_K, test_undistort_matches_oracle_and_forward_model as _undistort_check

pytestmark = pytest.mark.gpu

W, H = 752, 480


@pytest.fixture(scope="module", params=[0, 1, 2], ids=["auto", "scan", "walk"])
def matcher(pkg, request):
    m = pkg.ORBmatcher(0.8, True)
    m.set_scan_mode(request.param)
    yield m
    m.close()


def test_undistort_host_functions_on_the_gpu_box(pkg, oracle):
    """tests/test_undistort.py again under the gpu marker: the product library's host functions as built on the GPU box."""
    _undistort_check(pkg, oracle)


def border_keypoints(pkg, keys, desc, rng, n_extra=24):
    """A few extra keypoints on the image border and in its corners (copies of real ones, moved): their undistorted positions
    lie within half a grid cell of - or beyond - the bounds, which is where PosInGrid rejects (Frame.cc:821-822)."""
    src = rng.integers(0, len(keys), n_extra)
    ex = keys[src].copy()
    pos = [(0.0, 0.0), (W - 1.0, H - 1.0), (W - 0.5, H - 0.5), (W - 1.0, 0.5), (0.5, H - 1.0), (W - 0.25, 240.0), (376.0, H - 0.25)]
    for i in range(n_extra):
        if i < len(pos):
            ex["x"][i], ex["y"][i] = pos[i]
        else:   # anywhere on the outermost 3 pixels
            side = rng.integers(0, 4)
            t = rng.uniform(0, 1)
            ex["x"][i] = [t * (W - 1), t * (W - 1), rng.uniform(0, 3), W - 1 - rng.uniform(0, 3)][side]
            ex["y"][i] = [rng.uniform(0, 3), H - 1 - rng.uniform(0, 3), t * (H - 1), t * (H - 1)][side]
    return np.concatenate([keys, ex]), np.concatenate([desc, desc[src]])


def euroc_frame(pkg, oracle, keys, desc, sf, u_right=None, K=EUROC_K, D=EUROC_D):
    """(FrameView, OracleFrame, keys_un, bounds) built as the Frame constructor builds them; product and oracle each use their own
    undistortion and bounds, which must agree to the bit."""
    keys_un = pkg.undistort_keypoints(keys, K, D)
    ref = oracle.undistort_points(np.stack([keys["x"], keys["y"]], axis=1), K, D)
    assert np.array_equal(keys_un["x"].view(np.uint32), ref[:, 0].view(np.uint32)) and np.array_equal(keys_un["y"].view(np.uint32), ref[:, 1].view(np.uint32))
    bounds = pkg.image_bounds(W, H, K, D)
    assert bounds == oracle.image_bounds(W, H, K, D)
    F = pkg.FrameView(keys_un, desc, bounds, u_right=u_right)
    OF = oracle.OracleFrame(ref[:, 0], ref[:, 1], keys["octave"], keys["angle"], desc, bounds, sf, u_right=u_right)
    return F, OF, keys_un, bounds


def grid_facts(keys_un, bounds):
    """(keypoints PosInGrid rejects, the grid's inverse cell sizes) computed here in plain numpy float32 (Frame.cc:379-380, :815-825)."""
    iw = np.float32(64) / (np.float32(bounds[1]) - np.float32(bounds[0]))
    ih = np.float32(48) / (np.float32(bounds[3]) - np.float32(bounds[2]))
    gx = np.round((keys_un["x"] - np.float32(bounds[0])) * iw).astype(np.int64)
    gy = np.round((keys_un["y"] - np.float32(bounds[2])) * ih).astype(np.int64)
    out = (gx < 0) | (gx >= 64) | (gy < 0) | (gy >= 48)
    return out, float(iw), float(ih)


def extract_pair(oracle, synth, seed):
    frames, offs = synth.make_stream(seed, 2)
    o = oracle.OracleExtractor(**EUROC)
    _, k0, d0 = o.extract(frames[0])
    _, k1, d1 = o.extract(frames[1])
    return k0, d0, k1, d1, offs, o.scale_factors


def test_bounds_are_the_headline_geometry(pkg):
    b = pkg.image_bounds(W, H, EUROC_K, EUROC_D)
    assert b[0] < -20 and b[2] < -20 and b[1] > W + 20 and b[3] > H + 20          # negative origin, grid larger than the image
    iw = 64.0 / (b[1] - b[0])
    assert abs(1.0 / iw - round(1.0 / iw)) > 0.05                                  # non-integer cell width


@pytest.mark.parametrize("seed", [4100, 4101])
def test_search_by_projection_m2_distorted(pkg, oracle, synth, matcher, seed):
    """ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th) (ORBmatcher.cc:44-143) against a frame with undistorted
    keypoints and undistorted-corner bounds; some keypoints outside the grid, some windows clipped at cell 0."""
    k0, d0, k1, d1, offs, sf = extract_pair(oracle, synth, seed)
    rng = np.random.default_rng(seed)
    k1, d1 = border_keypoints(pkg, k1, d1, rng)
    F, OF, k1u, bounds = euroc_frame(pkg, oracle, k1, d1, sf)
    out, iw, ih = grid_facts(k1u, bounds)
    assert out.sum() >= 1, "no keypoint outside the grid: the test lost its point"
    k0u = pkg.undistort_keypoints(k0, EUROC_K, EUROC_D)
    nq = len(k0)
    projX = (k0u["x"] + np.float32(offs[0][0] - offs[1][0])).astype(np.float32)
    projY = (k0u["y"] + np.float32(offs[0][1] - offs[1][1])).astype(np.float32)
    # a few map points projected right onto the rejected keypoints and into the grid's first / last cells
    tgt = np.nonzero(out)[0][:8]
    projX[:len(tgt)] = k1u["x"][tgt]; projY[:len(tgt)] = k1u["y"][tgt]
    projX[8:12] = np.float32(bounds[0]) + np.float32([0.3, 1.0, 5.0, 9.0]); projY[8:12] = np.float32(bounds[2]) + np.float32([0.3, 2.0, 4.0, 8.0])
    viewCos = rng.choice(np.array([0.9, 0.9985, 1.0], dtype=np.float32), nq)
    level = k0["octave"].astype(np.int32)
    in_view = (rng.random(nq) < 0.9).astype(np.uint8); in_view[:12] = 1
    obs = (rng.random(nq) < 0.85).astype(np.uint8)
    # windows that reach below cell 0 of the (negative-origin) grid exist
    r = 4.0 * 3.0 * sf[np.clip(level, 0, 7)]
    assert (np.floor((projX - np.float32(bounds[0]) - r) * iw) < 0).any()
    total = 0
    for th in (1.0, 3.0):
        F.slot[:] = -1; F.slot_obs[:] = 0; OF.slot[:] = -1; OF.slot_obs[:] = 0
        n_gpu, moq_gpu, _ = matcher.SearchByProjection(F, in_view, d0, projX, projY, viewCos, level, sf, th=th, mp_obs=obs)
        n_ref, moq_ref = OF.search_by_projection_mp(in_view, d0, projX, projY, viewCos, level, th, 0.8, qobs=obs)
        assert n_gpu == n_ref
        assert np.array_equal(moq_gpu, moq_ref)
        assert np.array_equal(F.slot, OF.slot) and np.array_equal(F.slot_obs, OF.slot_obs)
        assert not np.isin(moq_ref[moq_ref >= 0], np.nonzero(out)[0]).any()        # a keypoint outside the grid is never a candidate
        total += n_ref
    assert total > 300


def test_search_stress_1000x1000_distorted(pkg, oracle, synth, matcher):
    """BASELINE config 3 stress setting (window = whole image, levels open, nnratio 0.8, TH_HIGH 100, sequential claims) on the
    distorted geometry: the open-window path of the scan must see exactly the keypoints PosInGrid accepts."""
    k0, d0, k1, d1, offs, sf = extract_pair(oracle, synth, 4200)
    rng = np.random.default_rng(7)
    k1, d1 = border_keypoints(pkg, k1, d1, rng, n_extra=40)
    # the rejected keypoints carry descriptors that WOULD be best matches: copies of query descriptors
    F, OF, k1u, bounds = euroc_frame(pkg, oracle, k1, d1, sf)
    out, _, _ = grid_facts(k1u, bounds)
    assert out.sum() >= 1
    d1 = d1.copy(); d1[np.nonzero(out)[0]] = d0[:int(out.sum())]
    F, OF, k1u, bounds = euroc_frame(pkg, oracle, k1, d1, sf)
    k0u = pkg.undistort_keypoints(k0, EUROC_K, EUROC_D)
    nq = len(k0)
    u = (k0u["x"] + np.float32(offs[0][0] - offs[1][0])).astype(np.float32)
    v = (k0u["y"] + np.float32(offs[0][1] - offs[1][1])).astype(np.float32)
    ml = np.full(nq, -1, np.int32)
    for radius in (1.0e4, 900.0):       # 900: the window covers the grid's cells but not "one cell beyond the bounds" for every query
        rad = np.full(nq, radius, np.float32)
        F.slot[:] = -1; F.slot_obs[:] = 0; OF.slot[:] = -1; OF.slot_obs[:] = 0
        n_gpu, moq_gpu, bd_gpu = matcher.search_window(F, d0, u, v, rad, ml, ml, nnratio=0.8, th_dist=100, use_second=True)
        n_ref, moq_ref, bd_ref = OF.search_by_projection_win(d0, u, v, rad, ml, ml, 0.8, 100, True)
        assert n_gpu == n_ref and n_ref > 300
        assert np.array_equal(moq_gpu, moq_ref) and np.array_equal(bd_gpu, bd_ref)
        assert np.array_equal(F.slot, OF.slot)
        assert (F.slot[out] == -1).all()


@pytest.mark.parametrize("stereo", [False, True])
def test_search_by_projection_last_frame_m3_distorted(pkg, oracle, synth, matcher, stereo):
    """ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) (ORBmatcher.cc:2027-2289): Pinhole projection (the
    undistorted model), bounds test against the undistorted-corner bounds (:2094-2099), search among undistorted keypoints."""
    k0, d0, k1, d1, offs, sf = extract_pair(oracle, synth, 4300 + int(stereo))
    rng = np.random.default_rng(4300)
    k1, d1 = border_keypoints(pkg, k1, d1, rng)
    fx, fy, cx, cy = [float(x) for x in EUROC_K]
    k0u = pkg.undistort_keypoints(k0, EUROC_K, EUROC_D)
    n0 = len(k0)
    z = np.float32(5.0)
    Xw = np.stack([(k0u["x"] - np.float32(cx)) / np.float32(fx) * z, (k0u["y"] - np.float32(cy)) / np.float32(fy) * z, np.full(n0, z, np.float32)], axis=1).astype(np.float32)
    Xw[rng.random(n0) < 0.03, 2] = -1.0
    # some map points that project between the image rectangle and the (larger) undistorted bounds: inside for the reference
    edge = rng.permutation(n0)[:30]
    Xw[edge, 0] = (np.float32(-15.0) - np.float32(cx)) / np.float32(fx) * z
    dx, dy = offs[0][0] - offs[1][0], offs[0][1] - offs[1][1]
    ang = 0.002
    Tcw = np.eye(4, dtype=np.float32)
    Tcw[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    Tcw[:3, 3] = [dx * z / fx, dy * z / fy, -0.3 if stereo else 0.0]
    Tlw = np.eye(4, dtype=np.float32)
    has_mp = (rng.random(n0) < 0.8).astype(np.uint8); has_mp[edge] = 1
    obs = (rng.random(n0) < 0.9).astype(np.uint8)
    params = EUROC_K.copy()
    u_right = None
    if stereo:
        k1u_tmp = pkg.undistort_keypoints(k1, EUROC_K, EUROC_D)
        u_right = np.where(rng.random(len(k1)) < 0.7, k1u_tmp["x"] - np.float32(47.9) / z, np.float32(-1)).astype(np.float32)
    total = 0
    for th in (15.0, 30.0):
        F, OF, k1u, bounds = euroc_frame(pkg, oracle, k1, d1, sf, u_right=u_right)
        mb, mbf = (0.11, 47.9) if stereo else (0.0, 0.0)
        n_gpu = matcher.SearchByProjectionLastFrame(F, sf, has_mp, Xw, d0, k0, Tcw, Tlw, 0, params, th, bMono=not stereo, mb=mb, mbf=mbf, mp_obs=obs)
        n_ref = OF.search_by_projection_ff(has_mp, Xw, d0, k0["octave"], k0["angle"], Tcw, Tlw, 0, params, th, mono=not stereo,
                                           check_ori=True, mb=mb, mbf=mbf, qobs=obs)
        assert n_gpu == n_ref
        assert np.array_equal(F.slot, OF.slot) and np.array_equal(F.slot_obs, OF.slot_obs)
        total += n_ref
    assert total > 100


@pytest.mark.parametrize("nfeatures,window", [(1000, 100), (5000, 100)])
def test_search_for_initialization_n2_distorted(pkg, oracle, synth, nfeatures, window):
    """ORBmatcher::SearchForInitialization (ORBmatcher.cc:722-837): both frames undistorted, F2's grid on the distorted bounds,
    vbPrevMatched = F1's undistorted keypoints (Tracking.cc:2441-2443)."""
    frames, offs = synth.make_stream(4400 + nfeatures, 2)
    cfg = dict(EUROC); cfg["nfeatures"] = nfeatures
    o = oracle.OracleExtractor(**cfg)
    (_, k0, d0), (_, k1, d1) = o.extract(frames[0]), o.extract(frames[1])
    rng = np.random.default_rng(nfeatures)
    k1, d1 = border_keypoints(pkg, k1, d1, rng)
    k0u = pkg.undistort_keypoints(k0, EUROC_K, EUROC_D)
    m = pkg.ORBmatcher(0.9, True)
    try:
        prev_gpu = np.stack([k0u["x"], k0u["y"]], axis=1).astype(np.float32).copy()
        prev_ref = prev_gpu.copy()
        total = 0
        for rnd in range(2):
            F2, OF2, k1u, bounds = euroc_frame(pkg, oracle, k1, d1, o.scale_factors)
            F1 = pkg.FrameView(k0u, d0, bounds)
            n_gpu, m_gpu = m.SearchForInitialization(F1, F2, prev_gpu, window)
            n_ref, m_ref = oracle.search_for_initialization(k0u, d0, OF2, prev_ref, window, 0.9, True)
            assert n_gpu == n_ref
            assert np.array_equal(m_gpu, m_ref)
            assert np.array_equal(prev_gpu, prev_ref)
            total += n_ref
        assert total > 50
    finally:
        m.close()


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["auto", "scan", "walk"])
def test_batch_device_distorted(pkg, oracle, synth, mode):
    """The batch entry points on the distorted geometry, 12 frame pairs per launch (one lane per query in k_match_walk; the
    single-frame tests above run it with four): orbm_undistort_keypoints_batch_device on the extractor's device output, then
    orbm_search_by_projection_batch_device with the undistorted-corner bounds.  Even pairs search tracking-sized windows, odd
    pairs the whole frame.  Checked per pair against the oracle (and the device undistortion against the host function)."""
    import torch
    npairs = 12
    frames, offs = synth.make_stream(4500, npairs + 1)
    o = oracle.OracleExtractor(**EUROC)
    ext = [o.extract(f)[1:] for f in frames]
    sf = np.asarray(o.scale_factors, dtype=np.float32)
    rng = np.random.default_rng(45)
    ext = [border_keypoints(pkg, k, d, rng, n_extra=12) for k, d in ext]
    cap = max(len(k) for k, _ in ext) + 5
    kp = np.zeros((npairs + 1, cap, 7), dtype=np.float32)
    de = np.zeros((npairs + 1, cap, 32), dtype=np.uint8)
    cnt = np.zeros((npairs + 1, 2), dtype=np.int32)
    for i, (k, d) in enumerate(ext):
        n = len(k)
        kp[i, :n] = np.ascontiguousarray(k).view(np.float32).reshape(n, 7)
        de[i, :n] = d
        cnt[i, 0] = n
    dev = "cuda"
    t = lambda a: torch.from_numpy(a).to(dev)
    d_kp, d_de, d_cnt = t(kp), t(de), t(cnt)
    d_kpu = torch.zeros_like(d_kp)
    m = pkg.ORBmatcher(0.8, True)
    m.set_scan_mode(mode)
    try:
        m.undistort_batch_device(d_kp.data_ptr(), cap, d_cnt.data_ptr(), 2, npairs + 1, EUROC_K, EUROC_D, d_kpu.data_ptr())
        torch.cuda.synchronize()
        kpu = d_kpu.cpu().numpy()
        ext_u = []
        for i, (k, d) in enumerate(ext):
            n = len(k)
            ku = pkg.undistort_keypoints(k, EUROC_K, EUROC_D)
            assert kpu[i, :n].tobytes() == np.ascontiguousarray(ku).view(np.float32).reshape(n, 7).tobytes(), "device undistortion differs from the host function, frame %d" % i
            assert not kpu[i, n:].any()
            ext_u.append(ku)
        bounds = pkg.image_bounds(W, H, EUROC_K, EUROC_D)
        u = np.zeros((npairs, cap), np.float32); v = np.zeros((npairs, cap), np.float32); rad = np.zeros((npairs, cap), np.float32)
        lo = np.zeros((npairs, cap), np.int32); hi = np.zeros((npairs, cap), np.int32)
        for p in range(npairs):
            k = ext_u[p]; n = len(k)
            u[p, :n] = k["x"] + np.float32(offs[p][0] - offs[p + 1][0]); v[p, :n] = k["y"] + np.float32(offs[p][1] - offs[p + 1][1])
            lvl = k["octave"].astype(np.int32)
            if p % 2 == 0:
                rad[p, :n] = 15.0 * sf[lvl]; lo[p, :n] = lvl - 1; hi[p, :n] = lvl + 1
            else:
                rad[p, :n] = 1.0e4; lo[p, :n] = -1; hi[p, :n] = -1
        d_u, d_v, d_r, d_lo, d_hi = t(u), t(v), t(rad), t(lo), t(hi)
        slot = torch.full((npairs, cap), -1, dtype=torch.int32, device=dev); sobs = torch.zeros((npairs, cap), dtype=torch.uint8, device=dev)
        moq = torch.full((npairs, cap), -7, dtype=torch.int32, device=dev); bd = torch.zeros((npairs, cap), dtype=torch.int32, device=dev)
        nm = torch.zeros((npairs,), dtype=torch.int32, device=dev)
        fs = pkg.FrameStruct(cap, d_kpu[1:].data_ptr(), d_de[1:].data_ptr(), None, *bounds)
        qs = pkg.QueryStruct(cap, d_de.data_ptr(), d_u.data_ptr(), d_v.data_ptr(), d_r.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(), None, None)
        rc = m.L.orbm_search_by_projection_batch_device(m.m, C.byref(fs), cap, C.c_void_p(d_cnt[1:].data_ptr()), 2, C.byref(qs), cap,
                                                        C.c_void_p(d_cnt.data_ptr()), 2, npairs, C.c_float(0.8), 100, 1,
                                                        C.c_void_p(slot.data_ptr()), C.c_void_p(sobs.data_ptr()), C.c_void_p(moq.data_ptr()),
                                                        C.c_void_p(bd.data_ptr()), C.c_void_p(nm.data_ptr()), None)
        assert rc == 0, m.L.orbm_last_error(m.m)
        torch.cuda.synchronize()
        moq_h, bd_h, nm_h, slot_h = moq.cpu().numpy(), bd.cpu().numpy(), nm.cpu().numpy(), slot.cpu().numpy()
        n_out = 0
        for p in range(npairs):
            (k0, d0), (k1, d1) = ext[p], ext[p + 1]
            k1u = ext_u[p + 1]
            n0, n1 = len(k0), len(k1)
            n_out += int(grid_facts(k1u, bounds)[0].sum())
            OF = oracle.OracleFrame(k1u["x"], k1u["y"], k1u["octave"], k1u["angle"], d1, bounds, o.scale_factors)
            n_ref, moq_ref, bd_ref = OF.search_by_projection_win(d0, u[p, :n0], v[p, :n0], rad[p, :n0], lo[p, :n0], hi[p, :n0], 0.8, 100, True)
            assert nm_h[p] == n_ref and n_ref > 200, "pair %d" % p
            assert np.array_equal(moq_h[p, :n0], moq_ref) and np.array_equal(bd_h[p, :n0], bd_ref), "pair %d" % p
            assert np.array_equal(slot_h[p, :n1], OF.slot), "pair %d" % p
        assert n_out >= npairs
    finally:
        m.close()


def test_random_bounds_and_distortions(pkg, oracle, synth, matcher):
    """Bounds with a random negative origin and an extent beyond the image, distortion coefficients of either sign (pincushion
    moves border keypoints OUT of the corner-defined bounds): the window search on each against the oracle."""
    k0, d0, k1, d1, offs, sf = extract_pair(oracle, synth, 4600)
    rng = np.random.default_rng(46)
    k1b, d1b = border_keypoints(pkg, k1, d1, rng, n_extra=30)
    n_rejected = 0
    for case in range(6):
        D = np.array([rng.uniform(-0.35, 0.25), rng.uniform(-0.1, 0.1), rng.uniform(-1e-3, 1e-3), rng.uniform(-1e-3, 1e-3)], np.float32)
        if case == 0:
            D[0] = np.float32(0.2)      # pincushion: edge midpoints leave the bounds
        F, OF, k1u, bounds = euroc_frame(pkg, oracle, k1b, d1b, sf, D=D)
        if case >= 3:                   # arbitrary bounds (not from the corners): origin in [-80, 0], extent >= the image
            bounds = (float(np.float32(rng.uniform(-80, 0))), float(np.float32(W + rng.uniform(0, 80))), float(np.float32(rng.uniform(-80, 0))), float(np.float32(H + rng.uniform(0, 80))))
            F = pkg.FrameView(k1u, d1b, bounds)
            OF = oracle.OracleFrame(k1u["x"], k1u["y"], k1u["octave"], k1u["angle"], d1b, bounds, sf)
        n_rejected += int(grid_facts(k1u, bounds)[0].sum())
        k0u = pkg.undistort_keypoints(k0, EUROC_K, D)
        nq = len(k0)
        u = (k0u["x"] + np.float32(offs[0][0] - offs[1][0])).astype(np.float32)
        v = (k0u["y"] + np.float32(offs[0][1] - offs[1][1])).astype(np.float32)
        lvl = k0["octave"].astype(np.int32)
        radius = rng.choice(np.array([3, 15, 40, 200, 1e4], np.float32), nq).astype(np.float32)
        n_gpu, moq_gpu, bd_gpu = matcher.search_window(F, d0, u, v, radius, lvl - 1, lvl + 1, nnratio=0.8, th_dist=100, use_second=True)
        n_ref, moq_ref, bd_ref = OF.search_by_projection_win(d0, u, v, radius, lvl - 1, lvl + 1, 0.8, 100, True)
        assert n_gpu == n_ref and n_ref > 100, "case %d D=%s bounds=%s" % (case, D, bounds)
        assert np.array_equal(moq_gpu, moq_ref) and np.array_equal(bd_gpu, bd_ref), "case %d" % case
        assert np.array_equal(F.slot, OF.slot), "case %d" % case
    assert n_rejected > 0
