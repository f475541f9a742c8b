"""GPU parity tests of the extractor: HIP path (through the C ABI) vs the CPU oracle on the same seeded frames.
Bit-exact: every comparison is array_equal on bytes / float32 bit patterns."""
import numpy as np
import pytest

from conftest import EUROC, TUMVI

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ex(pkg):
    e = pkg.ORBextractor(**EUROC)
    yield e
    e.close()


@pytest.fixture(scope="module")
def oex(oracle):
    return oracle.OracleExtractor(**EUROC)


def assert_kps_equal(k1, k2):
    assert len(k1) == len(k2)
    for f in k1.dtype.names:
        a, b = k1[f], k2[f]
        assert np.array_equal(a.view(np.uint32) if a.dtype.kind == "f" else a, b.view(np.uint32) if b.dtype.kind == "f" else b), f


@pytest.mark.parametrize("seed", [1000, 1001])
def test_stage_parity(ex, oex, oracle, frame, seed):
    img = frame(seed)
    mono, kps, desc = ex(img, None, (0, 1000))
    pyr = oex.pyramid(img)
    q = oex.features_per_level
    for l in range(8):
        assert ex.level_shape(l) == pyr[l].shape
        assert np.array_equal(ex.image_pyramid_level(l), pyr[l]), "pyramid level %d" % l
        assert np.array_equal(ex.blurred_level(l), oracle.gaussian_blur7(pyr[l])), "blur level %d" % l
        c_ref = oex.level_candidates(pyr[l])
        c_gpu = ex.level_candidates(l)
        assert np.array_equal(c_gpu, c_ref), "FAST candidates level %d" % l
        h, w = pyr[l].shape
        k_ref = oracle.distribute_octtree(c_ref, 16, w - 16, 16, h - 16, q[l])
        k_gpu = ex.level_keypoints(l)
        assert np.array_equal(k_gpu, k_ref), "octree level %d" % l


def test_pyramid_border(ex, oex, oracle, frame):
    """mvImagePyramid with its 19-px BORDER_REFLECT_101 frame (ORBextractor.cc:1203-1215)."""
    img = frame(1000)
    ex(img)
    pyr = oex.pyramid(img)
    for l in (0, 3, 7):
        h, w = pyr[l].shape
        ref = np.zeros((h + 38, w + 38), dtype=np.uint8)
        import ctypes as C
        oracle.lib().orc_copy_make_border101(pyr[l].ctypes.data_as(C.c_void_p), w, h, C.c_size_t(w), ref.ctypes.data_as(C.c_void_p), 19, C.c_size_t(w + 38))
        assert np.array_equal(ex.image_pyramid_level(l, border=19), ref)
    # every level in one transfer (orbx_download_pyramid: what the adapter fills mvImagePyramid from)
    for border in (19, 0):
        packed = ex.image_pyramid(border=border)
        assert len(packed) == len(pyr)
        for l in range(len(pyr)):
            assert np.array_equal(packed[l], ex.image_pyramid_level(l, border=border)), "level %d border %d" % (l, border)


@pytest.mark.parametrize("seed", [1000, 1001, 1002, 1003, 1004])
@pytest.mark.parametrize("lap", [(0, 1000), (0, 0)])
def test_extract_parity_euroc(ex, oex, frame, seed, lap):
    img = frame(seed)
    mono, kps, desc = ex(img, None, lap)
    mono_r, kps_r, desc_r = oex.extract(img, lap)
    assert mono == mono_r
    assert_kps_equal(kps, kps_r)
    assert np.array_equal(desc, desc_r)
    assert len(kps) >= 1000


@pytest.mark.parametrize("seed", [2000, 2001])
def test_extract_parity_tumvi(pkg, oracle, frame, seed):
    img = frame(seed, 512, 512)
    e = pkg.ORBextractor(**TUMVI)
    o = oracle.OracleExtractor(**TUMVI)
    mono, kps, desc = e(img, None, (0, 1000))
    mono_r, kps_r, desc_r = o.extract(img, (0, 1000))
    assert mono == mono_r
    assert_kps_equal(kps, kps_r)
    assert np.array_equal(desc, desc_r)
    e.close()


def test_ini_extractor_5000(pkg, oracle, frame):
    """mpIniORBextractor = 5*nFeatures (Tracking.cc:844): exercises the large-N octree (LDS > 64 KB)."""
    img = frame(1000)
    e = pkg.ORBextractor(5000, 1.2, 8, 20, 7)
    o = oracle.OracleExtractor(5000, 1.2, 8, 20, 7)
    mono, kps, desc = e(img, None, (0, 1000))
    mono_r, kps_r, desc_r = o.extract(img, (0, 1000))
    assert mono == mono_r
    assert_kps_equal(kps, kps_r)
    assert np.array_equal(desc, desc_r)
    e.close()


def test_ini_extractor_10000(pkg, oracle):
    """5 * nFeatures for KITTI's 2000 features (Examples/Monocular/KITTI00-02.yaml): 2172 features on level 0, just inside
    the LDS-resident octree; noise image so that every quota is reached.  12000 is refused loudly."""
    img = np.random.default_rng(10000).integers(0, 256, (376, 1241), dtype=np.uint8)
    e = pkg.ORBextractor(10000, 1.2, 8, 20, 7)
    o = oracle.OracleExtractor(10000, 1.2, 8, 20, 7)
    try:
        mono, kps, desc = e(img, None, (0, 0))
        mono_r, kps_r, desc_r = o.extract(img, (0, 0))
        assert mono == mono_r and len(kps_r) > 9000
        assert_kps_equal(kps, kps_r)
        assert np.array_equal(desc, desc_r)
    finally:
        e.close()
    e2 = pkg.ORBextractor(12000, 1.2, 8, 20, 7)
    try:
        with pytest.raises((pkg.OrbError, ValueError)):
            e2(img, None, (0, 0))
    finally:
        e2.close()


def test_threshold_fallback_cells(pkg, oracle, frame):
    """ORBextractor.cc:820-828: a cell is detected at iniThFAST and, only when that returns nothing, again at minThFAST.  A frame
    whose left third keeps its contrast, whose middle third is compressed to a few grey levels (corners only below iniThFAST)
    and whose right third is flat has cells of all three kinds; k_fast runs its second detection only for the middle ones."""
    img = frame(1003).copy()
    w = img.shape[1]
    mid = img[:, w // 3:2 * w // 3].astype(np.float32)
    img[:, w // 3:2 * w // 3] = np.clip(128.0 + (mid - 128.0) * 0.18, 0, 255).astype(np.uint8)
    img[:, 2 * w // 3:] = 97
    for ini, mn in [(20, 7), (20, 20), (10, 25), (40, 3)]:
        e = pkg.ORBextractor(1000, 1.2, 8, ini, mn)
        o = oracle.OracleExtractor(1000, 1.2, 8, ini, mn)
        try:
            mono, kps, desc = e(img, None, (0, 1000))
            mono_r, kps_r, desc_r = o.extract(img, (0, 1000))
            assert mono == mono_r
            assert_kps_equal(kps, kps_r)
            assert np.array_equal(desc, desc_r)
            pyr = o.pyramid(img)
            weak = strong = 0
            for l in range(8):
                c_ref = o.level_candidates(pyr[l])
                assert np.array_equal(e.level_candidates(l), c_ref), "FAST candidates level %d (ini %d, min %d)" % (l, ini, mn)
                # response = score - 1... a candidate of a first-detection cell has S > iniThFAST
                weak += int((c_ref[:, 2] < ini).sum())
                strong += int((c_ref[:, 2] >= ini).sum())
            assert strong > 100
            if mn < ini:
                assert weak > 100, "the frame must hold cells that only the second detection fills"
            else:
                assert weak == 0
        finally:
            e.close()


def test_lapping_partial(ex, oex, frame):
    img = frame(1002)
    for lap in [(300, 500), (0, 375), (376, 2000)]:
        mono, kps, desc = ex(img, None, lap)
        mono_r, kps_r, desc_r = oex.extract(img, lap)
        assert mono == mono_r and 0 < mono < len(kps)
        assert_kps_equal(kps, kps_r)
        assert np.array_equal(desc, desc_r)


def test_edge_cases(pkg, oracle, ex, oex):
    # empty image -> -1 (ORBextractor.cc:1075-1076)
    mono, kps, desc = ex(np.zeros((0, 0), dtype=np.uint8))
    assert mono == -1 and len(kps) == 0
    # constant image: no corners, zero keypoints, descriptors released (:1100-1101)
    flat = np.full((480, 752), 77, dtype=np.uint8)
    mono, kps, desc = ex(flat)
    mono_r, kps_r, _ = oex.extract(flat)
    assert mono == mono_r == 0 and len(kps) == len(kps_r) == 0
    # non-contiguous rows (cv::Mat step > cols)
    rng = np.random.default_rng(5)
    big = rng.integers(0, 256, (480, 800), dtype=np.uint8)
    view = big[:, 10:10 + 752]
    mono, kps, desc = ex(view)
    mono_r, kps_r, desc_r = oex.extract(np.ascontiguousarray(view))
    assert mono == mono_r
    assert_kps_equal(kps, kps_r)
    assert np.array_equal(desc, desc_r)
    # pure noise (maximum candidate density) and odd sizes
    for (H, W) in [(480, 752), (241, 377), (100, 131)]:
        img = rng.integers(0, 256, (H, W), dtype=np.uint8)
        e = pkg.ORBextractor(**EUROC)
        mono, kps, desc = e(img)
        mono_r, kps_r, desc_r = oex.extract(img)
        assert mono == mono_r
        assert_kps_equal(kps, kps_r)
        assert np.array_equal(desc, desc_r)
        e.close()
    # wrong type
    with pytest.raises(ValueError):
        ex(np.zeros((480, 752), dtype=np.float32))


@pytest.mark.parametrize("B", [6, 131])
def test_batch_device_api(pkg, oex, frame, B):
    """orbx_extract_batch_device: B frames resident in HBM, outputs in HBM, compared frame by frame.
    B = 131 takes the large-batch launch path (whole pyramid of a frame per workgroup)."""
    import torch
    seeds = [1000 + (i % 6) for i in range(B)]
    imgs = np.stack([frame(s) for s in seeds])
    B, H, W = imgs.shape
    e = pkg.ORBextractor(**EUROC)
    cap = e.configure(H, W, B)
    d_img = torch.from_numpy(imgs).cuda()
    d_kps = torch.zeros((B, cap, 7), dtype=torch.int32, device="cuda")
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for lap in [(0, 1000), (0, 0)]:
        e.extract_batch_device(d_img.data_ptr(), H, W, W, H * W, B, d_kps.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap, lap, stream=s)
        torch.cuda.synchronize()
        cnt = d_cnt.cpu().numpy()
        kps_all = d_kps.cpu().numpy()
        desc_all = d_desc.cpu().numpy()
        refs = {}
        for b, seed in enumerate(seeds):
            if seed not in refs:
                refs[seed] = oex.extract(imgs[b], lap)
            mono_r, kps_r, desc_r = refs[seed]
            n = int(cnt[b, 0])
            assert n == len(kps_r) and int(cnt[b, 1]) == mono_r
            kps = kps_all[b, :n].copy().view(pkg.KP_DTYPE).reshape(-1)
            assert_kps_equal(kps, kps_r)
            assert np.array_equal(desc_all[b, :n], desc_r)
    e.close()


def test_full_size_properties(pkg, frame):
    """Size-independent properties at BASELINE's full batch size: determinism (same frames twice -> same bytes),
    frame independence (a frame's result does not depend on its batch neighbours), quota respected."""
    import torch
    B = 64
    base = np.stack([frame(1000 + (i % 8)) for i in range(B)])
    e = pkg.ORBextractor(**EUROC)
    cap = e.configure(480, 752, B)
    d_img = torch.from_numpy(base).cuda()
    outs = []
    for rep in range(2):
        d_kps = torch.zeros((B, cap, 7), dtype=torch.int32, device="cuda")
        d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
        d_cnt = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
        e.extract_batch_device(d_img.data_ptr(), 480, 752, 752, 480 * 752, B, d_kps.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap,
                               (0, 1000), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append((d_kps.cpu().numpy(), d_desc.cpu().numpy(), d_cnt.cpu().numpy()))
    assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
    kps, desc, cnt = outs[0]
    for i in range(8, B):
        n = cnt[i, 0]
        assert n == cnt[i % 8, 0] and np.array_equal(kps[i, :n], kps[i % 8, :n]) and np.array_equal(desc[i, :n], desc[i % 8, :n])
    assert (cnt[:, 0] >= 1000).all() and (cnt[:, 0] <= cap).all()
    e.close()


@pytest.mark.parametrize("cfg", [
    dict(H=376, W=1241, nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7),    # KITTI (Examples/Monocular/KITTI00-02.yaml)
    dict(H=480, W=640, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7),     # TUM RGB-D (TUM1.yaml)
    dict(H=300, W=402, nfeatures=300, scaleFactor=1.5, nlevels=5, iniThFAST=12, minThFAST=4),      # other scale factor / level count
    dict(H=256, W=300, nfeatures=64, scaleFactor=2.0, nlevels=3, iniThFAST=30, minThFAST=30),      # octave pyramid, ini == min threshold
    dict(H=200, W=260, nfeatures=20, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7),       # tiny quotas (some levels get 1-2 features)
    dict(H=130, W=150, nfeatures=500, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7),      # upper levels smaller than one cell
    dict(H=1080, W=1920, nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7),   # full HD: long rows in every staging loop
    dict(H=613, W=1021, nfeatures=1200, scaleFactor=1.3, nlevels=6, iniThFAST=15, minThFAST=5),    # odd sizes: unaligned level-0 rows, ragged row tails
    dict(H=500, W=700, nfeatures=300, scaleFactor=2.6, nlevels=3, iniThFAST=20, minThFAST=7),      # k_resize's one-pass form: a tile's source rows exceed its LDS stage
    dict(H=600, W=900, nfeatures=200, scaleFactor=3.3, nlevels=2, iniThFAST=20, minThFAST=7),      # ... and neighbouring columns more than 3 source bytes apart
])
def test_extract_parity_config_matrix(pkg, oracle, synth, cfg):
    """Geometry / parameter sweep: other datasets' image sizes, scale factors, level counts, thresholds, quotas."""
    cfg = dict(cfg)
    H, W = cfg.pop("H"), cfg.pop("W")
    img = synth.make_frame(4000 + H, H, W)
    e = pkg.ORBextractor(**cfg)
    o = oracle.OracleExtractor(**cfg)
    for lap in [(0, 1000), (0, 0)]:
        mono, kps, desc = e(img, None, lap)
        mono_r, kps_r, desc_r = o.extract(img, lap)
        assert mono == mono_r
        assert_kps_equal(kps, kps_r)
        assert np.array_equal(desc, desc_r)
    nl = cfg["nlevels"]
    pyr = o.pyramid(img)
    for l in range(nl):
        assert np.array_equal(e.image_pyramid_level(l), pyr[l])
    e.close()


@pytest.mark.parametrize("disp", [12, 37])
def test_compute_stereo_matches_n1(pkg, oracle, synth, disp):
    """Frame::ComputeStereoMatches (Frame.cc:901-1079): right image = the same scene shifted by `disp` px, both images
    extracted on the device (their pyramids stay resident), descriptor search + 11x11 SAD + parabola on the device."""
    big = synth.make_frame(4100 + disp, H=480, W=752 + 64)
    imgL = np.ascontiguousarray(big[:, 0:752])                       # scene column c is at xL = c ...
    imgR = np.ascontiguousarray(big[:, disp:disp + 752])             # ... and at xR = c - disp: disparity uL - uR = disp
    exL, exR = pkg.ORBextractor(**EUROC), pkg.ORBextractor(**EUROC)
    try:
        _, kL, dL = exL(imgL, None, (0, 0))
        _, kR, dR = exR(imgR, None, (0, 0))
        mb, mbf = 0.11, 47.9
        uR_gpu, z_gpu = exL.ComputeStereoMatches(exR, kL, dL, kR, dR, mb, mbf)
        o = oracle.OracleExtractor(**EUROC)
        uR_ref, z_ref = o.compute_stereo_matches(imgL, imgR, kL, dL, kR, dR, mb, mbf)
        assert np.array_equal(uR_gpu.view(np.uint32), uR_ref.view(np.uint32))
        assert np.array_equal(z_gpu.view(np.uint32), z_ref.view(np.uint32))
        ok = uR_ref >= 0
        assert ok.sum() > 200
        assert np.median(np.abs((kL["x"][ok] - uR_ref[ok]) - disp)) < 1.0   # the recovered disparity is the shift
    finally:
        exL.close(); exR.close()


def test_compute_stereo_matches_edge_cases(pkg, synth):
    """No right keypoints -> all -1; calling before any extraction is an argument error, not a crash."""
    img = synth.make_frame(4200)
    exL, exR = pkg.ORBextractor(**EUROC), pkg.ORBextractor(**EUROC)
    try:
        with pytest.raises(pkg.OrbError):
            exL.ComputeStereoMatches(exR, np.zeros(1, pkg.KP_DTYPE), np.zeros((1, 32), np.uint8), np.zeros(1, pkg.KP_DTYPE), np.zeros((1, 32), np.uint8), 0.11, 47.9)
        _, kL, dL = exL(img, None, (0, 0))
        _, kR, dR = exR(img, None, (0, 0))
        uR, depth = exL.ComputeStereoMatches(exR, kL, dL, kR[:0], dR[:0], 0.11, 47.9)
        assert (uR == -1).all() and (depth == -1).all()
        # identical images: every SAD is 0, so the median is 0, thDist = 0 and the filter `first < thDist` (:1066) keeps nothing
        uR, depth = exL.ComputeStereoMatches(exR, kL, dL, kR, dR, 0.11, 47.9)
        assert (uR == -1).all() and (depth == -1).all()
    finally:
        exL.close(); exR.close()


@pytest.mark.parametrize("ch,rgb", [(3, True), (3, False), (4, True), (4, False)])
def test_cvt_color_gray(pkg, ex, ch, rgb):
    """cvtColor to gray as Tracking::GrabImage* applies it (Tracking.cc:1122-1135): OpenCV's fixed-point weights, checked
    against the same integer formula in numpy (incl. a width that is not a multiple of 4)."""
    rng = np.random.default_rng(ch * 2 + rgb)
    for (H, W) in [(480, 752), (37, 101)]:
        im = rng.integers(0, 256, (H, W, ch), dtype=np.uint8)
        r, g, b = (im[..., 0], im[..., 1], im[..., 2]) if rgb else (im[..., 2], im[..., 1], im[..., 0])
        ref = ((r.astype(np.uint32) * 4899 + g.astype(np.uint32) * 9617 + b.astype(np.uint32) * 1868 + 8192) >> 14).astype(np.uint8)
        assert np.array_equal(ex.cvtColorGray(im, rgb), ref)
    white = np.full((8, 8, ch), 255, np.uint8)
    assert (ex.cvtColorGray(white, rgb) == 255).all()


def test_blur_saturation_and_borders(ex, oex, oracle):
    """cv::GaussianBlur's 8-bit fixed point reaches 257 on saturated areas (the taps sum to 257/256) and is clamped; white and
    black blocks up to the image border exercise the clamp in the packed store and the reflected borders on every level."""
    H, W = 480, 752
    yy, xx = np.mgrid[0:H, 0:W]
    img = (((yy // 37 + xx // 41) % 2) * 255).astype(np.uint8)
    img[:5, :] = 255; img[:, -6:] = 255; img[-3:, :] = 254; img[:, :2] = 1
    ex(img, None, (0, 0))
    pyr = oex.pyramid(img)
    for l in range(8):
        assert np.array_equal(ex.image_pyramid_level(l), pyr[l]), "pyramid level %d" % l
        ref = oracle.gaussian_blur7(pyr[l])
        assert np.array_equal(ex.blurred_level(l), ref), "blur level %d" % l
    assert (oracle.gaussian_blur7(pyr[0]) == 255).any()


def test_maximum_image_size(pkg, oracle, synth):
    """The largest supported image (4096 x 4096, 12-bit packed coordinates) extracts bit-exactly; one pixel more is refused."""
    rng = np.random.default_rng(4096)
    small = synth.make_frame(4096, 512, 512)
    img = np.kron(small, np.ones((8, 8), np.uint8))
    img = (img.astype(np.int32) + rng.integers(-12, 13, img.shape)).clip(0, 255).astype(np.uint8)
    e = pkg.ORBextractor(2000, 1.2, 8, 20, 7)
    o = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    try:
        mono, kps, desc = e(img, None, (0, 0))
        mono_r, kps_r, desc_r = o.extract(img, (0, 0))
        assert mono == mono_r and len(kps) > 1500
        assert_kps_equal(kps, kps_r)
        assert np.array_equal(desc, desc_r)
        with pytest.raises((pkg.OrbError, ValueError)):
            e(np.zeros((4097, 64), np.uint8), None, (0, 0))
    finally:
        e.close()
