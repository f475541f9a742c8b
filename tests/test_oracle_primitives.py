"""CPU: the oracle's OpenCV-primitive restatements checked from first principles,
and the extractor constants checked against the values SURVEY.md derives independently from the reference."""
import ctypes as C

import numpy as np
import pytest

from conftest import EUROC, TUMVI


def test_cvround_half_even(oracle):
    L = oracle.lib()
    assert [L.orc_cvRound(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def test_extractor_constants(oracle):
    e = oracle.OracleExtractor(**EUROC)   # SURVEY.md 8a X0/X1 [DERIVED]
    assert e.features_per_level == [217, 181, 151, 126, 105, 87, 73, 60]
    assert e.umax == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert [e.level_size(l, 752, 480) for l in range(8)] == [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]
    t = oracle.OracleExtractor(**TUMVI)
    assert t.features_per_level == [326, 271, 226, 189, 157, 131, 109, 91]
    assert [t.level_size(l, 512, 512)[0] for l in range(8)] == [512, 427, 356, 296, 247, 206, 171, 143]
    ini = oracle.OracleExtractor(5000, 1.2, 8, 20, 7)
    assert ini.features_per_level == [1086, 905, 754, 628, 524, 436, 364, 303]
    # 749-pixel disc
    assert sum(2 * u + 1 for u in e.umax[1:]) * 2 + 2 * e.umax[0] + 1 == 749


def test_gauss_kernel_and_blur(oracle):
    k = (C.c_int * 7)()
    oracle.lib().orc_gauss7_kernel(k)
    assert list(k) == [18, 34, 49, 55, 49, 34, 18]            # SURVEY.md A.5, sum 257 (not renormalised)
    img = np.zeros((15, 15), np.uint8)
    img[7, 7] = 255
    out = oracle.gaussian_blur7(img)
    kk = np.array(list(k))
    exp = (255 * np.outer(kk, kk) + 32768) >> 16
    assert np.array_equal(out[4:11, 4:11], exp) and out.sum() == exp.sum()
    const = np.full((20, 33), 100, np.uint8)
    assert (oracle.gaussian_blur7(const) == ((100 * 257 * 257 + 32768) >> 16)).all()
    assert (oracle.gaussian_blur7(np.full((9, 9), 255, np.uint8)) == 255).all()   # saturates
    # BORDER_REFLECT_101 at the edge: column x=-1 mirrors x=1
    ramp = np.tile(np.arange(40, dtype=np.uint8) * 5, (12, 1))
    out = oracle.gaussian_blur7(ramp)
    row = ramp[0].astype(np.int64)
    ext = np.concatenate([row[3:0:-1], row, row[-2:-5:-1]])
    h = np.array([np.dot(kk, ext[i:i + 7]) for i in range(40)])
    assert np.array_equal(out[5], np.minimum((h * 257 + 32768) >> 16, 255))


def test_resize_linear(oracle):
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (60, 90), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear(src, 90, 60), src)                  # identity
    assert (oracle.resize_linear(np.full((60, 90), 137, np.uint8), 75, 50) == 137).all()
    # against a float bilinear model (pixel-centre mapping), within 1 grey level
    dw, dh = 75, 50
    out = oracle.resize_linear(src, dw, dh).astype(np.float64)
    xs = (np.arange(dw) + 0.5) * (90 / dw) - 0.5
    ys = (np.arange(dh) + 0.5) * (60 / dh) - 0.5
    x0 = np.floor(xs).astype(int); fx = xs - x0
    y0 = np.floor(ys).astype(int); fy = ys - y0
    s = src.astype(np.float64)
    ref = ((s[y0][:, x0] * (1 - fx) + s[y0][:, x0 + 1] * fx) * (1 - fy)[:, None] +
           (s[y0 + 1][:, x0] * (1 - fx) + s[y0 + 1][:, x0 + 1] * fx) * fy[:, None])
    assert np.abs(out - ref).max() <= 1.0


CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def patch(center, circle_vals, size=7):
    p = np.full((size, size), center, np.uint8)
    c = size // 2
    for (dx, dy), v in zip(CIRCLE, circle_vals):
        p[c + dy, c + dx] = v
    return p


def test_fast_hand_built(oracle):
    # 9 contiguous darker pixels (50 vs centre 100): corner for t < 50, score = 49
    vals = [50] * 9 + [100] * 7
    p = patch(100, vals)
    kp = oracle.fast9_16(p, 20)
    assert kp.tolist() == [[3, 3, 49]]
    assert oracle.fast9_16(p, 49).tolist() == [[3, 3, 49]]
    assert len(oracle.fast9_16(p, 50)) == 0                                         # strict: 50 < 100-50 is false
    assert len(oracle.fast9_16(patch(100, [50] * 8 + [100] * 8), 20)) == 0          # 8 contiguous is not enough
    # wrap-around arc (positions 12..15,0..4) and brighter arc
    vals = [200] * 5 + [100] * 7 + [200] * 4
    assert oracle.fast9_16(patch(100, vals), 20).tolist() == [[3, 3, 99]]
    # score = min margin over the best arc
    vals = [40, 45, 50, 55, 60, 65, 70, 75, 79] + [100] * 7
    assert oracle.fast9_16(patch(100, vals), 20).tolist() == [[3, 3, 20]]           # min(v - c_k) = 21 -> score 20
    L = oracle.lib()
    pp = patch(100, vals)
    assert L.orc_fast_corner_score(pp.ctypes.data + 3 * 7 + 3, C.c_size_t(7), 7) == 20
    # non-maximum suppression: two adjacent corners with equal score suppress each other
    big = np.full((9, 12), 100, np.uint8)
    for cx in (4, 5):
        for (dx, dy) in CIRCLE[:9]:
            big[4 + dy, cx + dx] = 30
    res = oracle.fast9_16(big, 20)
    scores = {(x, y): s for x, y, s in res.tolist()}
    assert (4, 4) not in scores or (5, 4) not in scores


def test_fast_atan2(oracle):
    L = oracle.lib()
    assert L.orc_fast_atan2(0.0, 1.0) == 0.0
    assert abs(L.orc_fast_atan2(1.0, 0.0) - 90.0) < 1e-4
    assert abs(L.orc_fast_atan2(0.0, -1.0) - 180.0) < 1e-4
    assert abs(L.orc_fast_atan2(-1.0, 0.0) - 270.0) < 1e-4
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = rng.normal(0, 1e5, 2)
        ref = np.degrees(np.arctan2(y, x)) % 360.0
        got = L.orc_fast_atan2(float(np.float32(y)), float(np.float32(x)))
        assert min(abs(got - ref), 360 - abs(got - ref)) < 0.3                      # OpenCV documents ~0.3 degrees
        assert 0.0 <= got <= 360.0


def test_ic_angle_on_gradients(oracle):
    e = oracle.OracleExtractor(**EUROC)
    xx, yy = np.meshgrid(np.arange(64), np.arange(64))
    assert abs(e.ic_angle((xx * 3).astype(np.uint8), 32, 32)) < 0.01               # brighter to the right -> 0 deg
    assert abs(e.ic_angle((yy * 3).astype(np.uint8), 32, 32) - 90.0) < 0.01        # brighter downwards -> 90 deg
    a = e.ic_angle(((xx + yy) * 2).astype(np.uint8), 32, 32)
    assert abs(a - 45.0) < 0.3


def test_descriptor_rotation_consistency(oracle):
    """Steered BRIEF: a patch and its 90-degree rotation give the same descriptor when the angle follows."""
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (81, 81), dtype=np.uint8)
    rot = np.ascontiguousarray(np.rot90(img, k=-1))         # clockwise in image coordinates (y down): angle + 90
    d0 = oracle.compute_descriptor(img, 40, 40, 0.0)
    d1 = oracle.compute_descriptor(rot, 40, 40, 90.0)
    assert np.unpackbits(d0 ^ d1).sum() <= 2                # exact up to cos(90deg) not being exactly 0 in fp32


def test_octree_parallel_model_equals_literal(oracle):
    """The data-parallel DistributeOctTree formulation used by the HIP kernel (tests/octree_model.py) against the
    literal std::list restatement, including response ties and clustered points."""
    import octree_model as M
    rng = np.random.default_rng(0)
    done = 0
    while done < 60:
        w = int(rng.integers(80, 900)); h = int(rng.integers(70, 600))
        if round(float(np.float32(w - 32) / np.float32(h - 32))) < 1:
            continue
        n = int(rng.integers(1, 500))
        pts = set()
        while len(pts) < n:
            if rng.random() < 0.5:
                pts.add((int(rng.integers(3, w - 35)), int(rng.integers(3, h - 35))))
            else:
                cx, cy = rng.integers(3, w - 35), rng.integers(3, h - 35)
                pts.add((int(np.clip(cx + rng.integers(-6, 7), 3, w - 36)), int(np.clip(cy + rng.integers(-6, 7), 3, h - 36))))
        pts = sorted(pts, key=lambda p: (p[1], p[0]))
        c = np.array([(x, y, int(rng.integers(7, 12))) for x, y in pts], dtype=np.float32)
        N = int(rng.integers(1, 400))
        ref = oracle.distribute_octtree(c, 16, w - 16, 16, h - 16, N)
        idx = M.distribute(c[:, 0].astype(np.int64), c[:, 1].astype(np.int64), c[:, 2].astype(np.int64), 16, w - 16, 16, h - 16, N)
        assert np.array_equal(c[idx], ref)
        nIni = int(np.floor(np.float32(w - 32) / np.float32(h - 32) + np.float32(0.5)))
        assert len(ref) <= max(N + 3, 4 * nIni)        # the capacity bound liborbhip sizes its outputs with
        done += 1


def test_get_features_in_area_vs_bruteforce(oracle):
    rng = np.random.default_rng(4)
    N = 800
    kx = rng.uniform(-20, 770, N).astype(np.float32)
    ky = rng.uniform(-20, 500, N).astype(np.float32)
    octv = rng.integers(0, 8, N).astype(np.int32)
    sf = np.array([1.2 ** i for i in range(8)], np.float32)
    F = oracle.OracleFrame(kx, ky, octv, np.zeros(N, np.float32), np.zeros((N, 32), np.uint8), (0.0, 752.0, 0.0, 480.0), sf)
    start, idx = F.grid_csr()
    inv_w, inv_h = np.float32(64) / np.float32(752), np.float32(48) / np.float32(480)
    gx = np.floor(np.abs((kx - np.float32(0)) * inv_w) + np.float32(0.5)) * np.sign(kx)       # std::round
    gy = np.floor(np.abs((ky - np.float32(0)) * inv_h) + np.float32(0.5)) * np.sign(ky)
    in_grid = (gx >= 0) & (gx < 64) & (gy >= 0) & (gy < 48)
    assert start[-1] == in_grid.sum() and sorted(idx.tolist()) == np.nonzero(in_grid)[0].tolist()
    for _ in range(200):
        x, y, r = rng.uniform(0, 752), rng.uniform(0, 480), rng.choice([5.0, 30.0, 200.0])
        mn, mx = int(rng.integers(-1, 5)), int(rng.integers(-1, 8))
        got = F.features_in_area(x, y, r, mn, mx)
        keep = in_grid & (np.abs(kx - np.float32(x)) < np.float32(r)) & (np.abs(ky - np.float32(y)) < np.float32(r))
        if mn > 0 or mx >= 0:
            keep &= octv >= mn
            if mx >= 0:
                keep &= octv <= mx
        assert sorted(got.tolist()) == np.nonzero(keep)[0].tolist()
        # walk order: (cell x, cell y, index)
        key = [(int(gx[i]), int(gy[i]), int(i)) for i in got]
        assert key == sorted(key)


def test_three_maxima(oracle):
    assert oracle.three_maxima([0] * 30) == (-1, -1, -1)
    h = [0] * 30; h[3] = 10; h[7] = 9; h[1] = 8; h[20] = 7
    assert oracle.three_maxima(h) == (3, 7, 1)
    h = [0] * 30; h[5] = 100; h[6] = 9; h[7] = 8
    assert oracle.three_maxima(h) == (5, -1, -1)                                   # second < 10% of first
    h = [0] * 30; h[5] = 100; h[6] = 50; h[7] = 9
    assert oracle.three_maxima(h) == (5, 6, -1)
