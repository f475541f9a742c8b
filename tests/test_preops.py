"""N4 pre-path image operations (CLAHE of the TUM-VI examples, remap of the stereo examples): the C oracle against an independent
numpy restatement and against closed-form answers.  OpenCV itself is absent from this image: both restatements are [OPENCV-UNVERIFIED]."""
import importlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def clahe_numpy(im, clip=3.0, tiles=(8, 8)):
    """clahe.cpp restated on arrays (histogram / clip / redistribute / cumulative LUT / float32 bilinear blend)."""
    H, W = im.shape
    tX, tY = tiles
    eW = W if W % tX == 0 else W + tX - W % tX
    eH = H if H % tY == 0 else H + tY - H % tY
    ext = np.pad(im, ((0, eH - H), (0, eW - W)), mode="reflect")
    tw, th = eW // tX, eH // tY
    total = tw * th
    lut_scale = np.float32(255) / np.float32(total)
    limit = max(int(clip * total / 256), 1) if clip > 0 else 0
    lut = np.zeros((tY, tX, 256), np.uint8)
    for ty in range(tY):
        for tx in range(tX):
            h = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if limit > 0:
                clipped = int(np.maximum(h - limit, 0).sum())
                h = np.minimum(h, limit)
                batch, residual = divmod(clipped, 256)
                h += batch
                if residual:
                    step = max(256 // residual, 1)
                    idx = np.arange(0, 256, step)[:residual]
                    h[idx] += 1
            c = np.cumsum(h).astype(np.float32) * lut_scale
            lut[ty, tx] = np.clip(np.rint(c), 0, 255).astype(np.uint8)
    f32 = np.float32
    ys, xs = np.arange(H, dtype=f32), np.arange(W, dtype=f32)
    tyf = ys * (f32(1) / f32(th)) - f32(0.5)
    txf = xs * (f32(1) / f32(tw)) - f32(0.5)
    ty1 = np.floor(tyf).astype(np.int32); tx1 = np.floor(txf).astype(np.int32)
    ya = (tyf - ty1.astype(f32)).astype(f32); xa = (txf - tx1.astype(f32)).astype(f32)
    ya1, xa1 = f32(1) - ya, f32(1) - xa
    ty2 = np.minimum(ty1 + 1, tY - 1); tx2 = np.minimum(tx1 + 1, tX - 1)
    ty1 = np.maximum(ty1, 0); tx1 = np.maximum(tx1, 0)
    v = im.astype(np.int64)
    l11 = lut[ty1[:, None], tx1[None, :], v].astype(f32); l12 = lut[ty1[:, None], tx2[None, :], v].astype(f32)
    l21 = lut[ty2[:, None], tx1[None, :], v].astype(f32); l22 = lut[ty2[:, None], tx2[None, :], v].astype(f32)
    top = (l11 * xa1[None, :]).astype(f32) + (l12 * xa[None, :]).astype(f32)
    bot = (l21 * xa1[None, :]).astype(f32) + (l22 * xa[None, :]).astype(f32)
    res = (top * ya1[:, None]).astype(f32) + (bot * ya[:, None]).astype(f32)
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


def remap_numpy(im, mapx, mapy):
    """remap INTER_LINEAR / BORDER_CONSTANT(0) in closed form: 1/32-pixel coordinates, (sum of tap * fx' * fy' + 512) >> 10."""
    sx = np.rint(mapx.astype(np.float32) * np.float32(32)).astype(np.int64)
    sy = np.rint(mapy.astype(np.float32) * np.float32(32)).astype(np.int64)
    fx, fy = sx & 31, sy & 31
    ix, iy = sx >> 5, sy >> 5
    H, W = im.shape

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        return np.where(ok, im[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0).astype(np.int64)
    acc = tap(iy, ix) * (32 - fx) * (32 - fy) + tap(iy, ix + 1) * fx * (32 - fy) + tap(iy + 1, ix) * (32 - fx) * fy + tap(iy + 1, ix + 1) * fx * fy
    return ((acc + 512) >> 10).astype(np.uint8)


def rectify_maps(H, W, seed=0):
    """A plausible rectification map pair: small rotation + radial term, reaching outside the source near the corners."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    cx, cy, f = W / 2 + rng.uniform(-5, 5), H / 2 + rng.uniform(-5, 5), 0.6 * W
    a = np.deg2rad(rng.uniform(-2, 2))
    xn, yn = (xs - cx) / f, (ys - cy) / f
    xr, yr = np.cos(a) * xn - np.sin(a) * yn, np.sin(a) * xn + np.cos(a) * yn
    r2 = xr * xr + yr * yr
    d = 1 + 0.28 * r2 + 0.07 * r2 * r2
    return (xr * d * f + cx).astype(np.float32), (yr * d * f + cy).astype(np.float32)


@pytest.mark.parametrize("shape,tiles,clip", [((64, 96), (8, 8), 3.0), ((67, 101), (8, 8), 3.0), ((48, 40), (4, 6), 40.0), ((32, 32), (2, 2), 0.0)])
def test_clahe_oracle_vs_numpy(oracle, shape, tiles, clip):
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    base = rng.integers(0, 256, shape, dtype=np.uint8)
    smooth = (np.linspace(20, 120, shape[1])[None, :] + 30 * np.sin(np.arange(shape[0]) / 5.0)[:, None] + rng.normal(0, 6, shape)).clip(0, 255).astype(np.uint8)
    for im in (base, smooth):
        assert np.array_equal(oracle.clahe(im, clip, tiles), clahe_numpy(im, clip, tiles))


def test_clahe_constant_image_known_answer(oracle):
    """A constant tile: the one occupied bin is clipped to the limit and the excess spread evenly, so the cumulative LUT at grey level v
    is round((v+1) * (excess/256 share) + limit ...) -- computed here by hand for 512x512, 8x8 tiles, clip 3 (the TUM-VI setting)."""
    v, total, limit = 100, 64 * 64, int(3.0 * 64 * 64 / 256)            # limit = 48
    clipped = total - limit                                             # 4048 -> batch 15, residual 208, step 1: bins 0..207 get +1
    batch, residual = divmod(clipped, 256)
    assert (batch, residual) == (15, 208)
    cum = (v + 1) * batch + min(v + 1, residual) + limit                # bins 0..v inclusive
    expect = int(np.rint(np.float32(cum) * (np.float32(255) / np.float32(total))))
    out = oracle.clahe(np.full((512, 512), v, np.uint8))
    assert (out == expect).all()


def test_remap_oracle_vs_numpy_and_properties(oracle):
    rng = np.random.default_rng(5)
    im = rng.integers(0, 256, (120, 188), dtype=np.uint8)
    mx, my = rectify_maps(120, 188, 1)
    out = oracle.remap_linear(im, mx, my)
    assert np.array_equal(out, remap_numpy(im, mx, my))
    assert (out == 0).any() and (out != 0).any()                        # some of the map leaves the source: border value 0
    ys, xs = np.mgrid[0:120, 0:188].astype(np.float32)
    assert np.array_equal(oracle.remap_linear(im, xs, ys), im)          # identity map
    half = oracle.remap_linear(im, xs + np.float32(0.5), ys)            # half-pixel shift: (a + b + 1) >> 1, last column blends with the border
    a, b = im.astype(np.int32), np.concatenate([im[:, 1:], np.zeros((120, 1), np.uint8)], axis=1).astype(np.int32)
    assert np.array_equal(half, ((a + b + 1) >> 1).astype(np.uint8))
    # a different output size than the source, NaN / huge coordinates
    mx2 = np.full((10, 12), np.nan, np.float32); my2 = np.full((10, 12), 1e12, np.float32)
    assert (oracle.remap_linear(im, mx2, my2) == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,tiles,clip", [((480, 752), (8, 8), 3.0), ((512, 512), (8, 8), 3.0), ((67, 101), (8, 8), 3.0), ((300, 333), (5, 7), 2.0),
                                              ((64, 64), (16, 16), 40.0), ((96, 128), (8, 8), 0.0)])
def test_clahe_gpu_parity(pkg, oracle, synth, shape, tiles, clip):
    """orbx_clahe against the oracle, bit-exact: the TUM-VI setting on the EuRoC and TUM-VI image sizes, sizes that need the
    reflect-101 extension, other tile grids, no clipping."""
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    try:
        rng = np.random.default_rng(shape[0] + shape[1])
        frames, _ = synth.make_stream(4100, 1)
        scene = frames[0]
        fits = scene.shape[0] >= shape[0] and scene.shape[1] >= shape[1]
        ims = [rng.integers(0, 256, shape, dtype=np.uint8), np.full(shape, 77, np.uint8),
               (scene[:shape[0], :shape[1]] // 3 + 40).astype(np.uint8) if fits else rng.integers(90, 130, shape, dtype=np.uint8)]
        for im in ims:
            assert np.array_equal(ex.CLAHE(im, clip, tiles), oracle.clahe(im, clip, tiles))
        with pytest.raises(pkg.OrbError):
            ex.CLAHE(ims[0], 3.0, (32, 32))        # more than 256 tiles
    finally:
        ex.close()


@pytest.mark.gpu
def test_remap_gpu_parity(pkg, oracle, synth):
    """orbx_remap_linear against the oracle, bit-exact, on a rectification-like map pair (parts of it leave the source), with the
    maps reused from the device on the second call, an output size different from the source, and non-finite coordinates."""
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    try:
        frames, _ = synth.make_stream(4200, 2)
        H, W = frames[0].shape
        mx, my = rectify_maps(H, W, 3)
        ref0 = oracle.remap_linear(frames[0], mx, my)
        assert np.array_equal(ex.remap(frames[0], mx, my), ref0)
        assert (ref0 == 0).sum() > 100                                         # border region present
        assert np.array_equal(ex.remap(frames[1], size=(H, W)), oracle.remap_linear(frames[1], mx, my))   # maps kept on the device
        ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
        assert np.array_equal(ex.remap(frames[0], xs, ys), frames[0])          # identity
        mx2, my2 = rectify_maps(100, 130, 4)
        mx2, my2 = mx2 * 4, my2 * 4
        mx2[3, 4] = np.nan; my2[5, 6] = np.inf; mx2[7, 8] = -1e30; my2[9, 9] = 3e9
        assert np.array_equal(ex.remap(frames[0], mx2, my2), oracle.remap_linear(frames[0], mx2, my2))
        with pytest.raises(pkg.OrbError):
            ex.remap(frames[0], size=(50, 60))                                 # no maps of that size uploaded
    finally:
        ex.close()
