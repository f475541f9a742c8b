"""Deterministic synthetic frames for parity tests and bench.py ("Synthetic frame generator").

The reference ships no images and there is no network, so BASELINE.json's "synthetic frame
stream" is produced here.  Everything is a pure function of the seed (SplitMix64, counter mode), implemented with
numpy uint64 arithmetic so a 752x480 frame takes a few tens of milliseconds.

Normative definition (this file is the specification; CRC32 of seed 1000 is pinned in tests/test_synth.py):
  draw k (k = 0,1,2,...) = splitmix64(seed, k); integer uniform [a,b] = a + draw % (b-a+1)
  1. base: (gh+1)x(gw+1) nodes, gw = ceil(W/32), gh = ceil(H/32), values uniform [40,215], row-major draws;
     pixel (x,y) = bilinear interpolation of the 4 surrounding nodes at (x/32, y/32) in float32, rint().
  2. 600 rectangles: draws (cx in [0,W-1], cy in [0,H-1], w in [6,60], h in [6,60], delta in [-90,90]);
     covers x in [cx-w//2, cx-w//2+w) clipped to the image, likewise y; saturating add of delta.
  3. 300 discs: draws (cx, cy, r in [2,9], delta in [-70,70]); pixels with dx^2+dy^2 <= r^2; saturating add.
  4. per-pixel noise uniform [-3,3], row-major draws, saturating add.
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, idx):
    """k-th output (k = idx, array ok) of SplitMix64 seeded with `seed`."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx + np.uint64(1)) * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        z = z ^ (z >> np.uint64(31))
    return z


class _Stream:
    def __init__(self, seed):
        self.seed = seed
        self.k = 0

    def ints(self, n, lo, hi):
        r = splitmix64(self.seed, np.arange(self.k, self.k + n, dtype=np.uint64))
        self.k += n
        return (r % np.uint64(hi - lo + 1)).astype(np.int64) + lo


def make_frame(seed, H=480, W=752):
    """One synthetic uint8 frame of shape (H, W)."""
    s = _Stream(seed)
    gw, gh = -(-W // 32), -(-H // 32)
    nodes = s.ints((gh + 1) * (gw + 1), 40, 215).reshape(gh + 1, gw + 1).astype(np.float32)
    xs = np.arange(W, dtype=np.float32) / np.float32(32)
    ys = np.arange(H, dtype=np.float32) / np.float32(32)
    jx = np.floor(xs).astype(np.int64)
    jy = np.floor(ys).astype(np.int64)
    tx = (xs - jx.astype(np.float32))[None, :]
    ty = (ys - jy.astype(np.float32))[:, None]
    n00 = nodes[jy][:, jx]
    n01 = nodes[jy][:, jx + 1]
    n10 = nodes[jy + 1][:, jx]
    n11 = nodes[jy + 1][:, jx + 1]
    one = np.float32(1)
    top = n00 * (one - tx) + n01 * tx
    bot = n10 * (one - tx) + n11 * tx
    img = np.rint(top * (one - ty) + bot * ty).astype(np.int32)

    r = s.ints(600 * 5, 0, (1 << 31) - 1).reshape(600, 5)
    for cx, cy, w, h, d in r:
        cx, cy = int(cx % W), int(cy % H)
        w, h, d = 6 + int(w % 55), 6 + int(h % 55), int(d % 181) - 90
        x0, y0 = cx - w // 2, cy - h // 2
        xa, xb, ya, yb = max(x0, 0), min(x0 + w, W), max(y0, 0), min(y0 + h, H)
        if xa < xb and ya < yb:
            np.clip(img[ya:yb, xa:xb] + d, 0, 255, out=img[ya:yb, xa:xb])
    r = s.ints(300 * 4, 0, (1 << 31) - 1).reshape(300, 4)
    for cx, cy, rad, d in r:
        cx, cy = int(cx % W), int(cy % H)
        rad, d = 2 + int(rad % 8), int(d % 141) - 70
        xa, xb, ya, yb = max(cx - rad, 0), min(cx + rad + 1, W), max(cy - rad, 0), min(cy + rad + 1, H)
        yy, xx = np.mgrid[ya:yb, xa:xb]
        m = (xx - cx) ** 2 + (yy - cy) ** 2 <= rad * rad
        sub = img[ya:yb, xa:xb]
        sub[m] = np.clip(sub[m] + d, 0, 255)
    noise = s.ints(H * W, -3, 3).reshape(H, W)
    img = np.clip(img + noise, 0, 255)
    return np.ascontiguousarray(img.astype(np.uint8))


def make_stream(seed, nframes, H=480, W=752, margin=64, max_shift=8):
    """Synthetic frame stream for BASELINE config 3: frame t = frame t-1 shifted by an integer (dx,dy) in
    [-max_shift, max_shift]^2.  Frames are crops of one (H+2m)x(W+2m) canvas whose offset does a bounded
    random walk (reflected at the canvas edge).
    Returns (frames uint8 [nframes,H,W], offsets int32 [nframes,2] as (ox, oy)).
    A scene point seen at (x, y) in frame t-1 appears at (x + ox[t-1] - ox[t], y + oy[t-1] - oy[t]) in frame t."""
    canvas = make_frame(seed, H + 2 * margin, W + 2 * margin)
    s = _Stream(seed ^ 0x5DEECE66D)
    steps = s.ints(2 * nframes, -max_shift, max_shift).reshape(nframes, 2)
    frames = np.empty((nframes, H, W), dtype=np.uint8)
    offs = np.empty((nframes, 2), dtype=np.int32)
    ox = oy = margin
    for t in range(nframes):
        if t > 0:
            ox += int(steps[t, 0])
            oy += int(steps[t, 1])
            if ox < 0: ox = -ox
            if ox > 2 * margin: ox = 4 * margin - ox
            if oy < 0: oy = -oy
            if oy > 2 * margin: oy = 4 * margin - oy
        frames[t] = canvas[oy:oy + H, ox:ox + W]
        offs[t] = (ox, oy)
    return frames, offs


def make_descriptor_sets(seed, n=1000, flip_p=0.08, fresh_frac=0.2):
    """Matcher-only value distribution of SURVEY.md section 8d: candidates = n random 256-bit strings; queries =
    candidates with each bit flipped with probability flip_p for (1-fresh_frac) of the queries, fresh random
    strings for the rest.  Returns (queries [n,32] uint8, candidates [n,32] uint8)."""
    s = _Stream(seed)
    cand = (s.ints(n * 32, 0, 255)).astype(np.uint8).reshape(n, 32)
    fresh = (s.ints(n * 32, 0, 255)).astype(np.uint8).reshape(n, 32)
    thr = int(round(flip_p * 65536))
    flips = (s.ints(n * 256, 0, 65535) < thr).reshape(n, 32, 8)
    mask = np.zeros((n, 32), dtype=np.uint8)
    for b in range(8):
        mask |= (flips[:, :, b].astype(np.uint8) << b)
    q = cand ^ mask
    is_fresh = s.ints(n, 0, 999) < int(round(fresh_frac * 1000))
    q[is_fresh] = fresh[is_fresh]
    return np.ascontiguousarray(q), np.ascontiguousarray(cand)


# Examples/Monocular/TUM_512.yaml:9-19 (BASELINE config 5): KannalaBrandt8 fx, fy, cx, cy, k1..k4
TUMVI_KB8 = np.array([190.978477, 190.973307, 254.931706, 256.897442, 0.003482389402, 0.000715034845, -0.002053236141, 0.000202936736],
                     dtype=np.float32)


def kb8_unproject(params, u, v):
    """Unit rays of pixels (u, v) under the KannalaBrandt8 model [fx, fy, cx, cy, k0..k3]: r(theta) = theta + k0 theta^3 + ... is
    inverted by Newton iterations in float64.  Scene construction only (the callers project the points back with the code under
    test); it does not restate KannalaBrandt8::unproject."""
    p = np.asarray(params, dtype=np.float64)
    x = (np.asarray(u, np.float64) - p[2]) / p[0]
    y = (np.asarray(v, np.float64) - p[3]) / p[1]
    r = np.sqrt(x * x + y * y)
    th = r.copy()
    for _ in range(12):
        t2 = th * th
        f = th * (1 + t2 * (p[4] + t2 * (p[5] + t2 * (p[6] + t2 * p[7])))) - r
        df = 1 + t2 * (3 * p[4] + t2 * (5 * p[5] + t2 * (7 * p[6] + t2 * 9 * p[7])))
        th = th - f / df
    s = np.where(r > 1e-12, np.sin(th) / np.maximum(r, 1e-12), 1.0)
    return np.stack([x * s, y * s, np.cos(th)], axis=-1)


def pinhole_unproject(params, u, v):
    p = np.asarray(params, dtype=np.float64)
    x = (np.asarray(u, np.float64) - p[2]) / p[0]
    y = (np.asarray(v, np.float64) - p[3]) / p[1]
    d = np.stack([x, y, np.ones_like(x)], axis=-1)
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def make_last_frame_scene(cam_type, params, kx, ky, shift, seed, tz=0.0):
    """Map points for a last-frame projection search (ORBmatcher.cc:2027-2289) over a shifted synthetic stream: the keypoint of
    the last frame at (kx, ky) is seen at (kx, ky) + shift in the current frame, so its map point is placed on the CURRENT
    camera's ray through that pixel (un-projected through the camera model under test, depth 2-9 m) and moved to the world
    by the inverse of a small rigid current pose Tcw; Tlw = I.  Returns (Xw [n,3] f32, Tcw [4,4] f32, Tlw [4,4] f32).
    Projecting Tcw * Xw with the model lands within float rounding of the shifted pixel, wherever it lies in the field of view."""
    n = len(kx)
    r = splitmix64(seed ^ 0xA24BAED4963EE407, np.arange(n + 8, dtype=np.uint64))
    depth = 2.0 + (r[:n] % np.uint64(7001)).astype(np.float64) / 1000.0
    ang = (-0.02 + (float(r[n] % np.uint64(4001)) / 1e5), -0.02 + (float(r[n + 1] % np.uint64(4001)) / 1e5), -0.03 + (float(r[n + 2] % np.uint64(6001)) / 1e5))
    ca, sa, cb, sb, cg, sg = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
    Rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
    Ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    Rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]])
    R = (Rz @ Ry @ Rx).astype(np.float32).astype(np.float64)
    t = np.array([0.05 - float(r[n + 3] % np.uint64(101)) / 1e3, -0.04 + float(r[n + 4] % np.uint64(81)) / 1e3, tz], dtype=np.float32).astype(np.float64)
    u = np.asarray(kx, np.float64) + float(shift[0])
    v = np.asarray(ky, np.float64) + float(shift[1])
    rays = kb8_unproject(params, u, v) if cam_type == 1 else pinhole_unproject(params, u, v)
    Xc = rays * depth[:, None]
    Xw = (Xc - t[None, :]) @ R            # R^T (Xc - t), row-vector form
    Tcw = np.eye(4, dtype=np.float32)
    Tcw[:3, :3] = R.astype(np.float32)
    Tcw[:3, 3] = t.astype(np.float32)
    return np.ascontiguousarray(Xw.astype(np.float32)), Tcw, np.eye(4, dtype=np.float32)
