"""GPU parity tests of the matrix-pipe candidate scan (k_match_rank + k_match_scan_mfma, csrc/orb_match_mfma.h).

The open-window blocks of a batch launch compute their Hamming distances (ORBmatcher.cc:2463-2483) as exact int8 dot products
whose accumulator is the reduction key itself; frame pairs all of whose queries are open build their lists inside k_match_resolve
(fused form, claimed keypoints masked per 512-query super-chunk).  Every case runs with the fused engine, with the matrix-pipe scan
kernel alone and with the vector-ALU scan (orbm_set_hamming_engine 2 / 1 / 0) and all three must equal the CPU oracle's in-order loop (ORBmatcher.cc:99-130): distances, the
grid-walk tie order between equal distances, pre-occupied keypoints, keypoints outside the grid, dead queries, frames whose
keypoint count is not a multiple of the 32-candidate tile, blocks that mix open and windowed queries (served by k_match_scan)."""
import ctypes as C

import numpy as np
import pytest

from conftest import EUROC

pytestmark = pytest.mark.gpu

W, H = 752, 480


def run_batch(pkg, m, cand, qry, bounds, nnratio=0.8, th=100, second=True):
    """cand: list of dict(k, d, slot, sobs); qry: list of dict(d, u, v, r, lo, hi, flags).  One launch of
    orbm_search_by_projection_batch_device over len(cand) pairs; returns per-pair (nm, moq, bd, slot, sobs)."""
    import torch
    npairs = len(cand)
    capn = max(max(len(c["k"]) for c in cand), 1) + 3
    capq = max(max(len(q["u"]) for q in qry), 1) + 5
    kp = np.zeros((npairs, capn, 7), np.float32); de = np.zeros((npairs, capn, 32), np.uint8); cn = np.zeros((npairs, 2), np.int32)
    slot = np.full((npairs, capn), -1, np.int32); sobs = np.zeros((npairs, capn), np.uint8)
    qd = np.zeros((npairs, capq, 32), np.uint8); qn = np.zeros((npairs, 2), np.int32)
    u = np.zeros((npairs, capq), np.float32); v = np.zeros((npairs, capq), np.float32); r = np.zeros((npairs, capq), np.float32)
    lo = np.zeros((npairs, capq), np.int32); hi = np.zeros((npairs, capq), np.int32); fl = np.zeros((npairs, capq), np.uint8)
    for p, (c, q) in enumerate(zip(cand, qry)):
        n, nq = len(c["k"]), len(q["u"])
        kp[p, :n] = np.ascontiguousarray(c["k"]).view(np.float32).reshape(n, 7); de[p, :n] = c["d"]; cn[p, 0] = n
        slot[p, :n] = c["slot"]; sobs[p, :n] = c["sobs"]
        qd[p, :nq] = q["d"]; qn[p, 0] = nq
        u[p, :nq] = q["u"]; v[p, :nq] = q["v"]; r[p, :nq] = q["r"]; lo[p, :nq] = q["lo"]; hi[p, :nq] = q["hi"]; fl[p, :nq] = q["flags"]
    dev = "cuda"
    t = lambda a: torch.from_numpy(a).to(dev)
    d = dict(kp=t(kp), de=t(de), cn=t(cn), slot=t(slot), sobs=t(sobs), qd=t(qd), qn=t(qn), u=t(u), v=t(v), r=t(r), lo=t(lo), hi=t(hi), fl=t(fl))
    moq = torch.full((npairs, capq), -7, dtype=torch.int32, device=dev); bd = torch.zeros((npairs, capq), dtype=torch.int32, device=dev)
    nm = torch.zeros((npairs,), dtype=torch.int32, device=dev)
    fs = pkg.FrameStruct(capn, d["kp"].data_ptr(), d["de"].data_ptr(), None, *bounds)
    qs = pkg.QueryStruct(capq, d["qd"].data_ptr(), d["u"].data_ptr(), d["v"].data_ptr(), d["r"].data_ptr(), d["lo"].data_ptr(), d["hi"].data_ptr(), None, d["fl"].data_ptr())
    rc = m.L.orbm_search_by_projection_batch_device(m.m, C.byref(fs), capn, C.c_void_p(d["cn"].data_ptr()), 2, C.byref(qs), capq,
                                                    C.c_void_p(d["qn"].data_ptr()), 2, npairs, C.c_float(nnratio), int(th), int(second),
                                                    C.c_void_p(d["slot"].data_ptr()), C.c_void_p(d["sobs"].data_ptr()), C.c_void_p(moq.data_ptr()),
                                                    C.c_void_p(bd.data_ptr()), C.c_void_p(nm.data_ptr()), None)
    assert rc == 0, m.L.orbm_last_error(m.m)
    torch.cuda.synchronize()
    moq_h, bd_h, nm_h, slot_h, sobs_h = moq.cpu().numpy(), bd.cpu().numpy(), nm.cpu().numpy(), d["slot"].cpu().numpy(), d["sobs"].cpu().numpy()
    return [(int(nm_h[p]), moq_h[p, :len(qry[p]["u"])], bd_h[p, :len(qry[p]["u"])], slot_h[p, :len(cand[p]["k"])], sobs_h[p, :len(cand[p]["k"])]) for p in range(npairs)]


def oracle_pair(oracle, c, q, bounds, sf, nnratio, th, second):
    k = c["k"]
    OF = oracle.OracleFrame(k["x"], k["y"], k["octave"], k["angle"], c["d"], bounds, sf)
    OF.slot[:] = c["slot"]; OF.slot_obs[:] = c["sobs"]
    inv = (q["flags"] & 1).astype(np.uint8); obs = ((q["flags"] >> 1) & 1).astype(np.uint8)
    n, moq, bd = OF.search_by_projection_win(q["d"], q["u"], q["v"], q["r"], q["lo"], q["hi"], nnratio, th, second, qobs=obs, in_view=inv)
    return n, moq, bd, OF.slot.copy(), OF.slot_obs.copy()


def check(pkg, oracle, cand, qry, bounds, sf, nnratio=0.8, th=100, second=True, min_total=1):
    ref = [oracle_pair(oracle, c, q, bounds, sf, nnratio, th, second) for c, q in zip(cand, qry)]
    for engine in (2, 1, 0):
        m = pkg.ORBmatcher(nnratio, True)
        try:
            m.set_hamming_engine(engine)
            got = run_batch(pkg, m, cand, qry, bounds, nnratio, th, second)
        finally:
            m.close()
        for p, (g, r) in enumerate(zip(got, ref)):
            what = "engine %d pair %d (n=%d nq=%d)" % (engine, p, len(cand[p]["k"]), len(qry[p]["u"]))
            assert g[0] == r[0], what
            assert np.array_equal(g[1], r[1]) and np.array_equal(g[2], r[2]), what
            assert np.array_equal(g[3], r[3]) and np.array_equal(g[4], r[4]), what
    assert sum(r[0] for r in ref) >= min_total
    return ref


@pytest.fixture(scope="module")
def stream(oracle, synth):
    frames, offs = synth.make_stream(5100, 13)
    o = oracle.OracleExtractor(**EUROC)
    return [o.extract(f)[1:] for f in frames], offs, np.asarray(o.scale_factors, np.float32)


def open_queries(k0, d0, shift, rng=None, radius=1.0e4):
    nq = len(k0)
    return dict(d=d0, u=(k0["x"] + np.float32(shift[0])).astype(np.float32), v=(k0["y"] + np.float32(shift[1])).astype(np.float32),
                r=np.full(nq, radius, np.float32), lo=np.full(nq, -1, np.int32), hi=np.full(nq, -1, np.int32), flags=np.full(nq, 3, np.uint8))


def free_frame(k, d):
    return dict(k=k, d=d, slot=np.full(len(k), -1, np.int32), sobs=np.zeros(len(k), np.uint8))


def test_stress_batch_both_engines(pkg, oracle, stream):
    """BASELINE config 3's setting, 12 frame pairs in one launch (48 query blocks: batch mode, matrix-pipe scan)."""
    ext, offs, sf = stream
    bounds = (0.0, float(W), 0.0, float(H))
    cand = [free_frame(*ext[p + 1]) for p in range(12)]
    qry = [open_queries(ext[p][0], ext[p][1], (offs[p][0] - offs[p + 1][0], offs[p][1] - offs[p + 1][1])) for p in range(12)]
    ref = check(pkg, oracle, cand, qry, bounds, sf)
    assert min(r[0] for r in ref) > 300


def test_tile_tails_and_small_frames(pkg, oracle, stream):
    """Keypoint counts around the 32-candidate tile and the 16-candidate lane halves, query counts around the 256-query block and
    the 32-query tile; every pair still has its own vote."""
    ext, offs, sf = stream
    bounds = (0.0, float(W), 0.0, float(H))
    ns = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 257, 999]
    nqs = [1, 31, 32, 33, 255, 256, 257, 300, 511, 513, 7, 64, 1000]
    cand, qry = [], []
    for p, (n, nq) in enumerate(zip(ns, nqs)):
        k1, d1 = ext[(p + 1) % 13]
        k0, d0 = ext[p]
        n, nq = min(n, len(k1)), min(nq, len(k0))
        cand.append(free_frame(k1[:n], d1[:n]))
        # queries = the keypoints of the candidate set themselves (shift 0) for the small cases, so that matches exist
        src_k, src_d = (k1, d1) if n < 300 else (k0, d0)
        sh = (0, 0) if n < 300 else (offs[p][0] - offs[(p + 1) % 13][0], offs[p][1] - offs[(p + 1) % 13][1])
        qry.append(open_queries(src_k[:nq], src_d[:nq], sh))
    check(pkg, oracle, cand, qry, bounds, sf, min_total=100)


def test_ties_occupied_dead_and_mixed_blocks(pkg, oracle, stream):
    """Duplicated descriptors (equal distances: the grid-walk order decides), pre-occupied keypoints with and without
    observations, keypoints outside the grid, queries that are not in view, map points without observations, and blocks in which
    some queries search a small window (those blocks belong to k_match_scan, their neighbours to the matrix pipe)."""
    ext, offs, sf = stream
    rng = np.random.default_rng(52)
    bounds = (-40.0, W + 25.0, -30.0, H + 35.5)             # non-integer cells, negative origin
    cand, qry = [], []
    for p in range(12):
        k1, d1 = ext[p + 1]
        k0, d0 = ext[p]
        k1 = k1.copy(); d1 = d1.copy()
        n = len(k1)
        base = d1[rng.integers(0, n, 25)]
        dup = rng.random(n) < 0.5
        d1[dup] = base[rng.integers(0, 25, int(dup.sum()))]              # heavy duplication -> ties
        flip = rng.random((n, 32)) < 0.01
        d1[flip] ^= 1
        out = rng.permutation(n)[:6]
        k1["x"][out[:3]] = np.float32(-70.0); k1["y"][out[3:]] = np.float32(H + 60.0)   # PosInGrid false
        c = free_frame(k1, d1)
        occ = rng.random(n) < (0.0, 0.15, 0.5)[p % 3]
        c["slot"][occ] = 1 << 20
        c["sobs"][occ] = rng.random(int(occ.sum())) < 0.6
        cand.append(c)
        nq = len(k0)
        qi = rng.integers(0, n, nq)
        q = open_queries(k0, d1[qi].copy() if p % 2 else d0, (offs[p][0] - offs[p + 1][0], offs[p][1] - offs[p + 1][1]))
        inv = (rng.random(nq) < 0.93).astype(np.uint8); obs = (rng.random(nq) < 0.8).astype(np.uint8)
        q["flags"] = (inv | (obs << 1)).astype(np.uint8)
        if p in (3, 4):            # queries 256..511 (block 1) partly windowed: that block is not open
            sel = 256 + rng.permutation(256)[:40]
            sel = sel[sel < nq]
            q["r"][sel] = 25.0; q["lo"][sel] = 0; q["hi"][sel] = 3
        if p == 5:                 # the whole pair windowed: the device vote sends it to the grid-window walk
            q["r"][:] = 20.0
        if p == 6:                 # a block whose queries are all dead
            q["flags"][256:512] &= 0xFE
        qry.append(q)
    for nnratio, th, second in ((0.8, 100, True), (0.6, 50, True), (0.9, 255, False)):
        check(pkg, oracle, cand, qry, bounds, sf, nnratio=nnratio, th=th, second=second, min_total=500)


def test_2048_keypoints(pkg, oracle, synth):
    """The largest frame of the 32-bit-key regime (2048 keypoints = 64 tiles, rank fills its 11 bits) and one keypoint more
    (64-bit keys: vector-ALU scan whatever the engine)."""
    rng = np.random.default_rng(53)
    sf = np.array([1.2 ** i for i in range(8)], np.float32)
    bounds = (0.0, float(W), 0.0, float(H))
    cand, qry = [], []
    for p in range(13):
        N = 2048 if p < 12 else 2049
        kps = np.zeros(N, dtype=pkg.KP_DTYPE)
        kps["x"] = rng.uniform(1, W - 1, N).astype(np.float32); kps["y"] = rng.uniform(1, H - 1, N).astype(np.float32)
        kps["octave"] = rng.integers(0, 8, N); kps["angle"] = rng.uniform(0, 360, N).astype(np.float32)
        base = rng.integers(0, 256, (60, 32), dtype=np.uint8)
        desc = base[rng.integers(0, 60, N)].copy()
        desc[rng.random((N, 32)) < 0.03] ^= 4
        nq = 700
        qi = rng.integers(0, N, nq)
        qd = desc[qi].copy(); qd[rng.random((nq, 32)) < 0.02] ^= 16
        cand.append(free_frame(kps, desc))
        qry.append(dict(d=qd, u=kps["x"][qi], v=kps["y"][qi], r=np.full(nq, 5.0e3, np.float32), lo=np.full(nq, -1, np.int32), hi=np.full(nq, -1, np.int32),
                        flags=np.full(nq, 3, np.uint8)))
    check(pkg, oracle, cand[:12], qry[:12], bounds, sf, nnratio=0.7, th=60, min_total=2000)
    check(pkg, oracle, cand[12:] * 3, qry[12:] * 3, bounds, sf, nnratio=0.7, th=60, min_total=100)


def test_in_view_queries_whose_window_is_outside_the_grid(pkg, oracle, stream):
    """A query that is in view but whose window lies beyond the grid (Frame::GetFeaturesInArea's early returns, Frame.cc:757-777: a
    projection far outside the image) has no candidates.  It does not keep its frame pair from being fused (the vote counts queries
    with a window inside the grid), so the fused resolve itself must treat it as dead - found by tests/fuzz_parity.py with th 255,
    where such a query otherwise takes a keypoint."""
    ext, offs, sf = stream
    bounds = (-179.5, 909.66, -101.45, 593.57)
    cand = [free_frame(*ext[p + 1]) for p in range(12)]
    qry = []
    for p in range(12):
        q = open_queries(ext[p][0], ext[p][1], (offs[p][0] - offs[p + 1][0], offs[p][1] - offs[p + 1][1]))
        far = np.arange(7, len(q["u"]), 53)
        q["u"] = q["u"].copy(); q["v"] = q["v"].copy()
        q["u"][far] = np.float32(25581.0); q["v"][far[::2]] = np.float32(-17992.0)
        qry.append(q)
    for th in (255, 100):
        ref = check(pkg, oracle, cand, qry, bounds, sf, nnratio=0.6, th=th, second=False)
        assert all((r[1][np.arange(7, len(r[1]), 53)] == -1).all() for r in ref)


def _dist_matrix(q, c):
    return np.unpackbits(q[:, None, :] ^ c[None, :, :], axis=2).sum(axis=2).astype(np.int64)


@pytest.mark.parametrize("nq,nc", [(1000, 1000), (7, 301), (129, 257), (1, 1), (300, 2500)])
def test_hamming_matrix_both_engines(pkg, synth, nq, nc):
    """orbm_hamming_matrix on the matrix pipe (engines 1 / 2: k_hamming_matrix_mfma) and on the vector ALU (engine 0) against numpy:
    query slabs and candidate blocks with ragged tails, a single pair, more than one 256-candidate block."""
    n = max(nq, nc)
    q, c = synth.make_descriptor_sets(2100 + n, n=n)
    q, c = q[:nq], c[:nc]
    ref = _dist_matrix(q, c).astype(np.uint16)
    for engine in (2, 0):
        m = pkg.ORBmatcher(0.8, True)
        m.set_hamming_engine(engine)
        assert np.array_equal(m.hamming_matrix(q, c), ref), engine
        m.close()


@pytest.mark.parametrize("nq,nc", [(700, 705), (260, 33), (3, 1), (65, 2049), (40, 4500)])
def test_knn_match2_both_engines(pkg, synth, nq, nc):
    """knnMatch(k = 2) (Frame.cc:1246) on the matrix pipe (k_knn2_mfma: sorted pair per lane, 2048-descriptor super-blocks) and on the
    vector ALU: duplicated train descriptors (tied distances, ordered by train index) across tile, half-tile and super-block borders."""
    n = max(nq, nc)
    q, c = synth.make_descriptor_sets(5500 + n, n=n)
    q, c = q[:nq].copy(), c[:nc].copy()
    if nc > 40:
        c[-5:] = c[:5]                      # the same descriptor at both ends of the train set
        c[33] = c[31]; c[16] = c[15]        # across a tile border, across the two halves of a tile
        q[0] = c[15]                        # distance 0 twice
    D = _dist_matrix(q, c)
    order = np.argsort(D * (1 << 20) + np.arange(nc)[None, :], axis=1)[:, :2]
    for engine in (2, 0):
        m = pkg.ORBmatcher(0.8, True)
        m.set_hamming_engine(engine)
        idx, dist = m.knnMatch2(q, c)
        if nc >= 2:
            assert np.array_equal(idx, order.astype(np.int32)), engine
            assert np.array_equal(dist, np.take_along_axis(D, order, axis=1).astype(np.int32)), engine
        else:
            assert (idx[:, 0] == 0).all() and (idx[:, 1] == -1).all() and (dist[:, 1] == -1).all() and np.array_equal(dist[:, 0], D[:, 0]), engine
        m.close()
