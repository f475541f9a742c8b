"""GPU parity for BASELINE config 5 as ONE workload: 512x512 frames, 1500 features (Examples/Monocular/TUM_512.yaml:36-51), last-frame
SearchByProjection (ORBmatcher.cc:2027-2289) whose window centres come from KannalaBrandt8::project (KannalaBrandt8.cpp:29-45) with
the TUM-VI parameters (TUM_512.yaml:9-19).  The map points are built by KB8 UN-projection of the shifted keypoints, so that the
search finds most of the ~1500 correspondences wherever they lie in the fisheye image; everything is compared index-exact with
the oracle: extraction (keypoints, descriptors) and the search (slot array, counts) - through the host entry point and through the
batched device entry point the benchmark's tumvi mode runs."""
import ctypes as C

import numpy as np
import pytest

from conftest import TUMVI

pytestmark = pytest.mark.gpu

H = W = 512
BOUNDS = (0.0, 512.0, 0.0, 512.0)     # KannalaBrandt8 frames: mDistCoef = 0 -> bounds = image (Frame.cc:892-898)


def _extract_pair(pkg, oracle, synth, seed):
    frames, offs = synth.make_stream(seed, 2, H, W)
    ex, ox = pkg.ORBextractor(**TUMVI), oracle.OracleExtractor(**TUMVI)
    out = []
    for f in frames:
        mono, k, d = ex(f, None, (0, 1000))
        mono_r, k_r, d_r = ox.extract(f, (0, 1000))
        assert mono == mono_r and k.tobytes() == k_r.tobytes() and np.array_equal(d, d_r)
        out.append((k, d))
    ex.close()
    return out, offs, ox.scale_factors


@pytest.mark.parametrize("seed", [2100, 2101])
@pytest.mark.parametrize("th", [15.0, 30.0])       # Tracking.cc:2898-2915: th = 15 (mono), retry with 2 * th
def test_config5_extract_and_kb8_last_frame_search(pkg, oracle, synth, seed, th):
    (k0, d0), (k1, d1) = _extract_pair(pkg, oracle, synth, seed)[0]
    frames, offs = synth.make_stream(seed, 2, H, W)
    sf = oracle.OracleExtractor(**TUMVI).scale_factors
    assert len(k0) >= 1500 and len(k1) >= 1500
    shift = (offs[0][0] - offs[1][0], offs[0][1] - offs[1][1])
    P = synth.TUMVI_KB8
    Xw, Tcw, Tlw = synth.make_last_frame_scene(1, P, k0["x"], k0["y"], shift, seed)
    rng = np.random.default_rng(seed)
    has_mp = (rng.random(len(k0)) < 0.9).astype(np.uint8)
    obs = (rng.random(len(k0)) < 0.9).astype(np.uint8)
    Xw[rng.random(len(k0)) < 0.02] *= np.float32(-1)          # behind the camera: invzc < 0 (:2076-2079)
    m = pkg.ORBmatcher(0.9, True)                             # Tracking.cc:2872
    F = pkg.FrameView(k1, d1, BOUNDS)
    OF = oracle.OracleFrame(k1["x"], k1["y"], k1["octave"], k1["angle"], d1, BOUNDS, sf)
    n_gpu = m.SearchByProjectionLastFrame(F, sf, has_mp, Xw, d0, k0, Tcw, Tlw, 1, P, th, bMono=True, mp_obs=obs)
    n_ref = OF.search_by_projection_ff(has_mp, Xw, d0, k0["octave"], k0["angle"], Tcw, Tlw, 1, P, th, mono=True, check_ori=True, qobs=obs)
    m.close()
    assert n_gpu == n_ref
    assert np.array_equal(F.slot, OF.slot) and np.array_equal(F.slot_obs, OF.slot_obs)
    assert n_ref >= 600, n_ref                                # most of the ~1350 live map points are found again


@pytest.mark.parametrize("check_ori", [True, False])
def test_config5_batch_device(pkg, oracle, synth, check_ori):
    """orbm_search_by_projection_last_frame_batch_device: B frames of a 512x512 stream resident in HBM, extracted by
    orbx_extract_batch_device, pair p = (frame p, frame p+1); projection + search + rotation pruning on the device, compared
    pair by pair with the oracle run on the oracle's own extraction of the same frames."""
    import torch
    B = 7
    seed = 2200
    frames, offs = synth.make_stream(seed, B, H, W)
    dev = torch.device("cuda", 0)
    ex = pkg.ORBextractor(**TUMVI)
    ox = oracle.OracleExtractor(**TUMVI)
    sf = ox.scale_factors
    cap = ex.configure(H, W, B)
    d_img = torch.from_numpy(frames).to(dev)
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    ex.extract_batch_device(d_img.data_ptr(), H, W, W, H * W, B, d_kps.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap, (0, 1000), stream=0)
    torch.cuda.synchronize()
    cnt = d_cnt.cpu().numpy()
    kps = d_kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28)
    ref = [ox.extract(f, (0, 1000)) for f in frames]
    for b in range(B):
        assert cnt[b, 0] == len(ref[b][1]) and kps[b, :cnt[b, 0]].tobytes() == ref[b][1].tobytes()
    # caller side: map points of pair p from the keypoints of frame p (host numpy, then resident on the device)
    P = synth.TUMVI_KB8
    np_ = B - 1
    Xw = np.zeros((np_, cap, 3), np.float32)
    Tcw = np.zeros((np_, 16), np.float32)
    Tlw = np.zeros((np_, 16), np.float32)
    has = np.zeros((np_, cap), np.uint8)
    rng = np.random.default_rng(9)
    for p in range(np_):
        k0 = ref[p][1]
        shift = (offs[p][0] - offs[p + 1][0], offs[p][1] - offs[p + 1][1])
        x, T, Tl = synth.make_last_frame_scene(1, P, k0["x"], k0["y"], shift, seed + p)
        Xw[p, :len(k0)] = x
        Tcw[p], Tlw[p] = T.reshape(-1), Tl.reshape(-1)
        has[p, :len(k0)] = rng.random(len(k0)) < 0.92
    t = lambda a: torch.from_numpy(a).to(dev)
    d_Xw, d_Tcw, d_Tlw, d_has = t(Xw), t(Tcw), t(Tlw), t(has)
    d_slot = torch.full((np_, cap), -1, dtype=torch.int32, device=dev)
    d_sobs = torch.zeros((np_, cap), dtype=torch.uint8, device=dev)
    d_moq = torch.zeros((np_, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros((np_,), dtype=torch.int32, device=dev)
    m = pkg.ORBmatcher(0.9, check_ori)
    cur = pkg.FrameStruct(cap, d_kps[1:].data_ptr(), d_desc[1:].data_ptr(), None, *[C.c_float(b) for b in BOUNDS])
    last = pkg.LastFrameStruct(cap, d_has.data_ptr(), d_Xw.data_ptr(), d_desc.data_ptr(), d_kps.data_ptr(), None, d_Tcw.data_ptr(), d_Tlw.data_ptr())
    sfa = np.ascontiguousarray(sf, np.float32)
    rc = m.L.orbm_search_by_projection_last_frame_batch_device(
        m.m, C.byref(cur), cap, C.c_void_p(d_cnt[1:].data_ptr()), 2, C.byref(last), cap, C.c_void_p(d_cnt.data_ptr()), 2, np_,
        sfa.ctypes.data_as(C.c_void_p), len(sfa), 1, P.ctypes.data_as(C.c_void_p), C.c_float(0.0), C.c_float(0.0), C.c_float(15.0), 1, int(check_ori),
        C.c_void_p(d_slot.data_ptr()), C.c_void_p(d_sobs.data_ptr()), C.c_void_p(d_moq.data_ptr()), C.c_void_p(d_nm.data_ptr()), None)
    assert rc == 0, m.L.orbm_last_error(m.m)
    torch.cuda.synchronize()
    slot, sobs, moq, nm = d_slot.cpu().numpy(), d_sobs.cpu().numpy(), d_moq.cpu().numpy(), d_nm.cpu().numpy()
    total = 0
    for p in range(np_):
        (_, k0, d0), (_, k1, d1) = ref[p], ref[p + 1]
        OF = oracle.OracleFrame(k1["x"], k1["y"], k1["octave"], k1["angle"], d1, BOUNDS, sf)
        n_ref = OF.search_by_projection_ff(has[p, :len(k0)], Xw[p, :len(k0)], d0, k0["octave"], k0["angle"], Tcw[p].reshape(4, 4), Tlw[p].reshape(4, 4), 1, P,
                                           15.0, mono=True, check_ori=check_ori)
        assert nm[p] == n_ref, (p, nm[p], n_ref)
        assert np.array_equal(slot[p, :len(k1)], OF.slot) and np.array_equal(sobs[p, :len(k1)], OF.slot_obs)
        want = np.full(len(k0), -1, np.int32)                   # match_of_query = inverse of the slot array
        hit = np.nonzero(OF.slot >= 0)[0]
        want[OF.slot[hit]] = hit
        assert np.array_equal(moq[p, :len(k0)], want)
        total += n_ref
    m.close(); ex.close()
    assert total >= 600 * np_
