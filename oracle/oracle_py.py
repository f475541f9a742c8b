"""ctypes view of oracle/liborb_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product package
(3_orb_slam3_selfnote_amd) never does.  See oracle/orb_oracle.h for the reference file:line each function follows.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liborb_oracle.so")


def build(force=False):
    if force or not os.path.exists(_LIB) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB) for f in ("orb_oracle.c", "orb_oracle.h", "orb_cpu_bench.c", "Makefile")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB


class Extractor(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("nlevels", C.c_int), ("iniThFAST", C.c_int), ("minThFAST", C.c_int),
                ("scaleFactor", C.c_double),
                ("mvScaleFactor", C.c_float * 16), ("mvInvScaleFactor", C.c_float * 16),
                ("mvLevelSigma2", C.c_float * 16), ("mvInvLevelSigma2", C.c_float * 16),
                ("mnFeaturesPerLevel", C.c_int * 16), ("umax", C.c_int * 16)]


class Frame(C.Structure):
    _fields_ = [("N", C.c_int), ("kx", C.c_void_p), ("ky", C.c_void_p), ("octave", C.c_void_p), ("angle", C.c_void_p),
                ("desc", C.c_void_p), ("uRight", C.c_void_p),
                ("mnMinX", C.c_float), ("mnMaxX", C.c_float), ("mnMinY", C.c_float), ("mnMaxY", C.c_float),
                ("mfGridElementWidthInv", C.c_float), ("mfGridElementHeightInv", C.c_float),
                ("mvScaleFactors", C.c_void_p), ("nlevels", C.c_int),
                ("cell_start", C.c_int32 * (64 * 48 + 1)), ("cell_idx", C.c_void_p)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_fast_atan2.restype = C.c_float
        _lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.orc_ic_angle.restype = C.c_float
        _lib.orc_ic_angle.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float]
        _lib.orc_compute_descriptor.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_void_p]
        _lib.orc_libm_cosf.restype = C.c_float
        _lib.orc_libm_cosf.argtypes = [C.c_float]
        _lib.orc_libm_sinf.restype = C.c_float
        _lib.orc_libm_sinf.argtypes = [C.c_float]
        _lib.orc_radius_by_viewing_cos.restype = C.c_float
        _lib.orc_radius_by_viewing_cos.argtypes = [C.c_float]
        _lib.orc_cvRound.argtypes = [C.c_double]
        _lib.orc_get_features_in_area.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p]
        _lib.orc_project.argtypes = [C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        _lib.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_size_t]
        _lib.orc_gaussian_blur7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
        _lib.orc_fast9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_int]
        _lib.orc_fast_corner_score.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        _lib.orc_copy_make_border101.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t]
        _lib.orc_level_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int]
        _lib.orc_distribute_octtree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        _lib.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib.orc_search_by_projection_mp.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_float, C.c_float] + [C.c_void_p] * 3
        _lib.orc_search_by_projection_win.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 4
        _lib.orc_search_by_projection_ff.argtypes = ([C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p] +
                                                     [C.c_float] * 3 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p])
        _lib.orc_search_by_projection_mp_fisheye.argtypes = ([C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 12 + [C.c_float, C.c_float] + [C.c_void_p] * 4)
        _lib.orc_search_by_projection_ff_fisheye.argtypes = ([C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 9 + [C.c_int, C.c_void_p] +
                                                             [C.c_float] * 2 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p])
        _lib.orc_search_for_initialization.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_float, C.c_int, C.c_void_p]
        _lib.orc_compute_stereo_matches.argtypes = ([C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4 +
                                                    [C.c_float, C.c_float, C.c_void_p, C.c_void_p])
        _lib.orc_frame_init.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_float] * 4 + [C.c_void_p, C.c_int]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleExtractor:
    """ORBextractor restatement (ORBextractor.cc:408-468 ctor, :1071-1184 operator())."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7):
        self.L = lib()
        self.e = Extractor()
        self.L.orc_extractor_init(C.byref(self.e), int(nfeatures), C.c_float(scaleFactor), int(nlevels), int(iniThFAST), int(minThFAST))
        self.nlevels = nlevels
        self.nfeatures = nfeatures

    @property
    def scale_factors(self):
        return np.array(self.e.mvScaleFactor[:self.nlevels], dtype=np.float32)

    @property
    def features_per_level(self):
        return list(self.e.mnFeaturesPerLevel[:self.nlevels])

    @property
    def umax(self):
        return list(self.e.umax)

    def level_size(self, level, cols, rows):
        lc, lr = C.c_int(), C.c_int()
        self.L.orc_level_size(C.byref(self.e), level, cols, rows, C.byref(lc), C.byref(lr))
        return lc.value, lr.value

    def pyramid(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        rows, cols = img.shape
        levels = []
        for l in range(self.nlevels):
            lc, lr = self.level_size(l, cols, rows)
            levels.append(np.zeros((lr, lc), dtype=np.uint8))
        arr = (C.c_void_p * self.nlevels)(*[_p(a) for a in levels])
        self.L.orc_compute_pyramid(C.byref(self.e), _p(img), cols, rows, C.c_size_t(img.strides[0]), arr)
        return levels

    def compute_stereo_matches(self, imgL, imgR, keysL, descL, keysR, descR, mb, mbf):
        """Frame::ComputeStereoMatches (Frame.cc:901-1079) on the two rectified images; returns (mvuRight, mvDepth)."""
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        lvL, lvR = self.pyramid(imgL), self.pyramid(imgR)
        arrL = (C.c_void_p * self.nlevels)(*[_p(x) for x in lvL])
        arrR = (C.c_void_p * self.nlevels)(*[_p(x) for x in lvR])
        rows, cols = imgL.shape
        nL, nR = len(keysL), len(keysR)
        uR, depth = np.zeros(nL, np.float32), np.zeros(nL, np.float32)
        args = [a(keysL["x"], np.float32), a(keysL["y"], np.float32), a(keysL["octave"], np.int32), a(descL, np.uint8)]
        argsR = [a(keysR["x"], np.float32), a(keysR["y"], np.float32), a(keysR["octave"], np.int32), a(descR, np.uint8)]
        self.L.orc_compute_stereo_matches(C.byref(self.e), arrL, arrR, cols, rows, nL, *[_p(x) for x in args], nR, *[_p(x) for x in argsR],
                                          C.c_float(mb), C.c_float(mbf), _p(uR), _p(depth))
        return uR, depth

    def level_candidates(self, level_img):
        level_img = np.ascontiguousarray(level_img, dtype=np.uint8)
        h, w = level_img.shape
        cap = (w * h) // 4 + 16
        out = np.zeros((cap, 3), dtype=np.float32)
        n = self.L.orc_level_candidates(C.byref(self.e), _p(level_img), w, h, C.c_size_t(level_img.strides[0]), _p(out), cap)
        return out[:n].copy()

    def ic_angle(self, level_img, x, y):
        level_img = np.ascontiguousarray(level_img, dtype=np.uint8)
        return float(self.L.orc_ic_angle(C.byref(self.e), _p(level_img), C.c_size_t(level_img.strides[0]), C.c_float(x), C.c_float(y)))

    def extract(self, img, lap=(0, 1000), cap=None):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        rows, cols = img.shape
        if cap is None:
            cap = self.nfeatures + 3 * self.nlevels + 64
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = C.c_int(0)
        mono = self.L.orc_extract(C.byref(self.e), _p(img), rows, cols, C.c_size_t(img.strides[0]), int(lap[0]), int(lap[1]),
                                  _p(kps), _p(desc), cap, C.byref(n))
        if mono == -2:
            return self.extract(img, lap, cap=n.value + 8)
        return mono, kps[:n.value].copy(), desc[:n.value].copy()


def distribute_octtree(xyr, minX, maxX, minY, maxY, N):
    xyr = np.ascontiguousarray(xyr, dtype=np.float32)
    n = xyr.shape[0]
    out = np.zeros((n + 8, 3), dtype=np.float32)
    m = lib().orc_distribute_octtree(_p(xyr), n, minX, maxX, minY, maxY, N, _p(out), n + 8)
    return out[:m].copy()


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros((dh, dw), dtype=np.uint8)
    lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], C.c_size_t(src.strides[0]), _p(dst), dw, dh, C.c_size_t(dw))
    return dst


def gaussian_blur7(src):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros_like(src)
    lib().orc_gaussian_blur7(_p(src), src.shape[1], src.shape[0], C.c_size_t(src.strides[0]), _p(dst), C.c_size_t(dst.strides[0]))
    return dst


def fast9_16(img, threshold):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    cap = w * h // 4 + 16
    out = np.zeros((cap, 3), dtype=np.int32)
    n = lib().orc_fast9_16(_p(img), w, h, C.c_size_t(img.strides[0]), threshold, _p(out), cap)
    return out[:n].copy()


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def compute_descriptor(blurred, x, y, angle):
    blurred = np.ascontiguousarray(blurred, dtype=np.uint8)
    d = np.zeros(32, dtype=np.uint8)
    lib().orc_compute_descriptor(_p(blurred), C.c_size_t(blurred.strides[0]), C.c_float(x), C.c_float(y), C.c_float(angle), _p(d))
    return d


class OracleFrame:
    """The slice of Frame the matcher reads (Frame.cc:379-380, 434-465)."""

    def __init__(self, kx, ky, octave, angle, desc, bounds, scale_factors, u_right=None):
        self.L = lib()
        self.kx = np.ascontiguousarray(kx, dtype=np.float32)
        self.ky = np.ascontiguousarray(ky, dtype=np.float32)
        self.octave = np.ascontiguousarray(octave, dtype=np.int32)
        self.angle = np.ascontiguousarray(angle, dtype=np.float32)
        self.desc = np.ascontiguousarray(desc, dtype=np.uint8)
        self.sf = np.ascontiguousarray(scale_factors, dtype=np.float32)
        self.u_right = None if u_right is None else np.ascontiguousarray(u_right, dtype=np.float32)
        self.N = len(self.kx)
        self.f = Frame()
        self.L.orc_frame_init(C.byref(self.f), self.N, _p(self.kx), _p(self.ky), _p(self.octave), _p(self.angle),
                              _p(self.desc), _p(self.u_right), *[C.c_float(b) for b in bounds], _p(self.sf), len(self.sf))
        self.slot = np.full(self.N, -1, dtype=np.int32)
        self.slot_obs = np.zeros(self.N, dtype=np.uint8)

    def __del__(self):
        try:
            self.L.orc_frame_free(C.byref(self.f))
        except Exception:
            pass

    def grid_csr(self):
        start = np.array(self.f.cell_start[:], dtype=np.int32)
        n = int(start[-1])
        idx = np.ctypeslib.as_array(C.cast(self.f.cell_idx, C.POINTER(C.c_int32)), shape=(max(n, 1),))[:n].copy()
        return start, idx

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        out = np.zeros(max(self.N, 1), dtype=np.int32)
        n = self.L.orc_get_features_in_area(C.byref(self.f), C.c_float(x), C.c_float(y), C.c_float(r), min_level, max_level, _p(out))
        return out[:n].copy()

    def search_by_projection_mp(self, in_view, qdesc, projX, projY, viewCos, level, th, nnratio, qobs=None, projXR=None):
        nq = len(projX)
        a = lambda v, t: np.ascontiguousarray(v, dtype=t)
        in_view, qdesc = a(in_view, np.uint8), a(qdesc, np.uint8)
        projX, projY, viewCos, level = a(projX, np.float32), a(projY, np.float32), a(viewCos, np.float32), a(level, np.int32)
        projXR = a(projXR, np.float32) if projXR is not None else np.zeros(nq, np.float32)
        qobs = a(qobs, np.uint8) if qobs is not None else np.ones(nq, np.uint8)
        moq = np.full(nq, -1, dtype=np.int32)
        n = self.L.orc_search_by_projection_mp(C.byref(self.f), nq, _p(in_view), _p(qdesc), _p(projX), _p(projY), _p(projXR),
                                               _p(viewCos), _p(level), _p(qobs), C.c_float(th), C.c_float(nnratio),
                                               _p(self.slot), _p(self.slot_obs), _p(moq))
        return n, moq

    def search_by_projection_win(self, qdesc, u, v, radius, min_level, max_level, nnratio=0.8, th_high=100,
                                 mode_second=True, qobs=None, in_view=None):
        nq = len(u)
        if not 0 <= th_high <= 255:
            raise ValueError("th_high >= 256 accepts a query without candidates: index -1 in the reference (undefined)")
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        qdesc, u, v, radius = a(qdesc, np.uint8), a(u, np.float32), a(v, np.float32), a(radius, np.float32)
        min_level, max_level = a(min_level, np.int32), a(max_level, np.int32)
        qobs = a(qobs, np.uint8) if qobs is not None else np.ones(nq, np.uint8)
        in_view = a(in_view, np.uint8) if in_view is not None else np.ones(nq, np.uint8)
        moq = np.full(nq, -1, dtype=np.int32)
        bd = np.full(nq, 256, dtype=np.int32)
        n = self.L.orc_search_by_projection_win(C.byref(self.f), nq, _p(in_view), _p(qdesc), _p(u), _p(v), _p(radius),
                                                _p(min_level), _p(max_level), _p(qobs), C.c_float(nnratio), int(th_high),
                                                int(bool(mode_second)), _p(self.slot), _p(self.slot_obs), _p(moq), _p(bd))
        return n, moq, bd

    def search_by_projection_ff(self, has_mp, Xw, mpdesc, last_octave, last_angle, Tcw, Tlw, cam_type, cam_params,
                                th, mono=True, check_ori=True, mb=0.0, mbf=0.0, qobs=None):
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        n_last = len(has_mp)
        has_mp, Xw, mpdesc = a(has_mp, np.uint8), a(Xw, np.float32), a(mpdesc, np.uint8)
        last_octave, last_angle = a(last_octave, np.int32), a(last_angle, np.float32)
        Tcw, Tlw, cam_params = a(Tcw, np.float32), a(Tlw, np.float32), a(cam_params, np.float32)
        qobs = a(qobs, np.uint8) if qobs is not None else np.ones(n_last, np.uint8)
        return self.L.orc_search_by_projection_ff(C.byref(self.f), n_last, _p(has_mp), _p(Xw), _p(mpdesc), _p(last_octave),
                                                  _p(last_angle), _p(qobs), _p(Tcw), _p(Tlw), int(cam_type), _p(cam_params),
                                                  C.c_float(mb), C.c_float(mbf), C.c_float(th), int(mono), int(check_ori),
                                                  _p(self.slot), _p(self.slot_obs))


def search_for_initialization(keys1, desc1, F2, prev_matched, window_size, nnratio=0.9, check_ori=True):
    """ORBmatcher::SearchForInitialization (ORBmatcher.cc:722-837); F2 = OracleFrame; prev_matched (n1, 2) float32 in/out."""
    a = lambda x, t: np.ascontiguousarray(x, dtype=t)
    n1 = len(keys1)
    octave, angle, desc1 = a(keys1["octave"], np.int32), a(keys1["angle"], np.float32), a(desc1, np.uint8)
    m12 = np.full(n1, -1, np.int32)
    n = lib().orc_search_for_initialization(n1, _p(octave), _p(angle), _p(desc1), C.byref(F2.f), _p(prev_matched), int(window_size),
                                            C.c_float(nnratio), int(check_ori), _p(m12))
    return n, m12


class OracleFisheyeFrame:
    """A fisheye-stereo Frame (Nleft != -1): left grid over mvKeys, right grid over mvKeysRight, one mvpMapPoints array."""

    def __init__(self, left, right):
        self.L = lib()
        self.left, self.right = left, right            # OracleFrame each (same bounds)
        self.n_left = left.N
        self.slot = np.full(left.N + right.N, -1, dtype=np.int32)
        self.slot_obs = np.zeros(left.N + right.N, dtype=np.uint8)

    def search_by_projection_mp(self, l2r, r2l, in_view, in_view_r, qdesc, projX, projY, viewCos, level, projXR, projYR, viewCosR,
                                levelR, th, nnratio, qobs=None):
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        nmp = len(level)
        ml, mr = np.full(nmp, -1, np.int32), np.full(nmp, -1, np.int32)
        l2r, r2l = a(l2r, np.int32), a(r2l, np.int32)
        args = [a(in_view, np.uint8), a(in_view_r, np.uint8), a(qdesc, np.uint8), a(projX, np.float32), a(projY, np.float32),
                a(viewCos, np.float32), a(level, np.int32), a(projXR, np.float32), a(projYR, np.float32), a(viewCosR, np.float32),
                a(levelR, np.int32), a(qobs, np.uint8) if qobs is not None else np.ones(nmp, np.uint8)]
        n = self.L.orc_search_by_projection_mp_fisheye(C.byref(self.left.f), C.byref(self.right.f), _p(l2r), _p(r2l), nmp,
                                                       *[_p(x) for x in args], C.c_float(th), C.c_float(nnratio),
                                                       _p(self.slot), _p(self.slot_obs), _p(ml), _p(mr))
        return n, ml, mr

    def search_by_projection_ff(self, has_mp, Xw, mpdesc, last_octave, last_angle, Tcw, Tlw, Trl, cam_type, cam_params, th,
                                mono=False, check_ori=True, mb=0.0, qobs=None):
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        n_last = len(has_mp)
        has_mp, Xw, mpdesc = a(has_mp, np.uint8), a(Xw, np.float32), a(mpdesc, np.uint8)
        last_octave, last_angle = a(last_octave, np.int32), a(last_angle, np.float32)
        Tcw, Tlw, Trl, cam_params = a(Tcw, np.float32), a(Tlw, np.float32), a(Trl, np.float32), a(cam_params, np.float32)
        qobs = a(qobs, np.uint8) if qobs is not None else np.ones(n_last, np.uint8)
        return self.L.orc_search_by_projection_ff_fisheye(C.byref(self.left.f), C.byref(self.right.f), n_last, _p(has_mp), _p(Xw),
                                                          _p(mpdesc), _p(last_octave), _p(last_angle), _p(qobs), _p(Tcw), _p(Tlw),
                                                          _p(Trl), int(cam_type), _p(cam_params), C.c_float(mb), C.c_float(th),
                                                          int(mono), int(check_ori), _p(self.slot), _p(self.slot_obs))


class KeyFrame(C.Structure):
    _fields_ = [("N", C.c_int), ("kx", C.c_void_p), ("ky", C.c_void_p), ("octave", C.c_void_p), ("angle", C.c_void_p), ("desc", C.c_void_p),
                ("uRight", C.c_void_p), ("has_mp", C.c_void_p), ("n_nodes", C.c_int), ("node_id", C.c_void_p), ("node_start", C.c_void_p),
                ("node_idx", C.c_void_p), ("scaleFactors", C.c_void_p), ("levelSigma2", C.c_void_p)]


class OracleKeyFrame:
    """The slice of KeyFrame read by SearchForTriangulation (ORBmatcher.cc:981-1222)."""

    def __init__(self, keys_un, desc, feat_vec, scale_factors, level_sigma2, u_right=None, has_mp=None):
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        self.N = len(keys_un)
        self.kx, self.ky = a(keys_un["x"], np.float32), a(keys_un["y"], np.float32)
        self.octave, self.angle = a(keys_un["octave"], np.int32), a(keys_un["angle"], np.float32)
        self.desc = a(desc, np.uint8)
        self.u_right = np.full(self.N, -1, np.float32) if u_right is None else a(u_right, np.float32)
        self.has_mp = np.zeros(self.N, np.uint8) if has_mp is None else a(has_mp, np.uint8)
        ids = sorted(feat_vec.keys())
        self.node_id = np.array(ids, dtype=np.uint32)
        self.node_start = np.zeros(len(ids) + 1, dtype=np.int32)
        idx = []
        for k, nid in enumerate(ids):
            idx.extend(int(i) for i in feat_vec[nid])
            self.node_start[k + 1] = len(idx)
        self.node_idx = np.array(idx, dtype=np.int32) if idx else np.zeros(1, np.int32)
        self.sf, self.sigma2 = a(scale_factors, np.float32), a(level_sigma2, np.float32)
        self.k = KeyFrame(self.N, *[v.ctypes.data for v in (self.kx, self.ky, self.octave, self.angle, self.desc, self.u_right, self.has_mp)],
                          len(ids), self.node_id.ctypes.data, self.node_start.ctypes.data, self.node_idx.ctypes.data, self.sf.ctypes.data,
                          self.sigma2.ctypes.data)


def search_for_triangulation(K1, K2, R1w, t1w, R2w, t2w, Cw1, cam1, cam2, only_stereo=False, coarse=False, check_ori=False):
    a = lambda x: np.ascontiguousarray(x, dtype=np.float32)
    R1w, t1w, R2w, t2w, Cw1, cam1, cam2 = a(R1w), a(t1w), a(R2w), a(t2w), a(Cw1), a(cam1), a(cam2)
    m12 = np.full(max(K1.N, 1), -1, dtype=np.int32)
    L = lib()
    L.orc_search_for_triangulation.argtypes = [C.c_void_p] * 9 + [C.c_int] * 3 + [C.c_void_p]
    n = L.orc_search_for_triangulation(C.byref(K1.k), C.byref(K2.k), _p(R1w), _p(t1w), _p(R2w), _p(t2w), _p(Cw1), _p(cam1), _p(cam2),
                                       int(only_stereo), int(coarse), int(check_ori), _p(m12))
    m12 = m12[:K1.N]
    i1 = np.nonzero(m12 >= 0)[0]
    return n, np.stack([i1, m12[i1]], axis=1).astype(np.int64)


PAIR_PRED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)


def search_for_triangulation_pred(K1, K2, ep, epipole_gate, pred, only_stereo=False, coarse=False, check_ori=False):
    """SearchForTriangulation around an injected epipolar predicate pred(idx1, idx2) -> bool (ORBmatcher.cc:1148's virtual call)."""
    m12 = np.full(max(K1.N, 1), -1, dtype=np.int32)
    L = lib()
    cb = PAIR_PRED(lambda user, i1, i2: 1 if pred(i1, i2) else 0)
    L.orc_search_for_triangulation_pred.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, PAIR_PRED, C.c_void_p, C.c_void_p]
    n = L.orc_search_for_triangulation_pred(C.byref(K1.k), C.byref(K2.k), C.c_float(ep[0]), C.c_float(ep[1]), int(bool(epipole_gate)), int(only_stereo),
                                            int(coarse), int(check_ori), cb, None, _p(m12))
    m12 = m12[:K1.N]
    i1 = np.nonzero(m12 >= 0)[0]
    return n, np.stack([i1, m12[i1]], axis=1).astype(np.int64)


def triangulation_candidates(K1, K2, ep, epipole_gate, only_stereo=False):
    """(start[K1.N + 1], idx2[], dist[]): the candidate lists in front of the predicate, ordered (dist ascending, node position descending)."""
    L = lib()
    L.orc_triangulation_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    start = np.zeros(K1.N + 1, np.int32)
    cap = 1 << 16
    while True:
        idx2, dist = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        tot = L.orc_triangulation_candidates(C.byref(K1.k), C.byref(K2.k), C.c_float(ep[0]), C.c_float(ep[1]), int(bool(epipole_gate)), int(only_stereo),
                                             _p(start), _p(idx2), _p(dist), cap)
        if tot <= cap:
            return start, idx2[:tot].copy(), dist[:tot].copy()
        cap = tot


def pinhole_pair_geometry(R1w, t1w, R2w, t2w, Cw1, cam1, cam2):
    """(epipole in image 2, F12) of a Pinhole keyframe pair, ORBmatcher.cc:988-1010 + Pinhole.cpp:143-148."""
    a = lambda x: np.ascontiguousarray(x, dtype=np.float32)
    R1w, t1w, R2w, t2w, Cw1, cam1, cam2 = a(R1w), a(t1w), a(R2w), a(t2w), a(Cw1), a(cam1), a(cam2)
    ep, F12 = np.zeros(2, np.float32), np.zeros(9, np.float32)
    L = lib()
    L.orc_pinhole_pair_geometry.argtypes = [C.c_void_p] * 9
    L.orc_pinhole_pair_geometry(_p(R1w), _p(t1w), _p(R2w), _p(t2w), _p(Cw1), _p(cam1), _p(cam2), _p(ep), _p(F12))
    return ep, F12


def pinhole_epipolar_constrain(F12, x1, y1, x2, y2, unc):
    """Pinhole::epipolarConstrain (Pinhole.cpp:150-164) for one pair."""
    L = lib()
    L.orc_pinhole_epipolar_constrain.argtypes = [C.c_void_p] + [C.c_float] * 5
    F12 = np.ascontiguousarray(F12, dtype=np.float32)
    return bool(L.orc_pinhole_epipolar_constrain(_p(F12), C.c_float(x1), C.c_float(y1), C.c_float(x2), C.c_float(y2), C.c_float(unc)))


def search_by_bow(KF, F, nnratio=0.7, check_ori=True, n_left=-1):
    """SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (ORBmatcher.cc:273-469) on two OracleKeyFrame views; n_left = F.Nleft."""
    L = lib()
    L.orc_search_by_bow_kf_frame_stereo.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    m = np.full(max(F.N, 1), -1, dtype=np.int32)
    n = L.orc_search_by_bow_kf_frame_stereo(C.byref(KF.k), C.byref(F.k), int(n_left), C.c_float(nnratio), int(check_ori), _p(m))
    return n, m[:F.N]


def search_by_bow_keyframes(K1, K2, nnratio=0.8, check_ori=True):
    """SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (ORBmatcher.cc:839-979) on two OracleKeyFrame views."""
    L = lib()
    L.orc_search_by_bow_kf_kf.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    m = np.full(max(K1.N, 1), -1, dtype=np.int32)
    n = L.orc_search_by_bow_kf_kf(C.byref(K1.k), C.byref(K2.k), C.c_float(nnratio), int(check_ori), _p(m))
    return n, m[:K1.N]


def search_by_projection_kf(F, valid, Xw, mpdesc, kf_angle, max_dist, min_dist, Tcw, cam_type, cam_params, log_scale_factor, th, orb_dist,
                            check_ori=True):
    """M4 on an OracleFrame (ORBmatcher.cc:2291-2413); slot/slot_obs of F are updated in place."""
    a = lambda x, t: np.ascontiguousarray(x, dtype=t)
    valid, Xw, mpdesc = a(valid, np.uint8), a(Xw, np.float32), a(mpdesc, np.uint8)
    kf_angle, max_dist, min_dist = a(kf_angle, np.float32), a(max_dist, np.float32), a(min_dist, np.float32)
    Tcw, cam_params = a(Tcw, np.float32), a(cam_params, np.float32)
    L = lib()
    L.orc_search_by_projection_kf.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int,
                                                                                   C.c_void_p, C.c_void_p]
    return L.orc_search_by_projection_kf(C.byref(F.f), len(valid), _p(valid), _p(Xw), _p(mpdesc), _p(kf_angle), _p(max_dist), _p(min_dist),
                                         _p(Tcw), int(cam_type), _p(cam_params), C.c_float(log_scale_factor), C.c_float(th), int(orb_dist),
                                         int(check_ori), _p(F.slot), _p(F.slot_obs))


def search_by_projection_sim3(F, valid, Xw, normal, mpdesc, max_dist, min_dist, Scw, cam, log_scale_factor, th, ratio_hamming, cam_type=0):
    """M5 on an OracleFrame holding the KeyFrame's keypoints (ORBmatcher.cc:489-602); slot/slot_obs updated in place."""
    a = lambda x, t: np.ascontiguousarray(x, dtype=t)
    valid, Xw, normal, mpdesc = a(valid, np.uint8), a(Xw, np.float32), a(normal, np.float32), a(mpdesc, np.uint8)
    max_dist, min_dist, Scw, cam = a(max_dist, np.float32), a(min_dist, np.float32), a(Scw, np.float32), a(cam, np.float32)
    L = lib()
    L.orc_search_by_projection_sim3_cam.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    return L.orc_search_by_projection_sim3_cam(C.byref(F.f), len(valid), _p(valid), _p(Xw), _p(normal), _p(mpdesc), _p(max_dist), _p(min_dist),
                                               _p(Scw), int(cam_type), _p(cam), C.c_float(log_scale_factor), int(th), C.c_float(ratio_hamming), _p(F.slot), _p(F.slot_obs))


def fuse(F, valid, Xw, normal, mpdesc, max_dist, min_dist, Tcw, Ow, cam_type, cam, bf, inv_sigma2, log_scale_factor, th):
    """Search part of Fuse(KeyFrame*, vpMapPoints, th) (ORBmatcher.cc:1425-1658) on an OracleFrame of the keyframe."""
    a = lambda x, t: np.ascontiguousarray(x, dtype=t)
    n = len(valid)
    bi, bd = np.full(n, -1, np.int32), np.full(n, 256, np.int32)
    L = lib()
    L.orc_fuse.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    args = [a(valid, np.uint8), a(Xw, np.float32), a(normal, np.float32), a(mpdesc, np.uint8), a(max_dist, np.float32), a(min_dist, np.float32),
            a(Tcw, np.float32), a(Ow, np.float32)]
    cam, inv_sigma2 = a(cam, np.float32), a(inv_sigma2, np.float32)
    nf = L.orc_fuse(C.byref(F.f), n, *[_p(x) for x in args], int(cam_type), _p(cam), C.c_float(bf), _p(inv_sigma2), C.c_float(log_scale_factor),
                    C.c_float(th), _p(bi), _p(bd))
    return nf, bi, bd


def fuse_sim3(F, valid, Xw, normal, mpdesc, max_dist, min_dist, Scw, cam, log_scale_factor, th, cam_type=0):
    """Search part of Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) (ORBmatcher.cc:1660-1786)."""
    a = lambda x, t: np.ascontiguousarray(x, dtype=t)
    n = len(valid)
    bi, bd = np.full(n, -1, np.int32), np.full(n, 256, np.int32)
    L = lib()
    L.orc_fuse_sim3_cam.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    args = [a(valid, np.uint8), a(Xw, np.float32), a(normal, np.float32), a(mpdesc, np.uint8), a(max_dist, np.float32), a(min_dist, np.float32),
            a(Scw, np.float32)]
    cam = a(cam, np.float32)
    nf = L.orc_fuse_sim3_cam(C.byref(F.f), n, *[_p(x) for x in args], int(cam_type), _p(cam), C.c_float(log_scale_factor), C.c_float(th), _p(bi), _p(bd))
    return nf, bi, bd


def search_by_sim3(F1, log_sf1, valid1, Xw1, desc1, maxd1, mind1, R1w, t1w, F2, log_sf2, valid2, Xw2, desc2, maxd2, mind2, R2w, t2w, s12, R12, t12,
                   cam1, th):
    """ORBmatcher::SearchBySim3 (ORBmatcher.cc:1788-2012) on two OracleFrame views of the keyframes."""
    a = lambda x, t: np.ascontiguousarray(x, dtype=t)
    L = lib()
    L.orc_search_by_sim3.argtypes = ([C.c_void_p, C.c_float] + [C.c_void_p] * 7 + [C.c_void_p, C.c_float] + [C.c_void_p] * 7 +
                                     [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p])
    m12 = np.full(max(F1.N, 1), -1, np.int32)
    s1 = [a(valid1, np.uint8), a(Xw1, np.float32), a(desc1, np.uint8), a(maxd1, np.float32), a(mind1, np.float32), a(R1w, np.float32), a(t1w, np.float32)]
    s2 = [a(valid2, np.uint8), a(Xw2, np.float32), a(desc2, np.uint8), a(maxd2, np.float32), a(mind2, np.float32), a(R2w, np.float32), a(t2w, np.float32)]
    R12, t12, cam1 = a(R12, np.float32), a(t12, np.float32), a(cam1, np.float32)
    n = L.orc_search_by_sim3(C.byref(F1.f), C.c_float(log_sf1), *[_p(x) for x in s1], C.byref(F2.f), C.c_float(log_sf2), *[_p(x) for x in s2],
                             C.c_float(s12), _p(R12), _p(t12), _p(cam1), C.c_float(th), _p(m12))
    return n, m12[:F1.N]


def distinctive_descriptor(desc):
    """MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:350-436): BestIdx among the rows of desc (N x 32)."""
    desc = np.ascontiguousarray(desc, dtype=np.uint8)
    L = lib()
    L.orc_distinctive_descriptor.argtypes = [C.c_void_p, C.c_int]
    return L.orc_distinctive_descriptor(_p(desc), len(desc))


def undistort_points(xy, K, D):
    xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
    K, D = np.ascontiguousarray(K, dtype=np.float32), np.ascontiguousarray(D, dtype=np.float32)
    out = np.zeros_like(xy)
    L = lib()
    L.orc_undistort_points.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.orc_undistort_points(len(xy), _p(xy), _p(K), _p(D), len(D), _p(out))
    return out


def image_bounds(cols, rows, K, D):
    K, D = np.ascontiguousarray(K, dtype=np.float32), np.ascontiguousarray(D, dtype=np.float32)
    v = [C.c_float() for _ in range(4)]
    L = lib()
    L.orc_image_bounds.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4
    L.orc_image_bounds(cols, rows, _p(K), _p(D), len(D), *[C.byref(x) for x in v])
    return tuple(x.value for x in v)


def project(cam_type, params, X, Y, Z):
    params = np.ascontiguousarray(params, dtype=np.float32)
    u, v = C.c_float(), C.c_float()
    lib().orc_project(cam_type, _p(params), C.c_float(X), C.c_float(Y), C.c_float(Z), C.byref(u), C.byref(v))
    return u.value, v.value


def three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, dtype=np.int32)
    i1, i2, i3 = C.c_int(), C.c_int(), C.c_int()
    lib().orc_three_maxima(_p(sizes), len(sizes), C.byref(i1), C.byref(i2), C.byref(i3))
    return i1.value, i2.value, i3.value


def clahe(im, clip_limit=3.0, tiles=(8, 8)):
    """cv::createCLAHE(clip_limit, Size(tiles))->apply (mono_tum_vi.cc:101-109)."""
    L = lib()
    L.orc_clahe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    im = np.ascontiguousarray(im, dtype=np.uint8)
    out = np.empty_like(im)
    rc = L.orc_clahe(_p(im), im.shape[0], im.shape[1], im.shape[1], float(clip_limit), int(tiles[0]), int(tiles[1]), _p(out), im.shape[1])
    if rc != 0:
        raise ValueError("orc_clahe rc=%d" % rc)
    return out


def remap_linear(im, mapx, mapy):
    """cv::remap(im, out, mapx, mapy, INTER_LINEAR), CV_32FC1 maps, BORDER_CONSTANT 0 (stereo_euroc.cc:166-167)."""
    L = lib()
    L.orc_remap_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    im = np.ascontiguousarray(im, dtype=np.uint8)
    mapx = np.ascontiguousarray(mapx, dtype=np.float32); mapy = np.ascontiguousarray(mapy, dtype=np.float32)
    H, W = mapx.shape
    out = np.empty((H, W), np.uint8)
    rc = L.orc_remap_linear(_p(im), im.shape[0], im.shape[1], im.shape[1], _p(mapx), _p(mapy), W, H, W, _p(out), W)
    if rc != 0:
        raise ValueError("orc_remap_linear rc=%d" % rc)
    return out


class BenchCfg(C.Structure):  # orc_bench_cfg (orb_cpu_bench.c)
    _fields_ = [("nframes", C.c_int), ("count", C.c_int), ("rows", C.c_int), ("cols", C.c_int), ("threads", C.c_int), ("warmup", C.c_int),
                ("lap0", C.c_int), ("lap1", C.c_int), ("cap", C.c_int), ("mode", C.c_int), ("nnratio", C.c_float), ("th_high", C.c_int),
                ("cam_type", C.c_int), ("cam", C.c_void_p), ("Xw", C.c_void_p), ("has_mp", C.c_void_p), ("Tcw", C.c_void_p), ("Tlw", C.c_void_p),
                ("th", C.c_float), ("check_ori", C.c_int), ("bounds", C.c_float * 4),
                ("distort", C.c_int), ("K", C.c_float * 4), ("D", C.c_float * 5), ("nD", C.c_int)]


def bench_stream(ex, frames, offs, count, threads=1, warmup=50, lap=(0, 1000), cap=None, mode=0, nnratio=0.8, th_high=100, scene=None, distort=None):
    """BASELINE.md section 3 protocol over the oracle, natively timed (orb_cpu_bench.c): frames [n, H, W] uint8, offs [n, 2] int32.
    mode 0 = extract + config-3 stress match; mode 1 = extract + last-frame search with `scene` = dict(cam_type, cam, Xw [n, cap, 3],
    has_mp [n, cap], Tcw [n, 16], Tlw [n, 16], th, check_ori, bounds).  Returns a dict with the per-frame outputs and times."""
    L = lib()
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n, H, W = frames.shape
    offs = np.ascontiguousarray(offs, dtype=np.int32)
    if cap is None:
        cap = ex.nfeatures + 3 * ex.nlevels + 64
    cfg = BenchCfg()
    cfg.nframes, cfg.count, cfg.rows, cfg.cols, cfg.threads, cfg.warmup = n, int(count), H, W, int(threads), int(warmup)
    cfg.lap0, cfg.lap1, cfg.cap, cfg.mode, cfg.nnratio, cfg.th_high = int(lap[0]), int(lap[1]), int(cap), int(mode), float(nnratio), int(th_high)
    keep = []
    if distort is not None:   # (K[4], D[4 or 5]): mode 0 on undistorted keypoints with the undistorted-corner bounds (Frame.cc:837-899)
        Kd, Dd = distort
        cfg.distort, cfg.nD = 1, len(Dd)
        for i in range(4):
            cfg.K[i] = float(Kd[i])
        for i in range(len(Dd)):
            cfg.D[i] = float(Dd[i])
    if mode == 1:
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        cam, Xw, has, Tcw, Tlw = a(scene["cam"], np.float32), a(scene["Xw"], np.float32), a(scene["has_mp"], np.uint8), a(scene["Tcw"], np.float32), a(scene["Tlw"], np.float32)
        assert Xw.shape == (n, cap, 3) and has.shape == (n, cap) and Tcw.shape == (n, 16) and Tlw.shape == (n, 16)
        keep = [cam, Xw, has, Tcw, Tlw]
        cfg.cam_type, cfg.cam, cfg.Xw, cfg.has_mp, cfg.Tcw, cfg.Tlw = int(scene["cam_type"]), cam.ctypes.data, Xw.ctypes.data, has.ctypes.data, Tcw.ctypes.data, Tlw.ctypes.data
        cfg.th, cfg.check_ori = float(scene["th"]), int(scene["check_ori"])
        for i, b in enumerate(scene["bounds"]):
            cfg.bounds[i] = float(b)
    kps = np.zeros((n, cap), dtype=KP_DTYPE)
    desc = np.zeros((n, cap, 32), dtype=np.uint8)
    counts = np.zeros((n, 2), dtype=np.int32)
    moq = np.full((n, cap), -1, dtype=np.int32)
    nmatch = np.zeros(n, dtype=np.int32)
    ms_e, ms_m, wall = np.zeros(n), np.zeros(n), np.zeros(2)
    L.orc_bench_stream.argtypes = [C.c_void_p] * 12
    rc = L.orc_bench_stream(C.byref(ex.e), C.byref(cfg), _p(frames), _p(offs), _p(kps), _p(desc), _p(counts), _p(moq), _p(nmatch), _p(ms_e), _p(ms_m), _p(wall))
    if rc != 0:
        raise RuntimeError("orc_bench_stream rc=%d" % rc)
    del keep
    c = int(count)
    return dict(kps=kps, desc=desc, counts=counts, moq=moq, nmatch=nmatch, ms_extract=ms_e[:c], ms_match=ms_m[:c], wall_extract=float(wall[0]),
                wall_match=float(wall[1]), count=c, threads=int(threads), cap=int(cap))
