"""orb-frontend-mi355x: ORB-SLAM3's per-frame feature front end on MI355X.

Host-side mirror (Python, ctypes) of the C ABI in include/orbhip.h, keeping the reference's names:
`ORBextractor` (include/ORBextractor.h:43-109), `ORBmatcher` (include/ORBmatcher.h:35-108).  The C++ drop-in
classes with the reference's exact signatures live in csrc/adapter/.  This module is what tests/ and bench.py
drive; it contains no compute of its own and no CPU fallback: if liborbhip.so is missing or no HIP device is
usable, construction raises.

The package name starts with a digit (it is fixed by the project layout), so import it with
    importlib.import_module("3_orb_slam3_selfnote_amd")
"""
import ctypes as C
import os
import time

import numpy as np

from . import build as _build

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORBHIP_LIB", os.path.join(HERE, "liborbhip.so"))  # ORBHIP_LIB: diagnostic builds only

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

E_EMPTY, E_ARG, E_HIP, E_CAP = -1, -2, -3, -4

# every symbol include/orbhip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "orbx_create", "orbx_destroy", "orbx_last_error", "orbx_get_levels", "orbx_get_scale_factor",
    "orbx_get_scale_tables", "orbx_get_features_per_level", "orbx_configure", "orbx_max_keypoints", "orbx_extract",
    "orbx_extract_batch_device", "orbx_get_host_us", "orbx_level_info", "orbx_download_level", "orbx_download_pyramid", "orbx_download_blurred_level",
    "orbx_download_candidates", "orbx_download_level_keypoints", "orbx_set_profiling", "orbx_get_stage_ms",
    "orbx_ref_cosf", "orbx_ref_sinf", "orbx_ref_atanf", "orbx_ref_atan2f", "orbx_compute_stereo_matches", "orbx_cvt_color_gray", "orbx_cvt_color_gray_device",
    "orbx_clahe", "orbx_clahe_device", "orbx_remap_linear", "orbx_remap_linear_device",
    "orbm_create", "orbm_destroy", "orbm_last_error", "orbm_descriptor_distance", "orbm_search_by_projection",
    "orbm_search_by_projection_batch_device", "orbm_search_by_projection_fisheye", "orbm_search_by_projection_last_frame_fisheye", "orbm_search_by_projection_last_frame", "orbm_search_by_projection_last_frame_batch_device", "orbm_search_by_projection_keyframe", "orbm_search_by_projection_sim3", "orbm_search_by_projection_sim3_cam", "orbm_fuse_sim3_cam", "orbm_search_for_triangulation", "orbm_triangulation_candidates", "orbm_search_for_triangulation_pred", "orbm_search_for_initialization", "orbm_search_by_bow", "orbm_search_by_bow_fisheye", "orbm_search_by_bow_keyframes", "orbm_fuse", "orbm_fuse_sim3", "orbm_search_by_sim3", "orbm_distinctive_descriptors", "orbm_knn_match2", "orbm_hamming_matrix", "orbm_three_maxima",
    "orbm_radius_by_viewing_cos", "orbm_project", "orbm_undistort_keypoints", "orbm_image_bounds", "orbm_undistort_keypoints_batch_device", "orbm_set_profiling", "orbm_set_scan_mode", "orbm_set_hamming_engine", "orbm_get_last_ms", "orbm_get_stage_ms",
]


class FrameStruct(C.Structure):  # orbm_frame_t
    _fields_ = [("n", C.c_int32), ("keys_un", C.c_void_p), ("descriptors", C.c_void_p), ("u_right", C.c_void_p),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float)]


class KeyFrameStruct(C.Structure):  # orbm_keyframe_t
    _fields_ = [("n", C.c_int32), ("keys_un", C.c_void_p), ("descriptors", C.c_void_p), ("u_right", C.c_void_p),
                ("has_mappoint", C.c_void_p), ("n_nodes", C.c_int32), ("node_id", C.c_void_p), ("node_start", C.c_void_p),
                ("node_idx", C.c_void_p), ("scale_factors", C.c_void_p), ("level_sigma2", C.c_void_p), ("nlevels", C.c_int32)]


class LastFrameStruct(C.Structure):  # orbm_last_frame_t
    _fields_ = [("n", C.c_int32), ("has_mp", C.c_void_p), ("Xw", C.c_void_p), ("mpdesc", C.c_void_p), ("last_keys", C.c_void_p),
                ("obs", C.c_void_p), ("Tcw", C.c_void_p), ("Tlw", C.c_void_p)]


class QueryStruct(C.Structure):  # orbm_queries_t
    _fields_ = [("nq", C.c_int32), ("descriptors", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p),
                ("radius", C.c_void_p), ("min_level", C.c_void_p), ("max_level", C.c_void_p), ("u_r", C.c_void_p),
                ("flags", C.c_void_p)]


PAIR_PRED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)   # orbm_pair_predicate_t

_lib = None


def load(build_if_needed=True):
    """dlopen liborbhip.so (building it first when the sources are newer).  Raises if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  Two HIP
    # runtimes in one process cannot both own the device, so when torch is installed it is imported FIRST: the
    # loader then resolves liborbhip.so's NEEDED libamdhip64.so.7 to the copy torch already mapped, and torch
    # tensors, streams and liborbhip kernels share one runtime.  Without torch the system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if build_if_needed and os.path.exists("/opt/rocm/bin/hipcc") and "ORBHIP_LIB" not in os.environ:
        _build.build()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("liborbhip.so is not built (run __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.orbx_create.restype = vp
    L.orbx_create.argtypes = [i32, f32, i32, i32, i32, i32]
    L.orbx_destroy.argtypes = [vp]
    L.orbx_last_error.restype = C.c_char_p
    L.orbx_last_error.argtypes = [vp]
    L.orbx_get_levels.argtypes = [vp]
    L.orbx_get_scale_factor.restype = f32
    L.orbx_get_scale_factor.argtypes = [vp]
    L.orbx_get_scale_tables.argtypes = [vp, vp, vp, vp, vp]
    L.orbx_get_features_per_level.argtypes = [vp, vp]
    L.orbx_configure.argtypes = [vp, i32, i32, i32]
    L.orbx_max_keypoints.argtypes = [vp]
    L.orbx_extract.argtypes = [vp, vp, i32, i32, sz, i32, i32, vp, vp, i32, vp]
    L.orbx_extract_batch_device.argtypes = [vp, vp, i32, i32, sz, sz, i32, i32, i32, vp, vp, vp, i32, vp]
    L.orbx_get_host_us.argtypes = [vp, vp, i32]
    L.orbx_level_info.argtypes = [vp, i32, vp, vp]
    L.orbx_download_level.argtypes = [vp, i32, i32, i32, vp, sz]
    L.orbx_download_pyramid.argtypes = [vp, i32, i32, vp, sz, vp, vp]
    L.orbx_download_blurred_level.argtypes = [vp, i32, i32, vp, sz]
    L.orbx_download_candidates.argtypes = [vp, i32, i32, vp, i32]
    L.orbx_download_level_keypoints.argtypes = [vp, i32, i32, vp, i32]
    L.orbx_set_profiling.argtypes = [vp, i32]
    L.orbx_get_stage_ms.argtypes = [vp, vp, i32]
    L.orbx_compute_stereo_matches.argtypes = [vp, i32, vp, i32, i32, vp, vp, i32, vp, vp, f32, f32, vp, vp]
    L.orbx_cvt_color_gray.argtypes = [vp, vp, i32, i32, sz, i32, i32, vp, sz]
    L.orbx_cvt_color_gray_device.argtypes = [vp, i32, i32, sz, i32, i32, vp, sz, vp]
    L.orbx_clahe.argtypes = [vp, vp, i32, i32, sz, C.c_double, i32, i32, vp, sz]
    L.orbx_clahe_device.argtypes = [vp, i32, i32, sz, C.c_double, i32, i32, vp, vp, sz, vp]
    L.orbx_remap_linear.argtypes = [vp, vp, i32, i32, sz, vp, vp, i32, i32, vp, sz]
    L.orbx_remap_linear_device.argtypes = [vp, i32, i32, sz, vp, vp, sz, i32, i32, vp, sz, vp]
    L.orbx_ref_cosf.restype = f32
    L.orbx_ref_cosf.argtypes = [f32]
    L.orbx_ref_sinf.restype = f32
    L.orbx_ref_sinf.argtypes = [f32]
    L.orbx_ref_atanf.restype = f32
    L.orbx_ref_atanf.argtypes = [f32]
    L.orbx_ref_atan2f.restype = f32
    L.orbx_ref_atan2f.argtypes = [f32, f32]
    L.orbm_create.restype = vp
    L.orbm_create.argtypes = [i32]
    L.orbm_destroy.argtypes = [vp]
    L.orbm_last_error.restype = C.c_char_p
    L.orbm_last_error.argtypes = [vp]
    L.orbm_descriptor_distance.argtypes = [vp, vp]
    L.orbm_search_by_projection.argtypes = [vp, vp, vp, f32, i32, i32, vp, vp, vp, vp]
    L.orbm_search_by_projection_batch_device.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, i32, i32, f32, i32, i32,
                                                         vp, vp, vp, vp, vp, vp]
    L.orbm_search_by_projection_fisheye.argtypes = [vp, vp, i32, vp, vp, vp, f32, i32, vp, vp, vp, vp]
    L.orbm_search_by_projection_last_frame_fisheye.argtypes = [vp, vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, f32, f32,
                                                               i32, i32, vp, vp]
    L.orbm_search_for_initialization.argtypes = [vp, vp, vp, vp, i32, f32, i32, vp]
    L.orbm_fuse.argtypes = [vp, vp, vp, vp, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, f32, f32, vp, vp]
    L.orbm_fuse_sim3.argtypes = [vp, vp, vp, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp]
    L.orbm_fuse_sim3_cam.argtypes = [vp, vp, vp, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, f32, vp, vp]
    L.orbm_search_by_projection_sim3_cam.argtypes = [vp, vp, vp, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, f32, vp, vp]
    L.orbm_search_by_sim3.argtypes = [vp] + [vp, vp, i32, f32, vp, vp, vp, vp, vp, vp, vp] * 2 + [f32, vp, vp, vp, f32, vp]
    L.orbm_distinctive_descriptors.argtypes = [vp, i32, vp, vp, vp]
    L.orbm_knn_match2.argtypes = [vp, vp, i32, vp, i32, vp, vp]
    L.orbm_search_by_bow.argtypes = [vp, vp, vp, f32, i32, vp]
    L.orbm_search_by_bow_fisheye.argtypes = [vp, vp, vp, i32, f32, i32, vp]
    L.orbm_search_by_bow_keyframes.argtypes = [vp, vp, vp, f32, i32, vp]
    L.orbm_hamming_matrix.argtypes = [vp, vp, i32, vp, i32, vp]
    L.orbm_search_for_triangulation.argtypes = [vp] * 10 + [i32, i32, i32, vp]
    L.orbm_triangulation_candidates.argtypes = [vp, vp, vp, f32, f32, i32, i32, vp, vp, vp, i32]
    L.orbm_search_for_triangulation_pred.argtypes = [vp, vp, vp, f32, f32, i32, i32, i32, i32, PAIR_PRED, vp, vp]
    L.orbm_search_by_projection_sim3.argtypes = [vp, vp, vp, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, vp, vp]
    L.orbm_search_by_projection_keyframe.argtypes = [vp, vp, vp, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, f32, i32, i32, vp, vp]
    L.orbm_search_by_projection_last_frame.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, f32, f32, f32,
                                                       i32, i32, vp, vp]
    L.orbm_search_by_projection_last_frame_batch_device.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, i32, i32, vp, i32, i32, vp, f32, f32, f32,
                                                                    i32, i32, vp, vp, vp, vp, vp]
    L.orbm_three_maxima.argtypes = [vp, i32, vp, vp, vp]
    L.orbm_radius_by_viewing_cos.restype = f32
    L.orbm_radius_by_viewing_cos.argtypes = [f32]
    L.orbm_project.argtypes = [i32, vp, f32, f32, f32, vp, vp]
    L.orbm_undistort_keypoints.argtypes = [i32, vp, vp, vp, i32, vp]
    L.orbm_image_bounds.argtypes = [i32, i32, vp, vp, i32, vp, vp, vp, vp]
    L.orbm_undistort_keypoints_batch_device.argtypes = [vp, vp, i32, vp, i32, i32, i32, vp, vp, i32, vp, vp]
    L.orbm_set_profiling.argtypes = [vp, i32]
    L.orbm_set_scan_mode.argtypes = [vp, i32]
    L.orbm_set_scan_mode.restype = i32
    L.orbm_set_hamming_engine.argtypes = [vp, i32]
    L.orbm_set_hamming_engine.restype = i32
    L.orbm_get_last_ms.restype = f32
    L.orbm_get_last_ms.argtypes = [vp]
    L.orbm_get_stage_ms.argtypes = [vp, vp, i32]
    _lib = L
    return L


def _p(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


class OrbError(RuntimeError):
    pass


class ORBextractor:
    """ORB_SLAM3::ORBextractor (ORBextractor.h:43-109) on one MI355X.

    __call__(image, mask, vLappingArea) mirrors operator() (ORBextractor.cc:1071-1184) and returns
    (monoIndex, keypoints[KP_DTYPE], descriptors[n,32])."""

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device=0):
        self.L = load()
        self.h = self.L.orbx_create(int(nfeatures), C.c_float(scaleFactor), int(nlevels), int(iniThFAST), int(minThFAST), int(device))
        if not self.h:
            raise OrbError("orbx_create failed: no usable HIP device %d or bad parameters (no CPU fallback)" % device)
        self.nlevels = int(nlevels)
        self.nfeatures = int(nfeatures)
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.L.orbx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc == E_HIP:
            raise OrbError("%s: %s" % (what, self.L.orbx_last_error(self.h).decode()))
        if rc == E_ARG:
            raise ValueError("%s: bad argument (%s)" % (what, self.L.orbx_last_error(self.h).decode()))
        return rc

    # --- getters, ORBextractor.h:61-81 ---
    def GetLevels(self):
        return self.L.orbx_get_levels(self.h)

    def GetScaleFactor(self):
        return float(self.L.orbx_get_scale_factor(self.h))

    def _tables(self):
        t = [np.zeros(self.nlevels, dtype=np.float32) for _ in range(4)]
        self.L.orbx_get_scale_tables(self.h, *[_p(a) for a in t])
        return t

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        a = np.zeros(self.nlevels, dtype=np.int32)
        self.L.orbx_get_features_per_level(self.h, _p(a))
        return a.tolist()

    def configure(self, rows, cols, max_batch=1):
        return self._check(self.L.orbx_configure(self.h, int(rows), int(cols), int(max_batch)), "orbx_configure")

    def max_keypoints(self):
        return self.L.orbx_max_keypoints(self.h)

    def __call__(self, image, mask=None, vLappingArea=(0, 1000)):
        if image is None or image.size == 0:
            return -1, np.zeros(0, dtype=KP_DTYPE), np.zeros((0, 32), dtype=np.uint8)  # ORBextractor.cc:1075-1076
        if image.dtype != np.uint8 or image.ndim != 2:
            raise ValueError("image must be CV_8UC1 (ORBextractor.cc:1080)")
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        rows, cols = image.shape
        cap = self.configure(rows, cols, 1)
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = C.c_int(0)
        # the C call alone (what a C++ caller of the adapter pays), for tools/run_sequence.py.  The argument objects are made BEFORE the
        # clock starts: an allocation inside the timed region can start CPython's full garbage collection, which takes ~40 ms once a
        # large package such as PyTorch is imported - that was the "36 ms stall" of round 2's sequence runs (tools/stall_probe.py)
        args = (self.h, _p(image), rows, cols, C.c_size_t(image.strides[0]), int(vLappingArea[0]), int(vLappingArea[1]), _p(kps), _p(desc), cap, C.byref(n))
        t0 = time.perf_counter()
        rc = self.L.orbx_extract(*args)
        self.last_call_s = time.perf_counter() - t0
        if rc == E_EMPTY:
            return -1, kps[:0], desc[:0]
        if rc == E_CAP:
            raise OrbError("keypoint capacity bound violated: %d > %d" % (n.value, cap))
        self._check(rc, "orbx_extract")
        return rc, kps[:n.value].copy(), desc[:n.value].copy()

    def cvtColorGray(self, im, rgb=True):
        """cvtColor(im, gray, CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY) of Tracking::GrabImage* (Tracking.cc:1122-1135)."""
        im = np.ascontiguousarray(im, dtype=np.uint8)
        if im.ndim != 3 or im.shape[2] not in (3, 4):
            raise ValueError("expected an H x W x 3|4 uint8 image")
        H, W, ch = im.shape
        out = np.empty((H, W), np.uint8)
        rc = self.L.orbx_cvt_color_gray(self.h, _p(im), H, W, C.c_size_t(W * ch), ch, int(bool(rgb)), _p(out), C.c_size_t(W))
        if rc < 0:
            raise OrbError("orbx_cvt_color_gray rc=%d: %s" % (rc, self.L.orbx_last_error(self.h).decode()))
        return out

    def CLAHE(self, im, clip_limit=3.0, tiles=(8, 8)):
        """cv::createCLAHE(clip_limit, Size(tiles))->apply(im, im) of the TUM-VI examples (mono_tum_vi.cc:101-109)."""
        im = np.ascontiguousarray(im, dtype=np.uint8)
        if im.ndim != 2:
            raise ValueError("expected an H x W uint8 image")
        H, W = im.shape
        out = np.empty((H, W), np.uint8)
        rc = self.L.orbx_clahe(self.h, _p(im), H, W, C.c_size_t(W), C.c_double(clip_limit), int(tiles[0]), int(tiles[1]), _p(out), C.c_size_t(W))
        if rc < 0:
            raise OrbError("orbx_clahe rc=%d: %s" % (rc, self.L.orbx_last_error(self.h).decode()))
        return out

    def remap(self, im, mapx=None, mapy=None, size=None):
        """cv::remap(im, out, M1, M2, INTER_LINEAR) of the stereo examples (stereo_euroc.cc:166-167).  Maps are kept on the device:
        leave them out (and give size=(rows, cols)) to reuse the ones of the previous call."""
        im = np.ascontiguousarray(im, dtype=np.uint8)
        if mapx is not None:
            mapx = np.ascontiguousarray(mapx, dtype=np.float32); mapy = np.ascontiguousarray(mapy, dtype=np.float32)
            if mapx.shape != mapy.shape or mapx.ndim != 2:
                raise ValueError("mapx / mapy must be equal-shaped 2-D float32 arrays")
            size = mapx.shape
        H, W = size
        out = np.empty((H, W), np.uint8)
        rc = self.L.orbx_remap_linear(self.h, _p(im), im.shape[0], im.shape[1], C.c_size_t(im.shape[1]), _p(mapx) if mapx is not None else None,
                                      _p(mapy) if mapy is not None else None, H, W, _p(out), C.c_size_t(W))
        if rc < 0:
            raise OrbError("orbx_remap_linear rc=%d: %s" % (rc, self.L.orbx_last_error(self.h).decode()))
        return out

    def ComputeStereoMatches(self, right, keysL, descL, keysR, descR, mb, mbf, frame_l=0, frame_r=0):
        """Frame::ComputeStereoMatches (Frame.cc:901-1079).  self / right = mpORBextractorLeft / Right after extracting the two
        images (their pyramids are still on the device).  Returns (mvuRight, mvDepth)."""
        keysL, keysR = np.ascontiguousarray(keysL, dtype=KP_DTYPE), np.ascontiguousarray(keysR, dtype=KP_DTYPE)
        descL, descR = np.ascontiguousarray(descL, dtype=np.uint8), np.ascontiguousarray(descR, dtype=np.uint8)
        uR = np.full(len(keysL), -1, dtype=np.float32)
        depth = np.full(len(keysL), -1, dtype=np.float32)
        rc = self.L.orbx_compute_stereo_matches(self.h, int(frame_l), right.h, int(frame_r), len(keysL), _p(keysL), _p(descL), len(keysR),
                                                _p(keysR), _p(descR), C.c_float(mb), C.c_float(mbf), _p(uR), _p(depth))
        if rc < 0:
            raise OrbError("orbx_compute_stereo_matches rc=%d: %s" % (rc, self.L.orbx_last_error(self.h).decode()))
        return uR, depth

    def extract_batch_device(self, d_images, rows, cols, stride, frame_stride, nframes, d_kps, d_desc, d_counts, cap,
                             vLappingArea=(0, 1000), stream=None):
        """All pointers are device addresses (ints).  Asynchronous on `stream` (hipStream_t as int; None/0 = the
        device's default stream, which is also torch's default current stream)."""
        rc = self.L.orbx_extract_batch_device(self.h, C.c_void_p(d_images), int(rows), int(cols), C.c_size_t(stride),
                                              C.c_size_t(frame_stride), int(nframes), int(vLappingArea[0]),
                                              int(vLappingArea[1]), C.c_void_p(d_kps), C.c_void_p(d_desc),
                                              C.c_void_p(d_counts), int(cap), C.c_void_p(stream) if stream else None)
        self._check(rc, "orbx_extract_batch_device")
        if rc < 0:
            raise OrbError("orbx_extract_batch_device rc=%d" % rc)
        return rc

    # --- mvImagePyramid (ORBextractor.h:83) and stage taps ---
    def level_shape(self, level):
        r, c = C.c_int(), C.c_int()
        self._check(self.L.orbx_level_info(self.h, level, C.byref(r), C.byref(c)), "orbx_level_info")
        return r.value, c.value

    def image_pyramid_level(self, level, frame=0, border=0):
        r, c = self.level_shape(level)
        out = np.zeros((r + 2 * border, c + 2 * border), dtype=np.uint8)
        self._check(self.L.orbx_download_level(self.h, frame, level, border, _p(out), C.c_size_t(out.strides[0])), "orbx_download_level")
        return out

    def image_pyramid(self, frame=0, border=0):
        """mvImagePyramid of one frame in one transfer (orbx_download_pyramid): list of per-level arrays incl. the border frame."""
        off = np.zeros(self.nlevels, np.uint64); st = np.zeros(self.nlevels, np.uint64)
        nbytes = self._check(self.L.orbx_download_pyramid(self.h, int(frame), int(border), None, 0, _p(off), _p(st)), "orbx_download_pyramid")
        buf = np.zeros(max(nbytes, 1), np.uint8)
        self._check(self.L.orbx_download_pyramid(self.h, int(frame), int(border), _p(buf), C.c_size_t(buf.size), _p(off), _p(st)), "orbx_download_pyramid")
        out = []
        for l in range(self.nlevels):
            r, c = self.level_shape(l)
            rows, stride = r + 2 * border, int(st[l])
            out.append(buf[int(off[l]):int(off[l]) + rows * stride].reshape(rows, stride)[:, :c + 2 * border])
        return out

    def blurred_level(self, level, frame=0):
        r, c = self.level_shape(level)
        out = np.zeros((r, c), dtype=np.uint8)
        self._check(self.L.orbx_download_blurred_level(self.h, frame, level, _p(out), C.c_size_t(out.strides[0])), "orbx_download_blurred_level")
        return out

    def level_candidates(self, level, frame=0):
        r, c = self.level_shape(level)
        cap = (r * c) // 4 + 64
        out = np.zeros((cap, 3), dtype=np.float32)
        n = self._check(self.L.orbx_download_candidates(self.h, frame, level, _p(out), cap), "orbx_download_candidates")
        return out[:n].copy()

    def level_keypoints(self, level, frame=0):
        cap = self.max_keypoints() + 8
        out = np.zeros((cap, 3), dtype=np.float32)
        n = self._check(self.L.orbx_download_level_keypoints(self.h, frame, level, _p(out), cap), "orbx_download_level_keypoints")
        return out[:n].copy()

    def set_profiling(self, on=True):
        self.L.orbx_set_profiling(self.h, 1 if on else 0)

    def stage_ms(self):
        ms = np.zeros(5, dtype=np.float32)
        n = self.L.orbx_get_stage_ms(self.h, _p(ms), 5)
        return dict(zip(["pyramid", "fast", "octree", "blur", "describe"], ms[:n].tolist()))


class FrameView:
    """The slice of ORB_SLAM3::Frame the projection searches read (Frame.h mvKeysUn, mDescriptors, mvuRight,
    mnMinX..mnMaxY; Frame.cc:872-899), plus mvpMapPoints as (slot, slot_obs) arrays."""

    def __init__(self, keys_un, descriptors, bounds, u_right=None):
        self.keys_un = np.ascontiguousarray(keys_un, dtype=KP_DTYPE)
        self.descriptors = np.ascontiguousarray(descriptors, dtype=np.uint8)
        self.u_right = None if u_right is None else np.ascontiguousarray(u_right, dtype=np.float32)
        self.bounds = tuple(float(b) for b in bounds)  # mnMinX, mnMaxX, mnMinY, mnMaxY
        self.N = len(self.keys_un)
        self.slot = np.full(self.N, -1, dtype=np.int32)     # mvpMapPoints[i] as query id, -1 = NULL
        self.slot_obs = np.zeros(self.N, dtype=np.uint8)    # Observations()>0 of the holder

    def struct(self):
        return FrameStruct(self.N, _p(self.keys_un), _p(self.descriptors), _p(self.u_right), C.c_float(self.bounds[0]),
                           C.c_float(self.bounds[1]), C.c_float(self.bounds[2]), C.c_float(self.bounds[3]))


class KeyFrameView:
    """The slice of ORB_SLAM3::KeyFrame that SearchForTriangulation reads: mvKeysUn, mDescriptors, mvuRight, the map-point
    occupancy, mFeatVec (as {node id: [indices]}), mvScaleFactors, mvLevelSigma2."""

    def __init__(self, keys_un, descriptors, feat_vec, scale_factors, level_sigma2, u_right=None, has_mappoint=None):
        self.keys_un = np.ascontiguousarray(keys_un, dtype=KP_DTYPE)
        self.descriptors = np.ascontiguousarray(descriptors, dtype=np.uint8)
        self.N = len(self.keys_un)
        self.u_right = np.full(self.N, -1, np.float32) if u_right is None else np.ascontiguousarray(u_right, dtype=np.float32)
        self.has_mappoint = np.zeros(self.N, np.uint8) if has_mappoint is None else np.ascontiguousarray(has_mappoint, dtype=np.uint8)
        ids = sorted(feat_vec.keys())                      # std::map iteration order
        self.node_id = np.array(ids, dtype=np.uint32)
        self.node_start = np.zeros(len(ids) + 1, dtype=np.int32)
        idx = []
        for k, nid in enumerate(ids):
            idx.extend(int(i) for i in feat_vec[nid])
            self.node_start[k + 1] = len(idx)
        self.node_idx = np.array(idx, dtype=np.int32) if idx else np.zeros(1, np.int32)
        self.sf = np.ascontiguousarray(scale_factors, dtype=np.float32)
        self.sigma2 = np.ascontiguousarray(level_sigma2, dtype=np.float32)

    def struct(self):
        return KeyFrameStruct(self.N, _p(self.keys_un), _p(self.descriptors), _p(self.u_right), _p(self.has_mappoint), len(self.node_id),
                              _p(self.node_id), _p(self.node_start), _p(self.node_idx), _p(self.sf), _p(self.sigma2), len(self.sf))


class ORBmatcher:
    """ORB_SLAM3::ORBmatcher (ORBmatcher.h:35-108) -- the projection-search members, on one MI355X."""
    TH_LOW = 50
    TH_HIGH = 100
    HISTO_LENGTH = 30

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self.L = load()
        self.m = self.L.orbm_create(int(device))
        if not self.m:
            raise OrbError("orbm_create failed: no usable HIP device %d (no CPU fallback)" % device)
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)

    def close(self):
        if getattr(self, "m", None):
            self.L.orbm_destroy(self.m)
            self.m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc == E_HIP:
            raise OrbError("%s: %s" % (what, self.L.orbm_last_error(self.m).decode()))
        if rc == E_ARG:
            raise ValueError("%s: bad argument (%s)" % (what, self.L.orbm_last_error(self.m).decode()))
        return rc

    @staticmethod
    def DescriptorDistance(a, b):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        b = np.ascontiguousarray(b, dtype=np.uint8)
        return load().orbm_descriptor_distance(_p(a), _p(b))

    @staticmethod
    def RadiusByViewingCos(viewCos):
        return float(load().orbm_radius_by_viewing_cos(C.c_float(viewCos)))

    @staticmethod
    def ComputeThreeMaxima(histo_sizes):
        h = np.ascontiguousarray(histo_sizes, dtype=np.int32)
        i1, i2, i3 = C.c_int(), C.c_int(), C.c_int()
        load().orbm_three_maxima(_p(h), len(h), C.byref(i1), C.byref(i2), C.byref(i3))
        return i1.value, i2.value, i3.value

    def search_window(self, frame, qdesc, u, v, radius, min_level, max_level, flags=None, u_r=None, nnratio=None,
                      th_dist=None, use_second=True):
        """orbm_search_by_projection: the core shared by the SearchByProjection overloads."""
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        qdesc, u, v, radius = a(qdesc, np.uint8), a(u, np.float32), a(v, np.float32), a(radius, np.float32)
        min_level, max_level = a(min_level, np.int32), a(max_level, np.int32)
        nq = len(u)
        flags = None if flags is None else a(flags, np.uint8)
        u_r = None if u_r is None else a(u_r, np.float32)
        qs = QueryStruct(nq, _p(qdesc), _p(u), _p(v), _p(radius), _p(min_level), _p(max_level), _p(u_r), _p(flags))
        fs = frame.struct()
        moq = np.full(nq, -1, dtype=np.int32)
        bd = np.full(nq, 256, dtype=np.int32)
        args = (self.m, C.byref(fs), C.byref(qs), C.c_float(self.mfNNratio if nnratio is None else nnratio),
                int(self.TH_HIGH if th_dist is None else th_dist), int(bool(use_second)), _p(frame.slot), _p(frame.slot_obs), _p(moq), _p(bd))
        t0 = time.perf_counter()
        rc = self.L.orbm_search_by_projection(*args)
        self.last_call_s = time.perf_counter() - t0   # the C call alone (arguments made before the clock starts, see ORBextractor.__call__)
        self._check(rc, "orbm_search_by_projection")
        if rc < 0:
            raise OrbError("orbm_search_by_projection rc=%d" % rc)
        return rc, moq, bd

    def SearchByProjection(self, F, mp_in_view, mp_desc, mp_projX, mp_projY, mp_viewCos, mp_level, scale_factors, th=1.0,
                           mp_obs=None, mp_projXR=None):
        """SearchByProjection(Frame &F, const vector<MapPoint*>&, th, ...) -- ORBmatcher.cc:44-143 (mono /
        rectified-stereo half).  MapPoint fields arrive as arrays: mbTrackInView (&& !isBad && far-point test),
        GetDescriptor(), mTrackProjX/Y, mTrackViewCos, mnTrackScaleLevel, Observations()>0."""
        sf = np.asarray(scale_factors, dtype=np.float32)
        lvl = np.asarray(mp_level, dtype=np.int32)
        vc = np.asarray(mp_viewCos, dtype=np.float32)
        r = np.where(vc.astype(np.float64) > 0.998, np.float32(2.5), np.float32(4.0)).astype(np.float32)  # :216-222
        if float(np.float32(th)) != 1.0:
            r = (r * np.float32(th)).astype(np.float32)                                                    # :69-70
        radius = (r * sf[np.clip(lvl, 0, len(sf) - 1)]).astype(np.float32)                               # :73
        nq = len(lvl)
        obs = np.ones(nq, np.uint8) if mp_obs is None else np.asarray(mp_obs, dtype=np.uint8)
        flags = (np.asarray(mp_in_view, dtype=np.uint8) & 1) | ((obs & 1) << 1)
        return self.search_window(F, mp_desc, mp_projX, mp_projY, radius, lvl - 1, lvl, flags=flags, u_r=mp_projXR,
                                  nnratio=self.mfNNratio, th_dist=self.TH_HIGH, use_second=True)

    def SearchByProjectionLastFrame(self, CurrentFrame, scale_factors, has_mp, Xw, mp_desc, last_keys, Tcw, Tlw, cam_type,
                                    cam_params, th, bMono=True, mb=0.0, mbf=0.0, mp_obs=None):
        """SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) -- ORBmatcher.cc:2027-2289."""
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        sf = a(scale_factors, np.float32)
        has_mp, Xw, mp_desc = a(has_mp, np.uint8), a(Xw, np.float32), a(mp_desc, np.uint8)
        last_keys = a(last_keys, KP_DTYPE)
        Tcw, Tlw, cam_params = a(Tcw, np.float32), a(Tlw, np.float32), a(cam_params, np.float32)
        obs = None if mp_obs is None else a(mp_obs, np.uint8)
        fs = CurrentFrame.struct()
        rc = self.L.orbm_search_by_projection_last_frame(self.m, C.byref(fs), _p(sf), len(sf), len(has_mp), _p(has_mp), _p(Xw),
                                                         _p(mp_desc), _p(last_keys), _p(obs), _p(Tcw), _p(Tlw), int(cam_type),
                                                         _p(cam_params), C.c_float(mb), C.c_float(mbf), C.c_float(th), int(bool(bMono)),
                                                         int(self.mbCheckOrientation), _p(CurrentFrame.slot), _p(CurrentFrame.slot_obs))
        self._check(rc, "orbm_search_by_projection_last_frame")
        if rc < 0:
            raise OrbError("orbm_search_by_projection_last_frame rc=%d" % rc)
        return rc

    def SearchByProjectionFisheye(self, F, n_left, left_to_right, right_to_left, mp_desc, scale_factors, th,
                                  in_view, projX, projY, viewCos, level, in_view_r, projXR, projYR, viewCosR, levelR, mp_obs=None):
        """SearchByProjection(Frame &F, const vector<MapPoint*>&, th, ...) for a fisheye-stereo frame (Nleft != -1) --
        ORBmatcher.cc:44-214 complete.  F: FrameView over mvKeys ++ mvKeysRight / mDescriptors (N = Nleft + Nright).
        Returns (nmatches, match_left[nmp], match_right[nmp]) with right matches as indices into the right image."""
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        sf = a(scale_factors, np.float32)
        nmp = len(level)
        lvl, lvlR = a(level, np.int32), a(levelR, np.int32)

        def radius_of(vc, lv, with_th):
            r = np.where(a(vc, np.float32).astype(np.float64) > 0.998, np.float32(2.5), np.float32(4.0)).astype(np.float32)  # :216-222
            if with_th and float(np.float32(th)) != 1.0:
                r = (r * np.float32(th)).astype(np.float32)                                                                  # :69-70
            return (r * sf[np.clip(lv, 0, len(sf) - 1)]).astype(np.float32)

        obs = np.ones(nmp, np.uint8) if mp_obs is None else a(mp_obs, np.uint8)
        nq = 2 * nmp
        u, v, rad = np.zeros(nq, np.float32), np.zeros(nq, np.float32), np.zeros(nq, np.float32)
        minl, maxl, flags = np.zeros(nq, np.int32), np.zeros(nq, np.int32), np.zeros(nq, np.uint8)
        u[0::2], v[0::2], rad[0::2] = a(projX, np.float32), a(projY, np.float32), radius_of(viewCos, lvl, True)
        u[1::2], v[1::2], rad[1::2] = a(projXR, np.float32), a(projYR, np.float32), radius_of(viewCosR, lvlR, False)  # :148: no th
        minl[0::2], maxl[0::2] = lvl - 1, lvl
        minl[1::2], maxl[1::2] = lvlR - 1, lvlR
        flags[0::2] = (a(in_view, np.uint8) & 1) | ((obs & 1) << 1)
        flags[1::2] = ((a(in_view_r, np.uint8) & 1) & (lvlR != -1)) | ((obs & 1) << 1)                                 # :147
        qdesc = np.repeat(a(mp_desc, np.uint8).reshape(nmp, 32), 2, axis=0).copy()
        l2r = None if left_to_right is None else a(left_to_right, np.int32)
        r2l = None if right_to_left is None else a(right_to_left, np.int32)
        qs = QueryStruct(nq, _p(qdesc), _p(u), _p(v), _p(rad), _p(minl), _p(maxl), None, _p(flags))
        fs = F.struct()
        moq = np.full(nq, -1, dtype=np.int32)
        rc = self.L.orbm_search_by_projection_fisheye(self.m, C.byref(fs), int(n_left), _p(l2r), _p(r2l), C.byref(qs),
                                                      C.c_float(self.mfNNratio), int(self.TH_HIGH), _p(F.slot), _p(F.slot_obs), _p(moq), None)
        self._check(rc, "orbm_search_by_projection_fisheye")
        if rc < 0:
            raise OrbError("orbm_search_by_projection_fisheye rc=%d" % rc)
        mr = moq[1::2].copy()
        mr[mr >= 0] -= int(n_left)
        return rc, moq[0::2].copy(), mr

    def SearchByProjectionLastFrameFisheye(self, CurrentFrame, n_left, scale_factors, has_mp, Xw, mp_desc, last_keys, Tcw, Tlw, Trl,
                                           cam_type, cam_params, th, bMono=False, mb=0.0, mp_obs=None):
        """SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) for a fisheye-stereo current frame --
        ORBmatcher.cc:2027-2289 complete (incl. the right-camera pass :2189-2256)."""
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        sf = a(scale_factors, np.float32)
        has_mp, Xw, mp_desc = a(has_mp, np.uint8), a(Xw, np.float32), a(mp_desc, np.uint8)
        last_keys = a(last_keys, KP_DTYPE)
        Tcw, Tlw, Trl, cam_params = a(Tcw, np.float32), a(Tlw, np.float32), a(Trl, np.float32), a(cam_params, np.float32)
        obs = None if mp_obs is None else a(mp_obs, np.uint8)
        fs = CurrentFrame.struct()
        rc = self.L.orbm_search_by_projection_last_frame_fisheye(self.m, C.byref(fs), int(n_left), _p(sf), len(sf), len(has_mp), _p(has_mp),
                                                                 _p(Xw), _p(mp_desc), _p(last_keys), _p(obs), _p(Tcw), _p(Tlw), _p(Trl),
                                                                 int(cam_type), _p(cam_params), C.c_float(mb), C.c_float(th), int(bool(bMono)),
                                                                 int(self.mbCheckOrientation), _p(CurrentFrame.slot), _p(CurrentFrame.slot_obs))
        self._check(rc, "orbm_search_by_projection_last_frame_fisheye")
        if rc < 0:
            raise OrbError("orbm_search_by_projection_last_frame_fisheye rc=%d" % rc)
        return rc

    def SearchForInitialization(self, F1, F2, vbPrevMatched, windowSize=10):
        """SearchForInitialization(Frame &F1, Frame &F2, vbPrevMatched, vnMatches12, windowSize) -- ORBmatcher.cc:722-837.
        F1, F2: FrameView; vbPrevMatched: (N1, 2) float32, updated in place.  Returns (nmatches, vnMatches12)."""
        assert vbPrevMatched.dtype == np.float32 and vbPrevMatched.flags["C_CONTIGUOUS"] and vbPrevMatched.shape == (F1.N, 2)
        m12 = np.full(F1.N, -1, dtype=np.int32)
        f1, f2 = F1.struct(), F2.struct()
        rc = self.L.orbm_search_for_initialization(self.m, C.byref(f1), C.byref(f2), _p(vbPrevMatched), int(windowSize),
                                                   C.c_float(self.mfNNratio), int(self.mbCheckOrientation), _p(m12))
        self._check(rc, "orbm_search_for_initialization")
        if rc < 0:
            raise OrbError("orbm_search_for_initialization rc=%d" % rc)
        return rc, m12

    def SearchByBoW(self, KF, F, n_left=None):
        """SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches) -- ORBmatcher.cc:273-469 (Nleft == -1).
        KF, F: KeyFrameView (KF.has_mappoint = pMP && !isBad).  Returns (nmatches, matchF[F.N] = keyframe keypoint index or -1)."""
        m = np.full(max(F.N, 1), -1, dtype=np.int32)
        ks, fs = KF.struct(), F.struct()
        if n_left is not None:   # fisheye-stereo frame: F = mvKeys ++ mvKeysRight, Frame::Nleft = n_left
            rc = self.L.orbm_search_by_bow_fisheye(self.m, C.byref(ks), C.byref(fs), int(n_left), C.c_float(self.mfNNratio), int(self.mbCheckOrientation), _p(m))
        else:
            rc = self.L.orbm_search_by_bow(self.m, C.byref(ks), C.byref(fs), C.c_float(self.mfNNratio), int(self.mbCheckOrientation), _p(m))
        self._check(rc, "orbm_search_by_bow")
        if rc < 0:
            raise OrbError("orbm_search_by_bow rc=%d" % rc)
        return rc, m[:F.N]

    def SearchByBoWKeyFrames(self, KF1, KF2):
        """SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12) -- ORBmatcher.cc:839-979.
        Returns (nmatches, matches12[KF1.N] = keypoint index in KF2 or -1)."""
        m = np.full(max(KF1.N, 1), -1, dtype=np.int32)
        a, b = KF1.struct(), KF2.struct()
        rc = self.L.orbm_search_by_bow_keyframes(self.m, C.byref(a), C.byref(b), C.c_float(self.mfNNratio), int(self.mbCheckOrientation), _p(m))
        self._check(rc, "orbm_search_by_bow_keyframes")
        if rc < 0:
            raise OrbError("orbm_search_by_bow_keyframes rc=%d" % rc)
        return rc, m[:KF1.N]

    def Fuse(self, KF, scale_factors, inv_level_sigma2, log_scale_factor, valid, Xw, normal, mp_desc, max_dist, min_dist, Tcw, Ow,
             cam_type, cam_params, bf, th=3.0):
        """Search part of Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th) -- ORBmatcher.cc:1425-1658.
        KF: FrameView of the keyframe (u_right = mvuRight).  Returns (nFused, bestIdx[nP], bestDist[nP])."""
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        sf, is2 = a(scale_factors, np.float32), a(inv_level_sigma2, np.float32)
        valid, Xw, normal, mp_desc = a(valid, np.uint8), a(Xw, np.float32), a(normal, np.float32), a(mp_desc, np.uint8)
        max_dist, min_dist, Tcw, Ow, cam = a(max_dist, np.float32), a(min_dist, np.float32), a(Tcw, np.float32), a(Ow, np.float32), a(cam_params, np.float32)
        n = len(valid)
        bi, bd = np.full(max(n, 1), -1, np.int32), np.full(max(n, 1), 256, np.int32)
        fs = KF.struct()
        rc = self.L.orbm_fuse(self.m, C.byref(fs), _p(sf), _p(is2), len(sf), C.c_float(log_scale_factor), n, _p(valid), _p(Xw), _p(normal), _p(mp_desc),
                              _p(max_dist), _p(min_dist), _p(Tcw), _p(Ow), int(cam_type), _p(cam), C.c_float(bf), C.c_float(th), _p(bi), _p(bd))
        self._check(rc, "orbm_fuse")
        if rc < 0:
            raise OrbError("orbm_fuse rc=%d" % rc)
        return rc, bi[:n], bd[:n]

    def FuseSim3(self, KF, scale_factors, log_scale_factor, valid, Xw, normal, mp_desc, max_dist, min_dist, Scw, cam, th=4.0, cam_type=0):
        """Search part of Fuse(KeyFrame *pKF, cv::Mat Scw, vpPoints, th, vpReplacePoint) -- ORBmatcher.cc:1660-1786."""
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        sf = a(scale_factors, np.float32)
        valid, Xw, normal, mp_desc = a(valid, np.uint8), a(Xw, np.float32), a(normal, np.float32), a(mp_desc, np.uint8)
        max_dist, min_dist, Scw, cam = a(max_dist, np.float32), a(min_dist, np.float32), a(Scw, np.float32), a(cam, np.float32)
        n = len(valid)
        bi, bd = np.full(max(n, 1), -1, np.int32), np.full(max(n, 1), 256, np.int32)
        fs = KF.struct()
        rc = self.L.orbm_fuse_sim3_cam(self.m, C.byref(fs), _p(sf), len(sf), C.c_float(log_scale_factor), n, _p(valid), _p(Xw), _p(normal), _p(mp_desc),
                                       _p(max_dist), _p(min_dist), _p(Scw), int(cam_type), _p(cam), C.c_float(th), _p(bi), _p(bd))
        self._check(rc, "orbm_fuse_sim3")
        if rc < 0:
            raise OrbError("orbm_fuse_sim3 rc=%d" % rc)
        return rc, bi[:n], bd[:n]

    def SearchBySim3(self, KF1, side1, KF2, side2, s12, R12, t12, cam1, th):
        """SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12, s12, R12, t12, th) -- ORBmatcher.cc:1788-2012.
        KF1, KF2: FrameView of the keyframes; side_k = dict(sf, log_sf, valid, Xw, desc, max_dist, min_dist, Rw, tw).
        Returns (nFound, matches12[KF1.N])."""
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        keep = []

        def pack(KF, S):
            sf = a(S["sf"], np.float32)
            arrs = [a(S["valid"], np.uint8), a(S["Xw"], np.float32), a(S["desc"], np.uint8), a(S["max_dist"], np.float32), a(S["min_dist"], np.float32),
                    a(S["Rw"], np.float32), a(S["tw"], np.float32)]
            fs = KF.struct()
            keep.extend([sf, fs] + arrs)
            return [C.byref(fs), _p(sf), len(sf), C.c_float(S["log_sf"])] + [_p(x) for x in arrs]

        R12, t12, cam1 = a(R12, np.float32), a(t12, np.float32), a(cam1, np.float32)
        m12 = np.full(max(KF1.N, 1), -1, np.int32)
        rc = self.L.orbm_search_by_sim3(self.m, *pack(KF1, side1), *pack(KF2, side2), C.c_float(s12), _p(R12), _p(t12), _p(cam1), C.c_float(th), _p(m12))
        self._check(rc, "orbm_search_by_sim3")
        if rc < 0:
            raise OrbError("orbm_search_by_sim3 rc=%d" % rc)
        return rc, m12[:KF1.N]

    def ComputeDistinctiveDescriptors(self, groups):
        """MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:350-436) for many map points at once.
        groups: list of (N_p, 32) uint8 arrays (the descriptors observing map point p).  Returns BestIdx per map point."""
        start = np.zeros(len(groups) + 1, np.int32)
        for i, g in enumerate(groups):
            start[i + 1] = start[i] + len(g)
        desc = np.ascontiguousarray(np.concatenate([np.asarray(g, np.uint8).reshape(-1, 32) for g in groups]) if len(groups) else np.zeros((0, 32), np.uint8))
        best = np.full(max(len(groups), 1), -1, np.int32)
        rc = self.L.orbm_distinctive_descriptors(self.m, len(groups), _p(start), _p(desc), _p(best))
        self._check(rc, "orbm_distinctive_descriptors")
        if rc < 0:
            raise OrbError("orbm_distinctive_descriptors rc=%d" % rc)
        return best[:len(groups)]

    def knnMatch2(self, query, train):
        """cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, matches, 2) (Frame.cc:1246).  Returns (idx[nq, 2], dist[nq, 2])."""
        query, train = np.ascontiguousarray(query, dtype=np.uint8), np.ascontiguousarray(train, dtype=np.uint8)
        nq = len(query)
        idx, dist = np.full((max(nq, 1), 2), -1, np.int32), np.full((max(nq, 1), 2), -1, np.int32)
        rc = self.L.orbm_knn_match2(self.m, _p(query), nq, _p(train), len(train), _p(idx), _p(dist))
        self._check(rc, "orbm_knn_match2")
        if rc < 0:
            raise OrbError("orbm_knn_match2 rc=%d" % rc)
        return idx[:nq], dist[:nq]

    def SearchByProjectionKeyFrame(self, CurrentFrame, scale_factors, log_scale_factor, valid, Xw, mp_desc, kf_angle, max_dist,
                                   min_dist, Tcw, cam_type, cam_params, th, ORBdist):
        """SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*>&, th, ORBdist) -- ORBmatcher.cc:2291-2413."""
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        sf = a(scale_factors, np.float32)
        valid, Xw, mp_desc = a(valid, np.uint8), a(Xw, np.float32), a(mp_desc, np.uint8)
        kf_angle, max_dist, min_dist = a(kf_angle, np.float32), a(max_dist, np.float32), a(min_dist, np.float32)
        Tcw, cam_params = a(Tcw, np.float32), a(cam_params, np.float32)
        fs = CurrentFrame.struct()
        rc = self.L.orbm_search_by_projection_keyframe(self.m, C.byref(fs), _p(sf), len(sf), C.c_float(log_scale_factor), len(valid), _p(valid),
                                                       _p(Xw), _p(mp_desc), _p(kf_angle), _p(max_dist), _p(min_dist), _p(Tcw), int(cam_type),
                                                       _p(cam_params), C.c_float(th), int(ORBdist), int(self.mbCheckOrientation),
                                                       _p(CurrentFrame.slot), _p(CurrentFrame.slot_obs))
        self._check(rc, "orbm_search_by_projection_keyframe")
        if rc < 0:
            raise OrbError("orbm_search_by_projection_keyframe rc=%d" % rc)
        return rc

    def SearchByProjectionSim3(self, KF, scale_factors, log_scale_factor, valid, Xw, normal, mp_desc, max_dist, min_dist, Scw, cam, th,
                               ratioHamming=1.0, cam_type=0):
        """SearchByProjection(KeyFrame *pKF, cv::Mat Scw, vpPoints, vpMatched, th, ratioHamming) -- ORBmatcher.cc:489-720.
        KF is a FrameView of the keyframe's keypoints; KF.slot plays vpMatched."""
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        sf = a(scale_factors, np.float32)
        valid, Xw, normal, mp_desc = a(valid, np.uint8), a(Xw, np.float32), a(normal, np.float32), a(mp_desc, np.uint8)
        max_dist, min_dist, Scw, cam = a(max_dist, np.float32), a(min_dist, np.float32), a(Scw, np.float32), a(cam, np.float32)
        fs = KF.struct()
        rc = self.L.orbm_search_by_projection_sim3_cam(self.m, C.byref(fs), _p(sf), len(sf), C.c_float(log_scale_factor), len(valid), _p(valid),
                                                       _p(Xw), _p(normal), _p(mp_desc), _p(max_dist), _p(min_dist), _p(Scw), int(cam_type), _p(cam), int(th),
                                                       C.c_float(ratioHamming), _p(KF.slot), _p(KF.slot_obs))
        self._check(rc, "orbm_search_by_projection_sim3")
        if rc < 0:
            raise OrbError("orbm_search_by_projection_sim3 rc=%d" % rc)
        return rc

    def SearchForTriangulation(self, KF1, KF2, R1w, t1w, R2w, t2w, Cw1, cam1, cam2, bOnlyStereo=False, bCoarse=False):
        """SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse) -- ORBmatcher.cc:981-1222.
        Returns (nmatches, vMatchedPairs as an int array [k, 2])."""
        a = lambda x: np.ascontiguousarray(x, dtype=np.float32)
        R1w, t1w, R2w, t2w, Cw1, cam1, cam2 = a(R1w), a(t1w), a(R2w), a(t2w), a(Cw1), a(cam1), a(cam2)
        m12 = np.full(max(KF1.N, 1), -1, dtype=np.int32)
        s1, s2 = KF1.struct(), KF2.struct()
        rc = self.L.orbm_search_for_triangulation(self.m, C.byref(s1), C.byref(s2), _p(R1w), _p(t1w), _p(R2w), _p(t2w), _p(Cw1), _p(cam1),
                                                  _p(cam2), int(bool(bOnlyStereo)), int(bool(bCoarse)), int(self.mbCheckOrientation), _p(m12))
        self._check(rc, "orbm_search_for_triangulation")
        if rc < 0:
            raise OrbError("orbm_search_for_triangulation rc=%d" % rc)
        m12 = m12[:KF1.N]
        i1 = np.nonzero(m12 >= 0)[0]
        return rc, np.stack([i1, m12[i1]], axis=1).astype(np.int64)

    def TriangulationCandidates(self, KF1, KF2, ep, epipole_gate, bOnlyStereo=False):
        """orbm_triangulation_candidates: (start[KF1.N + 1], idx2[], dist[]) - per keypoint of KF1 the same-node keypoints of KF2 that
        pass every gate in front of the epipolar predicate (ORBmatcher.cc:1080-1113), ordered (dist ascending, node position descending)."""
        s1, s2 = KF1.struct(), KF2.struct()
        start = np.zeros(KF1.N + 1, np.int32)
        cap = 4096
        while True:
            idx2, dist = np.zeros(max(cap, 1), np.int32), np.zeros(max(cap, 1), np.int32)
            tot = self.L.orbm_triangulation_candidates(self.m, C.byref(s1), C.byref(s2), C.c_float(ep[0]), C.c_float(ep[1]), int(bool(epipole_gate)),
                                                       int(bool(bOnlyStereo)), _p(start), _p(idx2), _p(dist), cap)
            self._check(tot, "orbm_triangulation_candidates")
            if tot < 0:
                raise OrbError("orbm_triangulation_candidates rc=%d" % tot)
            if tot <= cap:
                return start, idx2[:tot].copy(), dist[:tot].copy()
            cap = tot

    def SearchForTriangulationPred(self, KF1, KF2, ep, epipole_gate, pred, bOnlyStereo=False, bCoarse=False):
        """SearchForTriangulation for camera models whose epipolarConstrain is not Pinhole's (KannalaBrandt8, rigs): pred(idx1, idx2) is
        the caller's predicate (ORBmatcher.cc:1148).  Returns (nmatches, vMatchedPairs [k, 2])."""
        s1, s2 = KF1.struct(), KF2.struct()
        m12 = np.full(max(KF1.N, 1), -1, dtype=np.int32)
        cb = PAIR_PRED(lambda user, i1, i2: 1 if pred(i1, i2) else 0)
        rc = self.L.orbm_search_for_triangulation_pred(self.m, C.byref(s1), C.byref(s2), C.c_float(ep[0]), C.c_float(ep[1]), int(bool(epipole_gate)),
                                                       int(bool(bOnlyStereo)), int(bool(bCoarse)), int(self.mbCheckOrientation), cb, None, _p(m12))
        self._check(rc, "orbm_search_for_triangulation_pred")
        if rc < 0:
            raise OrbError("orbm_search_for_triangulation_pred rc=%d" % rc)
        m12 = m12[:KF1.N]
        i1 = np.nonzero(m12 >= 0)[0]
        return rc, np.stack([i1, m12[i1]], axis=1).astype(np.int64)

    def hamming_matrix(self, q, c):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        c = np.ascontiguousarray(c, dtype=np.uint8)
        d = np.zeros((len(q), len(c)), dtype=np.uint16)
        self._check(self.L.orbm_hamming_matrix(self.m, _p(q), len(q), _p(c), len(c), _p(d)), "orbm_hamming_matrix")
        return d

    def undistort_batch_device(self, d_keys, key_stride, d_counts, count_stride, nframes, K, D, d_keys_un, stream=None, n_const=0):
        """Frame::UndistortKeyPoints (Frame.cc:837-870) on keypoints resident in HBM; pointers are device addresses (ints)."""
        K, D = np.ascontiguousarray(K, dtype=np.float32), np.ascontiguousarray(D, dtype=np.float32)
        rc = self.L.orbm_undistort_keypoints_batch_device(self.m, C.c_void_p(d_keys), int(key_stride), C.c_void_p(d_counts) if d_counts else None,
                                                          int(count_stride), int(n_const), int(nframes), _p(K), _p(D), len(D), C.c_void_p(d_keys_un),
                                                          C.c_void_p(stream) if stream else None)
        self._check(rc, "orbm_undistort_keypoints_batch_device")
        if rc < 0:
            raise OrbError("orbm_undistort_keypoints_batch_device rc=%d" % rc)

    def set_profiling(self, on=True):
        self.L.orbm_set_profiling(self.m, 1 if on else 0)

    def set_scan_mode(self, mode):
        """0 = per frame pair on the device (default), 1 = k_match_scan, 2 = k_match_walk; results do not depend on it."""
        self._check(self.L.orbm_set_scan_mode(self.m, int(mode)), "orbm_set_scan_mode")

    def set_hamming_engine(self, engine):
        """0 = vector ALU, 1 = open-window blocks on the matrix pipe (k_match_scan_mfma), 2 (default) = as 1 + the lists of all-open frame
        pairs built inside k_match_resolve (fused form); results do not depend on it."""
        self._check(self.L.orbm_set_hamming_engine(self.m, int(engine)), "orbm_set_hamming_engine")

    def last_ms(self):
        return float(self.L.orbm_get_last_ms(self.m))

    def stage_ms(self):
        ms = np.zeros(2, dtype=np.float32)
        n = self.L.orbm_get_stage_ms(self.m, _p(ms), 2)
        return dict(zip(["match_scan", "match_resolve"], ms[:n].tolist()))


def undistort_keypoints(keys, K, D):
    """Frame::UndistortKeyPoints (Frame.cc:837-870): mvKeys -> mvKeysUn."""
    keys = np.ascontiguousarray(keys, dtype=KP_DTYPE)
    K, D = np.ascontiguousarray(K, dtype=np.float32), np.ascontiguousarray(D, dtype=np.float32)
    out = np.zeros_like(keys)
    load().orbm_undistort_keypoints(len(keys), _p(keys), _p(K), _p(D), len(D), _p(out))
    return out


def image_bounds(cols, rows, K, D):
    """Frame::ComputeImageBounds (Frame.cc:872-899): (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    K, D = np.ascontiguousarray(K, dtype=np.float32), np.ascontiguousarray(D, dtype=np.float32)
    v = [C.c_float() for _ in range(4)]
    load().orbm_image_bounds(int(cols), int(rows), _p(K), _p(D), len(D), *[C.byref(x) for x in v])
    return tuple(x.value for x in v)


def project(cam_type, params, X, Y, Z):
    """GeometricCamera::project: 0 = Pinhole (Pinhole.cpp:46-49), 1 = KannalaBrandt8 (KannalaBrandt8.cpp:29-45)."""
    params = np.ascontiguousarray(params, dtype=np.float32)
    u, v = C.c_float(), C.c_float()
    load().orbm_project(int(cam_type), _p(params), C.c_float(X), C.c_float(Y), C.c_float(Z), C.byref(u), C.byref(v))
    return u.value, v.value
