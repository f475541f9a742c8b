// ORBextractor_hip.cc -- drop-in replacement of the reference's src/ORBextractor.cc.
//
// Compiled INSIDE the ORB-SLAM3 tree (it includes the reference's own, unmodified include/ORBextractor.h and
// OpenCV); it cannot be built in this repository's image, which has neither (DESIGN.md "Boundary").
// It defines exactly the members of ORB_SLAM3::ORBextractor that other translation units reference:
//   ORBextractor::ORBextractor(int,float,int,int,int)                          (ORBextractor.cc:408-468)
//   int ORBextractor::operator()(InputArray, InputArray, vector<KeyPoint>&, OutputArray, vector<int>&)  (:1071-1184)
// and keeps the inline getters of ORBextractor.h:61-81 and the public mvImagePyramid (:83) working by filling the
// members they read.  Frame.cc:469-483 (Frame::ExtractORB) and Tracking.cc:838-844 call it unchanged.
//
// Per call the adapter does what tools/adapter_harness.c times: orbx_configure (a no-op after the first frame), orbx_extract into
// the caller's keypoint vector and a descriptor block kept between frames, one n x 32 copy into the caller's matrix - no lock, no
// allocation besides the two output containers the reference allocates as well (:1100-1108).
//
// The class layout is untouched and ~ORBextractor(){} is inline and empty in the header, so the liborbhip handle is kept in the
// one member no other translation unit can see or needs: the protected `pattern` vector (the 512 test points of
// ORBextractor.cc:150-405, which live inside liborbhip here).  Its single element holds the handle's address - a per-object
// slot, read without a lock from whichever thread runs the extractor (the stereo pipeline starts a new thread per frame for the
// right image, Frame.cc:120-123).  A registry under a mutex exists only for construction and for destroying the handles at exit.
//
// mvImagePyramid (ORBextractor.h:83) is read by ONE caller, the reference's Frame::ComputeStereoMatches (Frame.cc:908-1017), and
// only on rectified-stereo input; monocular, RGB-D and IMU configurations never touch it, and with the member snippet of
// INTEGRATION.md 3b (orbx_compute_stereo_matches) the stereo search runs on the device-resident pyramid as well.  So it is NOT
// filled by default.  A build that keeps the reference's own ComputeStereoMatches switches it on once at start-up:
// ORBHIP_SetFillPyramid(true), or ORBHIP_FILL_PYRAMID=1 in the environment.  The fill is then one packed transfer
// (orbx_download_pyramid) into a block the extractor keeps between frames; the cv::Mat headers of mvImagePyramid point into it
// with the reference's "ROI of a bordered Mat" shape (:1194-1196), no allocation per frame.
#include "ORBextractor.h"  // the reference's header

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <vector>

#include "orbhip.h"

namespace ORB_SLAM3 {

namespace {
static_assert(sizeof(cv::Point) >= sizeof(void *), "cv::Point (two ints) must hold a pointer");
static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint_t), "cv::KeyPoint must be the 28-byte POD of OpenCV 2.4/3.x");

struct Entry {
  orbx_t *h = nullptr;
  std::vector<uint8_t> desc;                // descriptor rows of the last frame before they are copied into the caller's matrix
  std::vector<uint8_t> pyr;                 // packed bordered pyramid of the last frame (fill mode only)
  std::vector<size_t> off, stride;
};

struct Registry {                            // construction and process exit only
  std::mutex mu;
  std::vector<Entry *> all;
  ~Registry() {
    for (Entry *e : all) { if (e->h) orbx_destroy(e->h); delete e; }
  }
};
Registry &registry() { static Registry r; return r; }

std::atomic<int> g_fill_pyramid{-1};         // -1: not decided (environment), 0 / 1
bool fill_pyramid() {
  int v = g_fill_pyramid.load(std::memory_order_relaxed);
  if (v < 0) {
    const char *e = std::getenv("ORBHIP_FILL_PYRAMID");
    v = (e && std::atoi(e) != 0) ? 1 : 0;
    g_fill_pyramid.store(v, std::memory_order_relaxed);
  }
  return v != 0;
}
int device_from_env() {
  const char *e = std::getenv("ORBHIP_DEVICE");
  return e ? std::atoi(e) : 0;
}
inline Entry *entry_of(const std::vector<cv::Point> &pattern) {
  Entry *e = nullptr;
  if (!pattern.empty()) std::memcpy(&e, &pattern[0], sizeof(e));
  return e;
}
}  // namespace

// mvImagePyramid on the host, for builds that keep the reference's Frame::ComputeStereoMatches (see the header comment)
void ORBHIP_SetFillPyramid(bool on) { g_fill_pyramid.store(on ? 1 : 0, std::memory_order_relaxed); }

// the liborbhip handle behind an extractor, for Frame::ComputeStereoMatches -> orbx_compute_stereo_matches
struct ORBHIP_Access : ORBextractor {
  static orbx_t *handle(const ORBextractor *e) {
    Entry *en = entry_of(static_cast<const ORBHIP_Access *>(e)->pattern);
    return en ? en->h : nullptr;
  }
};
orbx_t *ORBHIP_Handle(const ORBextractor *e) { return ORBHIP_Access::handle(e); }

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST) {
  orbx_t *h = orbx_create(_nfeatures, _scaleFactor, _nlevels, _iniThFAST, _minThFAST, device_from_env());
  if (!h) throw std::runtime_error("ORBextractor: orbx_create failed (no usable HIP device; there is no CPU fallback)");
  mvScaleFactor.resize(nlevels);
  mvInvScaleFactor.resize(nlevels);
  mvLevelSigma2.resize(nlevels);
  mvInvLevelSigma2.resize(nlevels);
  orbx_get_scale_tables(h, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data());
  mnFeaturesPerLevel.resize(nlevels);
  orbx_get_features_per_level(h, mnFeaturesPerLevel.data());
  mvImagePyramid.resize(nlevels);
  Entry *e = new Entry();
  e->h = h;
  e->off.resize(nlevels); e->stride.resize(nlevels);
  pattern.assign(1, cv::Point(0, 0));
  std::memcpy(&pattern[0], &e, sizeof(e));
  Registry &r = registry();
  std::lock_guard<std::mutex> lk(r.mu);
  r.all.push_back(e);
}

int ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask: ignored by the reference too*/,
                             std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors,
                             std::vector<int> &vLappingArea) {
  if (_image.empty()) return -1;  // ORBextractor.cc:1075-1076
  cv::Mat image = _image.getMat();
  if (image.type() != CV_8UC1) throw std::runtime_error("ORBextractor: image must be CV_8UC1");  // assert at :1080
  Entry *e = entry_of(pattern);
  orbx_t *h = e->h;
  const int cap = orbx_configure(h, image.rows, image.cols, 1);
  if (cap < 0) throw std::runtime_error(std::string("ORBextractor: ") + orbx_last_error(h));
  _keypoints.resize(cap);
  if (e->desc.size() < (size_t)cap * 32) e->desc.resize((size_t)cap * 32);   // kept between frames
  int n = 0;
  const int rc = orbx_extract(h, image.data, image.rows, image.cols, image.step, vLappingArea[0], vLappingArea[1],
                              reinterpret_cast<orbx_keypoint_t *>(_keypoints.data()), e->desc.data(), cap, &n);
  if (rc == ORBX_E_EMPTY) return -1;
  if (rc < 0) throw std::runtime_error(std::string("ORBextractor: ") + orbx_last_error(h));
  _keypoints.resize(n);
  if (n == 0) {
    _descriptors.release();  // :1100-1101
  } else {
    _descriptors.create(n, 32, CV_8U);  // :1105-1108
    cv::Mat out = _descriptors.getMat();
    for (int i = 0; i < n; i++) std::memcpy(out.data + (size_t)i * out.step, e->desc.data() + (size_t)i * 32, 32);
  }
  if (fill_pyramid()) {
    const int B = 19;  // EDGE_THRESHOLD: the reference's "ROI of a bordered Mat" shape, :1194-1196
    const int need = orbx_download_pyramid(h, 0, B, NULL, 0, e->off.data(), e->stride.data());
    if (need < 0) throw std::runtime_error(std::string("ORBextractor: ") + orbx_last_error(h));
    if (e->pyr.size() < (size_t)need) e->pyr.resize((size_t)need);
    if (orbx_download_pyramid(h, 0, B, e->pyr.data(), e->pyr.size(), e->off.data(), e->stride.data()) < 0)
      throw std::runtime_error(std::string("ORBextractor: ") + orbx_last_error(h));
    for (int l = 0; l < nlevels; ++l) {
      int r, c;
      orbx_level_info(h, l, &r, &c);
      cv::Mat temp(r + 2 * B, c + 2 * B, CV_8UC1, e->pyr.data() + e->off[l], e->stride[l]);
      mvImagePyramid[l] = temp(cv::Rect(B, B, c, r));
    }
  }
  return rc;  // monoIndex, :1183
}

}  // namespace ORB_SLAM3
