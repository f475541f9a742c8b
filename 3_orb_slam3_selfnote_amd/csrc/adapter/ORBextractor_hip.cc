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
// The class layout is untouched, so the liborbhip handle lives in a side table keyed by `this`.
#include "ORBextractor.h"  // the reference's header

#include <cstring>
#include <mutex>
#include <stdexcept>
#include <unordered_map>

#include "orbhip.h"

namespace ORB_SLAM3 {

namespace {
std::mutex g_mu;
std::unordered_map<const ORBextractor *, orbx_t *> g_handles;  // ~ORBextractor(){} is inline and empty in the header:
                                                                // handles live until process exit (Tracking owns 3)
bool g_fill_pyramid = true;  // mvImagePyramid is read by Frame::ComputeStereoMatches (Frame.cc:908-1017)

orbx_t *handle_of(const ORBextractor *self) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_handles.find(self);
  return it == g_handles.end() ? nullptr : it->second;
}
int device_from_env() {
  const char *e = std::getenv("ORBHIP_DEVICE");
  return e ? std::atoi(e) : 0;
}
}  // namespace

// optional knob for monocular pipelines that never read mvImagePyramid: skips the 1.1 MB/frame download
void ORBHIP_SetFillPyramid(bool on) { g_fill_pyramid = on; }

// the liborbhip handle behind an extractor, for Frame::ComputeStereoMatches -> orbx_compute_stereo_matches
orbx_t *ORBHIP_Handle(const ORBextractor *e) { return handle_of(e); }

static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint_t), "cv::KeyPoint must be the 28-byte POD of OpenCV 2.4/3.x");

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
  std::lock_guard<std::mutex> lk(g_mu);
  g_handles[this] = h;
}

int ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask: ignored by the reference too*/,
                             std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors,
                             std::vector<int> &vLappingArea) {
  if (_image.empty()) return -1;  // ORBextractor.cc:1075-1076
  cv::Mat image = _image.getMat();
  if (image.type() != CV_8UC1) throw std::runtime_error("ORBextractor: image must be CV_8UC1");  // assert at :1080
  orbx_t *h = handle_of(this);
  const int cap = orbx_configure(h, image.rows, image.cols, 1);
  if (cap < 0) throw std::runtime_error(std::string("ORBextractor: ") + orbx_last_error(h));
  _keypoints.resize(cap);
  cv::Mat desc(cap, 32, CV_8U);
  int n = 0;
  const int rc = orbx_extract(h, image.data, image.rows, image.cols, image.step, vLappingArea[0], vLappingArea[1],
                              reinterpret_cast<orbx_keypoint_t *>(_keypoints.data()), desc.data, cap, &n);
  if (rc == ORBX_E_EMPTY) return -1;
  if (rc < 0) throw std::runtime_error(std::string("ORBextractor: ") + orbx_last_error(h));
  _keypoints.resize(n);
  if (n == 0) {
    _descriptors.release();  // :1100-1101
  } else {
    _descriptors.create(n, 32, CV_8U);  // :1105-1108
    desc.rowRange(0, n).copyTo(_descriptors.getMat());
  }
  if (g_fill_pyramid) {
    const int B = 19;  // EDGE_THRESHOLD: keep the reference's "ROI of a bordered Mat" shape, :1194-1196
    for (int l = 0; l < nlevels; ++l) {
      int r, c;
      orbx_level_info(h, l, &r, &c);
      cv::Mat temp(r + 2 * B, c + 2 * B, CV_8UC1);
      orbx_download_level(h, 0, l, B, temp.data, temp.step);
      mvImagePyramid[l] = temp(cv::Rect(B, B, c, r));
    }
  }
  return rc;  // monoIndex, :1183
}

}  // namespace ORB_SLAM3
