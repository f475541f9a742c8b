// MapPoint_ComputeDistinctiveDescriptors_hip.cc -- drop-in body of void MapPoint::ComputeDistinctiveDescriptors(), replacing
// src/MapPoint.cc:350-436.
//
// As with Frame::ComputeStereoMatches the member lives in a translation unit that holds the whole class: delete (or #if 0)
// lines 350-436 of src/MapPoint.cc and add this file to the library's sources.  The observation walk (:353-383) and the
// write-back (:432-435) stay as written; the all-pairs Hamming distances and the median selection (:389-430) run on the device
// (orbm_distinctive_descriptors: one wavefront per map point, the median by bisection on ballot counts - no N x N matrix).
// More than 1024 observations of one map point (ORBM limit, include/orbhip.h) are refused with an exception.
#include "MapPoint.h"

#include <stdexcept>
#include <vector>

#include "KeyFrame.h"
#include "orbhip.h"

namespace ORB_SLAM3 {

namespace {
orbm_t *distinctive_matcher() {   // one handle per thread: LocalMapping and Tracking both update descriptors
  thread_local orbm_t *m = nullptr;
  if (!m) {
    const char *e = std::getenv("ORBHIP_DEVICE");
    m = orbm_create(e ? std::atoi(e) : 0);
    if (!m) throw std::runtime_error("MapPoint::ComputeDistinctiveDescriptors: orbm_create failed (no usable HIP device; there is no CPU fallback)");
  }
  return m;
}
}  // namespace

void MapPoint::ComputeDistinctiveDescriptors() {
  // Retrieve all observed descriptors, :353-383
  std::vector<cv::Mat> vDescriptors;
  std::map<KeyFrame *, std::tuple<int, int> > observations;
  {
    std::unique_lock<std::mutex> lock1(mMutexFeatures);
    if (mbBad) return;
    observations = mObservations;
  }
  if (observations.empty()) return;
  vDescriptors.reserve(observations.size());
  for (std::map<KeyFrame *, std::tuple<int, int> >::iterator mit = observations.begin(), mend = observations.end(); mit != mend; mit++) {
    KeyFrame *pKF = mit->first;
    if (!pKF->isBad()) {
      const int leftIndex = std::get<0>(mit->second), rightIndex = std::get<1>(mit->second);
      if (leftIndex != -1) vDescriptors.push_back(pKF->mDescriptors.row(leftIndex));
      if (rightIndex != -1) vDescriptors.push_back(pKF->mDescriptors.row(rightIndex));
    }
  }
  if (vDescriptors.empty()) return;
  // :389-430 on the device
  const int N = (int)vDescriptors.size();
  std::vector<uint8_t> desc((size_t)N * 32);
  for (int i = 0; i < N; i++) std::memcpy(&desc[(size_t)i * 32], vDescriptors[i].ptr<uint8_t>(), 32);
  const int32_t start[2] = {0, N};
  int32_t best = 0;
  if (orbm_distinctive_descriptors(distinctive_matcher(), 1, start, desc.data(), &best) < 0 || best < 0)
    throw std::runtime_error(std::string("MapPoint::ComputeDistinctiveDescriptors: ") + orbm_last_error(distinctive_matcher()));
  {
    std::unique_lock<std::mutex> lock(mMutexFeatures);   // :432-435
    mDescriptor = vDescriptors[best].clone();
  }
}

}  // namespace ORB_SLAM3
