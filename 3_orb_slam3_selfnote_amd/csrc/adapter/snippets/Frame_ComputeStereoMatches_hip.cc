// Frame_ComputeStereoMatches_hip.cc -- drop-in body of void Frame::ComputeStereoMatches(), replacing src/Frame.cc:901-1079.
//
// Frame.cc holds the whole Frame class, so this member cannot be swapped by exchanging a translation unit: delete (or #if 0)
// lines 901-1079 of src/Frame.cc and add this file to the library's sources.  The row-band
// descriptor search, the 11x11 SAD refinement at 11 offsets and the parabola fit (:937-1050) run on the device on the two
// extractors' still-resident pyramids (one wavefront per left keypoint); the median filter over the accepted matches
// (:1060-1073) runs inside orbx_compute_stereo_matches on the host.  No image crosses PCIe a second time, and the pyramid
// download into mvImagePyramid - whose only consumer this function is - can be switched off (ORBHIP_SetFillPyramid(false)).
#include "Frame.h"

#include <stdexcept>

#include "orbhip.h"

namespace ORB_SLAM3 {

orbx_t *ORBHIP_Handle(const ORBextractor *e);  // ORBextractor_hip.cc: the liborbhip handle behind an extractor instance

void Frame::ComputeStereoMatches() {
  mvuRight = std::vector<float>(N, -1.0f);   // :903-904
  mvDepth = std::vector<float>(N, -1.0f);
  orbx_t *hl = ORBHIP_Handle(mpORBextractorLeft), *hr = ORBHIP_Handle(mpORBextractorRight);
  if (!hl || !hr) throw std::runtime_error("Frame::ComputeStereoMatches: extractors without a liborbhip handle");
  const int rc = orbx_compute_stereo_matches(hl, 0, hr, 0, N, reinterpret_cast<const orbx_keypoint_t *>(mvKeys.data()), mDescriptors.data,
                                             (int)mvKeysRight.size(), reinterpret_cast<const orbx_keypoint_t *>(mvKeysRight.data()),
                                             mDescriptorsRight.data, mb, mbf, mvuRight.data(), mvDepth.data());
  if (rc < 0) throw std::runtime_error(std::string("Frame::ComputeStereoMatches: ") + orbx_last_error(hl));
}

}  // namespace ORB_SLAM3
