// ORBmatcher_hip.cc -- liborbhip-backed definitions of the projection-search members of ORB_SLAM3::ORBmatcher.
//
// Compiled INSIDE the ORB-SLAM3 tree next to the reference's src/ORBmatcher.cc, which is built with
// -DORB_HIP_FRONTEND so that its own definitions of the members below are skipped (INTEGRATION.md shows the
// four #ifndef guards).  Needs the reference's Frame.h / MapPoint.h (OpenCV, Eigen, boost, DBoW2, g2o), so it cannot
// be built in this repository's image.
//
// Replaced members (mono / rectified-stereo / RGB-D frames, Frame::Nleft == -1):
//   int  ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, float th, bool bFarPoints, float thFarPoints)  :44-214
//   int  ORBmatcher::SearchByProjection(Frame &Cur, const Frame &Last, float th, bool bMono)                             :2027-2289
//   int  ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&)                                                  :2463-2483
//   void ORBmatcher::ComputeThreeMaxima(vector<int>*, int, int&, int&, int&)                                             :2416-2458
//   float ORBmatcher::RadiusByViewingCos(const float&)                                                                   :216-222
// Fisheye-stereo frames (Nleft != -1) are forwarded to the reference implementation, which INTEGRATION.md keeps
// available under the name SearchByProjection_ref.
#include "ORBmatcher.h"  // the reference's header

#include <stdexcept>
#include <vector>

#include "Frame.h"
#include "MapPoint.h"
#include "orbhip.h"

namespace ORB_SLAM3 {

namespace {
orbm_t *matcher() {  // one handle per thread: Tracking, LocalMapping and LoopClosing match concurrently
  thread_local orbm_t *m = nullptr;
  if (!m) {
    const char *e = std::getenv("ORBHIP_DEVICE");
    m = orbm_create(e ? std::atoi(e) : 0);
    if (!m) throw std::runtime_error("ORBmatcher: orbm_create failed (no usable HIP device; there is no CPU fallback)");
  }
  return m;
}

// Frame -> orbm_frame_t.  mvKeysUn is a std::vector<cv::KeyPoint>: its data() already has the 28-byte layout.
orbm_frame_t view_of(const Frame &F) {
  orbm_frame_t f;
  f.n = F.N;
  f.keys_un = reinterpret_cast<const orbx_keypoint_t *>(F.mvKeysUn.data());
  f.descriptors = F.mDescriptors.data;  // N x 32, continuous (created by ORBextractor::operator())
  f.u_right = F.mvuRight.empty() ? nullptr : F.mvuRight.data();
  f.min_x = Frame::mnMinX; f.max_x = Frame::mnMaxX; f.min_y = Frame::mnMinY; f.max_y = Frame::mnMaxY;
  return f;
}

// F.mvpMapPoints <-> (slot, slot_obs).  Pre-existing occupants get id 2^30 so they are never confused with a query index.
void slots_of(const Frame &F, std::vector<int32_t> &slot, std::vector<uint8_t> &obs) {
  slot.assign(F.N, -1);
  obs.assign(F.N, 0);
  for (int i = 0; i < F.N; i++)
    if (F.mvpMapPoints[i]) { slot[i] = 1 << 30; obs[i] = F.mvpMapPoints[i]->Observations() > 0; }
}
}  // namespace

int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return orbm_descriptor_distance(a.ptr<uint8_t>(), b.ptr<uint8_t>()); }

float ORBmatcher::RadiusByViewingCos(const float &viewCos) { return orbm_radius_by_viewing_cos(viewCos); }

void ORBmatcher::ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3) {
  std::vector<int> sizes(L);
  for (int i = 0; i < L; i++) sizes[i] = (int)histo[i].size();
  orbm_three_maxima(sizes.data(), L, &ind1, &ind2, &ind3);
}

int ORBmatcher::SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th, const bool bFarPoints,
                                   const float thFarPoints) {
  if (F.Nleft != -1) return SearchByProjection_ref(F, vpMapPoints, th, bFarPoints, thFarPoints);
  const int nq = (int)vpMapPoints.size();
  std::vector<uint8_t> qdesc((size_t)nq * 32), flags(nq, 0);
  std::vector<float> u(nq, 0.f), v(nq, 0.f), rad(nq, 0.f), ur(nq, 0.f);
  std::vector<int32_t> minl(nq, -1), maxl(nq, -1);
  const bool bFactor = th != 1.0;
  for (int q = 0; q < nq; q++) {
    MapPoint *pMP = vpMapPoints[q];
    if (!pMP->mbTrackInView) continue;                          // :52 (mbTrackInViewR only matters when Nleft != -1)
    if (bFarPoints && pMP->mTrackDepth > thFarPoints) continue; // :55
    if (pMP->isBad()) continue;                                 // :58
    const int lvl = pMP->mnTrackScaleLevel;
    float r = RadiusByViewingCos(pMP->mTrackViewCos);           // :67
    if (bFactor) r *= th;                                       // :69-70
    rad[q] = r * F.mvScaleFactors[lvl];                         // :73
    u[q] = pMP->mTrackProjX; v[q] = pMP->mTrackProjY; ur[q] = pMP->mTrackProjXR;
    minl[q] = lvl - 1; maxl[q] = lvl;
    const cv::Mat d = pMP->GetDescriptor();
    std::memcpy(&qdesc[(size_t)q * 32], d.ptr<uint8_t>(), 32);
    flags[q] = (uint8_t)(1u | ((pMP->Observations() > 0 ? 1u : 0u) << 1));
  }
  std::vector<int32_t> slot, moq(nq);
  std::vector<uint8_t> sobs;
  slots_of(F, slot, sobs);
  const orbm_frame_t f = view_of(F);
  orbm_queries_t qs{nq, qdesc.data(), u.data(), v.data(), rad.data(), minl.data(), maxl.data(), ur.data(), flags.data()};
  const int n = orbm_search_by_projection(matcher(), &f, &qs, mfNNratio, TH_HIGH, 1, slot.data(), sobs.data(), moq.data(), nullptr);
  if (n < 0) throw std::runtime_error(orbm_last_error(matcher()));
  for (int q = 0; q < nq; q++)
    if (moq[q] >= 0 && slot[moq[q]] == q) F.mvpMapPoints[moq[q]] = vpMapPoints[q];  // :130 (last claimer wins)
  return n;
}

int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono) {
  if (CurrentFrame.Nleft != -1 || LastFrame.Nleft != -1) return SearchByProjection_ref(CurrentFrame, LastFrame, th, bMono);
  const int nLast = LastFrame.N;
  std::vector<uint8_t> has(nLast, 0), obs(nLast, 0), desc((size_t)nLast * 32);
  std::vector<float> Xw((size_t)nLast * 3, 0.f);
  std::vector<orbx_keypoint_t> lk(nLast);
  for (int i = 0; i < nLast; i++) {
    std::memcpy(&lk[i], &LastFrame.mvKeysUn[i], sizeof(orbx_keypoint_t));
    lk[i].octave = LastFrame.mvKeys[i].octave;
    MapPoint *pMP = LastFrame.mvpMapPoints[i];
    if (!pMP || LastFrame.mvbOutlier[i]) continue;              // :2058-2061
    has[i] = 1;
    obs[i] = pMP->Observations() > 0;
    const cv::Mat x3Dw = pMP->GetWorldPos();
    for (int k = 0; k < 3; k++) Xw[(size_t)i * 3 + k] = x3Dw.at<float>(k);
    const cv::Mat d = pMP->GetDescriptor();
    std::memcpy(&desc[(size_t)i * 32], d.ptr<uint8_t>(), 32);
  }
  float Tcw[16], Tlw[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) { Tcw[r * 4 + c] = CurrentFrame.mTcw.at<float>(r, c); Tlw[r * 4 + c] = LastFrame.mTcw.at<float>(r, c); }
  const int camType = CurrentFrame.mpCamera->GetType();         // GeometricCamera::CAM_PINHOLE = 0, CAM_FISHEYE = 1
  std::vector<float> params(camType == 0 ? 4 : 8);
  for (size_t k = 0; k < params.size(); k++) params[k] = CurrentFrame.mpCamera->getParameter((int)k);
  std::vector<int32_t> slot;
  std::vector<uint8_t> sobs;
  slots_of(CurrentFrame, slot, sobs);
  const orbm_frame_t f = view_of(CurrentFrame);
  const int n = orbm_search_by_projection_last_frame(matcher(), &f, CurrentFrame.mvScaleFactors.data(), (int)CurrentFrame.mvScaleFactors.size(),
                                                     nLast, has.data(), Xw.data(), desc.data(), lk.data(), obs.data(), Tcw, Tlw, camType,
                                                     params.data(), CurrentFrame.mb, CurrentFrame.mbf, th, bMono ? 1 : 0,
                                                     mbCheckOrientation ? 1 : 0, slot.data(), sobs.data());
  if (n < 0) throw std::runtime_error(orbm_last_error(matcher()));
  for (int i = 0; i < CurrentFrame.N; i++) {
    if (slot[i] >= 0 && slot[i] < nLast) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[slot[i]];      // :2162
    else if (slot[i] == -1) CurrentFrame.mvpMapPoints[i] = static_cast<MapPoint *>(NULL);                     // :2279 (pruned)
  }
  return n;
}

}  // namespace ORB_SLAM3
