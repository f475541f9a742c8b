// ORBmatcher_hip.cc -- liborbhip-backed definitions of the projection-search members of ORB_SLAM3::ORBmatcher.
//
// Compiled INSIDE the ORB-SLAM3 tree next to the reference's src/ORBmatcher.cc, which is built with
// -DORB_HIP_FRONTEND so that its own definitions of the members below are skipped (INTEGRATION.md shows the
// four #ifndef guards).  Needs the reference's Frame.h / MapPoint.h (OpenCV, Eigen, boost, DBoW2, g2o), so it cannot
// be built in this repository's image.
//
// Replaced members (the two Frame overloads of SearchByProjection for every frame type, fisheye-stereo included; the
// KeyFrame overload for Frame::Nleft == -1):
//   int  ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, float th, bool bFarPoints, float thFarPoints)  :44-214
//   int  ORBmatcher::SearchByProjection(Frame &Cur, const Frame &Last, float th, bool bMono)                             :2027-2289
//   int  ORBmatcher::SearchByProjection(Frame &Cur, KeyFrame *pKF, const set<MapPoint*>&, float th, int ORBdist)       :2291-2413
//   int  ORBmatcher::SearchForTriangulation(KeyFrame*, KeyFrame*, cv::Mat F12, vector<pair<size_t,size_t>>&, bool, bool) :981-1222
//   int  ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f>&, vector<int>&, int windowSize)  :722-837
//   int  ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches)                          :273-469
//   int  ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12)                         :839-979
//   int  ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&)                                                  :2463-2483
//   void ORBmatcher::ComputeThreeMaxima(vector<int>*, int, int&, int&, int&)                                             :2416-2458
//   float ORBmatcher::RadiusByViewingCos(const float&)                                                                   :216-222
// The KeyFrame overload forwards fisheye-stereo frames (Nleft != -1) to the reference implementation, which
// INTEGRATION.md keeps available under the name SearchByProjection_ref.
#include "ORBmatcher.h"  // the reference's header

#include <stdexcept>
#include <vector>

#include "Frame.h"
#include "KeyFrame.h"
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

// Fisheye-stereo frame (Nleft != -1): GetFeaturesInArea reads mvKeys / mvKeysRight (Frame.cc:791-793) and mDescriptors
// already holds both images (Frame.cc:1201); the keypoints are concatenated here.
orbm_frame_t view_of_fisheye(const Frame &F, std::vector<orbx_keypoint_t> &keys) {
  keys.resize(F.N);
  if (F.Nleft > 0) std::memcpy(keys.data(), F.mvKeys.data(), sizeof(orbx_keypoint_t) * (size_t)F.Nleft);
  if (F.Nright > 0) std::memcpy(keys.data() + F.Nleft, F.mvKeysRight.data(), sizeof(orbx_keypoint_t) * (size_t)F.Nright);
  orbm_frame_t f;
  f.n = F.N;
  f.keys_un = keys.data();
  f.descriptors = F.mDescriptors.data;
  f.u_right = nullptr;  // not tested when Nleft != -1 (ORBmatcher.cc:93, :2139)
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
  if (F.Nleft != -1) {
    // two queries per map point: 2j = left image, 2j+1 = right image (ORBmatcher.cc:145-211)
    const int nmp = (int)vpMapPoints.size(), nq = 2 * nmp;
    std::vector<uint8_t> qdesc((size_t)nq * 32), flags(nq, 0);
    std::vector<float> u(nq, 0.f), v(nq, 0.f), rad(nq, 0.f);
    std::vector<int32_t> minl(nq, -1), maxl(nq, -1);
    const bool bFactor = th != 1.0;
    for (int j = 0; j < nmp; j++) {
      MapPoint *pMP = vpMapPoints[j];
      if (!pMP->mbTrackInView && !pMP->mbTrackInViewR) continue;  // :52
      if (bFarPoints && pMP->mTrackDepth > thFarPoints) continue; // :55
      if (pMP->isBad()) continue;                                 // :58
      const uint8_t ob = (uint8_t)((pMP->Observations() > 0 ? 1u : 0u) << 1);
      const cv::Mat d = pMP->GetDescriptor();
      std::memcpy(&qdesc[(size_t)(2 * j) * 32], d.ptr<uint8_t>(), 32);
      std::memcpy(&qdesc[(size_t)(2 * j + 1) * 32], d.ptr<uint8_t>(), 32);
      if (pMP->mbTrackInView) {
        const int lvl = pMP->mnTrackScaleLevel;
        float r = RadiusByViewingCos(pMP->mTrackViewCos);
        if (bFactor) r *= th;
        rad[2 * j] = r * F.mvScaleFactors[lvl];
        u[2 * j] = pMP->mTrackProjX; v[2 * j] = pMP->mTrackProjY;
        minl[2 * j] = lvl - 1; maxl[2 * j] = lvl;
        flags[2 * j] = (uint8_t)(1u | ob);
      }
      if (pMP->mbTrackInViewR && pMP->mnTrackScaleLevelR != -1) { // :145-147
        const int lvl = pMP->mnTrackScaleLevelR;
        const float r = RadiusByViewingCos(pMP->mTrackViewCosR);  // :148, no th factor
        rad[2 * j + 1] = r * F.mvScaleFactors[lvl];
        u[2 * j + 1] = pMP->mTrackProjXR; v[2 * j + 1] = pMP->mTrackProjYR;
        minl[2 * j + 1] = lvl - 1; maxl[2 * j + 1] = lvl;
        flags[2 * j + 1] = (uint8_t)(1u | ob);
      }
    }
    std::vector<int32_t> slot, moq(nq);
    std::vector<uint8_t> sobs;
    slots_of(F, slot, sobs);
    std::vector<orbx_keypoint_t> keys;
    const orbm_frame_t f = view_of_fisheye(F, keys);
    orbm_queries_t qs{nq, qdesc.data(), u.data(), v.data(), rad.data(), minl.data(), maxl.data(), nullptr, flags.data()};
    const int n = orbm_search_by_projection_fisheye(matcher(), &f, F.Nleft, F.mvLeftToRightMatch.data(), F.mvRightToLeftMatch.data(), &qs,
                                                    mfNNratio, TH_HIGH, slot.data(), sobs.data(), moq.data(), nullptr);
    if (n < 0) throw std::runtime_error(orbm_last_error(matcher()));
    for (int i = 0; i < F.N; i++)  // slots written by this call hold a query id: map point = id / 2 (own and partner writes alike)
      if (slot[i] >= 0 && slot[i] < nq) F.mvpMapPoints[i] = vpMapPoints[slot[i] >> 1];
    return n;
  }
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
  const bool fisheye = CurrentFrame.Nleft != -1;
  const int nLast = LastFrame.N;
  std::vector<uint8_t> has(nLast, 0), obs(nLast, 0), desc((size_t)nLast * 32);
  std::vector<float> Xw((size_t)nLast * 3, 0.f);
  std::vector<orbx_keypoint_t> lk(nLast);
  for (int i = 0; i < nLast; i++) {
    // angle: :2168-2170 / octave: :2100 - mvKeysUn / mvKeys, or mvKeys / mvKeysRight for a fisheye-stereo last frame
    const cv::KeyPoint &kpA = LastFrame.Nleft == -1 ? LastFrame.mvKeysUn[i] : (i < LastFrame.Nleft ? LastFrame.mvKeys[i] : LastFrame.mvKeysRight[i - LastFrame.Nleft]);
    const cv::KeyPoint &kpO = (LastFrame.Nleft == -1 || i < LastFrame.Nleft) ? LastFrame.mvKeys[i] : LastFrame.mvKeysRight[i - LastFrame.Nleft];
    std::memcpy(&lk[i], &kpA, sizeof(orbx_keypoint_t));
    lk[i].octave = kpO.octave;
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
  std::vector<orbx_keypoint_t> keys;
  const orbm_frame_t f = fisheye ? view_of_fisheye(CurrentFrame, keys) : view_of(CurrentFrame);
  int n;
  if (fisheye) {
    float Trl[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) Trl[r * 4 + c] = CurrentFrame.mTrl.at<float>(r, c);  // :2190
    n = orbm_search_by_projection_last_frame_fisheye(matcher(), &f, CurrentFrame.Nleft, CurrentFrame.mvScaleFactors.data(),
                                                     (int)CurrentFrame.mvScaleFactors.size(), nLast, has.data(), Xw.data(), desc.data(),
                                                     lk.data(), obs.data(), Tcw, Tlw, Trl, camType, params.data(), CurrentFrame.mb, th,
                                                     bMono ? 1 : 0, mbCheckOrientation ? 1 : 0, slot.data(), sobs.data());
  } else {
    n = orbm_search_by_projection_last_frame(matcher(), &f, CurrentFrame.mvScaleFactors.data(), (int)CurrentFrame.mvScaleFactors.size(),
                                             nLast, has.data(), Xw.data(), desc.data(), lk.data(), obs.data(), Tcw, Tlw, camType,
                                             params.data(), CurrentFrame.mb, CurrentFrame.mbf, th, bMono ? 1 : 0,
                                             mbCheckOrientation ? 1 : 0, slot.data(), sobs.data());
  }
  if (n < 0) throw std::runtime_error(orbm_last_error(matcher()));
  for (int i = 0; i < CurrentFrame.N; i++) {
    if (slot[i] >= 0 && slot[i] < nLast) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[slot[i]];      // :2162
    else if (slot[i] == -1) CurrentFrame.mvpMapPoints[i] = static_cast<MapPoint *>(NULL);                     // :2279 (pruned)
  }
  return n;
}

int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12,
                                        int windowSize) {
  vnMatches12.assign(F1.mvKeysUn.size(), -1);                    // :725
  orbm_frame_t f1 = view_of(F1), f2 = view_of(F2);
  f1.n = (int)F1.mvKeysUn.size(); f2.n = (int)F2.mvKeysUn.size();
  static_assert(sizeof(cv::Point2f) == 2 * sizeof(float), "cv::Point2f is two floats");
  const int n = orbm_search_for_initialization(matcher(), &f1, &f2, reinterpret_cast<float *>(vbPrevMatched.data()), windowSize, mfNNratio,
                                               mbCheckOrientation ? 1 : 0, vnMatches12.data());
  if (n < 0) throw std::runtime_error(orbm_last_error(matcher()));
  return n;
}

int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th,
                                   const int ORBdist) {
  if (CurrentFrame.Nleft != -1) return SearchByProjection_ref(CurrentFrame, pKF, sAlreadyFound, th, ORBdist);
  const std::vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();
  const int nKF = (int)vpMPs.size();
  std::vector<uint8_t> valid(nKF, 0), desc((size_t)nKF * 32);
  std::vector<float> Xw((size_t)nKF * 3, 0.f), ang(nKF, 0.f), dmax(nKF, 0.f), dmin(nKF, 0.f);
  for (int i = 0; i < nKF; i++) {
    MapPoint *pMP = vpMPs[i];
    if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;       // :2308-2312
    valid[i] = 1;
    const cv::Mat x3Dw = pMP->GetWorldPos();
    for (int k = 0; k < 3; k++) Xw[(size_t)i * 3 + k] = x3Dw.at<float>(k);
    std::memcpy(&desc[(size_t)i * 32], pMP->GetDescriptor().ptr<uint8_t>(), 32);
    ang[i] = pKF->mvKeysUn[i].angle;
    dmax[i] = pMP->GetMaxDistanceInvariance() / 1.2f;                     // the C ABI takes mfMaxDistance / mfMinDistance
    dmin[i] = pMP->GetMinDistanceInvariance() / 0.8f;
  }
  float Tcw[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) Tcw[r * 4 + c] = CurrentFrame.mTcw.at<float>(r, c);
  const int camType = CurrentFrame.mpCamera->GetType();
  std::vector<float> params(camType == 0 ? 4 : 8);
  for (size_t k = 0; k < params.size(); k++) params[k] = CurrentFrame.mpCamera->getParameter((int)k);
  std::vector<int32_t> slot(CurrentFrame.N, -1);
  std::vector<uint8_t> sobs(CurrentFrame.N, 0);
  for (int i = 0; i < CurrentFrame.N; i++)
    if (CurrentFrame.mvpMapPoints[i]) { slot[i] = 1 << 30; sobs[i] = 1; }  // any occupant blocks, :2355-2356
  const orbm_frame_t f = view_of(CurrentFrame);
  const int n = orbm_search_by_projection_keyframe(matcher(), &f, CurrentFrame.mvScaleFactors.data(), (int)CurrentFrame.mvScaleFactors.size(),
                                                   CurrentFrame.mfLogScaleFactor, nKF, valid.data(), Xw.data(), desc.data(), ang.data(),
                                                   dmax.data(), dmin.data(), Tcw, camType, params.data(), th, ORBdist,
                                                   mbCheckOrientation ? 1 : 0, slot.data(), sobs.data());
  if (n < 0) throw std::runtime_error(orbm_last_error(matcher()));
  for (int i = 0; i < CurrentFrame.N; i++)
    if (slot[i] >= 0 && slot[i] < nKF) CurrentFrame.mvpMapPoints[i] = vpMPs[slot[i]];  // :2373
  return n;
}

int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches) {
  // Two-camera (fisheye) rigs: keypoint k of a view is mvKeys[k] for k < Nleft and mvKeysRight[k - Nleft] after it (:383-395);
  // the C ABI takes that concatenation.  A rig on one side only would mix mvKeysUn with raw keys: reference path.
  if ((F.Nleft != -1) != (pKF->mpCamera2 != nullptr)) return SearchByBoW_ref(pKF, F, vpMapPointMatches);
  const std::vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();
  vpMapPointMatches.assign(F.N, static_cast<MapPoint *>(NULL));  // :277
  struct Flat { std::vector<uint8_t> has; std::vector<uint32_t> id; std::vector<int32_t> start, idx; orbm_keyframe_t k; };
  auto flatten = [](const DBoW2::FeatureVector &fv, Flat &X) {
    X.start.push_back(0);
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
      X.id.push_back(it->first);
      for (unsigned v : it->second) X.idx.push_back((int32_t)v);
      X.start.push_back((int32_t)X.idx.size());
    }
    X.k.n_nodes = (int32_t)X.id.size();
    X.k.node_id = X.id.data(); X.k.node_start = X.start.data(); X.k.node_idx = X.idx.data();
    X.k.u_right = nullptr; X.k.scale_factors = nullptr; X.k.level_sigma2 = nullptr; X.k.nlevels = 0;
  };
  Flat A, B;
  flatten(pKF->mFeatVec, A);
  flatten(F.mFeatVec, B);
  A.has.resize(pKF->N);
  for (int i = 0; i < pKF->N; i++) A.has[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();  // :307-313
  std::vector<cv::KeyPoint> keysKF, keysF;
  const bool rig = F.Nleft != -1;
  if (rig) {
    keysKF = pKF->mvKeys; keysKF.insert(keysKF.end(), pKF->mvKeysRight.begin(), pKF->mvKeysRight.end());
    keysF = F.mvKeys; keysF.insert(keysF.end(), F.mvKeysRight.begin(), F.mvKeysRight.end());
  }
  A.k.n = pKF->N; A.k.keys_un = reinterpret_cast<const orbx_keypoint_t *>(rig ? keysKF.data() : pKF->mvKeysUn.data());
  A.k.descriptors = pKF->mDescriptors.data; A.k.has_mappoint = A.has.data();
  B.has.assign(F.N, 0);
  B.k.n = F.N; B.k.keys_un = reinterpret_cast<const orbx_keypoint_t *>(rig ? keysF.data() : F.mvKeys.data());  // angle of F.mvKeys, :395
  B.k.descriptors = F.mDescriptors.data; B.k.has_mappoint = B.has.data();
  std::vector<int32_t> mF(F.N, -1);
  const int n = rig ? orbm_search_by_bow_fisheye(matcher(), &A.k, &B.k, F.Nleft, mfNNratio, mbCheckOrientation ? 1 : 0, mF.data())
                    : orbm_search_by_bow(matcher(), &A.k, &B.k, mfNNratio, mbCheckOrientation ? 1 : 0, mF.data());
  if (n < 0) throw std::runtime_error(orbm_last_error(matcher()));
  for (int i = 0; i < F.N; i++)
    if (mF[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[mF[i]];  // :389
  return n;
}

int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12) {
  // Two-camera rigs: the reference skips every index >= mvKeysUn.size(), i.e. the right image's keypoints (:874-876, :894-896);
  // clearing has_mappoint for them has the same effect (`if(!pMP) continue`, :879 / :900).
  const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
  vpMatches12.assign(vpMapPoints1.size(), static_cast<MapPoint *>(NULL));  // :852
  struct Flat { std::vector<uint8_t> has; std::vector<uint32_t> id; std::vector<int32_t> start, idx; std::vector<cv::KeyPoint> keys; orbm_keyframe_t k; };
  auto flatten = [](KeyFrame *pKF, const std::vector<MapPoint *> &mps, Flat &X) {
    X.start.push_back(0);
    for (DBoW2::FeatureVector::const_iterator it = pKF->mFeatVec.begin(); it != pKF->mFeatVec.end(); ++it) {
      X.id.push_back(it->first);
      for (unsigned v : it->second) X.idx.push_back((int32_t)v);
      X.start.push_back((int32_t)X.idx.size());
    }
    const int nUn = (int)pKF->mvKeysUn.size();
    X.has.resize(pKF->N);
    for (int i = 0; i < pKF->N; i++) X.has[i] = (pKF->NLeft == -1 || i < nUn) && mps[i] && !mps[i]->isBad();
    X.keys = pKF->mvKeysUn;
    X.keys.resize(pKF->N);   // rigs: N counts both images, the padding is never read (has_mappoint = 0 there)
    X.k.n = pKF->N; X.k.keys_un = reinterpret_cast<const orbx_keypoint_t *>(X.keys.data());
    X.k.descriptors = pKF->mDescriptors.data; X.k.has_mappoint = X.has.data();
    X.k.n_nodes = (int32_t)X.id.size();
    X.k.node_id = X.id.data(); X.k.node_start = X.start.data(); X.k.node_idx = X.idx.data();
    X.k.u_right = nullptr; X.k.scale_factors = nullptr; X.k.level_sigma2 = nullptr; X.k.nlevels = 0;
  };
  Flat A, B;
  flatten(pKF1, vpMapPoints1, A);
  flatten(pKF2, vpMapPoints2, B);
  std::vector<int32_t> m12(pKF1->N, -1);
  const int n = orbm_search_by_bow_keyframes(matcher(), &A.k, &B.k, mfNNratio, mbCheckOrientation ? 1 : 0, m12.data());
  if (n < 0) throw std::runtime_error(orbm_last_error(matcher()));
  for (int i = 0; i < pKF1->N; i++)
    if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];  // :927
  return n;
}

int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs,
                                       const bool bOnlyStereo, const bool bCoarse) {
  if (pKF1->mpCamera2 || pKF2->mpCamera2 || pKF1->mpCamera->GetType() != 0 || pKF2->mpCamera->GetType() != 0)
    return SearchForTriangulation_ref(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse);  // fisheye / two-camera rigs: reference path
  struct Flat {
    std::vector<uint8_t> has;
    std::vector<uint32_t> id;
    std::vector<int32_t> start, idx;
    orbm_keyframe_t k;
  };
  auto flatten = [](KeyFrame *pKF, Flat &F) {
    F.has.resize(pKF->N);
    for (int i = 0; i < pKF->N; i++) F.has[i] = pKF->GetMapPoint(i) != NULL;
    F.start.push_back(0);
    for (DBoW2::FeatureVector::const_iterator it = pKF->mFeatVec.begin(); it != pKF->mFeatVec.end(); ++it) {  // std::map: ascending ids
      F.id.push_back(it->first);
      for (unsigned v : it->second) F.idx.push_back((int32_t)v);
      F.start.push_back((int32_t)F.idx.size());
    }
    F.k.n = pKF->N;
    F.k.keys_un = reinterpret_cast<const orbx_keypoint_t *>(pKF->mvKeysUn.data());
    F.k.descriptors = pKF->mDescriptors.data;
    F.k.u_right = pKF->mvuRight.data();
    F.k.has_mappoint = F.has.data();
    F.k.n_nodes = (int32_t)F.id.size();
    F.k.node_id = F.id.data(); F.k.node_start = F.start.data(); F.k.node_idx = F.idx.data();
    F.k.scale_factors = pKF->mvScaleFactors.data(); F.k.level_sigma2 = pKF->mvLevelSigma2.data();
    F.k.nlevels = (int32_t)pKF->mvScaleFactors.size();
  };
  Flat A, B;
  flatten(pKF1, A);
  flatten(pKF2, B);
  auto m33 = [](const cv::Mat &M, float *o) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o[3 * r + c] = M.at<float>(r, c); };
  auto v3 = [](const cv::Mat &M, float *o) { for (int r = 0; r < 3; r++) o[r] = M.at<float>(r); };
  float R1w[9], R2w[9], t1w[3], t2w[3], Cw[3], cam1[4], cam2[4];
  m33(pKF1->GetRotation(), R1w); m33(pKF2->GetRotation(), R2w);
  v3(pKF1->GetTranslation(), t1w); v3(pKF2->GetTranslation(), t2w); v3(pKF1->GetCameraCenter(), Cw);
  for (int k = 0; k < 4; k++) { cam1[k] = pKF1->mpCamera->getParameter(k); cam2[k] = pKF2->mpCamera->getParameter(k); }
  std::vector<int32_t> m12(pKF1->N, -1);
  const int n = orbm_search_for_triangulation(matcher(), &A.k, &B.k, R1w, t1w, R2w, t2w, Cw, cam1, cam2, bOnlyStereo, bCoarse,
                                              mbCheckOrientation ? 1 : 0, m12.data());
  if (n < 0) throw std::runtime_error(orbm_last_error(matcher()));
  vMatchedPairs.clear();
  vMatchedPairs.reserve(n);
  for (size_t i = 0; i < m12.size(); i++)
    if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));  // :1211-1219
  return n;
}

}  // namespace ORB_SLAM3
