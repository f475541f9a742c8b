// ORBmatcher_hip.cc -- liborbhip-backed definition of EVERY member of ORB_SLAM3::ORBmatcher (include/ORBmatcher.h:35-108).
//
// Whole-translation-unit replacement, like ORBextractor_hip.cc: in the reference's CMakeLists.txt this file takes the place
// of src/ORBmatcher.cc (line 90); no reference source or header is edited.  It is compiled INSIDE
// the ORB-SLAM3 tree: it needs the reference's Frame.h / KeyFrame.h / MapPoint.h (OpenCV, Eigen, boost, DBoW2), which this
// repository's image lacks; tests/test_adapter_typecheck.py type-checks it against those unmodified headers with
// declaration-only doubles of the third-party headers.
//
// The adapter flattens the live object graph (Frame, KeyFrame, MapPoint) into the arrays the C ABI takes and writes the
// results back into the objects; every descriptor distance, window test and claim runs on the device.  Members and the
// reference lines they replace:
//   ORBmatcher(float, bool), TH_LOW / TH_HIGH / HISTO_LENGTH                                        ORBmatcher.cc:36-42
//   SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)               :44-214   (all frame types)
//   SearchByProjection(Frame &Cur, const Frame &Last, th, bMono)                                    :2027-2289 (all frame types)
//   SearchByProjection(Frame &Cur, KeyFrame*, const set<MapPoint*>&, th, ORBdist)                   :2291-2413 (all frame types)
//   SearchByProjection(KeyFrame*, cv::Mat Scw, vpPoints, vpMatched, th, ratioHamming)               :489-602
//   SearchByProjection(KeyFrame*, cv::Mat Scw, vpPoints, vpPointsKFs, vpMatched, vpMatchedKF, ...)  :604-720
//   SearchByBoW(KeyFrame*, Frame&, ...) / SearchByBoW(KeyFrame*, KeyFrame*, ...)                    :273-469 / :839-979
//   SearchForInitialization                                                                         :722-837
//   SearchForTriangulation(..., bOnlyStereo, bCoarse)                                               :981-1222  (Pinhole keyframes)
//   SearchForTriangulation(..., bOnlyStereo, vMatchedPoints)                                        :1224-1413 (no call site)
//   SearchBySim3                                                                                    :1788-2012
//   Fuse(KeyFrame*, vpMapPoints, th, bRight) / Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint)   :1425-1658 / :1660-1786
//   DescriptorDistance, ComputeThreeMaxima, RadiusByViewingCos                                      :2463-2483, :2416-2458, :216-222
//   CheckDistEpipolarLine, CheckDistEpipolarLine2                                                   :225-267   (no call site)
// SearchForTriangulation between keyframes whose camera is not a Pinhole model or that carry a second camera: the epipolar test is the
// reference's own virtual pCamera1->epipolarConstrain (KannalaBrandt8's triangulates with cv::SVD), called
// as a predicate on the candidate pairs the device delivers (SearchForTriangulationGeneric below); everything else is on the device.
#include "ORBmatcher.h"  // the reference's header, unmodified

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <vector>

#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"
#include "orbhip.h"

namespace ORB_SLAM3 {

const int ORBmatcher::TH_HIGH = 100;     // ORBmatcher.cc:36
const int ORBmatcher::TH_LOW = 50;       // :37
const int ORBmatcher::HISTO_LENGTH = 30; // :38

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}  // :40-42

namespace {
orbm_t *matcher() {  // one handle per thread: Tracking, LocalMapping and LoopClosing match concurrently (System.cc:193-214)
  thread_local orbm_t *m = nullptr;
  if (!m) {
    const char *e = std::getenv("ORBHIP_DEVICE");
    m = orbm_create(e ? std::atoi(e) : 0);
    if (!m) throw std::runtime_error("ORBmatcher: orbm_create failed (no usable HIP device; there is no CPU fallback)");
  }
  return m;
}
int checked(int rc) {
  if (rc < 0) throw std::runtime_error(std::string("ORBmatcher (liborbhip): ") + orbm_last_error(matcher()));
  return rc;
}

// MapPoint::mfMaxDistance / mfMinDistance are protected and only exposed scaled (GetMaxDistanceInvariance() = 1.2f * mfMaxDistance,
// MapPoint.cc:552-563), while MapPoint::PredictScale (MapPoint.cc:570-602) divides the RAW mfMaxDistance.  (1.2f * x) / 1.2f is
// not always x in fp32, so the raw members are read through a pointer-to-member formed in a derived class - no header is edited.
struct MapPointRaw : MapPoint {
  static float MapPoint::*max_distance() { return &MapPointRaw::mfMaxDistance; }
  static float MapPoint::*min_distance() { return &MapPointRaw::mfMinDistance; }
  static std::mutex MapPoint::*mutex_pos() { return &MapPointRaw::mMutexPos; }
};
// both distances under mMutexPos, as the getters read them (MapPoint.cc:552-563): LocalMapping's UpdateNormalAndDepth writes them
inline void raw_distances(MapPoint *p, float &dmax, float &dmin) {
  std::unique_lock<std::mutex> lock(p->*MapPointRaw::mutex_pos());
  dmax = p->*MapPointRaw::max_distance();
  dmin = p->*MapPointRaw::min_distance();
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

// KeyFrame -> orbm_frame_t (mvKeysUn, mDescriptors, mvuRight, KeyFrame::mnMinX.. as IsInImage / GetFeaturesInArea use them)
orbm_frame_t view_of(const KeyFrame *pKF) {
  orbm_frame_t f;
  f.n = (int)pKF->mvKeysUn.size();
  f.keys_un = reinterpret_cast<const orbx_keypoint_t *>(pKF->mvKeysUn.data());
  f.descriptors = pKF->mDescriptors.data;
  f.u_right = pKF->mvuRight.empty() ? nullptr : pKF->mvuRight.data();
  f.min_x = (float)pKF->mnMinX; f.max_x = (float)pKF->mnMaxX; f.min_y = (float)pKF->mnMinY; f.max_y = (float)pKF->mnMaxY;
  return f;
}

// F.mvpMapPoints <-> (slot, slot_obs).  Pre-existing occupants get id 2^30 so they are never confused with a query index.
void slots_of(const Frame &F, std::vector<int32_t> &slot, std::vector<uint8_t> &obs) {
  slot.assign(F.N, -1);
  obs.assign(F.N, 0);
  for (int i = 0; i < F.N; i++)
    if (F.mvpMapPoints[i]) { slot[i] = 1 << 30; obs[i] = F.mvpMapPoints[i]->Observations() > 0; }
}

void camera_of(GeometricCamera *cam, int &type, std::vector<float> &params) {
  type = (int)cam->GetType();                       // GeometricCamera::CAM_PINHOLE = 0, CAM_FISHEYE = 1
  params.resize(type == 0 ? 4 : 8);
  for (size_t k = 0; k < params.size(); k++) params[k] = cam->getParameter((int)k);
}
void mat44(const cv::Mat &M, float *o, int rows = 4) {
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) o[r * 4 + c] = r < rows ? M.at<float>(r, c) : (r == c ? 1.f : 0.f);
}
void mat33(const cv::Mat &M, float *o) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o[3 * r + c] = M.at<float>(r, c); }
void vec3(const cv::Mat &M, float *o) { for (int r = 0; r < 3; r++) o[r] = M.at<float>(r); }

// The map-point side of the Sim3 / Fuse / relocalisation searches, flattened.
struct Points {
  std::vector<uint8_t> valid, desc;
  std::vector<float> Xw, normal, dmax, dmin;
  explicit Points(size_t n) : valid(n, 0), desc(n * 32), Xw(n * 3, 0.f), normal(n * 3, 0.f), dmax(n, 0.f), dmin(n, 0.f) {}
  void set(size_t i, MapPoint *pMP) {
    valid[i] = 1;
    vec3(pMP->GetWorldPos(), &Xw[3 * i]);
    vec3(pMP->GetNormal(), &normal[3 * i]);
    std::memcpy(&desc[32 * i], pMP->GetDescriptor().ptr<uint8_t>(), 32);
    raw_distances(pMP, dmax[i], dmin[i]);
  }
};

// DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) in key order
struct Nodes {
  std::vector<uint32_t> id;
  std::vector<int32_t> start, idx;
  explicit Nodes(const DBoW2::FeatureVector &fv) {
    start.push_back(0);
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
      id.push_back(it->first);
      for (size_t k = 0; k < it->second.size(); k++) idx.push_back((int32_t)it->second[k]);
      start.push_back((int32_t)idx.size());
    }
  }
  void into(orbm_keyframe_t &k) const {
    k.n_nodes = (int32_t)id.size();
    k.node_id = id.data(); k.node_start = start.data(); k.node_idx = idx.data();
  }
};
}  // namespace

int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return orbm_descriptor_distance(a.ptr<uint8_t>(), b.ptr<uint8_t>()); }

float ORBmatcher::RadiusByViewingCos(const float &viewCos) { return orbm_radius_by_viewing_cos(viewCos); }

void ORBmatcher::ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3) {
  std::vector<int> sizes(L);
  for (int i = 0; i < L; i++) sizes[i] = (int)histo[i].size();
  orbm_three_maxima(sizes.data(), L, &ind1, &ind2, &ind3);
}

// ORBmatcher.cc:225-267.  Neither helper has a call site in the reference (dead code there as well); they are defined because
// the class declares them.  Plain fp32 arithmetic as written in the reference, no device work.
bool ORBmatcher::CheckDistEpipolarLine(const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const cv::Mat &F12, const KeyFrame *pKF2, const bool b1) {
  const float a = kp1.pt.x * F12.at<float>(0, 0) + kp1.pt.y * F12.at<float>(1, 0) + F12.at<float>(2, 0);
  const float b = kp1.pt.x * F12.at<float>(0, 1) + kp1.pt.y * F12.at<float>(1, 1) + F12.at<float>(2, 1);
  const float c = kp1.pt.x * F12.at<float>(0, 2) + kp1.pt.y * F12.at<float>(1, 2) + F12.at<float>(2, 2);
  const float num = a * kp2.pt.x + b * kp2.pt.y + c;
  const float den = a * a + b * b;
  if (den == 0) return false;
  const float dsqr = num * num / den;
  if (!b1) return dsqr < 3.84 * pKF2->mvLevelSigma2[kp2.octave];
  return dsqr < 6.63 * pKF2->mvLevelSigma2[kp2.octave];
}
bool ORBmatcher::CheckDistEpipolarLine2(const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const cv::Mat &F12, const KeyFrame *pKF2, const float unc) {
  const float a = kp1.pt.x * F12.at<float>(0, 0) + kp1.pt.y * F12.at<float>(1, 0) + F12.at<float>(2, 0);
  const float b = kp1.pt.x * F12.at<float>(0, 1) + kp1.pt.y * F12.at<float>(1, 1) + F12.at<float>(2, 1);
  const float c = kp1.pt.x * F12.at<float>(0, 2) + kp1.pt.y * F12.at<float>(1, 2) + F12.at<float>(2, 2);
  const float num = a * kp2.pt.x + b * kp2.pt.y + c;
  const float den = a * a + b * b;
  if (den == 0) return false;
  const float dsqr = num * num / den;
  if (unc == 1.f) return dsqr < 3.84 * pKF2->mvLevelSigma2[kp2.octave];
  return dsqr < 3.84 * pKF2->mvLevelSigma2[kp2.octave] * unc;
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
    const int n = checked(orbm_search_by_projection_fisheye(matcher(), &f, F.Nleft, F.mvLeftToRightMatch.data(), F.mvRightToLeftMatch.data(), &qs,
                                                            mfNNratio, TH_HIGH, slot.data(), sobs.data(), moq.data(), nullptr));
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
  const int n = checked(orbm_search_by_projection(matcher(), &f, &qs, mfNNratio, TH_HIGH, 1, slot.data(), sobs.data(), moq.data(), nullptr));
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
    vec3(pMP->GetWorldPos(), &Xw[(size_t)i * 3]);
    std::memcpy(&desc[(size_t)i * 32], pMP->GetDescriptor().ptr<uint8_t>(), 32);
  }
  float Tcw[16], Tlw[16];
  mat44(CurrentFrame.mTcw, Tcw);
  mat44(LastFrame.mTcw, Tlw);
  int camType;
  std::vector<float> params;
  camera_of(CurrentFrame.mpCamera, camType, params);
  std::vector<int32_t> slot;
  std::vector<uint8_t> sobs;
  slots_of(CurrentFrame, slot, sobs);
  std::vector<orbx_keypoint_t> keys;
  const orbm_frame_t f = fisheye ? view_of_fisheye(CurrentFrame, keys) : view_of(CurrentFrame);
  int n;
  if (fisheye) {
    float Trl[16];
    mat44(CurrentFrame.mTrl, Trl, 3);  // :2190
    n = checked(orbm_search_by_projection_last_frame_fisheye(matcher(), &f, CurrentFrame.Nleft, CurrentFrame.mvScaleFactors.data(),
                                                             (int)CurrentFrame.mvScaleFactors.size(), nLast, has.data(), Xw.data(), desc.data(),
                                                             lk.data(), obs.data(), Tcw, Tlw, Trl, camType, params.data(), CurrentFrame.mb, th,
                                                             bMono ? 1 : 0, mbCheckOrientation ? 1 : 0, slot.data(), sobs.data()));
  } else {
    n = checked(orbm_search_by_projection_last_frame(matcher(), &f, CurrentFrame.mvScaleFactors.data(), (int)CurrentFrame.mvScaleFactors.size(),
                                                     nLast, has.data(), Xw.data(), desc.data(), lk.data(), obs.data(), Tcw, Tlw, camType,
                                                     params.data(), CurrentFrame.mb, CurrentFrame.mbf, th, bMono ? 1 : 0,
                                                     mbCheckOrientation ? 1 : 0, slot.data(), sobs.data()));
  }
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
  return checked(orbm_search_for_initialization(matcher(), &f1, &f2, reinterpret_cast<float *>(vbPrevMatched.data()), windowSize, mfNNratio,
                                                mbCheckOrientation ? 1 : 0, vnMatches12.data()));
}

int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th,
                                   const int ORBdist) {
  const std::vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();
  const int nKF = (int)vpMPs.size();
  std::vector<uint8_t> valid(nKF, 0), desc((size_t)nKF * 32);
  std::vector<float> Xw((size_t)nKF * 3, 0.f), ang(nKF, 0.f), dmax(nKF, 0.f), dmin(nKF, 0.f);
  for (int i = 0; i < nKF; i++) {
    MapPoint *pMP = vpMPs[i];
    if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;       // :2308-2312
    valid[i] = 1;
    vec3(pMP->GetWorldPos(), &Xw[(size_t)i * 3]);
    std::memcpy(&desc[(size_t)i * 32], pMP->GetDescriptor().ptr<uint8_t>(), 32);
    ang[i] = pKF->mvKeysUn[i].angle;                                      // :2378
    raw_distances(pMP, dmax[i], dmin[i]);                                 // the library applies the 1.2f / 0.8f of the getters (:2327-2328)
  }
  float Tcw[16];
  mat44(CurrentFrame.mTcw, Tcw);
  int camType;
  std::vector<float> params;
  camera_of(CurrentFrame.mpCamera, camType, params);
  // A fisheye-stereo frame (Nleft != -1) takes part with its LEFT image only: GetFeaturesInArea(..., bRight = false) walks mGrid
  // over mvKeys (Frame.cc:781-793) and mvKeysUn == mvKeys there (Frame.cc:839-843, :1211), so the same device path applies to
  // the first Nleft keypoints / descriptor rows / slots.
  orbm_frame_t f = view_of(CurrentFrame);
  const int n_cur = CurrentFrame.Nleft != -1 ? CurrentFrame.Nleft : CurrentFrame.N;
  f.n = n_cur;
  if (CurrentFrame.Nleft != -1) { f.keys_un = reinterpret_cast<const orbx_keypoint_t *>(CurrentFrame.mvKeys.data()); f.u_right = nullptr; }
  std::vector<int32_t> slot(n_cur, -1);
  std::vector<uint8_t> sobs(n_cur, 0);
  for (int i = 0; i < n_cur; i++)
    if (CurrentFrame.mvpMapPoints[i]) { slot[i] = 1 << 30; sobs[i] = 1; }  // any occupant blocks, :2355-2356
  const int n = checked(orbm_search_by_projection_keyframe(matcher(), &f, CurrentFrame.mvScaleFactors.data(), (int)CurrentFrame.mvScaleFactors.size(),
                                                           CurrentFrame.mfLogScaleFactor, nKF, valid.data(), Xw.data(), desc.data(), ang.data(),
                                                           dmax.data(), dmin.data(), Tcw, camType, params.data(), th, ORBdist,
                                                           mbCheckOrientation ? 1 : 0, slot.data(), sobs.data()));
  for (int i = 0; i < n_cur; i++)
    if (slot[i] >= 0 && slot[i] < nKF) CurrentFrame.mvpMapPoints[i] = vpMPs[slot[i]];  // :2373 (matches pruned by the rotation check never appear)
  return n;
}

// Shared by the two Sim3 overloads (:489-602, :604-720): the second only stores the keyframe of each matched point as well.
static int sim3_projection(orbm_t *m, KeyFrame *pKF, const cv::Mat &Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th,
                           float ratioHamming, std::vector<int32_t> &slot) {
  const int nP = (int)vpPoints.size();
  std::set<MapPoint *> spAlreadyFound(vpMatched.begin(), vpMatched.end());   // :506-507
  spAlreadyFound.erase(static_cast<MapPoint *>(NULL));
  Points P(nP);
  for (int i = 0; i < nP; i++) {
    MapPoint *pMP = vpPoints[i];
    if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;                 // :516-517
    P.set(i, pMP);
  }
  float S[16];
  mat44(Scw, S);
  int camType;
  std::vector<float> params;
  camera_of(pKF->mpCamera, camType, params);
  const orbm_frame_t f = view_of(pKF);
  slot.assign(f.n, -1);
  std::vector<uint8_t> sobs(f.n, 0);
  for (int i = 0; i < f.n && i < (int)vpMatched.size(); i++)
    if (vpMatched[i]) { slot[i] = 1 << 30; sobs[i] = 1; }                    // :577-578: any occupant blocks
  const int n = orbm_search_by_projection_sim3_cam(m, &f, pKF->mvScaleFactors.data(), (int)pKF->mvScaleFactors.size(), pKF->mfLogScaleFactor, nP,
                                                   P.valid.data(), P.Xw.data(), P.normal.data(), P.desc.data(), P.dmax.data(), P.dmin.data(), S, camType,
                                                   params.data(), th, ratioHamming, slot.data(), sobs.data());
  return n;
}

int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th,
                                   float ratioHamming) {
  std::vector<int32_t> slot;
  const int n = checked(sim3_projection(matcher(), pKF, Scw, vpPoints, vpMatched, th, ratioHamming, slot));
  for (size_t i = 0; i < slot.size(); i++)
    if (slot[i] >= 0 && slot[i] < (int)vpPoints.size()) vpMatched[i] = vpPoints[slot[i]];            // :595
  return n;
}

int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, const std::vector<KeyFrame *> &vpPointsKFs,
                                   std::vector<MapPoint *> &vpMatched, std::vector<KeyFrame *> &vpMatchedKF, int th, float ratioHamming) {
  std::vector<int32_t> slot;
  const int n = checked(sim3_projection(matcher(), pKF, Scw, vpPoints, vpMatched, th, ratioHamming, slot));
  for (size_t i = 0; i < slot.size(); i++)
    if (slot[i] >= 0 && slot[i] < (int)vpPoints.size()) { vpMatched[i] = vpPoints[slot[i]]; vpMatchedKF[i] = vpPointsKFs[slot[i]]; }  // :712-713
  return n;
}

int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches) {
  // Two-camera (fisheye) rigs: keypoint k of a view is mvKeys[k] for k < Nleft and mvKeysRight[k - Nleft] after it (:383-395);
  // the C ABI takes that concatenation.  Frames and keyframes of one system come from the same sensor configuration.
  if ((F.Nleft != -1) != (pKF->mpCamera2 != nullptr)) throw std::logic_error("ORBmatcher::SearchByBoW: frame and keyframe from different camera rigs");
  const std::vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();
  vpMapPointMatches.assign(F.N, static_cast<MapPoint *>(NULL));  // :277
  const Nodes nA(pKF->mFeatVec), nB(F.mFeatVec);
  orbm_keyframe_t A, B;
  std::memset(&A, 0, sizeof(A)); std::memset(&B, 0, sizeof(B));
  nA.into(A); nB.into(B);
  std::vector<uint8_t> hasA(pKF->N), hasB(F.N, 0);
  for (int i = 0; i < pKF->N; i++) hasA[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();  // :307-313
  std::vector<cv::KeyPoint> keysKF, keysF;
  const bool rig = F.Nleft != -1;
  if (rig) {
    keysKF = pKF->mvKeys; keysKF.insert(keysKF.end(), pKF->mvKeysRight.begin(), pKF->mvKeysRight.end());
    keysF = F.mvKeys; keysF.insert(keysF.end(), F.mvKeysRight.begin(), F.mvKeysRight.end());
  }
  A.n = pKF->N; A.keys_un = reinterpret_cast<const orbx_keypoint_t *>(rig ? keysKF.data() : pKF->mvKeysUn.data());
  A.descriptors = pKF->mDescriptors.data; A.has_mappoint = hasA.data();
  B.n = F.N; B.keys_un = reinterpret_cast<const orbx_keypoint_t *>(rig ? keysF.data() : F.mvKeys.data());  // angle of F.mvKeys, :395
  B.descriptors = F.mDescriptors.data; B.has_mappoint = hasB.data();
  std::vector<int32_t> mF(F.N, -1);
  const int n = checked(rig ? orbm_search_by_bow_fisheye(matcher(), &A, &B, F.Nleft, mfNNratio, mbCheckOrientation ? 1 : 0, mF.data())
                            : orbm_search_by_bow(matcher(), &A, &B, mfNNratio, mbCheckOrientation ? 1 : 0, mF.data()));
  for (int i = 0; i < F.N; i++)
    if (mF[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[mF[i]];  // :389
  return n;
}

int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12) {
  // Two-camera rigs: the reference skips every index >= mvKeysUn.size(), i.e. the right image's keypoints (:874-876, :894-896);
  // clearing has_mappoint for them has the same effect (`if(!pMP) continue`, :879 / :900).
  const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
  vpMatches12.assign(vpMapPoints1.size(), static_cast<MapPoint *>(NULL));  // :852
  struct Side {
    Nodes nodes; std::vector<uint8_t> has; std::vector<cv::KeyPoint> keys; orbm_keyframe_t k;
    Side(KeyFrame *pKF, const std::vector<MapPoint *> &mps) : nodes(pKF->mFeatVec), has(pKF->N), keys(pKF->mvKeysUn) {
      std::memset(&k, 0, sizeof(k));
      const int nUn = (int)pKF->mvKeysUn.size();
      for (int i = 0; i < pKF->N; i++) has[i] = (pKF->NLeft == -1 || i < nUn) && mps[i] && !mps[i]->isBad();
      keys.resize(pKF->N);   // rigs: N counts both images, the padding is never read (has_mappoint = 0 there)
      nodes.into(k);
      k.n = pKF->N; k.keys_un = reinterpret_cast<const orbx_keypoint_t *>(keys.data());
      k.descriptors = pKF->mDescriptors.data; k.has_mappoint = has.data();
    }
  };
  Side A(pKF1, vpMapPoints1), B(pKF2, vpMapPoints2);
  std::vector<int32_t> m12(pKF1->N, -1);
  const int n = checked(orbm_search_by_bow_keyframes(matcher(), &A.k, &B.k, mfNNratio, mbCheckOrientation ? 1 : 0, m12.data()));
  for (int i = 0; i < pKF1->N; i++)
    if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];  // :927
  return n;
}

// SearchForTriangulation for KannalaBrandt8 keyframes and two-camera rigs (ORBmatcher.cc:981-1222 with a non-Pinhole
// epipolarConstrain at :1148).  KannalaBrandt8::epipolarConstrain triangulates with cv::SVD and stays the reference's own code
// (SURVEY.md section 2 row 4); everything in front of it - the vocabulary-node walk, the Hamming distances, TH_LOW, the stereo
// filter, the epipole gate - runs on the device (orbm_search_for_triangulation_pred), which then asks this predicate about the
// surviving pairs in the reference's order of preference.
namespace {
struct TriangulationPredicate {
  KeyFrame *pKF1, *pKF2;
  bool rig;
  cv::Mat R12, t12, Rll, Rlr, Rrl, Rrr, tll, tlr, trl, trr;
  static int call(void *user, int idx1, int idx2) {
    TriangulationPredicate *c = static_cast<TriangulationPredicate *>(user);
    KeyFrame *pKF1 = c->pKF1, *pKF2 = c->pKF2;
    const cv::KeyPoint &kp1 = (pKF1->NLeft == -1) ? pKF1->mvKeysUn[idx1] : (idx1 < pKF1->NLeft) ? pKF1->mvKeys[idx1] : pKF1->mvKeysRight[idx1 - pKF1->NLeft];  // :1064-1066
    const cv::KeyPoint &kp2 = (pKF2->NLeft == -1) ? pKF2->mvKeysUn[idx2] : (idx2 < pKF2->NLeft) ? pKF2->mvKeys[idx2] : pKF2->mvKeysRight[idx2 - pKF2->NLeft];  // :1099-1101
    const bool bRight1 = !(pKF1->NLeft == -1 || idx1 < pKF1->NLeft), bRight2 = !(pKF2->NLeft == -1 || idx2 < pKF2->NLeft);
    GeometricCamera *pCamera1 = pKF1->mpCamera, *pCamera2 = pKF2->mpCamera;
    const cv::Mat *R = &c->R12, *t = &c->t12;
    if (pKF1->mpCamera2 && pKF2->mpCamera2) {                                        // :1115-1145
      if (bRight1 && bRight2) { R = &c->Rrr; t = &c->trr; pCamera1 = pKF1->mpCamera2; pCamera2 = pKF2->mpCamera2; }
      else if (bRight1 && !bRight2) { R = &c->Rrl; t = &c->trl; pCamera1 = pKF1->mpCamera2; pCamera2 = pKF2->mpCamera; }
      else if (!bRight1 && bRight2) { R = &c->Rlr; t = &c->tlr; pCamera1 = pKF1->mpCamera; pCamera2 = pKF2->mpCamera2; }
      else { R = &c->Rll; t = &c->tll; }
    }
    return pCamera1->epipolarConstrain(pCamera2, kp1, kp2, *R, *t, pKF1->mvLevelSigma2[kp1.octave], pKF2->mvLevelSigma2[kp2.octave]) ? 1 : 0;  // :1148
  }
};

int SearchForTriangulationGeneric(ORBmatcher *self, KeyFrame *pKF1, KeyFrame *pKF2, std::vector<std::pair<size_t, size_t> > &vMatchedPairs,
                                  const bool bOnlyStereo, const bool bCoarse, const bool checkOri) {
  (void)self;
  // epipole in the second image and the relative poses, exactly as the reference forms them (:988-1023)
  cv::Mat Cw = pKF1->GetCameraCenter();
  cv::Mat R2w = pKF2->GetRotation();
  cv::Mat t2w = pKF2->GetTranslation();
  cv::Mat C2 = R2w * Cw + t2w;
  const cv::Point2f ep = pKF2->mpCamera->project(C2);
  cv::Mat R1w = pKF1->GetRotation();
  cv::Mat t1w = pKF1->GetTranslation();
  TriangulationPredicate P;
  P.pKF1 = pKF1; P.pKF2 = pKF2;
  P.rig = pKF1->mpCamera2 != NULL || pKF2->mpCamera2 != NULL;
  if (!pKF1->mpCamera2 && !pKF2->mpCamera2) {
    P.R12 = R1w * R2w.t();
    P.t12 = -R1w * R2w.t() * t2w + t1w;
  } else {
    P.Rll = pKF1->GetRotation() * pKF2->GetRotation().t();
    P.Rlr = pKF1->GetRotation() * pKF2->GetRightRotation().t();
    P.Rrl = pKF1->GetRightRotation() * pKF2->GetRotation().t();
    P.Rrr = pKF1->GetRightRotation() * pKF2->GetRightRotation().t();
    P.tll = pKF1->GetRotation() * (-pKF2->GetRotation().t() * pKF2->GetTranslation()) + pKF1->GetTranslation();
    P.tlr = pKF1->GetRotation() * (-pKF2->GetRightRotation().t() * pKF2->GetRightTranslation()) + pKF1->GetTranslation();
    P.trl = pKF1->GetRightRotation() * (-pKF2->GetRotation().t() * pKF2->GetTranslation()) + pKF1->GetRightTranslation();
    P.trr = pKF1->GetRightRotation() * (-pKF2->GetRightRotation().t() * pKF2->GetRightTranslation()) + pKF1->GetRightTranslation();
  }
  struct Side {
    Nodes nodes; std::vector<uint8_t> has; std::vector<cv::KeyPoint> keys; std::vector<float> ur; orbm_keyframe_t k;
    explicit Side(KeyFrame *pKF) : nodes(pKF->mFeatVec), has(pKF->N) {
      std::memset(&k, 0, sizeof(k));
      for (int i = 0; i < pKF->N; i++) has[i] = pKF->GetMapPoint(i) != NULL;
      if (pKF->NLeft != -1) { keys = pKF->mvKeys; keys.insert(keys.end(), pKF->mvKeysRight.begin(), pKF->mvKeysRight.end()); }  // :1064-1066
      else keys = pKF->mvKeysUn;
      keys.resize(pKF->N);
      // bStereo = !mpCamera2 && mvuRight[idx] >= 0 (:1059, :1086): always false on a rig
      if (pKF->mpCamera2 || (int)pKF->mvuRight.size() < pKF->N) ur.assign(pKF->N, -1.0f); else ur = pKF->mvuRight;
      nodes.into(k);
      k.n = pKF->N;
      k.keys_un = reinterpret_cast<const orbx_keypoint_t *>(keys.data());
      k.descriptors = pKF->mDescriptors.data;
      k.u_right = ur.data();
      k.has_mappoint = has.data();
      k.scale_factors = pKF->mvScaleFactors.data(); k.level_sigma2 = pKF->mvLevelSigma2.data();
      k.nlevels = (int32_t)pKF->mvScaleFactors.size();
    }
  };
  Side A(pKF1), B(pKF2);
  std::vector<int32_t> m12(pKF1->N > 0 ? pKF1->N : 1, -1);
  const int n = checked(orbm_search_for_triangulation_pred(matcher(), &A.k, &B.k, ep.x, ep.y, pKF1->mpCamera2 ? 0 : 1, bOnlyStereo ? 1 : 0, bCoarse ? 1 : 0,
                                                           checkOri ? 1 : 0, &TriangulationPredicate::call, &P, m12.data()));
  vMatchedPairs.clear();
  vMatchedPairs.reserve(n);
  for (int i = 0; i < pKF1->N; i++)
    if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));  // :1211-1219
  return n;
}
}  // namespace

int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs,
                                       const bool bOnlyStereo, const bool bCoarse) {
  (void)F12;  // unused by the reference as well (it appears only in the signature, :981)
  if (pKF1->mpCamera2 || pKF2->mpCamera2 || pKF1->mpCamera->GetType() != 0 || pKF2->mpCamera->GetType() != 0)
    return SearchForTriangulationGeneric(this, pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse, mbCheckOrientation);
  struct Side {
    Nodes nodes; std::vector<uint8_t> has; orbm_keyframe_t k;
    explicit Side(KeyFrame *pKF) : nodes(pKF->mFeatVec), has(pKF->N) {
      std::memset(&k, 0, sizeof(k));
      for (int i = 0; i < pKF->N; i++) has[i] = pKF->GetMapPoint(i) != NULL;
      nodes.into(k);
      k.n = pKF->N;
      k.keys_un = reinterpret_cast<const orbx_keypoint_t *>(pKF->mvKeysUn.data());
      k.descriptors = pKF->mDescriptors.data;
      k.u_right = pKF->mvuRight.data();
      k.has_mappoint = has.data();
      k.scale_factors = pKF->mvScaleFactors.data(); k.level_sigma2 = pKF->mvLevelSigma2.data();
      k.nlevels = (int32_t)pKF->mvScaleFactors.size();
    }
  };
  Side A(pKF1), B(pKF2);
  float R1w[9], R2w[9], t1w[3], t2w[3], Cw[3], cam1[4], cam2[4];
  mat33(pKF1->GetRotation(), R1w); mat33(pKF2->GetRotation(), R2w);
  vec3(pKF1->GetTranslation(), t1w); vec3(pKF2->GetTranslation(), t2w); vec3(pKF1->GetCameraCenter(), Cw);
  for (int k = 0; k < 4; k++) { cam1[k] = pKF1->mpCamera->getParameter(k); cam2[k] = pKF2->mpCamera->getParameter(k); }
  std::vector<int32_t> m12(pKF1->N, -1);
  const int n = checked(orbm_search_for_triangulation(matcher(), &A.k, &B.k, R1w, t1w, R2w, t2w, Cw, cam1, cam2, bOnlyStereo, bCoarse,
                                                      mbCheckOrientation ? 1 : 0, m12.data()));
  vMatchedPairs.clear();
  vMatchedPairs.reserve(n);
  for (size_t i = 0; i < m12.size(); i++)
    if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));  // :1211-1219
  return n;
}

// ORBmatcher.cc:1224-1413 ("matchAndtriangulate").  No call site in the reference: LocalMapping.cc:583 and Tracking.cc:4288 both
// bind the overload above.  Defined because the class declares it; it has no device path and says so instead of guessing.
int ORBmatcher::SearchForTriangulation(KeyFrame *, KeyFrame *, cv::Mat, std::vector<std::pair<size_t, size_t> > &, const bool, std::vector<cv::Mat> &) {
  throw std::runtime_error("ORBmatcher::SearchForTriangulation(..., vMatchedPoints): this overload has no call site in ORB-SLAM3 and no device path");
}

int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12,
                             const cv::Mat &t12, const float th) {
  const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
  const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
  std::vector<bool> done1(N1, false), done2(N2, false);           // vbAlreadyMatched1 / 2, :1813-1826
  for (int i = 0; i < N1; i++) {
    MapPoint *pMP = vpMatches12[i];
    if (!pMP) continue;
    done1[i] = true;
    const int idx2 = std::get<0>(pMP->GetIndexInKeyFrame(pKF2));
    if (idx2 >= 0 && idx2 < N2) done2[idx2] = true;
  }
  Points P1(N1), P2(N2);
  for (int i = 0; i < N1; i++) { MapPoint *p = vpMapPoints1[i]; if (p && !done1[i] && !p->isBad()) P1.set(i, p); }   // :1835-1842
  for (int i = 0; i < N2; i++) { MapPoint *p = vpMapPoints2[i]; if (p && !done2[i] && !p->isBad()) P2.set(i, p); }   // :1915-1922
  float R1w[9], R2w[9], t1w[3], t2w[3], r12[9], tt12[3];
  mat33(pKF1->GetRotation(), R1w); mat33(pKF2->GetRotation(), R2w);
  vec3(pKF1->GetTranslation(), t1w); vec3(pKF2->GetTranslation(), t2w);
  mat33(R12, r12); vec3(t12, tt12);
  const float cam1[4] = {pKF1->fx, pKF1->fy, pKF1->cx, pKF1->cy};  // :1791-1794: both projections use pKF1's pinhole parameters
  const orbm_frame_t f1 = view_of(pKF1), f2 = view_of(pKF2);
  std::vector<int32_t> m12(N1 > 0 ? N1 : 1, -1);
  const int n = checked(orbm_search_by_sim3(matcher(), &f1, pKF1->mvScaleFactors.data(), (int)pKF1->mvScaleFactors.size(), pKF1->mfLogScaleFactor,
                                            P1.valid.data(), P1.Xw.data(), P1.desc.data(), P1.dmax.data(), P1.dmin.data(), R1w, t1w, &f2,
                                            pKF2->mvScaleFactors.data(), (int)pKF2->mvScaleFactors.size(), pKF2->mfLogScaleFactor, P2.valid.data(),
                                            P2.Xw.data(), P2.desc.data(), P2.dmax.data(), P2.dmin.data(), R2w, t2w, s12, r12, tt12, cam1, th, m12.data()));
  for (int i = 0; i < N1; i++)
    if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];        // :2003
  return n;
}

int ORBmatcher::Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th, const bool bRight) {
  // bRight (:1430-1443): the right camera of a two-camera rig - its pose, its camera model, its keypoints (mvKeysRight, the
  // descriptor rows behind NLeft); the matched index is shifted by NLeft afterwards (:1588).
  cv::Mat Rcw = bRight ? pKF->GetRightRotation() : pKF->GetRotation();
  cv::Mat tcw = bRight ? pKF->GetRightTranslation() : pKF->GetTranslation();
  cv::Mat Ow = bRight ? pKF->GetRightCameraCenter() : pKF->GetCameraCenter();
  GeometricCamera *pCamera = bRight ? pKF->mpCamera2 : pKF->mpCamera;
  const int nMPs = (int)vpMapPoints.size();
  Points P(nMPs);
  for (int i = 0; i < nMPs; i++) {
    MapPoint *pMP = vpMapPoints[i];
    if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;  // :1464-1481
    P.set(i, pMP);
  }
  float T[16] = {0}, ow[3];
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T[4 * r + c] = Rcw.at<float>(r, c); T[4 * r + 3] = tcw.at<float>(r); }
  T[15] = 1.f;
  vec3(Ow, ow);
  int camType;
  std::vector<float> params;
  camera_of(pCamera, camType, params);
  orbm_frame_t f = view_of(pKF);
  const int shift = bRight ? pKF->NLeft : 0;
  if (pKF->NLeft != -1) {  // rigs: GetFeaturesInArea / the level test read the raw keypoints of the chosen image (:1539-1541)
    const std::vector<cv::KeyPoint> &keys = bRight ? pKF->mvKeysRight : pKF->mvKeys;
    f.n = (int)keys.size();
    f.keys_un = reinterpret_cast<const orbx_keypoint_t *>(keys.data());
    f.descriptors = pKF->mDescriptors.data + (size_t)shift * 32;
    f.u_right = pKF->mvuRight.empty() ? nullptr : pKF->mvuRight.data();  // indexed by the unshifted idx in the reference as well (:1549)
  }
  std::vector<int32_t> bi(nMPs > 0 ? nMPs : 1, -1), bd(nMPs > 0 ? nMPs : 1, 256);
  checked(orbm_fuse(matcher(), &f, pKF->mvScaleFactors.data(), pKF->mvInvLevelSigma2.data(), (int)pKF->mvScaleFactors.size(), pKF->mfLogScaleFactor, nMPs,
                    P.valid.data(), P.Xw.data(), P.normal.data(), P.desc.data(), P.dmax.data(), P.dmin.data(), T, ow, camType, params.data(), pKF->mbf, th,
                    bi.data(), bd.data()));
  int nFused = 0;
  for (int i = 0; i < nMPs; i++) {                                 // :1620-1645, on the live objects, in order
    if (bi[i] < 0) continue;
    const int bestIdx = bi[i] + shift;
    MapPoint *pMP = vpMapPoints[i];
    if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;          // a pointer listed twice: the reference's per-iteration test (:1464-1481)
    MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx);
    if (pMPinKF) {
      if (!pMPinKF->isBad()) {
        if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
        else pMPinKF->Replace(pMP);
      }
    } else {
      pMP->AddObservation(pKF, bestIdx);
      pKF->AddMapPoint(pMP, bestIdx);
    }
    nFused++;
  }
  return nFused;
}

int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint) {
  const std::set<MapPoint *> spAlreadyFound = pKF->GetMapPoints();  // :1678
  const int nPoints = (int)vpPoints.size();
  Points P(nPoints);
  for (int i = 0; i < nPoints; i++) {
    MapPoint *pMP = vpPoints[i];
    if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;        // :1689-1690
    P.set(i, pMP);
  }
  float S[16];
  mat44(Scw, S);
  int camType;
  std::vector<float> params;
  camera_of(pKF->mpCamera, camType, params);
  const orbm_frame_t f = view_of(pKF);
  std::vector<int32_t> bi(nPoints > 0 ? nPoints : 1, -1), bd(nPoints > 0 ? nPoints : 1, 256);
  checked(orbm_fuse_sim3_cam(matcher(), &f, pKF->mvScaleFactors.data(), (int)pKF->mvScaleFactors.size(), pKF->mfLogScaleFactor, nPoints, P.valid.data(),
                             P.Xw.data(), P.normal.data(), P.desc.data(), P.dmax.data(), P.dmin.data(), S, camType, params.data(), th, bi.data(), bd.data()));
  int nFused = 0;
  for (int i = 0; i < nPoints; i++) {                               // :1766-1782
    if (bi[i] < 0) continue;
    MapPoint *pMP = vpPoints[i];
    MapPoint *pMPinKF = pKF->GetMapPoint(bi[i]);
    if (pMPinKF) {
      if (!pMPinKF->isBad()) vpReplacePoint[i] = pMPinKF;
    } else {
      pMP->AddObservation(pKF, bi[i]);
      pKF->AddMapPoint(pMP, bi[i]);
    }
    nFused++;
  }
  return nFused;
}

}  // namespace ORB_SLAM3
