/*
 * orb_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the reference's ORB front end:
 *   /root/reference/src/ORBextractor.cc  (extractor, X0..X9 of SURVEY.md section 8a)
 *   /root/reference/src/ORBmatcher.cc    (M1, M2, M3, M4, M7)
 *   /root/reference/src/Frame.cc         (G1 grid: AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea)
 *   /root/reference/src/CameraModels/{Pinhole,KannalaBrandt8}.cpp (C1, C2)
 * plus the OpenCV 3.4.x primitives those files call (resize, FAST, GaussianBlur,
 * fastAtan2, cvRound), restated from their published algorithms because OpenCV is an
 * external, un-vendored dependency of the reference (CMakeLists.txt:43-56).
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this
 * path, and neither it nor OpenCV can be built in this image, so this restatement is
 * the oracle of record (see DESIGN.md "Oracle").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The shipped library (liborbhip.so) never links, loads or calls anything in oracle/.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::KeyPoint layout, 28 bytes. */
typedef struct {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} orc_keypoint;

#define ORC_MAX_LEVELS 16

/* ORBextractor constructor state, ORBextractor.cc:408-468. */
typedef struct {
  int nfeatures, nlevels, iniThFAST, minThFAST;
  double scaleFactor;
  float mvScaleFactor[ORC_MAX_LEVELS], mvInvScaleFactor[ORC_MAX_LEVELS];
  float mvLevelSigma2[ORC_MAX_LEVELS], mvInvLevelSigma2[ORC_MAX_LEVELS];
  int mnFeaturesPerLevel[ORC_MAX_LEVELS];
  int umax[16];
} orc_extractor;

void orc_extractor_init(orc_extractor *e, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh);

/* Level geometry, ORBextractor.cc:1192-1193. */
void orc_level_size(const orc_extractor *e, int level, int cols, int rows, int *lcols, int *lrows);

/* --- OpenCV primitives ------------------------------------ */
int orc_cvRound(double v);
/* cv::resize INTER_LINEAR 8UC1 (A.3). */
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, size_t sstride, uint8_t *dst, int dw, int dh, size_t dstride);
/* cv::GaussianBlur 7x7 sigma 2 BORDER_REFLECT_101 8UC1 (A.5). src and dst may not alias. */
void orc_gaussian_blur7(const uint8_t *src, int w, int h, size_t sstride, uint8_t *dst, size_t dstride);
void orc_gauss7_kernel(int k[7]);
/* cv::FAST(img, kps, threshold, true) type 9_16 (A.1). Returns number of keypoints written
 * (x, y, score triplets, raster order); cap = capacity in keypoints. */
int orc_fast9_16(const uint8_t *img, int w, int h, size_t stride, int threshold, int *xys, int cap);
/* FAST corner score for one pixel (cornerScore<16>), for unit tests. */
int orc_fast_corner_score(const uint8_t *p, size_t stride, int threshold);
/* cv::fastAtan2 (A.6). */
float orc_fast_atan2(float y, float x);
/* cv::copyMakeBorder BORDER_REFLECT_101 (A.2). */
void orc_copy_make_border101(const uint8_t *src, int w, int h, size_t sstride, uint8_t *dst, int border, size_t dstride);

/* --- extractor stages ------------------------------------------------------------- */
/* IC_Angle, ORBextractor.cc:75-102. img points at the level image origin. */
float orc_ic_angle(const orc_extractor *e, const uint8_t *img, size_t stride, float ptx, float pty);
/* computeOrbDescriptor, ORBextractor.cc:106-145. */
void orc_compute_descriptor(const uint8_t *blurred, size_t stride, float ptx, float pty, float angle_deg, uint8_t desc[32]);

/* Candidate generation of ComputeKeyPointsOctTree for one level (ORBextractor.cc:771-854):
 * returns count; out[i] = (x, y, response) relative to (minBorderX, minBorderY). */
int orc_level_candidates(const orc_extractor *e, const uint8_t *img, int w, int h, size_t stride, float *xyr, int cap);
/* DistributeOctTree, ORBextractor.cc:537-761. in/out are (x,y,response) triplets; returns count. */
int orc_distribute_octtree(const float *xyr, int n, int minX, int maxX, int minY, int maxY, int N, float *out, int cap);

/* Whole pyramid into caller-provided buffers: levels[l] must hold lrows*lcols bytes (tight stride). */
void orc_compute_pyramid(const orc_extractor *e, const uint8_t *img, int cols, int rows, size_t stride, uint8_t **levels);

/* ORBextractor::operator(), ORBextractor.cc:1071-1184.
 * Returns monoIndex, or -1 for an empty image. *n_out = number of keypoints. */
int orc_extract(const orc_extractor *e, const uint8_t *img, int rows, int cols, size_t stride, int lap0, int lap1,
                orc_keypoint *kps, uint8_t *desc, int cap, int *n_out);

/* glibc sinf/cosf as called by the reference (ORBextractor.cc:111), exported so tests can compare
 * the product's replica against the host libm. */
float orc_libm_cosf(float x);
float orc_libm_sinf(float x);
float orc_libm_atanf(float x);
float orc_libm_atan2f(float y, float x);
/* Sweeps of a replica (function pointer) against this host's libm; see orb_oracle.c.  Return the mismatch count. */
long orc_sweep_unary(float (*replica)(float), int which, uint32_t lo, uint32_t hi, uint32_t step, int both_signs);
long orc_sweep_atan2f(float (*replica)(float, float), uint64_t seed, long n, float scale);

/* --- matcher ---------------------------------------------------------------------- */
/* ORBmatcher::DescriptorDistance, ORBmatcher.cc:2463-2483. */
int orc_descriptor_distance(const uint8_t *a, const uint8_t *b);

#define ORC_GRID_COLS 64 /* Frame.h:38 */
#define ORC_GRID_ROWS 48 /* Frame.h:39 */

/* The slice of Frame that the matcher reads (mono / rectified-stereo; Nleft == -1). */
typedef struct {
  int N;
  const float *kx, *ky;       /* mvKeysUn[i].pt */
  const int32_t *octave;      /* mvKeysUn[i].octave */
  const float *angle;         /* mvKeysUn[i].angle */
  const uint8_t *desc;        /* mDescriptors, N x 32 */
  const float *uRight;        /* mvuRight or NULL (treated as all -1) */
  float mnMinX, mnMaxX, mnMinY, mnMaxY;
  float mfGridElementWidthInv, mfGridElementHeightInv;
  const float *mvScaleFactors;
  int nlevels;
  /* grid: CSR in (ix, iy) cell-major order = mGrid[ix][iy] vectors */
  int32_t cell_start[ORC_GRID_COLS * ORC_GRID_ROWS + 1];
  int32_t *cell_idx; /* N entries, owned */
} orc_frame;

/* Frame.cc:379-380 + AssignFeaturesToGrid :434-465 + PosInGrid :815-825. */
void orc_frame_init(orc_frame *f, int N, const float *kx, const float *ky, const int32_t *octave, const float *angle,
                    const uint8_t *desc, const float *uRight, float minX, float maxX, float minY, float maxY,
                    const float *scaleFactors, int nlevels);
void orc_frame_free(orc_frame *f);
/* Frame::GetFeaturesInArea (bRight=false), Frame.cc:744-813. Returns count. */
int orc_get_features_in_area(const orc_frame *f, float x, float y, float r, int minLevel, int maxLevel, int32_t *out);

/* Generic projection search core shared by M2/M3/M4 restatements.
 * slot[i]      : query id occupying keypoint i, or -1 (mvpMapPoints[i] == NULL)
 * slot_obs[i]  : 1 if that occupant has Observations()>0 */
/* M2: SearchByProjection(Frame&, vector<MapPoint*>&, th, ...) ORBmatcher.cc:44-214 (left/mono half).
 * Per query q: in_view[q] (mbTrackInView && !isBad && far-point test), desc, projX/projY, projXR, viewCos,
 * level (mnTrackScaleLevel), obs[q] = Observations()>0 of that map point. Returns nmatches. */
int orc_search_by_projection_mp(orc_frame *f, int nq, const uint8_t *in_view, const uint8_t *qdesc,
                                const float *projX, const float *projY, const float *projXR,
                                const float *viewCos, const int32_t *level, const uint8_t *qobs,
                                float th, float nnratio, int32_t *slot, uint8_t *slot_obs, int32_t *match_of_query);

/* Generic windowed variant used for the BASELINE config-3 stress case and by tests: per-query radius and
 * level window given explicitly (GetFeaturesInArea arguments), otherwise M2 semantics. */
int orc_search_by_projection_win(orc_frame *f, int nq, const uint8_t *in_view, const uint8_t *qdesc,
                                 const float *u, const float *v, const float *radius, const int32_t *minLevel,
                                 const int32_t *maxLevel, const uint8_t *qobs, float nnratio, int th_high,
                                 int mode_second, int32_t *slot, uint8_t *slot_obs, int32_t *match_of_query,
                                 int32_t *best_dist_out);

/* Camera models. type 0 = Pinhole (Pinhole.cpp:46-49), 1 = KannalaBrandt8 (KannalaBrandt8.cpp:29-45). */
void orc_project(int type, const float *params, float X, float Y, float Z, float *u, float *v);

/* M3: SearchByProjection(Frame &Cur, const Frame &Last, th, bMono), ORBmatcher.cc:2027-2289, Nleft==-1 path.
 * Last-frame side flattened: for i in 0..nLast: has_mp[i] (mvpMapPoints[i] && !mvbOutlier[i]), world pos Xw[3i..],
 * descriptor of the map point, last octave, last angle, obs flag. Tcw row-major 4x4 (only 3x4 used).
 * Returns nmatches (after rotation-histogram pruning when checkOri). */
int orc_search_by_projection_ff(orc_frame *cur, int nLast, const uint8_t *has_mp, const float *Xw, const uint8_t *mpdesc,
                                const int32_t *lastOctave, const float *lastAngle, const uint8_t *qobs,
                                const float *Tcw, const float *Tlw, int camType, const float *camParams, float mb, float mbf,
                                float th, int bMono, int checkOri, int32_t *slot, uint8_t *slot_obs);

/* M2 / M3 with a fisheye-stereo frame (Nleft != -1): left grid over mvKeys, right grid over mvKeysRight, one slot
 * array of Nleft + Nright entries (F.mvpMapPoints).  ORBmatcher.cc:44-214 resp. :2027-2289 complete. */
int orc_search_by_projection_mp_fisheye(orc_frame *fl, orc_frame *fr, const int32_t *leftToRight,
                                        const int32_t *rightToLeft, int nmp, const uint8_t *in_view,
                                        const uint8_t *in_view_r, const uint8_t *qdesc, const float *projX,
                                        const float *projY, const float *viewCos, const int32_t *level,
                                        const float *projXR, const float *projYR, const float *viewCosR,
                                        const int32_t *levelR, const uint8_t *qobs, float th, float nnratio,
                                        int32_t *slot, uint8_t *slot_obs, int32_t *match_left, int32_t *match_right);
int orc_search_by_projection_ff_fisheye(orc_frame *cl, orc_frame *cr, int nLast, const uint8_t *has_mp, const float *Xw,
                                        const uint8_t *mpdesc, const int32_t *lastOctave, const float *lastAngle,
                                        const uint8_t *qobs, const float *Tcw, const float *Tlw, const float *Trl,
                                        int camType, const float *camParams, float mb, float th, int bMono, int checkOri,
                                        int32_t *slot, uint8_t *slot_obs);

/* N1: Frame::ComputeStereoMatches, Frame.cc:901-1079. */
void orc_compute_stereo_matches(const orc_extractor *e, uint8_t *const *levelsL, uint8_t *const *levelsR, int cols, int rows,
                                int N, const float *kxL, const float *kyL, const int32_t *octL, const uint8_t *descL,
                                int Nr, const float *kxR, const float *kyR, const int32_t *octR, const uint8_t *descR,
                                float mb, float mbf, float *mvuRight, float *mvDepth);

/* N2: ORBmatcher::SearchForInitialization, ORBmatcher.cc:722-837. */
int orc_search_for_initialization(int n1, const int32_t *octave1, const float *angle1, const uint8_t *desc1, orc_frame *F2,
                                  float *prevMatched, int windowSize, float nnratio, int checkOri, int32_t *vnMatches12);

/* The slice of KeyFrame that SearchForTriangulation reads. */
typedef struct {
  int N;
  const float *kx, *ky;          /* mvKeysUn[i].pt */
  const int32_t *octave;
  const float *angle;
  const uint8_t *desc;
  const float *uRight;           /* mvuRight (never NULL here; -1 = mono) */
  const uint8_t *has_mp;         /* GetMapPoint(i) != NULL */
  int n_nodes;                   /* DBoW2::FeatureVector: nodes ascending by id */
  const uint32_t *node_id;
  const int32_t *node_start;     /* n_nodes + 1 */
  const int32_t *node_idx;
  const float *scaleFactors, *levelSigma2;
} orc_keyframe;

/* N3 (first member): SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&), ORBmatcher.cc:273-469 (Nleft == -1). */
int orc_search_by_bow_kf_frame(const orc_keyframe *KF, const orc_keyframe *F, float nnratio, int checkOri, int32_t *matchF);
int orc_search_by_bow_kf_frame_stereo(const orc_keyframe *KF, const orc_keyframe *F, int Nleft, float nnratio, int checkOri, int32_t *matchF);
/* N3: the search part of the two ORBmatcher::Fuse overloads, ORBmatcher.cc:1425-1658 and :1660-1786. */
int orc_fuse(orc_frame *kf, int nP, const uint8_t *valid, const float *Xw, const float *normal, const uint8_t *mpdesc,
             const float *maxDist, const float *minDist, const float *Tcw, const float *Ow, int camType, const float *cam, float bf,
             const float *invLevelSigma2, float logScaleFactor, float th, int32_t *bestIdx, int32_t *bestDist);
int orc_fuse_sim3(orc_frame *kf, int nP, const uint8_t *valid, const float *Xw, const float *normal, const uint8_t *mpdesc,
                  const float *maxDist, const float *minDist, const float *Scw, const float *cam, float logScaleFactor, float th,
                  int32_t *bestIdx, int32_t *bestDist);
/* N3: MapPoint::ComputeDistinctiveDescriptors, MapPoint.cc:350-436 (BestIdx of one group). */
int orc_distinctive_descriptor(const uint8_t *desc, int N);
/* N3: ORBmatcher::SearchBySim3, ORBmatcher.cc:1788-2012. */
int orc_search_by_sim3(orc_frame *kf1, float logSf1, const uint8_t *valid1, const float *Xw1, const uint8_t *mpdesc1,
                       const float *maxDist1, const float *minDist1, const float *R1w, const float *t1w, orc_frame *kf2, float logSf2,
                       const uint8_t *valid2, const float *Xw2, const uint8_t *mpdesc2, const float *maxDist2, const float *minDist2,
                       const float *R2w, const float *t2w, float s12, const float *R12, const float *t12, const float *cam1, float th,
                       int32_t *matches12);
/* N3: SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&), ORBmatcher.cc:839-979. */
int orc_search_by_bow_kf_kf(const orc_keyframe *K1, const orc_keyframe *K2, float nnratio, int checkOri, int32_t *matches12);

/* M6: SearchForTriangulation(KF1, KF2, F12, pairs, bOnlyStereo, bCoarse), ORBmatcher.cc:981-1222, both cameras
 * Pinhole and no second camera (mpCamera2 == NULL).  R?w row-major 3x3, t?w 3, Cw1 = pKF1->GetCameraCenter().
 * cam? = [fx, fy, cx, cy].  matches12[N1] out (vMatches12).  Returns nmatches. */
int orc_search_for_triangulation(const orc_keyframe *k1, const orc_keyframe *k2, const float *R1w, const float *t1w,
                                 const float *R2w, const float *t2w, const float *Cw1, const float *cam1, const float *cam2,
                                 int bOnlyStereo, int bCoarse, int checkOri, int32_t *matches12);
/* the same member around an injected epipolar predicate (the KannalaBrandt8 / rig call sites, ORBmatcher.cc:1148), the candidate lists
 * in front of it, and the Pinhole pieces a test needs to inject the reference's own Pinhole predicate elsewhere */
typedef int (*orc_pair_predicate)(void *user, int idx1, int idx2);
int orc_search_for_triangulation_pred(const orc_keyframe *k1, const orc_keyframe *k2, float epx, float epy, int epipole_gate,
                                      int bOnlyStereo, int bCoarse, int checkOri, orc_pair_predicate pred, void *user, int32_t *vMatches12);
int orc_triangulation_candidates(const orc_keyframe *k1, const orc_keyframe *k2, float epx, float epy, int epipole_gate, int bOnlyStereo,
                                 int32_t *start, int32_t *cidx2, int32_t *cdist, int cap);
int orc_pinhole_epipolar_constrain(const float *F12, float x1, float y1, float x2, float y2, float unc);
void orc_pinhole_pair_geometry(const float *R1w, const float *t1w, const float *R2w, const float *t2w, const float *Cw1, const float *cam1,
                               const float *cam2, float *ep, float *F12);
/* Pinhole::epipolarConstrain inputs: F12 = K1^-T [t12]x R12 K2^-1 (Pinhole.cpp:143-148), exported for tests. */
void orc_pinhole_F12(const float *R12, const float *t12, const float *cam1, const float *cam2, float *F12);

/* M4: SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist),
 * ORBmatcher.cc:2291-2413.  KeyFrame side flattened, i in [0, nKF): valid[i] = pMP && !isBad && !sAlreadyFound.count(pMP),
 * Xw, descriptor, kfAngle[i] = pKF->mvKeysUn[i].angle, maxDist[i]/minDist[i] = mfMaxDistance / mfMinDistance (raw; the
 * 1.2 / 0.8 invariance factors of MapPoint.cc:552-563 are applied here).  logScaleFactor = Frame::mfLogScaleFactor. */
int orc_search_by_projection_kf(orc_frame *cur, int nKF, const uint8_t *valid, const float *Xw, const uint8_t *mpdesc,
                                const float *kfAngle, const float *maxDist, const float *minDist, const float *Tcw,
                                int camType, const float *camParams, float logScaleFactor, float th, int ORBdist,
                                int checkOri, int32_t *slot, uint8_t *slot_obs);

/* M5: SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched,
 * int th, float ratioHamming), ORBmatcher.cc:489-602 (the :604-720 overload only stores one more pointer per match).
 * kf = the keyframe's keypoints/grid; i in [0, nP): valid[i] = !isBad && !spAlreadyFound.count(pMP); Xw, normal (GetNormal()),
 * descriptor, maxDist/minDist raw.  Scw row-major 4x4.  Pinhole camera [fx,fy,cx,cy].  slot = vpMatched (any occupant blocks). */
int orc_search_by_projection_sim3(orc_frame *kf, int nP, const uint8_t *valid, const float *Xw, const float *normal,
                                  const uint8_t *mpdesc, const float *maxDist, const float *minDist, const float *Scw,
                                  const float *cam, float logScaleFactor, int th, float ratioHamming, int32_t *slot,
                                  uint8_t *slot_obs);

/* M5 / Fuse(Sim3) with the keyframe's own camera model (pKF->mpCamera->project): camType as in orc_project. */
int orc_search_by_projection_sim3_cam(orc_frame *kf, int nP, const uint8_t *valid, const float *Xw, const float *normal,
                                      const uint8_t *mpdesc, const float *maxDist, const float *minDist, const float *Scw, int camType,
                                      const float *cam, float logScaleFactor, int th, float ratioHamming, int32_t *slot,
                                      uint8_t *slot_obs);
int orc_fuse_sim3_cam(orc_frame *kf, int nP, const uint8_t *valid, const float *Xw, const float *normal, const uint8_t *mpdesc,
                      const float *maxDist, const float *minDist, const float *Scw, int camType, const float *cam, float logScaleFactor, float th,
                      int32_t *bestIdx, int32_t *bestDist);

/* G0: Frame::UndistortKeyPoints (Frame.cc:837-870) / ComputeImageBounds (:872-899) =
 * cv::undistortPoints(src, dst, K, D=(k1,k2,p1,p2[,k3]), R=I, P=K) restated from SURVEY.md A.9: per point in double,
 * 5 fixed-point iterations, result stored as float.  K = [fx,fy,cx,cy]; D has nD = 4 or 5 coefficients.
 * If D[0] == 0 the reference copies the keypoints unchanged (Frame.cc:839-843). */
void orc_undistort_points(int n, const float *xy_in, const float *K, const float *D, int nD, float *xy_out);
void orc_image_bounds(int cols, int rows, const float *K, const float *D, int nD, float *minX, float *maxX, float *minY, float *maxY);

/* M7: ComputeThreeMaxima, ORBmatcher.cc:2416-2458, on bin sizes. */
void orc_three_maxima(const int *histo_sizes, int L, int *ind1, int *ind2, int *ind3);
/* RadiusByViewingCos, ORBmatcher.cc:216-222. */
float orc_radius_by_viewing_cos(float viewCos);

/* N4: cv::CLAHE::apply (mono_tum_vi.cc:101-109) and cv::remap INTER_LINEAR with float maps (stereo_euroc.cc:166-167), CV_8UC1.
 * OpenCV algorithms restated from the published 3.4/4.x sources: [OPENCV-UNVERIFIED]. */
int orc_clahe(const uint8_t *src, int rows, int cols, size_t sstride, double clipLimit, int tilesX, int tilesY, uint8_t *dst, size_t dstride);
int orc_remap_linear(const uint8_t *src, int srows, int scols, size_t sstride, const float *mapx, const float *mapy, size_t mstride, int rows,
                     int cols, uint8_t *dst, size_t dstride);

#ifdef __cplusplus
}
#endif
#endif
