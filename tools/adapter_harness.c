/*
 * adapter_harness.c -- what one frame costs a caller of the C++ adapters (csrc/adapter/), measured without OpenCV.
 *
 * The adapters cannot be compiled in this image (no OpenCV / Eigen), so this program does EXACTLY their per-call work on
 * malloc'ed buffers shaped like theirs and times it with CLOCK_MONOTONIC around each call, the way the reference times
 * ExtractORB (src/Frame.cc:333-343):
 *
 *   ORBextractor::operator()  (ORBextractor_hip.cc):  orbx_configure (no-op after frame 0), keypoint vector resized to the
 *       capacity, orbx_extract into it and into the descriptor block kept between frames, vector shrunk to n, a fresh n x 32
 *       descriptor matrix (cv::Mat::create) filled row by row; with --fill-pyramid also orbx_download_pyramid into the block
 *       kept between frames and one (pointer, step) header per level.
 *   Frame constructor's share in front of the matcher:  orbm_undistort_keypoints (EuRoC coefficients), bounds once.
 *   ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)  (ORBmatcher_hip.cc:285-334):  per keypoint of the last
 *       frame a pointer chase into a heap-allocated map-point object (has / outlier test, Observations(), world position,
 *       32-byte descriptor copy), keypoint copy with the octave patched, poses as 4 x 4 floats, the slot arrays built from the
 *       current frame's map-point pointers, orbm_search_by_projection_last_frame, the pointer scatter back.
 *
 * Input: a raw file of 752 x 480 8-bit frames (tools/adapter_harness.py writes the synthetic stream and builds / runs this).
 * Output: one JSON line with median / mean / p99 per part and for the frame.
 *
 *   gcc -O2 -I include tools/adapter_harness.c -o build/adapter_harness -L 3_orb_slam3_selfnote_amd -lorbhip -Wl,-rpath,... -lm
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "orbhip.h"

typedef struct { float pos[3]; uint8_t desc[32]; int nobs; int bad; } MapPointObj;   /* what the adapter reads of a MapPoint */

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}
static int cmp_d(const void *a, const void *b) { const double x = *(const double *)a, y = *(const double *)b; return x < y ? -1 : x > y; }
static void stats(double *v, int n, double *med, double *mean, double *p99, double *mx) {
  double s = 0;
  for (int i = 0; i < n; i++) s += v[i];
  qsort(v, (size_t)n, sizeof(double), cmp_d);
  *med = v[n / 2]; *mean = s / n; *p99 = v[(int)(0.99 * (n - 1))]; *mx = v[n - 1];
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s frames.raw nframes [--fill-pyramid] [--shifts file]\n", argv[0]); return 2; }
  const int W = 752, H = 480, nframes = atoi(argv[2]);
  int fill = 0;
  const char *shift_file = NULL;
  for (int i = 3; i < argc; i++) { if (!strcmp(argv[i], "--fill-pyramid")) fill = 1; else if (!strcmp(argv[i], "--shifts") && i + 1 < argc) shift_file = argv[++i]; }
  uint8_t *frames = (uint8_t *)malloc((size_t)W * H * nframes);
  FILE *f = fopen(argv[1], "rb");
  if (!f || fread(frames, (size_t)W * H, (size_t)nframes, f) != (size_t)nframes) { fprintf(stderr, "cannot read %d frames from %s\n", nframes, argv[1]); return 2; }
  fclose(f);
  int32_t *shifts = (int32_t *)calloc((size_t)nframes * 2, sizeof(int32_t));
  if (shift_file) { FILE *g = fopen(shift_file, "rb"); if (!g || fread(shifts, 8, (size_t)nframes, g) != (size_t)nframes) { fprintf(stderr, "cannot read shifts\n"); return 2; } fclose(g); }

  orbx_t *ex = orbx_create(1000, 1.2f, 8, 20, 7, 0);
  orbm_t *mt = orbm_create(0);
  if (!ex || !mt) { fprintf(stderr, "no usable HIP device (there is no CPU fallback)\n"); return 3; }
  float sf[8], isf[8], s2[8], is2[8];
  orbx_get_scale_tables(ex, sf, isf, s2, is2);
  const float K[4] = {458.654f, 457.296f, 367.215f, 248.375f}, D[4] = {-0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f};   /* EuRoC.yaml:9-17 */
  orbm_frame_t cur;
  memset(&cur, 0, sizeof(cur));
  orbm_image_bounds(W, H, K, D, 4, &cur.min_x, &cur.max_x, &cur.min_y, &cur.max_y);

  /* state kept between frames, as the adapters and the Frame objects keep it */
  uint8_t *desc_block = NULL; size_t desc_block_bytes = 0;
  uint8_t *pyr_block = NULL; size_t pyr_block_bytes = 0;
  size_t off[ORBX_MAX_LEVELS], stride[ORBX_MAX_LEVELS];
  struct { uint8_t *p; size_t step; int rows, cols; } pyr_hdr[ORBX_MAX_LEVELS];
  orbx_keypoint_t *last_keys = NULL, *last_keys_un = NULL; uint8_t *last_desc = NULL; int last_n = 0;
  MapPointObj **last_mps = NULL;          /* LastFrame.mvpMapPoints */
  double *t_ex = (double *)malloc(sizeof(double) * nframes), *t_un = (double *)malloc(sizeof(double) * nframes), *t_ma = (double *)malloc(sizeof(double) * nframes),
         *t_fr = (double *)malloc(sizeof(double) * nframes);
  int nt = 0;
  long total_matches = 0, total_kp = 0;
  for (int t = 0; t < nframes; t++) {
    const uint8_t *img = frames + (size_t)t * W * H;
    /* ---- ORBextractor::operator() */
    double t0 = now_ms();
    const int cap = orbx_configure(ex, H, W, 1);
    if (cap < 0) { fprintf(stderr, "configure: %s\n", orbx_last_error(ex)); return 3; }
    orbx_keypoint_t *keys = (orbx_keypoint_t *)malloc(sizeof(orbx_keypoint_t) * (size_t)cap);           /* _keypoints.resize(cap) */
    if (desc_block_bytes < (size_t)cap * 32) { desc_block = (uint8_t *)realloc(desc_block, (size_t)cap * 32); desc_block_bytes = (size_t)cap * 32; }
    int n = 0;
    const int rc = orbx_extract(ex, img, H, W, (size_t)W, 0, 1000, keys, desc_block, cap, &n);
    if (rc < 0) { fprintf(stderr, "extract: %s\n", orbx_last_error(ex)); return 3; }
    keys = (orbx_keypoint_t *)realloc(keys, sizeof(orbx_keypoint_t) * (size_t)(n > 0 ? n : 1));          /* _keypoints.resize(n) */
    uint8_t *desc = (uint8_t *)malloc((size_t)(n > 0 ? n : 1) * 32);                                    /* _descriptors.create(n, 32, CV_8U) */
    for (int i = 0; i < n; i++) memcpy(desc + (size_t)i * 32, desc_block + (size_t)i * 32, 32);
    if (fill) {
      const int need = orbx_download_pyramid(ex, 0, 19, NULL, 0, off, stride);
      if ((size_t)need > pyr_block_bytes) { pyr_block = (uint8_t *)realloc(pyr_block, (size_t)need); pyr_block_bytes = (size_t)need; }
      if (orbx_download_pyramid(ex, 0, 19, pyr_block, pyr_block_bytes, off, stride) < 0) { fprintf(stderr, "pyramid: %s\n", orbx_last_error(ex)); return 3; }
      for (int l = 0; l < 8; l++) { int r, c; orbx_level_info(ex, l, &r, &c); pyr_hdr[l].p = pyr_block + off[l] + 19 * stride[l] + 19; pyr_hdr[l].step = stride[l]; pyr_hdr[l].rows = r; pyr_hdr[l].cols = c; }
    }
    double t1 = now_ms();
    /* ---- Frame::UndistortKeyPoints (Frame.cc:837-870) */
    orbx_keypoint_t *keys_un = (orbx_keypoint_t *)malloc(sizeof(orbx_keypoint_t) * (size_t)(n > 0 ? n : 1));
    orbm_undistort_keypoints(n, keys, K, D, 4, keys_un);
    double t2 = now_ms();
    /* ---- ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) */
    MapPointObj **cur_mps = (MapPointObj **)calloc((size_t)(n > 0 ? n : 1), sizeof(MapPointObj *));      /* CurrentFrame.mvpMapPoints = NULL */
    int nm = 0;
    if (t > 0 && last_n > 0 && n > 0) {
      uint8_t *has = (uint8_t *)calloc((size_t)last_n, 1), *obs = (uint8_t *)calloc((size_t)last_n, 1), *qd = (uint8_t *)malloc((size_t)last_n * 32);
      float *Xw = (float *)calloc((size_t)last_n * 3, sizeof(float));
      orbx_keypoint_t *lk = (orbx_keypoint_t *)malloc(sizeof(orbx_keypoint_t) * (size_t)last_n);
      for (int i = 0; i < last_n; i++) {
        memcpy(&lk[i], &last_keys_un[i], sizeof(orbx_keypoint_t));
        lk[i].octave = last_keys[i].octave;
        MapPointObj *mp = last_mps[i];
        if (!mp || mp->bad) continue;
        has[i] = 1; obs[i] = mp->nobs > 0;
        memcpy(&Xw[(size_t)i * 3], mp->pos, 12);
        memcpy(qd + (size_t)i * 32, mp->desc, 32);
      }
      /* the current pose: the stream's shift as a translation at depth 5 (pure shift scene) */
      const float z = 5.0f;
      float Tcw[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, Tlw[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      Tcw[3] = (float)(shifts[2 * (t - 1)] - shifts[2 * t]) * z / K[0];
      Tcw[7] = (float)(shifts[2 * (t - 1) + 1] - shifts[2 * t + 1]) * z / K[1];
      int32_t *slot = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
      uint8_t *sobs = (uint8_t *)calloc((size_t)n, 1);
      for (int i = 0; i < n; i++) { slot[i] = cur_mps[i] ? (1 << 30) : -1; sobs[i] = cur_mps[i] ? (uint8_t)(cur_mps[i]->nobs > 0) : 0; }   /* slots_of() */
      cur.n = n; cur.keys_un = keys_un; cur.descriptors = desc; cur.u_right = NULL;
      nm = orbm_search_by_projection_last_frame(mt, &cur, sf, 8, last_n, has, Xw, qd, lk, obs, Tcw, Tlw, 0, K, 0.0f, 0.0f, 15.0f, 1, 1, slot, sobs);
      if (nm < 0) { fprintf(stderr, "search: %s\n", orbm_last_error(mt)); return 3; }
      for (int i = 0; i < n; i++) cur_mps[i] = (slot[i] >= 0 && slot[i] < last_n) ? last_mps[slot[i]] : NULL;
      free(has); free(obs); free(qd); free(Xw); free(lk); free(slot); free(sobs);
    }
    double t3 = now_ms();
    if (t >= 20) { t_ex[nt] = t1 - t0; t_un[nt] = t2 - t1; t_ma[nt] = t3 - t2; t_fr[nt] = t3 - t0; nt++; total_matches += nm; total_kp += n; }
    /* ---- untimed: the tracker's own bookkeeping: every keypoint without a map point gets one (un-projected at depth 5) */
    for (int i = 0; i < n; i++) {
      if (cur_mps[i]) { memcpy(cur_mps[i]->desc, desc + (size_t)i * 32, 32); cur_mps[i]->nobs++; continue; }
      MapPointObj *mp = (MapPointObj *)malloc(sizeof(MapPointObj));
      mp->pos[0] = (keys_un[i].x - K[2]) / K[0] * 5.0f; mp->pos[1] = (keys_un[i].y - K[3]) / K[1] * 5.0f; mp->pos[2] = 5.0f;
      /* the scene is static in the IMAGE of the previous pose: keep the map in the current camera's frame (Tlw = I) */
      memcpy(mp->desc, desc + (size_t)i * 32, 32); mp->nobs = 1; mp->bad = 0;
      cur_mps[i] = mp;
    }
    for (int i = 0; i < n; i++) {   /* re-express every point in the current camera frame, so that the next pair's Tlw is the identity again */
      cur_mps[i]->pos[0] = (keys_un[i].x - K[2]) / K[0] * 5.0f; cur_mps[i]->pos[1] = (keys_un[i].y - K[3]) / K[1] * 5.0f; cur_mps[i]->pos[2] = 5.0f;
    }
    free(last_keys); free(last_keys_un); free(last_desc); free(last_mps);   /* (map points themselves leak: a harness) */
    last_keys = keys; last_keys_un = keys_un; last_desc = desc; last_mps = cur_mps; last_n = n;
  }
  double me, mn, p9, mx, ume, umn, up9, umx, mme, mmn, mp9, mmx, fme, fmn, fp9, fmx;
  stats(t_ex, nt, &me, &mn, &p9, &mx); stats(t_un, nt, &ume, &umn, &up9, &umx); stats(t_ma, nt, &mme, &mmn, &mp9, &mmx); stats(t_fr, nt, &fme, &fmn, &fp9, &fmx);
  printf("{\"frames\": %d, \"warmup_frames\": 20, \"fill_pyramid\": %d, \"mean_keypoints\": %.1f, \"mean_matches\": %.1f, "
         "\"extract_ms\": {\"median\": %.4f, \"mean\": %.4f, \"p99\": %.4f, \"max\": %.4f}, "
         "\"undistort_ms\": {\"median\": %.4f, \"mean\": %.4f, \"p99\": %.4f, \"max\": %.4f}, "
         "\"match_ms\": {\"median\": %.4f, \"mean\": %.4f, \"p99\": %.4f, \"max\": %.4f}, "
         "\"frame_ms\": {\"median\": %.4f, \"mean\": %.4f, \"p99\": %.4f, \"max\": %.4f}}\n",
         nt, fill, (double)total_kp / nt, (double)total_matches / nt, me, mn, p9, mx, ume, umn, up9, umx, mme, mmn, mp9, mmx, fme, fmn, fp9, fmx);
  (void)pyr_hdr; (void)isf; (void)s2; (void)is2;
  orbm_destroy(mt); orbx_destroy(ex);
  return 0;
}
