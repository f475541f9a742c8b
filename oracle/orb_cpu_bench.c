/*
 * orb_cpu_bench.c -- CPU-baseline driver over the ORACLE (test infrastructure, NOT the product): BASELINE.md section 3.
 *
 * Times the plain-C restatement of the reference's path (orb_oracle.c) on a frame stream the way the reference times itself:
 * a monotonic clock (std::chrono::steady_clock = CLOCK_MONOTONIC) around each call, as src/Frame.cc:333-343 does around
 * ExtractORB and Examples/Monocular/mono_euroc.cc:105-120 around TrackMonocular; untimed warm-up frames first; per-frame
 * extract and match times returned so that the caller reports median and mean.
 *   threads == 1   the reference's own threading: one extractor instance on one thread (src/Frame.cc:339)
 *   threads  > 1   frame-parallel: one frame per thread from a shared counter (the fair throughput comparison for a batch)
 * The outputs of every frame (keypoints, descriptors, match indices) are returned as well: bench.py compares them byte for
 * byte with what the GPU produced for the same frames inside its timed region ("verified_frames").
 *
 * Two workloads (the `mode` field):
 *   0  BASELINE config 3: every frame extracted and matched against its predecessor with the SearchByProjection core in the
 *      "1000x1000" stress setting (window = whole image, levels open, best/second ratio test, sequential claims)
 *   1  BASELINE config 5: last-frame SearchByProjection (ORBmatcher.cc:2027-2289) with the camera model's projection
 * Pair t = (last = frame t-1, current = frame t); pair 0 wraps to the last frame of the stream.
 */
#define _GNU_SOURCE
#include "orb_oracle.h"
#include <malloc.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
  int nframes;            /* frames in the stream (pair 0 takes frame nframes-1 as its last frame) */
  int count;              /* frames 0 .. count-1 are processed and timed (count <= nframes) */
  int rows, cols;
  int threads, warmup;
  int lap0, lap1, cap;
  int mode;
  float nnratio;
  int th_high;
  /* mode 1 */
  int cam_type;
  const float *cam;       /* 4 (Pinhole) or 8 (KannalaBrandt8) parameters */
  const float *Xw;        /* [nframes][cap][3]: map points of pair t = keypoints of frame t-1 */
  const uint8_t *has_mp;  /* [nframes][cap] */
  const float *Tcw, *Tlw; /* [nframes][16] */
  float th;
  int check_ori;
  float bounds[4];        /* mnMinX, mnMaxX, mnMinY, mnMaxY of the current frame */
  /* mode 0 with distortion (the EuRoC calibration is pinhole + radial-tangential, EuRoC.yaml:9-17): the Frame constructor's
   * UndistortKeyPoints / ComputeImageBounds (Frame.cc:837-899) run between extraction and search, the grid and the search work on
   * mvKeysUn, and the projected features of the last frame are its undistorted keypoints plus the stream's shift */
  int distort;            /* 0: mvKeysUn = mvKeys, bounds = image rectangle */
  float K[4], D[5]; int nD;
} orc_bench_cfg;

typedef struct {
  const orc_extractor *e;
  const orc_bench_cfg *c;
  const uint8_t *frames;
  const int32_t *offs;
  orc_keypoint *kps; uint8_t *desc; int32_t *counts; int32_t *moq; int32_t *nmatch;
  double *ms_extract, *ms_match;
  volatile int next;
  int phase;              /* 0 extract, 1 match */
  int first, last;
} job_t;

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

static void extract_one(job_t *J, int t, int timed) {
  const orc_bench_cfg *c = J->c;
  int n = 0;
  const double t0 = now_ms();
  const int mono = orc_extract(J->e, J->frames + (size_t)t * c->rows * c->cols, c->rows, c->cols, (size_t)c->cols, c->lap0, c->lap1,
                               J->kps + (size_t)t * c->cap, J->desc + (size_t)t * c->cap * 32, c->cap, &n);
  const double t1 = now_ms();
  J->counts[2 * t] = n;
  J->counts[2 * t + 1] = mono;
  if (timed) J->ms_extract[t] = t1 - t0;
}

static void match_one(job_t *J, int t) {
  const orc_bench_cfg *c = J->c;
  const int tl = (t + c->nframes - 1) % c->nframes;
  const int n1 = J->counts[2 * t], n0 = J->counts[2 * tl];
  const orc_keypoint *k1 = J->kps + (size_t)t * c->cap, *k0 = J->kps + (size_t)tl * c->cap;
  const uint8_t *d1 = J->desc + (size_t)t * c->cap * 32, *d0 = J->desc + (size_t)tl * c->cap * 32;
  int32_t *out = J->moq + (size_t)t * c->cap;
  const int m = n0 > n1 ? n0 : n1;
  float *kx = (float *)malloc(sizeof(float) * (size_t)(m + 1) * 6);
  float *ky = kx + (m + 1), *ang = ky + (m + 1), *u = ang + (m + 1), *v = u + (m + 1), *rad = v + (m + 1);
  int32_t *oct = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m + 1) * 4);
  int32_t *lvl = oct + (m + 1), *slot = lvl + (m + 1), *oct0 = slot + (m + 1);
  uint8_t *sobs = (uint8_t *)calloc((size_t)(m + 1), 1);
  const double t0 = now_ms();
  /* the Frame constructor's share that the matcher needs: keypoint SoA + the 64x48 grid (Frame.cc:434-465) */
  for (int i = 0; i < n1; i++) { kx[i] = k1[i].x; ky[i] = k1[i].y; oct[i] = k1[i].octave; ang[i] = k1[i].angle; slot[i] = -1; }
  orc_frame F;
  float *xy = NULL;
  if (c->mode == 0 && c->distort) {   /* Frame.cc:837-870 (current frame), :872-899 */
    xy = (float *)malloc(sizeof(float) * 4 * (size_t)(m + 1));
    for (int i = 0; i < n1; i++) { xy[2 * i] = k1[i].x; xy[2 * i + 1] = k1[i].y; }
    orc_undistort_points(n1, xy, c->K, c->D, c->nD, xy + 2 * (size_t)(m + 1));
    for (int i = 0; i < n1; i++) { kx[i] = xy[2 * (size_t)(m + 1) + 2 * i]; ky[i] = xy[2 * (size_t)(m + 1) + 2 * i + 1]; }
    float b[4];
    orc_image_bounds(c->cols, c->rows, c->K, c->D, c->nD, &b[0], &b[1], &b[2], &b[3]);
    orc_frame_init(&F, n1, kx, ky, oct, ang, d1, NULL, b[0], b[1], b[2], b[3], J->e->mvScaleFactor, J->e->nlevels);
  } else if (c->mode == 0) orc_frame_init(&F, n1, kx, ky, oct, ang, d1, NULL, 0.0f, (float)c->cols, 0.0f, (float)c->rows, J->e->mvScaleFactor, J->e->nlevels);
  else orc_frame_init(&F, n1, kx, ky, oct, ang, d1, NULL, c->bounds[0], c->bounds[1], c->bounds[2], c->bounds[3], J->e->mvScaleFactor, J->e->nlevels);
  int nm = 0;
  if (n1 > 0 && n0 > 0) {
    if (c->mode == 0) {
      const float sx = (float)(J->offs[2 * tl] - J->offs[2 * t]), sy = (float)(J->offs[2 * tl + 1] - J->offs[2 * t + 1]);
      if (c->distort) {   /* the last frame's mvKeysUn (computed by ITS constructor in the reference; timed here with the pair) */
        for (int i = 0; i < n0; i++) { xy[2 * i] = k0[i].x; xy[2 * i + 1] = k0[i].y; }
        orc_undistort_points(n0, xy, c->K, c->D, c->nD, xy + 2 * (size_t)(m + 1));
        for (int i = 0; i < n0; i++) { u[i] = xy[2 * (size_t)(m + 1) + 2 * i] + sx; v[i] = xy[2 * (size_t)(m + 1) + 2 * i + 1] + sy; rad[i] = 1.0e4f; lvl[i] = -1; }
      } else
      for (int i = 0; i < n0; i++) { u[i] = k0[i].x + sx; v[i] = k0[i].y + sy; rad[i] = 1.0e4f; lvl[i] = -1; }
      nm = orc_search_by_projection_win(&F, n0, NULL, d0, u, v, rad, lvl, lvl, NULL, c->nnratio, c->th_high, 1, slot, sobs, out, NULL);
    } else {
      float *a0 = u; /* last-frame angles */
      for (int i = 0; i < n0; i++) { oct0[i] = k0[i].octave; a0[i] = k0[i].angle; }
      nm = orc_search_by_projection_ff(&F, n0, c->has_mp + (size_t)t * c->cap, c->Xw + (size_t)t * c->cap * 3, d0, oct0, a0, NULL,
                                       c->Tcw + (size_t)t * 16, c->Tlw + (size_t)t * 16, c->cam_type, c->cam, 0.0f, 0.0f, c->th, 1, c->check_ori,
                                       slot, sobs);
      for (int i = 0; i < n1; i++) out[i] = slot[i];   /* mode 1 returns the slot array of the current frame */
    }
  }
  orc_frame_free(&F);
  const double t1 = now_ms();
  J->nmatch[t] = nm;
  J->ms_match[t] = t1 - t0;
  free(kx); free(oct); free(sobs); free(xy);
}

static void *worker(void *arg) {
  job_t *J = (job_t *)arg;
  for (;;) {
    const int t = __atomic_fetch_add(&J->next, 1, __ATOMIC_RELAXED);
    if (t >= J->last) break;
    if (J->phase == 0) extract_one(J, t, 1);
    else match_one(J, t);
  }
  return NULL;
}

static double run_phase(job_t *J, int phase, int first, int last, int threads) {
  J->phase = phase; J->first = first; J->last = last;
  J->next = first;
  const double t0 = now_ms();
  if (threads <= 1) worker(J);
  else {
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, worker, J);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    free(th);
  }
  return (now_ms() - t0) * 1e-3;
}

/* Returns 0, or -1 on bad arguments.  orc_search_by_projection_win's in_view / qobs accept NULL (= all 1). */
int orc_bench_stream(const orc_extractor *e, const orc_bench_cfg *c, const uint8_t *frames, const int32_t *offs, orc_keypoint *kps,
                     uint8_t *desc, int32_t *counts, int32_t *moq, int32_t *nmatch, double *ms_extract, double *ms_match, double *wall) {
  if (!e || !c || !frames || !kps || !desc || !counts || !moq || !nmatch || !ms_extract || !ms_match || !wall) return -1;
  if (c->count < 1 || c->count > c->nframes || c->cap < 1 || (c->mode == 0 && !offs)) return -1;
  if (c->mode == 1 && (!c->cam || !c->Xw || !c->has_mp || !c->Tcw || !c->Tlw)) return -1;
  /* the oracle allocates its per-frame scratch (pyramid, candidate lists) with malloc: keep those blocks on the heap instead of
   * mmap/munmap per frame, whose page faults serialise the threads of the frame-parallel run in the kernel */
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  job_t J;
  memset(&J, 0, sizeof(J));
  J.e = e; J.c = c; J.frames = frames; J.offs = offs;
  J.kps = kps; J.desc = desc; J.counts = counts; J.moq = moq; J.nmatch = nmatch; J.ms_extract = ms_extract; J.ms_match = ms_match;
  for (size_t i = 0; i < (size_t)c->nframes * c->cap; i++) moq[i] = -1;
  /* warm-up (untimed): the first `warmup` frames, and the wrap-around predecessor of frame 0 when it is outside the timed range */
  for (int w = 0; w < c->warmup; w++) extract_one(&J, w % c->count, 0);
  if (c->count < c->nframes) extract_one(&J, c->nframes - 1, 0);
  wall[0] = run_phase(&J, 0, 0, c->count, c->threads);
  wall[1] = run_phase(&J, 1, 0, c->count, c->threads);
  return 0;
}
