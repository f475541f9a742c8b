// orbhip.hip -- liborbhip.so: host side of the C ABI in include/orbhip.h + kernel launches.
//
// Build (see 3_orb_slam3_selfnote_amd/build.py):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared orbhip.hip -o liborbhip.so
// -ffp-contract=off is part of the contract: the reference arithmetic is unfused IEEE.
//
// There is deliberately no CPU fallback in this file: every entry point that computes runs HIP kernels and returns
// ORBX_E_HIP when no device is usable.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <mutex>
#include <chrono>

#include "../../include/orbhip.h"
#include "orb_kernels.h"
#include "orb_match_kernels.h"
#include "orb_match_mfma.h"
#include "orb_project_kernels.h"

static_assert(sizeof(orbx_keypoint_t) == 28, "cv::KeyPoint layout");
static_assert(sizeof(KpOut) == 28, "cv::KeyPoint layout");

namespace {

inline int cv_round(double v) { return (int)lrint(v); }  // cvRound: round-half-even
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#define PROF_DEPTH 32   // batches whose stage events are kept for orbx_get_stage_ms / orbm_get_stage_ms (averaged)

// ORBHIP_TRACE_ALLOC=1: every growth of a device / pinned buffer is reported on stderr with the time its free and its allocation took
// (diagnostic: a re-allocation inside the per-frame path is a latency spike of tens of milliseconds)
static bool trace_alloc() { static const bool on = getenv("ORBHIP_TRACE_ALLOC") != nullptr; return on; }
static double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  hipError_t reserve(size_t need) {
    if (need <= bytes) return hipSuccess;
    const double t0 = trace_alloc() ? wall_ms() : 0.0;
    if (p) (void)hipFree(p);
    const double t1 = trace_alloc() ? wall_ms() : 0.0;
    p = nullptr;
    const size_t old = bytes;
    bytes = 0;
    hipError_t e = hipMalloc(&p, need);
    if (e == hipSuccess) bytes = need;
    if (trace_alloc()) fprintf(stderr, "orbhip-alloc: device buffer %zu -> %zu bytes: hipFree %.3f ms, hipMalloc %.3f ms\n", old, need, t1 - t0, wall_ms() - t1);
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
};

}  // namespace

// --------------------------------------------------------------------------------------------------------------
// extractor handle
// --------------------------------------------------------------------------------------------------------------
struct orbx_handle {
  int device = 0;
  // ORBextractor members (ORBextractor.h:95-109)
  int nfeatures = 0, nlevels = 0, iniThFAST = 0, minThFAST = 0;
  double scaleFactor = 0;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  std::vector<int> mnFeaturesPerLevel;
  int umax[ORB_HALF_PATCH + 1];
  // configured geometry
  int rows = 0, cols = 0, max_batch = 0;
  std::vector<LevelGeom> geom;
  size_t pyr_fs = 0, blur_fs = 0, slot_fs = 0;
  bool octCellsLds = false;
  int chainTiles = 0, chainBuf0 = 0, chainBuf1 = 0;   // k_pyramid_chain (single-frame calls); chainTiles = 0: not available
  size_t chainLds = 0;
  size_t octLds = 0;
  int cell_fs = 0, cand_fs = 0, lkp_fs = 0, totalTiles = 0, totalCells = 0, totalGroups = 0, totalKp = 0, octCap = 0;
  int maxKeypoints = 0;
  // device memory
  DevBuf d_pyr, d_blur, d_cellCnt, d_cellOff, d_slots, d_cand, d_knode, d_lkp, d_lrank, d_lcnt, d_candCnt, d_cells, d_groups, d_tiles, d_xtab,
      d_ytab, d_disc, d_chain;
  DevBuf d_img, d_okps;  // staging for the host entry point (d_okps: counts + keypoints + descriptors, one block)
  hipStream_t stream = nullptr;
  float host_us[4] = {0, 0, 0, 0};          // orbx_extract's last call: staging copy, submission, wait, copy-out (orbx_get_host_us)
  // last call
  FrameParams last{};
  bool have_last = false;
  hipEvent_t last_done = nullptr;   // recorded on the extraction's stream behind its last kernel (asynchronous entry point only)
  bool last_pending = false;        // ... and not yet known to have completed
  bool in_capture = false;          // orbx_extract is capturing its per-frame graph: nothing but the frame's own work is enqueued
  DevBuf stereo[7];  // grow-only buffers of orbx_compute_stereo_matches
  DevBuf maps[2];    // rectification maps of orbx_remap_linear, kept between calls
  int maps_rows = 0, maps_cols = 0;
  // pinned host staging of the single-frame entry point (orbx_extract): pageable copies would serialise on HIP's own staging
  void *pin_in = nullptr, *pin_out = nullptr;
  size_t pin_in_bytes = 0, pin_out_bytes = 0;
  DevBuf d_pyrpack; void *pin_pyr = nullptr; size_t pin_pyr_bytes = 0;   // orbx_download_pyramid: bordered levels, packed
  // the per-frame sequence of orbx_extract (H2D, 5 launches, 1 D2H) captured once per configuration as a hipGraph:
  // one submission per frame instead of 7
  hipGraphExec_t graph = nullptr;
  struct { int rows = 0, cols = 0, lap0 = 0, lap1 = 0, icap = 0; const void *pin_in = nullptr, *pin_out = nullptr, *d_img = nullptr, *d_okps = nullptr, *d_pyr = nullptr; } graph_key;
  bool graph_ok = true;   // cleared (for good) if capture / instantiation fails: the plain path is used instead
  // profiling
  bool profiling = false;
  hipEvent_t ev[PROF_DEPTH][6] = {};   // ring of event sets: one per batch since profiling was switched on
  int prof_head = 0;                   // batches recorded since then
  bool ev_ok = false;
  float stage_ms[5] = {0, 0, 0, 0, 0};
  bool stage_valid = false;
  std::string err;
};

// Synchronous copies go through the handle's own non-blocking stream, never through the legacy default stream: another host
// thread may be capturing its per-frame graph (orbx_extract), and a legacy-stream operation would implicitly join - and
// invalidate - that capture ("operation would make the legacy stream depend on a capturing blocking stream").
static hipError_t copy_on(hipStream_t s, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s);
  return e != hipSuccess ? e : hipStreamSynchronize(s);
}
// waits (on the handle's stream) for the last extraction of this handle, whatever stream it ran on
static hipError_t join_last(orbx_handle *h) {
  if (h->last_pending) { hipError_t e = hipStreamWaitEvent(h->stream, h->last_done, 0); if (e != hipSuccess) return e; }
  return hipStreamSynchronize(h->stream);
}

#define XCHECK(h, call)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
      return ORBX_E_HIP;                                                                      \
    }                                                                                         \
  } while (0)

// The dynamic-LDS limit is an attribute of the kernel FUNCTION (per device), not of a handle or a launch: handles on
// several host threads (Tracking / LocalMapping / LoopClosing each own a matcher, stereo runs two extractors) would race
// a per-launch "set to this launch's size" against each other's launches.  It is raised once per device, to the CU's whole
// 160 KiB, for every kernel that can ask for more than the 64 KiB default; launches only check their own size against it.
#define ORB_LDS_LIMIT (160 * 1024)
static hipError_t raise_lds_limits(int device) {
  static std::mutex mu;
  static bool done[64] = {};
  std::lock_guard<std::mutex> lock(mu);
  if (device < 0 || device >= 64) return hipErrorInvalidDevice;
  if (done[device]) return hipSuccess;
  const void *fns[] = {reinterpret_cast<const void *>(&k_match_walk<Key32, SCAN_PLAIN, 1>), reinterpret_cast<const void *>(&k_match_walk<Key32, SCAN_UR, 1>),
                       reinterpret_cast<const void *>(&k_match_walk<Key32, SCAN_FISHEYE, 1>), reinterpret_cast<const void *>(&k_match_walk<Key32, SCAN_FUSE, 1>),
                       reinterpret_cast<const void *>(&k_match_walk<Key32, SCAN_PLAIN, 4>), reinterpret_cast<const void *>(&k_match_walk<Key32, SCAN_UR, 4>),
                       reinterpret_cast<const void *>(&k_match_walk<Key32, SCAN_FISHEYE, 4>), reinterpret_cast<const void *>(&k_match_walk<Key32, SCAN_FUSE, 4>),
                       reinterpret_cast<const void *>(&k_octree<256, true>), reinterpret_cast<const void *>(&k_octree<256, false>),
                       reinterpret_cast<const void *>(&k_octree<1024, true>), reinterpret_cast<const void *>(&k_octree<1024, false>),
                       reinterpret_cast<const void *>(&k_resize),
                       reinterpret_cast<const void *>(&k_match_resolve<Key32, true>), reinterpret_cast<const void *>(&k_match_resolve<Key32, false>),
                       reinterpret_cast<const void *>(&k_match_resolve<Key64, true>), reinterpret_cast<const void *>(&k_match_resolve<Key64, false>),
                       reinterpret_cast<const void *>(&k_match_resolve<Key32, true, true>), reinterpret_cast<const void *>(&k_pyramid_chain)};
  for (const void *fn : fns) {
    hipFuncAttributes a;
    hipError_t e = hipFuncGetAttributes(&a, fn);
    if (e != hipSuccess) return e;
    const int room = ORB_LDS_LIMIT - (int)a.sharedSizeBytes;   // static __shared__ of the kernel counts against the same 160 KiB
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, room);
    if (e != hipSuccess) return e;
  }
  done[device] = true;
  return hipSuccess;
}

extern "C" {

float orbx_ref_cosf(float x) { return orbsc::ref_cosf(x); }
float orbx_ref_sinf(float x) { return orbsc::ref_sinf(x); }
float orbx_ref_atanf(float x) { return orbat::ref_atanf(x); }
float orbx_ref_atan2f(float y, float x) { return orbat::ref_atan2f(y, x); }

orbx_t *orbx_create(int nfeatures, float scaleFactor_, int nlevels, int iniThFAST, int minThFAST, int device) {
  if (nfeatures < 0 || nlevels < 1 || nlevels > ORBX_MAX_LEVELS || !(scaleFactor_ > 1.0f)) return nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return nullptr;
  if (hipSetDevice(device) != hipSuccess || raise_lds_limits(device) != hipSuccess) return nullptr;
  orbx_handle *h = new orbx_handle();
  h->device = device;
  h->nfeatures = nfeatures;
  h->nlevels = nlevels;
  h->iniThFAST = std::min(std::max(iniThFAST, 0), 255);  // cv::FAST clamps the threshold to [0,255]
  h->minThFAST = std::min(std::max(minThFAST, 0), 255);
  h->scaleFactor = (double)scaleFactor_;  // double member initialised from float, ORBextractor.h:96
  // ORBextractor.cc:413-429
  h->mvScaleFactor.resize(nlevels);
  h->mvLevelSigma2.resize(nlevels);
  h->mvInvScaleFactor.resize(nlevels);
  h->mvInvLevelSigma2.resize(nlevels);
  h->mvScaleFactor[0] = 1.0f;
  h->mvLevelSigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) {
    h->mvScaleFactor[i] = (float)((double)h->mvScaleFactor[i - 1] * h->scaleFactor);
    h->mvLevelSigma2[i] = h->mvScaleFactor[i] * h->mvScaleFactor[i];
  }
  for (int i = 0; i < nlevels; i++) {
    h->mvInvScaleFactor[i] = 1.0f / h->mvScaleFactor[i];
    h->mvInvLevelSigma2[i] = 1.0f / h->mvLevelSigma2[i];
  }
  // per-level quotas, ORBextractor.cc:432-444
  h->mnFeaturesPerLevel.resize(nlevels);
  const float factor = (float)(1.0 / h->scaleFactor);
  float nDesired = (float)nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; l++) {
    h->mnFeaturesPerLevel[l] = cv_round(nDesired);
    sum += h->mnFeaturesPerLevel[l];
    nDesired *= factor;
  }
  h->mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);
  // circular patch row ends, ORBextractor.cc:452-467
  {
    int v, v0;
    const int vmax = (int)floor(ORB_HALF_PATCH * sqrtf(2.f) / 2 + 1);
    const int vmin = (int)ceil(ORB_HALF_PATCH * sqrtf(2.f) / 2);
    const double hp2 = ORB_HALF_PATCH * ORB_HALF_PATCH;
    for (v = 0; v <= ORB_HALF_PATCH; v++) h->umax[v] = 0;
    for (v = 0; v <= vmax; ++v) h->umax[v] = cv_round(sqrt(hp2 - v * v));
    for (v = ORB_HALF_PATCH, v0 = 0; v >= vmin; --v) {
      while (h->umax[v0] == h->umax[v0 + 1]) ++v0;
      h->umax[v] = v0;
      ++v0;
    }
  }
  std::vector<int8_t> disc;
  for (int v = -ORB_HALF_PATCH; v <= ORB_HALF_PATCH; v++) {
    const int d = h->umax[v < 0 ? -v : v];
    for (int u = -d; u <= d; u++) { disc.push_back((int8_t)u); disc.push_back((int8_t)v); }
  }
  if ((int)disc.size() != 2 * ORB_DISC_PIXELS) { delete h; return nullptr; }
  {
    const DiscTab ref = make_disc_tab();  // the kernels' compile-time table must equal what ORBextractor.cc:452-467 derives
    for (int i = 0; i < ORB_DISC_PIXELS; i++) {
      const int lane = i & 63, k = i >> 6;
      const int8_t u = (int8_t)(ref.w[lane][k >> 2] >> (8 * (k & 3))), v = (int8_t)(ref.w[lane][3 + (k >> 2)] >> (8 * (k & 3)));
      if (u != disc[2 * i] || v != disc[2 * i + 1]) { delete h; return nullptr; }
    }
  }
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return nullptr; }
  if (h->d_disc.reserve(disc.size()) != hipSuccess ||
      copy_on(h->stream, h->d_disc.p, disc.data(), disc.size(), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipStreamDestroy(h->stream);
    delete h;
    return nullptr;
  }
  return h;
}

void orbx_destroy(orbx_t *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  DevBuf *bufs[] = {&h->d_pyr, &h->d_blur, &h->d_cellCnt, &h->d_cellOff, &h->d_slots, &h->d_cand, &h->d_knode, &h->d_lkp,
                    &h->d_lrank, &h->d_lcnt, &h->d_candCnt, &h->d_cells, &h->d_groups, &h->d_tiles, &h->d_xtab, &h->d_ytab, &h->d_disc, &h->d_chain, &h->d_img, &h->d_okps};
  for (DevBuf *b : bufs) b->release();
  for (DevBuf &b : h->stereo) b.release();
  for (DevBuf &b : h->maps) b.release();
  if (h->graph) (void)hipGraphExecDestroy(h->graph);
  if (h->last_done) (void)hipEventDestroy(h->last_done);
  if (h->pin_in) (void)hipHostFree(h->pin_in);
  if (h->pin_pyr) (void)hipHostFree(h->pin_pyr);
  h->d_pyrpack.release();
  if (h->pin_out) (void)hipHostFree(h->pin_out);
  if (h->ev_ok)
    for (auto &set : h->ev)
      for (auto &e : set) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

const char *orbx_last_error(const orbx_t *h) { return h ? h->err.c_str() : "null handle"; }
int orbx_get_levels(const orbx_t *h) { return h ? h->nlevels : ORBX_E_ARG; }
float orbx_get_scale_factor(const orbx_t *h) { return h ? (float)h->scaleFactor : 0.f; }

int orbx_get_scale_tables(const orbx_t *h, float *sf, float *isf, float *s2, float *is2) {
  if (!h) return ORBX_E_ARG;
  for (int i = 0; i < h->nlevels; i++) {
    if (sf) sf[i] = h->mvScaleFactor[i];
    if (isf) isf[i] = h->mvInvScaleFactor[i];
    if (s2) s2[i] = h->mvLevelSigma2[i];
    if (is2) is2[i] = h->mvInvLevelSigma2[i];
  }
  return h->nlevels;
}

int orbx_get_features_per_level(const orbx_t *h, int *n) {
  if (!h || !n) return ORBX_E_ARG;
  for (int i = 0; i < h->nlevels; i++) n[i] = h->mnFeaturesPerLevel[i];
  return h->nlevels;
}

// floor(2^32 / n) for xcd_map's division by multiplication (n == 1: the estimate q-1 is corrected on the device)
static uint32_t magic_div(uint32_t n) { return n <= 1 ? 0xffffffffu : (uint32_t)((1ull << 32) / n); }

#if defined(FAST_STAMPS) || defined(OCT_STAMPS)
// diagnostic builds only (tools/fast_stamps.py, tools/octree_stamps.py): device buffer [workgroups][8] of u32 cycle deltas per k_fast section
int orbx_debug_fast_stamps(void *d_buf) {
  unsigned int *p = (unsigned int *)d_buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_fast_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

int orbx_max_keypoints(const orbx_t *h) { return h ? h->maxKeypoints : ORBX_E_ARG; }

int orbx_configure(orbx_t *h, int rows, int cols, int max_batch) {
  if (!h || rows <= 0 || cols <= 0 || max_batch <= 0) return ORBX_E_ARG;
  if (rows > 4096 || cols > 4096) { h->err = "image larger than 4096x4096 (12-bit packed coordinates)"; return ORBX_E_ARG; }
  if (max_batch > 65535) { h->err = "batch larger than 65535"; return ORBX_E_ARG; }
  if (h->rows == rows && h->cols == cols && h->max_batch >= max_batch) return h->maxKeypoints;
  XCHECK(h, hipSetDevice(h->device));
  XCHECK(h, hipStreamSynchronize(h->stream));
  const int nl = h->nlevels;
  std::vector<LevelGeom> g(nl);
  std::vector<int2> xtab, ytab;
  size_t pyr = 0, blur = 0;
  int cells = 0, slots = 0, kps = 0, tiles = 0, octCap = 8;
  for (int l = 0; l < nl; l++) {
    LevelGeom &G = g[l];
    memset(&G, 0, sizeof(G));
    const float inv = h->mvInvScaleFactor[l];
    G.w = cv_round((float)cols * inv);  // ORBextractor.cc:1192-1193
    G.h = cv_round((float)rows * inv);
    if (G.w < 1 || G.h < 1) { h->err = "pyramid level collapses to zero size"; return ORBX_E_ARG; }
    G.pitch = (int)align_up((size_t)G.w, 64);
    G.bpitch = G.pitch;
    G.off = pyr;
    if (l > 0) pyr += align_up((size_t)G.pitch * G.h, 256);
    G.boff = blur;
    blur += align_up((size_t)G.bpitch * G.h, 256);
    // FAST grid, ORBextractor.cc:771-785
    G.maxBorderX = G.w - ORB_EDGE_THRESHOLD + 3;
    G.maxBorderY = G.h - ORB_EDGE_THRESHOLD + 3;
    const float width = (float)(G.maxBorderX - ORB_MIN_BORDER), height = (float)(G.maxBorderY - ORB_MIN_BORDER);
    const float W = 30;
    int nCols = width > 0 ? (int)(width / W) : 0, nRows = height > 0 ? (int)(height / W) : 0;
    if (nCols <= 0 || nRows <= 0) {  // the reference divides by zero here; defined as "no keypoints on this level"
      nCols = nRows = 0;
      G.wCell = G.hCell = 1;
    } else {
      G.wCell = (int)ceilf(width / nCols);
      G.hCell = (int)ceilf(height / nRows);
    }
    G.nCols = nCols;
    G.nRows = nRows;
    G.cellCap = ((G.wCell + 1) / 2) * ((G.hCell + 1) / 2);
    G.cellBase = cells;
    G.slotBase = slots;
    G.candBase = slots;
    G.candCap = nCols * nRows * G.cellCap;
    cells += nCols * nRows;
    slots += G.candCap;
    // octree, ORBextractor.cc:541-543
    G.N = h->mnFeaturesPerLevel[l];
    G.nIni = 0;
    G.hX = 1.f;
    if (nCols > 0) {
      G.nIni = (int)roundf(width / height);
      if (G.nIni < 1) { h->err = "level taller than 2x its width: the reference's DistributeOctTree has no root node"; return ORBX_E_ARG; }
      if (G.nIni > 64) { h->err = "aspect ratio above 64 not supported"; return ORBX_E_ARG; }
      G.hX = width / (float)G.nIni;
    }
    G.kpCap = std::max(G.N + 3, 4 * G.nIni) + 1;
    G.kpBase = kps;
    kps += G.kpCap;
    octCap = std::max(octCap, G.kpCap + 3);
    G.scale = h->mvScaleFactor[l];
    G.kpsize = (float)(int)((float)ORB_PATCH_SIZE * h->mvScaleFactor[l]);  // :862
    // blur tiles
    G.tilesX = (G.w + BLUR_TX - 1) / BLUR_TX;
    G.tilesY = (G.h + BLUR_TY - 1) / BLUR_TY;
    G.tileBase = tiles;
    tiles += G.tilesX * G.tilesY;
    // resize tables, SURVEY.md A.3 (cv::resize -> hal::resize: scale = 1./(dsize/ssize))
    G.xtabBase = (int)xtab.size();
    G.ytabBase = (int)ytab.size();
    if (l > 0) {
      const int sw = g[l - 1].w, sh = g[l - 1].h;
      const double inv_x = (double)G.w / sw, inv_y = (double)G.h / sh;
      const double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
      for (int dx = 0; dx < G.w; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        int a0 = std::min(std::max(cv_round((1.f - fx) * 2048), -32768), 32767);
        int a1 = std::min(std::max(cv_round(fx * 2048), -32768), 32767);
        xtab.push_back(make_int2(sx, (a0 & 0xffff) | (a1 << 16)));
      }
      // k_resize reads the table two columns at a time with 16-byte loads and computes whole output quads: a level's entries start at
      // an even index (xtabBase, below) and are followed by copies of the last one up to a multiple of four columns
      while ((xtab.size() - (size_t)G.xtabBase) % 4) xtab.push_back(xtab.back());
      for (int dy = 0; dy < G.h; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floor(fy);
        fy -= sy;
        int b0 = std::min(std::max(cv_round((1.f - fy) * 2048), -32768), 32767);
        int b1 = std::min(std::max(cv_round(fy * 2048), -32768), 32767);
        ytab.push_back(make_int2(sy, (b0 & 0xffff) | (b1 << 16)));
      }
      // k_resize's tile height: 16 output rows while the LDS stage of a tile (its source rows + their horizontally interpolated rows)
      // stays below 64 KB - two workgroups per CU beside everything else -, else 8
      auto src_rows = [&](int rows) {
        int m = 0;
        for (int dy0 = 0; dy0 < G.h; dy0 += rows) {   // as k_resize derives a tile's source rows
          const int nrows = std::min(rows, G.h - dy0);
          const int f = std::min(std::max(ytab[G.ytabBase + dy0].x, 0), sh - 1), la = std::min(std::max(ytab[G.ytabBase + dy0 + nrows - 1].x + 1, 0), sh - 1);
          m = std::max(m, la - f + 1);
        }
        return m;
      };
      const size_t rowStage = align_up((size_t)sw + 4, 16) + align_up((size_t)((G.w + 3) & ~3) * 2, 8);
      G.resizeRows = RESIZE_ROWS_MAX;
      G.resizeSrcRows = src_rows(G.resizeRows);
      if ((size_t)G.resizeSrcRows * rowStage > 64 * 1024 || G.resizeSrcRows > RESIZE_MAXSRC) { G.resizeRows = 8; G.resizeSrcRows = src_rows(8); }
    } else { G.resizeRows = RESIZE_ROWS_MAX; G.resizeSrcRows = 0; }
  }
  if (slots > (1 << 30)) { h->err = "workspace too large"; return ORBX_E_ARG; }
  // FAST cell records (k_fast): window of every cell as the reference's double loop derives it, ORBextractor.cc:787-803
  std::vector<uint32_t> cellrec((size_t)std::max(cells, 1) * 8, 0u);
  for (int l = 0; l < nl; l++) {
    LevelGeom &G = g[l];
    G.rowTileMagic = magic_div((uint32_t)((G.h + G.resizeRows - 1) / G.resizeRows));
    for (int ci = 0; ci < G.nRows; ci++)
      for (int cj = 0; cj < G.nCols; cj++) {
        const int c = ci * G.nCols + cj;
        uint32_t *R = &cellrec[(size_t)(G.cellBase + c) * 8];
        const int iniX = ORB_MIN_BORDER + cj * G.wCell, iniY = ORB_MIN_BORDER + ci * G.hCell;
        const int maxX = std::min(iniX + G.wCell + 6, G.maxBorderX), maxY = std::min(iniY + G.hCell + 6, G.maxBorderY);
        const int tw = maxX - iniX, th = maxY - iniY;
        const bool valid = !(iniX >= G.maxBorderX - 6 || iniY >= G.maxBorderY - 3 || tw - 6 <= 0 || th - 6 <= 0);
        if (valid && (tw > FAST_TILE_COLS || th > FAST_TILE_ROWS)) { h->err = "FAST cell larger than the LDS tile"; return ORBX_E_ARG; }
        R[0] = (uint32_t)iniX | ((uint32_t)iniY << 16);
        R[1] = valid ? ((uint32_t)tw | ((uint32_t)th << 8) | ((uint32_t)l << 16) | (1u << 24)) : ((uint32_t)l << 16);
        R[2] = (uint32_t)(cj * G.wCell) | ((uint32_t)(ci * G.hCell) << 16);
        R[3] = (uint32_t)(G.slotBase + c * G.cellCap);
        R[4] = (uint32_t)G.cellCap;
        R[5] = (uint32_t)G.pitch;
        R[6] = (uint32_t)(G.off & 0xffffffffu);
        R[7] = (uint32_t)((uint64_t)G.off >> 32);
      }
  }
  // k_fast's workgroups: groups of up to 2 x 2 neighbouring cells of a level that fit one LDS tile (80-byte rows from the first cell's
  // column rounded down to 4, FAST_TILE_ROWS rows); levels with larger cells keep one cell per group in that direction
  std::vector<uint32_t> grouprec;
  for (int l = 0; l < nl; l++) {
    const LevelGeom &G = g[l];
    const int sx = (3 + 2 * G.wCell + 6 <= FAST_TILE_PITCH) ? 2 : 1, sy = (2 * G.hCell + 6 <= FAST_TILE_ROWS) ? 2 : 1;
    for (int ci = 0; ci < G.nRows; ci += sy)
      for (int cj = 0; cj < G.nCols; cj += sx) {
        const int gx = std::min(sx, G.nCols - cj), gy = std::min(sy, G.nRows - ci);
        const int first = G.cellBase + ci * G.nCols + cj;
        const uint32_t *F = &cellrec[(size_t)first * 8];
        const int tileX = (int)(F[0] & 0xffffu), tileY = (int)(F[0] >> 16);
        int maxX = 0, maxY = 0, nvalid = 0;
        for (int a = 0; a < gy; a++)
          for (int b = 0; b < gx; b++) {
            const uint32_t *R = &cellrec[(size_t)(first + a * G.nCols + b) * 8];
            if (!(R[1] >> 24)) continue;
            nvalid++;
            maxX = std::max(maxX, (int)(R[0] & 0xffffu) + (int)(R[1] & 0xffu));
            maxY = std::max(maxY, (int)(R[0] >> 16) + (int)((R[1] >> 8) & 0xffu));
          }
        const int gtw = nvalid ? maxX - tileX : 0, gth = nvalid ? maxY - tileY : 0;
        if (nvalid && ((tileX & 3) + gtw > FAST_TILE_PITCH || gth > FAST_TILE_ROWS)) { h->err = "FAST cell group larger than the LDS tile"; return ORBX_E_ARG; }
        grouprec.push_back((uint32_t)first);
        grouprec.push_back((uint32_t)gx | ((uint32_t)gy << 8) | ((uint32_t)G.nCols << 16));
        grouprec.push_back((uint32_t)tileX | ((uint32_t)tileY << 16));
        grouprec.push_back((uint32_t)gtw | ((uint32_t)gth << 8) | ((uint32_t)l << 16) | (nvalid ? 1u << 24 : 0u));
      }
  }
  h->totalGroups = (int)(grouprec.size() / 4);
  // blur tile records (k_blur)
  std::vector<uint32_t> tilerec((size_t)std::max(tiles, 1) * 8, 0u);
  for (int l = 0; l < nl; l++) {
    const LevelGeom &G = g[l];
    if (G.pitch > 65535 || G.bpitch > 65535) { h->err = "level pitch above 65535"; return ORBX_E_ARG; }
    if (l > 0 && (size_t)((G.w + 3) & ~3) * 8 > ORB_LDS_LIMIT - 1024) { h->err = "pyramid level wider than 20000 columns"; return ORBX_E_ARG; }   // k_resize's x table
    for (int ty = 0; ty < G.tilesY; ty++)
      for (int tx = 0; tx < G.tilesX; tx++) {
        uint32_t *R = &tilerec[(size_t)(G.tileBase + ty * G.tilesX + tx) * 8];
        R[0] = (uint32_t)(tx * BLUR_TX) | ((uint32_t)(ty * BLUR_TY) << 16);
        R[1] = (uint32_t)l;
        R[2] = (uint32_t)G.w | ((uint32_t)G.h << 16);
        R[3] = (uint32_t)G.pitch | ((uint32_t)G.bpitch << 16);
        R[4] = (uint32_t)(G.off & 0xffffffffu); R[5] = (uint32_t)((uint64_t)G.off >> 32);
        R[6] = (uint32_t)(G.boff & 0xffffffffu); R[7] = (uint32_t)((uint64_t)G.boff >> 32);
      }
  }
  h->geom = g;
  h->pyr_fs = std::max<size_t>(pyr, 256);
  h->blur_fs = blur;
  h->cell_fs = std::max(cells, 1);
  h->slot_fs = (size_t)std::max(slots, 1);
  h->cand_fs = std::max(slots, 1);
  h->lkp_fs = kps;
  h->totalTiles = tiles;
  h->totalCells = cells;
  h->totalKp = kps;
  h->octCap = octCap;
  h->maxKeypoints = kps;
  const size_t B = (size_t)max_batch;
  XCHECK(h, h->d_pyr.reserve(h->pyr_fs * B));
  XCHECK(h, h->d_blur.reserve(h->blur_fs * B));
  XCHECK(h, h->d_cellCnt.reserve(sizeof(uint32_t) * h->cell_fs * B));
  XCHECK(h, h->d_cellOff.reserve(sizeof(uint32_t) * h->cell_fs * B));
  XCHECK(h, h->d_slots.reserve(sizeof(uint32_t) * h->slot_fs * B));
  XCHECK(h, h->d_cand.reserve(sizeof(uint32_t) * (size_t)h->cand_fs * B));
  XCHECK(h, h->d_knode.reserve(sizeof(uint16_t) * (size_t)h->cand_fs * B));
  XCHECK(h, h->d_lkp.reserve(sizeof(uint32_t) * (size_t)h->lkp_fs * B));
  XCHECK(h, h->d_lrank.reserve(sizeof(uint16_t) * (size_t)h->lkp_fs * B));
  XCHECK(h, h->d_lcnt.reserve(sizeof(int32_t) * 2 * nl * B));
  XCHECK(h, h->d_candCnt.reserve(sizeof(int32_t) * nl * B));
  XCHECK(h, h->d_cells.reserve(sizeof(uint32_t) * cellrec.size()));
  XCHECK(h, copy_on(h->stream, h->d_cells.p, cellrec.data(), sizeof(uint32_t) * cellrec.size(), hipMemcpyHostToDevice));
  XCHECK(h, h->d_groups.reserve(sizeof(uint32_t) * std::max<size_t>(grouprec.size(), 4)));
  if (!grouprec.empty()) XCHECK(h, copy_on(h->stream, h->d_groups.p, grouprec.data(), sizeof(uint32_t) * grouprec.size(), hipMemcpyHostToDevice));
  XCHECK(h, h->d_tiles.reserve(sizeof(uint32_t) * tilerec.size()));
  XCHECK(h, copy_on(h->stream, h->d_tiles.p, tilerec.data(), sizeof(uint32_t) * tilerec.size(), hipMemcpyHostToDevice));
  XCHECK(h, h->d_xtab.reserve(sizeof(int2) * std::max<size_t>(xtab.size(), 1)));
  XCHECK(h, h->d_ytab.reserve(sizeof(int2) * std::max<size_t>(ytab.size(), 1)));
  if (!xtab.empty()) XCHECK(h, copy_on(h->stream, h->d_xtab.p, xtab.data(), sizeof(int2) * xtab.size(), hipMemcpyHostToDevice));
  if (!ytab.empty()) XCHECK(h, copy_on(h->stream, h->d_ytab.p, ytab.data(), sizeof(int2) * ytab.size(), hipMemcpyHostToDevice));
  // k_pyramid_chain's tile records: for every 32x32 tile of a level L >= 1 the rectangles of levels L-1 .. 0 it depends on,
  // walked down through the resize tables exactly as k_resize indexes them (x: sx, min(sx+1, w-1); y: clamp(sy), clamp(sy+1))
  {
    std::vector<ChainTile> tilesC;
    size_t b0 = 0, b1 = 0, tabMax = 0;
    for (int L = nl - 1; L >= 1; L--)   // highest level first: its workgroups have the longest chains and must not start last
      for (int ty = 0; ty < g[L].h; ty += CHAIN_TILE)
        for (int tx = 0; tx < g[L].w; tx += CHAIN_TILE) {
          ChainTile T;
          memset(&T, 0, sizeof(T));
          T.level = (uint16_t)L;
          int x0 = tx, x1 = std::min(tx + CHAIN_TILE, g[L].w) - 1, y0 = ty, y1 = std::min(ty + CHAIN_TILE, g[L].h) - 1;   // inclusive
          size_t tab = 0;
          for (int l = L; l >= 0; l--) {
            T.x[l] = (uint16_t)x0; T.y[l] = (uint16_t)y0; T.w[l] = (uint16_t)(x1 - x0 + 1); T.h[l] = (uint16_t)(y1 - y0 + 1);
            const size_t area = (size_t)T.w[l] * T.h[l];
            if (l < L) { if (l & 1) b1 = std::max(b1, area); else b0 = std::max(b0, area); }
            if (l == 0) break;
            tab += (size_t)T.w[l] + T.h[l];
            const int sw = g[l - 1].w, sh = g[l - 1].h;
            const int2 *xt = &xtab[g[l].xtabBase], *yt = &ytab[g[l].ytabBase];
            int nx0 = sw, nx1 = 0, ny0 = sh, ny1 = 0;
            for (int x = x0; x <= x1; x++) { nx0 = std::min(nx0, xt[x].x); nx1 = std::max(nx1, std::min(xt[x].x + 1, sw - 1)); }
            for (int y = y0; y <= y1; y++) {
              ny0 = std::min(ny0, std::min(std::max(yt[y].x, 0), sh - 1));
              ny1 = std::max(ny1, std::min(std::max(yt[y].x + 1, 0), sh - 1));
            }
            x0 = nx0; x1 = nx1; y0 = ny0; y1 = ny1;
          }
          tabMax = std::max(tabMax, tab);
          tilesC.push_back(T);
        }
    b0 = align_up(b0, 16); b1 = align_up(b1, 16);
    const size_t lds = b0 + b1 + sizeof(int2) * tabMax;
    // stage sizes stay below 2^14 pixels (the kernel's index split) and the whole state inside 64 KiB; otherwise (scale factors far
    // from 1.2) single-frame calls keep the level-by-level form
    h->chainTiles = 0;
    if (!tilesC.empty() && lds <= 64 * 1024 && b1 <= 16384 && b0 <= 65536 && tilesC.size() < 65536) {
      bool ok = true;
      for (const ChainTile &T : tilesC)
        for (int l = 1; l <= T.level; l++) ok = ok && (size_t)T.w[l] * T.h[l] < 16384;
      if (ok) {
        XCHECK(h, h->d_chain.reserve(sizeof(ChainTile) * tilesC.size()));
        XCHECK(h, copy_on(h->stream, h->d_chain.p, tilesC.data(), sizeof(ChainTile) * tilesC.size(), hipMemcpyHostToDevice));
        h->chainTiles = (int)tilesC.size(); h->chainBuf0 = (int)b0; h->chainBuf1 = (int)b1; h->chainLds = lds;
      }
    }
  }
  // k_octree LDS: node arrays (72 B per node) + scan scratch, plus the level's cell offsets when they fit
  int maxCells = 1;
  for (int l = 0; l < nl; l++) maxCells = std::max(maxCells, g[l].nCols * g[l].nRows);
  size_t lds = 72 * (size_t)octCap + 128;
  if (lds > 160 * 1024 - 256) { h->err = "nfeatures too large for the LDS-resident octree"; return ORBX_E_ARG; }
  h->octCellsLds = lds + 4 * ((size_t)maxCells + 1) <= 96 * 1024;
  if (h->octCellsLds) lds += 4 * ((size_t)maxCells + 1);
  h->octLds = lds;
  if (lds > ORB_LDS_LIMIT) { h->err = "octree LDS state exceeds 160 KiB"; return ORBX_E_ARG; }   // limit raised once in orbx_create
  h->rows = rows;
  h->cols = cols;
  h->max_batch = max_batch;
  h->have_last = false;
  return h->maxKeypoints;
}

void orbx_set_profiling(orbx_t *h, int enable) {
  if (!h) return;
  h->profiling = enable != 0;
  h->prof_head = 0;
  h->stage_valid = false;
  if (h->profiling && !h->ev_ok) {
    (void)hipSetDevice(h->device);
    bool ok = true;
    for (auto &set : h->ev)
      for (auto &e : set) ok = ok && hipEventCreate(&e) == hipSuccess;
    h->ev_ok = ok;
  }
}

// Average per stage over the batches launched since profiling was switched on (the PROF_DEPTH most recent ones).
int orbx_get_stage_ms(orbx_t *h, float *ms, int cap) {
  if (!h || !ms || !h->profiling || !h->ev_ok || !h->stage_valid || h->prof_head <= 0) return 0;
  const int nb = std::min(h->prof_head, PROF_DEPTH), n = std::min(cap, 5);
  if (hipEventSynchronize(h->ev[(h->prof_head - 1) % PROF_DEPTH][5]) != hipSuccess) return 0;
  for (int i = 0; i < n; i++) ms[i] = 0.f;
  for (int b = 0; b < nb; b++) {
    const int slot = (h->prof_head - 1 - b) % PROF_DEPTH;
    for (int i = 0; i < n; i++) {
      float t = 0;
      if (hipEventElapsedTime(&t, h->ev[slot][i], h->ev[slot][i + 1]) != hipSuccess) return 0;
      ms[i] += t / (float)nb;
    }
  }
  return n;
}

int orbx_extract_batch_device(orbx_t *h, const uint8_t *d_images, int rows, int cols, size_t stride, size_t frame_stride,
                              int nframes, int lap0, int lap1, orbx_keypoint_t *d_kps, uint8_t *d_desc, int32_t *d_counts,
                              int cap, void *stream_) {
  if (!h) return ORBX_E_ARG;
  if (!d_images || rows <= 0 || cols <= 0 || nframes <= 0) return ORBX_E_EMPTY;
  if (!d_kps || !d_desc || !d_counts || cap <= 0 || stride < (size_t)cols) return ORBX_E_ARG;
  int rc = orbx_configure(h, rows, cols, std::max(nframes, h->rows == rows && h->cols == cols ? h->max_batch : 1));
  if (rc < 0) return rc;
  // asynchronous: an overflow could not be reported, and d_counts feeds the matcher as a live count bounded by its frame stride,
  // so a capacity below the octree's worst case is refused up front
  if (cap < h->maxKeypoints) { h->err = "cap below orbx_configure()'s bound"; return ORBX_E_CAP; }
  XCHECK(h, hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream_;  // verbatim: NULL is the device's default stream
  FrameParams P;
  memset(&P, 0, sizeof(P));
  for (int l = 0; l < h->nlevels; l++) P.geom[l] = h->geom[l];
  P.nlevels = h->nlevels;
  P.nframes = nframes;
  P.iniTh = h->iniThFAST;
  P.minTh = h->minThFAST;
  P.lap0 = lap0;
  P.lap1 = lap1;
  P.img0 = d_images;
  P.img0_stride = stride;
  P.img0_frame_stride = frame_stride;
  P.pyr = (uint8_t *)h->d_pyr.p;       P.pyr_fs = h->pyr_fs;
  P.blur = (uint8_t *)h->d_blur.p;     P.blur_fs = h->blur_fs;
  P.cellCnt = (uint32_t *)h->d_cellCnt.p; P.cell_fs = h->cell_fs;
  P.slots = (uint32_t *)h->d_slots.p;  P.slot_fs = h->slot_fs;
  P.cand = (uint32_t *)h->d_cand.p;    P.cand_fs = h->cand_fs;
  P.knode = (uint16_t *)h->d_knode.p;
  P.lkp = (uint32_t *)h->d_lkp.p;      P.lkp_fs = h->lkp_fs;
  P.lrank = (uint16_t *)h->d_lrank.p;
  P.lcnt = (int32_t *)h->d_lcnt.p;
  P.candCnt = (int32_t *)h->d_candCnt.p;
  P.xtab = (const int2 *)h->d_xtab.p;
  P.ytab = (const int2 *)h->d_ytab.p;
  P.disc = (const int8_t *)h->d_disc.p;
  P.totalTiles = h->totalTiles;
  P.totalCells = h->totalCells;
  P.cells = (const uint32_t *)h->d_cells.p;
  P.groups = (const uint32_t *)h->d_groups.p;
  P.totalGroups = h->totalGroups;
  P.magicGroups = magic_div((uint32_t)std::max(h->totalGroups, 1));
  P.tiles = (const uint32_t *)h->d_tiles.p;
  P.magicCells = magic_div((uint32_t)std::max(h->totalCells, 1));
  P.magicTiles = magic_div((uint32_t)std::max(h->totalTiles, 1));
  P.magicKpBlk = magic_div((uint32_t)std::max((h->totalKp + 3) / 4, 1));
  P.totalKp = h->totalKp;
  P.octCap = h->octCap;
  P.chain = (const ChainTile *)h->d_chain.p; P.chainBuf0 = h->chainBuf0; P.chainBuf1 = h->chainBuf1;
  P.out_kps = d_kps;
  P.out_desc = d_desc;
  P.out_counts = d_counts;
  P.cap = cap;
  const bool prof = h->profiling && h->ev_ok;
  hipEvent_t *pev = h->ev[h->prof_head % PROF_DEPTH];
  if (prof) XCHECK(h, hipEventRecord(pev[0], s));
  // One workgroup per (frame, level) in k_octree and one launch per level here: a batch fills the chip that way; a single frame is
  // all launch gaps and one long workgroup, so few-frame calls ("wide") take the one-launch pyramid and 1024-thread octree.
  const bool wide = (long long)h->nlevels * nframes <= 32;
  if (wide && h->chainTiles > 0) hipLaunchKernelGGL(k_pyramid_chain, dim3(h->chainTiles * nframes), dim3(CHAIN_NT), h->chainLds, s, P);
  else
  for (int l = 1; l < h->nlevels; l++) {
    const LevelGeom &G = h->geom[l], &Gs = h->geom[l - 1];
    const int rowBytes = (int)align_up((size_t)Gs.w + 4, 16);
    // Two-pass form (rows and their horizontal interpolation staged in LDS) whenever a tile's source rows fit the stage and pass 1's
    // precondition holds (two neighbouring columns' source bytes inside one aligned 8-byte window: horizontal scale factor below 3);
    // otherwise the one-pass form straight from global memory (tPitch 0), which stages the x table only.
    const int wq = (G.w + 3) & ~3;
    const int tPitch2 = (int)align_up((size_t)wq * 2, 8);                  // horizontally interpolated rows, 16 bits per column
    const size_t lds2 = (size_t)G.resizeSrcRows * (rowBytes + tPitch2) + 16 * RESIZE_ROWS_MAX;
    const bool twoPass = G.resizeSrcRows <= RESIZE_MAXSRC && (double)Gs.w / G.w < 3.0 && lds2 <= ORB_LDS_LIMIT - 1024;   // (very wide images: one-pass)
    const int tPitch = twoPass ? tPitch2 : 0;
    const size_t lds = twoPass ? lds2 : (size_t)wq * 8;
    hipLaunchKernelGGL(k_resize, dim3(((G.h + G.resizeRows - 1) / G.resizeRows) * nframes), dim3(256), lds, s, P, l, rowBytes, tPitch, G.resizeSrcRows);
  }
  if (prof) XCHECK(h, hipEventRecord(pev[1], s));
  if (h->totalGroups > 0) hipLaunchKernelGGL(k_fast, dim3(h->totalGroups * nframes), dim3(FAST_NT), 0, s, P);
  if (prof) XCHECK(h, hipEventRecord(pev[2], s));
  // (k_octree: one workgroup per (frame, level); a single frame has only nlevels of them and the longest, level 0, is the critical
  // path of the whole call, so few-frame launches use 1024 threads per workgroup)
  if (wide) {
    if (h->octCellsLds) hipLaunchKernelGGL((k_octree<1024, true>), dim3(h->nlevels * nframes), dim3(1024), h->octLds, s, P, (uint32_t *)h->d_cellOff.p);
    else hipLaunchKernelGGL((k_octree<1024, false>), dim3(h->nlevels * nframes), dim3(1024), h->octLds, s, P, (uint32_t *)h->d_cellOff.p);
  } else {
    if (h->octCellsLds) hipLaunchKernelGGL((k_octree<256, true>), dim3(h->nlevels * nframes), dim3(256), h->octLds, s, P, (uint32_t *)h->d_cellOff.p);
    else hipLaunchKernelGGL((k_octree<256, false>), dim3(h->nlevels * nframes), dim3(256), h->octLds, s, P, (uint32_t *)h->d_cellOff.p);
  }
  if (prof) XCHECK(h, hipEventRecord(pev[3], s));
  hipLaunchKernelGGL(k_blur, dim3(h->totalTiles * nframes), dim3(256), 0, s, P);
  if (prof) XCHECK(h, hipEventRecord(pev[4], s));
  hipLaunchKernelGGL(k_describe, dim3(std::max((h->totalKp + 3) / 4, 1) * nframes), dim3(256), 0, s, P);   // also writes the frame totals (>= 1 workgroup per frame: nfeatures == 0)
  if (prof) { XCHECK(h, hipEventRecord(pev[5], s)); h->prof_head++; h->stage_valid = true; }
  XCHECK(h, hipGetLastError());
  h->last = P;
  h->have_last = true;
  // consumers of this batch on OTHER streams (orbx_compute_stereo_matches, orbx_download_*) order themselves behind it
  if (!h->in_capture) {
    if (!h->last_done) XCHECK(h, hipEventCreateWithFlags(&h->last_done, hipEventDisableTiming));
    XCHECK(h, hipEventRecord(h->last_done, s));
    h->last_pending = true;
  }
  return 0;
}

int orbx_extract(orbx_t *h, const uint8_t *image, int rows, int cols, size_t stride, int lap0, int lap1,
                 orbx_keypoint_t *keypoints, uint8_t *descriptors, int cap, int *n_out) {
  if (!h) return ORBX_E_ARG;
  if (n_out) *n_out = 0;
  if (!image || rows <= 0 || cols <= 0) return ORBX_E_EMPTY;  // ORBextractor.cc:1075-1076
  if (stride < (size_t)cols || !n_out) return ORBX_E_ARG;
  int rc = orbx_configure(h, rows, cols, h->rows == rows && h->cols == cols ? h->max_batch : 1);
  if (rc < 0) return rc;
  const int icap = h->maxKeypoints;
  const size_t dstride = align_up((size_t)cols, 64);
  XCHECK(h, hipSetDevice(h->device));
  XCHECK(h, h->d_img.reserve(dstride * rows));
  // pinned staging: one H2D, all kernels, one D2H (counts, keypoints and descriptors are one device block) and ONE synchronisation
  // per frame
  const size_t in_bytes = dstride * rows, kp_bytes = sizeof(orbx_keypoint_t) * (size_t)icap, de_bytes = 32 * (size_t)icap;
  const size_t kp_off = 64, de_off = kp_off + align_up(kp_bytes, 64), out_bytes = de_off + de_bytes;   // [counts][keypoints][descriptors], 64-byte aligned parts
  XCHECK(h, h->d_okps.reserve(out_bytes));
  if (h->pin_in_bytes < in_bytes) {
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    h->pin_in = nullptr; h->pin_in_bytes = 0;
    XCHECK(h, hipHostMalloc(&h->pin_in, in_bytes, hipHostMallocDefault));
    h->pin_in_bytes = in_bytes;
  }
  if (h->pin_out_bytes < out_bytes) {
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    h->pin_out = nullptr; h->pin_out_bytes = 0;
    XCHECK(h, hipHostMalloc(&h->pin_out, out_bytes, hipHostMallocDefault));
    h->pin_out_bytes = out_bytes;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (int y = 0; y < rows; y++) memcpy((uint8_t *)h->pin_in + (size_t)y * dstride, image + (size_t)y * stride, (size_t)cols);
  const auto t1 = std::chrono::steady_clock::now();
  uint8_t *po = (uint8_t *)h->pin_out;
  auto enqueue = [&]() -> int {  // the whole per-frame sequence on h->stream
    XCHECK(h, hipMemcpyAsync(h->d_img.p, h->pin_in, in_bytes, hipMemcpyHostToDevice, h->stream));
    uint8_t *dout = (uint8_t *)h->d_okps.p;
    const int r = orbx_extract_batch_device(h, (const uint8_t *)h->d_img.p, rows, cols, dstride, dstride * rows, 1, lap0, lap1,
                                            (orbx_keypoint_t *)(dout + kp_off), dout + de_off, (int32_t *)dout, icap, h->stream);
    if (r < 0) return r;
    XCHECK(h, hipMemcpyAsync(po, dout, out_bytes, hipMemcpyDeviceToHost, h->stream));
    return 0;
  };
  bool launched = false;
  if (h->graph_ok && !h->profiling) {
    auto &K = h->graph_key;
    const bool same = h->graph && K.rows == rows && K.cols == cols && K.lap0 == lap0 && K.lap1 == lap1 && K.icap == icap && K.pin_in == h->pin_in &&
                      K.pin_out == h->pin_out && K.d_img == h->d_img.p && K.d_okps == h->d_okps.p && K.d_pyr == h->d_pyr.p;
    if (!same) {
      if (h->graph) { (void)hipGraphExecDestroy(h->graph); h->graph = nullptr; }
      hipGraph_t g = nullptr;
      if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        h->in_capture = true;
        const int r = enqueue();
        h->in_capture = false;
        const hipError_t e = hipStreamEndCapture(h->stream, &g);
        if (r == 0 && e == hipSuccess && g && hipGraphInstantiate(&h->graph, g, nullptr, nullptr, 0) == hipSuccess) {
          K.rows = rows; K.cols = cols; K.lap0 = lap0; K.lap1 = lap1; K.icap = icap; K.pin_in = h->pin_in; K.pin_out = h->pin_out;
          K.d_img = h->d_img.p; K.d_okps = h->d_okps.p; K.d_pyr = h->d_pyr.p;
        } else {
          h->graph = nullptr;
          h->graph_ok = false;
          (void)hipGetLastError();
        }
        if (g) (void)hipGraphDestroy(g);
      } else {
        h->graph_ok = false;
        (void)hipGetLastError();
      }
    }
    if (h->graph) {
      XCHECK(h, hipGraphLaunch(h->graph, h->stream));
      launched = true;
    }
  }
  if (!launched) { rc = enqueue(); if (rc < 0) return rc; }
  const auto t2 = std::chrono::steady_clock::now();
  XCHECK(h, hipStreamSynchronize(h->stream));
  const auto t3 = std::chrono::steady_clock::now();
  h->last_pending = false;   // this entry point returns with the frame's work complete
  int32_t counts[2];
  memcpy(counts, po, sizeof(counts));
  *n_out = counts[0];
  if (counts[0] > cap) return ORBX_E_CAP;
  if (counts[0] > 0) {
    if (!keypoints || !descriptors) return ORBX_E_ARG;
    memcpy(keypoints, po + kp_off, sizeof(orbx_keypoint_t) * (size_t)counts[0]);
    memcpy(descriptors, po + de_off, 32 * (size_t)counts[0]);
  }
  const auto t4 = std::chrono::steady_clock::now();
  auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<float, std::micro>(b - a).count(); };
  h->host_us[0] = us(t0, t1); h->host_us[1] = us(t1, t2); h->host_us[2] = us(t2, t3); h->host_us[3] = us(t3, t4);
  return counts[1];
}

// Host-side split of the last orbx_extract call, microseconds: copy of the image into pinned staging, submission of the frame's
// sequence (graph launch), wait for its completion, copy of the results to the caller's arrays.
int orbx_get_host_us(const orbx_t *h, float *us, int cap) {
  if (!h || !us || cap < 4) return ORBX_E_ARG;
  for (int i = 0; i < 4; i++) us[i] = h->host_us[i];
  return 4;
}

int orbx_cvt_color_gray_device(const uint8_t *d_src, int rows, int cols, size_t src_stride, int channels, int rgb_order, uint8_t *d_dst,
                               size_t dst_stride, void *stream) {
  if (!d_src || !d_dst || rows <= 0 || cols <= 0 || (channels != 3 && channels != 4) || src_stride < (size_t)cols * channels || dst_stride < (size_t)cols)
    return ORBX_E_ARG;
  hipLaunchKernelGGL(k_cvt_gray, dim3((cols + 1023) / 1024, rows), dim3(256), 0, (hipStream_t)stream, d_src, rows, cols, src_stride, channels,
                     rgb_order ? 1 : 0, d_dst, dst_stride);
  return hipGetLastError() == hipSuccess ? 0 : ORBX_E_HIP;
}

int orbx_cvt_color_gray(orbx_t *h, const uint8_t *src, int rows, int cols, size_t src_stride, int channels, int rgb_order, uint8_t *dst,
                        size_t dst_stride) {
  if (!h || !src || !dst || rows <= 0 || cols <= 0 || (channels != 3 && channels != 4) || src_stride < (size_t)cols * channels || dst_stride < (size_t)cols)
    return ORBX_E_ARG;
  XCHECK(h, hipSetDevice(h->device));
  const size_t sbytes = (size_t)cols * channels, dpitch = align_up((size_t)cols, 64);
  XCHECK(h, h->stereo[0].reserve(sbytes * rows));
  XCHECK(h, h->stereo[1].reserve(dpitch * rows));
  XCHECK(h, hipMemcpy2DAsync(h->stereo[0].p, sbytes, src, src_stride, sbytes, (size_t)rows, hipMemcpyHostToDevice, h->stream));
  const int rc = orbx_cvt_color_gray_device((const uint8_t *)h->stereo[0].p, rows, cols, sbytes, channels, rgb_order, (uint8_t *)h->stereo[1].p, dpitch, h->stream);
  if (rc < 0) return rc;
  XCHECK(h, hipMemcpy2DAsync(dst, dst_stride, h->stereo[1].p, dpitch, (size_t)cols, (size_t)rows, hipMemcpyDeviceToHost, h->stream));
  XCHECK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int orbx_clahe_device(const uint8_t *d_src, int rows, int cols, size_t src_stride, double clip_limit, int tiles_x, int tiles_y, uint8_t *d_lut,
                      uint8_t *d_dst, size_t dst_stride, void *stream) {
  if (!d_src || !d_dst || !d_lut || rows <= 0 || cols <= 0 || tiles_x <= 0 || tiles_y <= 0 || tiles_x * tiles_y > 256 || cols < tiles_x || rows < tiles_y ||
      src_stride < (size_t)cols || dst_stride < (size_t)cols || !(clip_limit >= 0.0))
    return ORBX_E_ARG;
  // CLAHE_Impl::apply: tile size of the (possibly extended) image, lutScale and the integer clip limit
  const int ecols = cols % tiles_x == 0 ? cols : cols + tiles_x - cols % tiles_x, erows = rows % tiles_y == 0 ? rows : rows + tiles_y - rows % tiles_y;
  const int tw = ecols / tiles_x, th = erows / tiles_y, tileSizeTotal = tw * th;
  const float lutScale = (float)255 / (float)tileSizeTotal;
  int clipLimit = 0;
  if (clip_limit > 0.0) {
    clipLimit = (int)(clip_limit * tileSizeTotal / 256);
    if (clipLimit < 1) clipLimit = 1;
  }
  hipLaunchKernelGGL(k_clahe_lut, dim3(tiles_x * tiles_y), dim3(256), 0, (hipStream_t)stream, d_src, rows, cols, src_stride, tiles_x, tw, th, clipLimit,
                     lutScale, d_lut);
  const int rowsPerBlock = 4;
  hipLaunchKernelGGL(k_clahe_interp, dim3((rows + rowsPerBlock - 1) / rowsPerBlock), dim3(256), (size_t)tiles_x * tiles_y * 256, (hipStream_t)stream, d_src,
                     rows, cols, src_stride, tiles_x, tiles_y, 1.0f / (float)tw, 1.0f / (float)th, (const uint8_t *)d_lut, d_dst, dst_stride, rowsPerBlock);
  return hipGetLastError() == hipSuccess ? 0 : ORBX_E_HIP;
}

int orbx_clahe(orbx_t *h, const uint8_t *src, int rows, int cols, size_t src_stride, double clip_limit, int tiles_x, int tiles_y, uint8_t *dst,
               size_t dst_stride) {
  if (!h || !src || !dst || rows <= 0 || cols <= 0 || src_stride < (size_t)cols || dst_stride < (size_t)cols) return ORBX_E_ARG;
  XCHECK(h, hipSetDevice(h->device));
  const size_t pitch = align_up((size_t)cols, 64);
  XCHECK(h, h->stereo[0].reserve(pitch * rows));
  XCHECK(h, h->stereo[1].reserve(pitch * rows));
  XCHECK(h, h->stereo[2].reserve(256 * 256));
  XCHECK(h, hipMemcpy2DAsync(h->stereo[0].p, pitch, src, src_stride, (size_t)cols, (size_t)rows, hipMemcpyHostToDevice, h->stream));
  const int rc = orbx_clahe_device((const uint8_t *)h->stereo[0].p, rows, cols, pitch, clip_limit, tiles_x, tiles_y, (uint8_t *)h->stereo[2].p,
                                   (uint8_t *)h->stereo[1].p, pitch, h->stream);
  if (rc < 0) return rc;
  XCHECK(h, hipMemcpy2DAsync(dst, dst_stride, h->stereo[1].p, pitch, (size_t)cols, (size_t)rows, hipMemcpyDeviceToHost, h->stream));
  XCHECK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int orbx_remap_linear_device(const uint8_t *d_src, int src_rows, int src_cols, size_t src_stride, const float *d_mapx, const float *d_mapy,
                             size_t map_stride_elems, int rows, int cols, uint8_t *d_dst, size_t dst_stride, void *stream) {
  if (!d_src || !d_mapx || !d_mapy || !d_dst || src_rows <= 0 || src_cols <= 0 || rows <= 0 || cols <= 0 || src_stride < (size_t)src_cols ||
      map_stride_elems < (size_t)cols || dst_stride < (size_t)cols || src_cols > 32767 || src_rows > 32767)
    return ORBX_E_ARG;
  hipLaunchKernelGGL(k_remap_linear, dim3((cols + 255) / 256, rows), dim3(256), 0, (hipStream_t)stream, d_src, src_rows, src_cols, src_stride, d_mapx,
                     d_mapy, map_stride_elems, rows, cols, d_dst, dst_stride);
  return hipGetLastError() == hipSuccess ? 0 : ORBX_E_HIP;
}

int orbx_remap_linear(orbx_t *h, const uint8_t *src, int src_rows, int src_cols, size_t src_stride, const float *mapx, const float *mapy, int rows,
                      int cols, uint8_t *dst, size_t dst_stride) {
  if (!h || !src || !dst || src_rows <= 0 || src_cols <= 0 || rows <= 0 || cols <= 0 || src_stride < (size_t)src_cols || dst_stride < (size_t)cols ||
      ((mapx == nullptr) != (mapy == nullptr)))
    return ORBX_E_ARG;
  XCHECK(h, hipSetDevice(h->device));
  if (mapx) {   // the maps are fixed per camera (computed once, stereo_euroc.cc:113-114): uploaded when given, kept for later calls
    XCHECK(h, h->maps[0].reserve(sizeof(float) * (size_t)rows * cols));
    XCHECK(h, h->maps[1].reserve(sizeof(float) * (size_t)rows * cols));
    XCHECK(h, hipMemcpyAsync(h->maps[0].p, mapx, sizeof(float) * (size_t)rows * cols, hipMemcpyHostToDevice, h->stream));
    XCHECK(h, hipMemcpyAsync(h->maps[1].p, mapy, sizeof(float) * (size_t)rows * cols, hipMemcpyHostToDevice, h->stream));
    h->maps_rows = rows;
    h->maps_cols = cols;
  } else if (h->maps_rows != rows || h->maps_cols != cols) {
    h->err = "orbx_remap_linear: no maps of this size uploaded yet";
    return ORBX_E_ARG;
  }
  const size_t spitch = align_up((size_t)src_cols, 64), dpitch = align_up((size_t)cols, 64);
  XCHECK(h, h->stereo[0].reserve(spitch * src_rows));
  XCHECK(h, h->stereo[1].reserve(dpitch * rows));
  XCHECK(h, hipMemcpy2DAsync(h->stereo[0].p, spitch, src, src_stride, (size_t)src_cols, (size_t)src_rows, hipMemcpyHostToDevice, h->stream));
  const int rc = orbx_remap_linear_device((const uint8_t *)h->stereo[0].p, src_rows, src_cols, spitch, (const float *)h->maps[0].p,
                                          (const float *)h->maps[1].p, (size_t)cols, rows, cols, (uint8_t *)h->stereo[1].p, dpitch, h->stream);
  if (rc < 0) return rc;
  XCHECK(h, hipMemcpy2DAsync(dst, dst_stride, h->stereo[1].p, dpitch, (size_t)cols, (size_t)rows, hipMemcpyDeviceToHost, h->stream));
  XCHECK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int orbx_level_info(const orbx_t *h, int level, int *rows, int *cols) {
  if (!h || level < 0 || level >= h->nlevels || h->geom.empty()) return ORBX_E_ARG;
  if (rows) *rows = h->geom[level].h;
  if (cols) *cols = h->geom[level].w;
  return 0;
}

static int download_plane(orbx_t *h, const uint8_t *src, size_t spitch, int w, int hh, int border, uint8_t *dst, size_t dst_stride) {
  std::vector<uint8_t> tmp((size_t)w * hh);
  XCHECK(h, join_last(h));
  XCHECK(h, hipMemcpy2DAsync(tmp.data(), (size_t)w, src, spitch, (size_t)w, (size_t)hh, hipMemcpyDeviceToHost, h->stream));
  XCHECK(h, hipStreamSynchronize(h->stream));
  auto refl = [](int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
  };
  for (int y = -border; y < hh + border; y++) {
    const uint8_t *S = tmp.data() + (size_t)refl(y, hh) * w;
    uint8_t *D = dst + (size_t)(y + border) * dst_stride;
    for (int x = -border; x < w + border; x++) D[x + border] = S[refl(x, w)];
  }
  return 0;
}

int orbx_download_level(orbx_t *h, int frame, int level, int border, uint8_t *dst, size_t dst_stride) {
  if (!h || !h->have_last || !dst || level < 0 || level >= h->nlevels || frame < 0 || frame >= h->last.nframes || border < 0) return ORBX_E_ARG;
  XCHECK(h, hipSetDevice(h->device));
  const LevelGeom &G = h->geom[level];
  if (dst_stride < (size_t)(G.w + 2 * border)) return ORBX_E_ARG;
  if (level == 0)
    return download_plane(h, h->last.img0 + (size_t)frame * h->last.img0_frame_stride, h->last.img0_stride, G.w, G.h, border, dst, dst_stride);
  return download_plane(h, h->last.pyr + (size_t)frame * h->last.pyr_fs + G.off, (size_t)G.pitch, G.w, G.h, border, dst, dst_stride);
}

// mvImagePyramid of one frame in ONE transfer: every level with its BORDER_REFLECT_101 frame (ORBextractor.cc:1203-1215) is
// written by one kernel into a packed device buffer, copied once into pinned memory and from there into the caller's block.
struct PackLevel { const uint8_t *src; int spitch, w, h, dstride; unsigned doff, dbytes; };
struct PackParams { PackLevel L[ORBX_MAX_LEVELS]; int nlevels, border; uint8_t *dst; };
}  // extern "C"
__global__ __launch_bounds__(256) void k_pyramid_pack(PackParams P) {
  const PackLevel G = P.L[blockIdx.y];
  auto refl = [](int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
  };
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < G.dbytes; i += gridDim.x * 256u) {
    const int y = (int)(i / (unsigned)G.dstride) - P.border, x = (int)(i % (unsigned)G.dstride) - P.border;
    P.dst[G.doff + i] = G.src[(size_t)refl(y, G.h) * G.spitch + refl(x, G.w)];
  }
}
extern "C" {
int orbx_download_pyramid(orbx_t *h, int frame, int border, uint8_t *dst, size_t dst_bytes, size_t *offsets, size_t *strides) {
  if (!h || !h->have_last || frame < 0 || frame >= h->last.nframes || border < 0 || border > 64 || !offsets || !strides) return ORBX_E_ARG;
  XCHECK(h, hipSetDevice(h->device));
  PackParams P;
  memset(&P, 0, sizeof(P));
  P.nlevels = h->nlevels; P.border = border;
  size_t total = 0;
  for (int l = 0; l < h->nlevels; l++) {
    const LevelGeom &G = h->geom[l];
    PackLevel &L = P.L[l];
    L.src = l == 0 ? h->last.img0 + (size_t)frame * h->last.img0_frame_stride : h->last.pyr + (size_t)frame * h->last.pyr_fs + G.off;
    L.spitch = l == 0 ? (int)h->last.img0_stride : G.pitch;
    L.w = G.w; L.h = G.h; L.dstride = G.w + 2 * border;
    L.doff = (unsigned)total; L.dbytes = (unsigned)((size_t)L.dstride * (size_t)(G.h + 2 * border));
    offsets[l] = total; strides[l] = (size_t)L.dstride;
    total += ((size_t)L.dbytes + 63) & ~(size_t)63;
  }
  if (!dst) return (int)std::min<size_t>(total, 0x7fffffff);   // size query
  if (dst_bytes < total) return ORBX_E_CAP;
  XCHECK(h, h->d_pyrpack.reserve(total));
  if (h->pin_pyr_bytes < total) {
    if (h->pin_pyr) (void)hipHostFree(h->pin_pyr);
    h->pin_pyr = nullptr; h->pin_pyr_bytes = 0;
    XCHECK(h, hipHostMalloc(&h->pin_pyr, total, hipHostMallocDefault));
    h->pin_pyr_bytes = total;
  }
  P.dst = (uint8_t *)h->d_pyrpack.p;
  if (h->last_pending) XCHECK(h, hipStreamWaitEvent(h->stream, h->last_done, 0));
  hipLaunchKernelGGL(k_pyramid_pack, dim3(64, h->nlevels), dim3(256), 0, h->stream, P);
  XCHECK(h, hipGetLastError());
  XCHECK(h, hipMemcpyAsync(h->pin_pyr, h->d_pyrpack.p, total, hipMemcpyDeviceToHost, h->stream));
  XCHECK(h, hipStreamSynchronize(h->stream));
  memcpy(dst, h->pin_pyr, total);
  return (int)std::min<size_t>(total, 0x7fffffff);
}

int orbx_download_blurred_level(orbx_t *h, int frame, int level, uint8_t *dst, size_t dst_stride) {
  if (!h || !h->have_last || !dst || level < 0 || level >= h->nlevels || frame < 0 || frame >= h->last.nframes) return ORBX_E_ARG;
  XCHECK(h, hipSetDevice(h->device));
  const LevelGeom &G = h->geom[level];
  if (dst_stride < (size_t)G.w) return ORBX_E_ARG;
  return download_plane(h, h->last.blur + (size_t)frame * h->last.blur_fs + G.boff, (size_t)G.bpitch, G.w, G.h, 0, dst, dst_stride);
}

static int download_packed(orbx_t *h, const uint32_t *src, int n, float *xyr, int cap) {
  std::vector<uint32_t> tmp((size_t)std::max(n, 1));
  if (n > 0) XCHECK(h, copy_on(h->stream, tmp.data(), src, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n && i < cap; i++) {
    xyr[3 * i] = (float)(tmp[i] & 0xfff);
    xyr[3 * i + 1] = (float)((tmp[i] >> 12) & 0xfff);
    xyr[3 * i + 2] = (float)(tmp[i] >> 24);
  }
  return n;
}

int orbx_download_candidates(orbx_t *h, int frame, int level, float *xyr, int cap) {
  if (!h || !h->have_last || !xyr || level < 0 || level >= h->nlevels || frame < 0 || frame >= h->last.nframes) return ORBX_E_ARG;
  XCHECK(h, hipSetDevice(h->device));
  XCHECK(h, join_last(h));
  int32_t n = 0;
  XCHECK(h, copy_on(h->stream, &n, h->last.candCnt + (size_t)frame * h->nlevels + level, sizeof(n), hipMemcpyDeviceToHost));
  return download_packed(h, h->last.cand + (size_t)frame * h->last.cand_fs + h->geom[level].candBase, n, xyr, cap);
}

int orbx_download_level_keypoints(orbx_t *h, int frame, int level, float *xyr, int cap) {
  if (!h || !h->have_last || !xyr || level < 0 || level >= h->nlevels || frame < 0 || frame >= h->last.nframes) return ORBX_E_ARG;
  XCHECK(h, hipSetDevice(h->device));
  XCHECK(h, join_last(h));
  int32_t n[2] = {0, 0};
  XCHECK(h, copy_on(h->stream, n, h->last.lcnt + ((size_t)frame * h->nlevels + level) * 2, sizeof(n), hipMemcpyDeviceToHost));
  return download_packed(h, h->last.lkp + (size_t)frame * h->last.lkp_fs + h->geom[level].kpBase, n[0], xyr, cap);
}

}  // extern "C"

// --------------------------------------------------------------------------------------------------------------
// matcher
// --------------------------------------------------------------------------------------------------------------
struct orbm_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  DevBuf d_kp, d_desc, d_ur, d_qdesc, d_qf[4], d_qi[2], d_qfl, d_slot, d_sobs, d_moq, d_bd, d_nm, d_a, d_b, d_c, d_topk;
  DevBuf d_partner, d_qside, d_qany;
  DevBuf d_lfq;       // query arrays written by k_lastframe_project (orbm_search_by_projection_last_frame_batch_device)
  DevBuf d_rank;      // k_match_rank's two tables (accumulator seeds, Key32 tie-break bits) for k_match_scan_mfma
  int hamming_engine = 2;   // orbm_set_hamming_engine: 0 vector ALU, 1 k_match_scan_mfma, 2 (default) + lists built inside k_match_resolve for all-open pairs
  DevBuf d_block;     // inputs + outputs of one host-pointer search, one block (see search_host)
  int scan_mode = 0;  // SCAN_AUTO / SCAN_DENSE / SCAN_WALK of the projection searches (orbm_set_scan_mode); next_scan_mode: one search only
  int next_scan_mode = -1;
  void *pin = nullptr; size_t pin_bytes = 0;   // its pinned host mirror
  DevBuf scratch[8];  // grow-only buffers of the per-node / per-map-point entry points (SearchByBoW, ...)
  // fisheye-stereo options of the NEXT projection search (set by the *_fisheye entry points, consumed and cleared by
  // orbm_search_by_projection_batch_device): device pointers
  struct { int nleft = 0; const int32_t *partner = nullptr; const uint8_t *qside = nullptr; int couple = 0; int serial = 0; uint8_t *qany = nullptr; int init_th_low = -1; const float *fuse_inv_sigma2 = nullptr; } ext;
  bool profiling = false;
  hipEvent_t ev[PROF_DEPTH][3] = {};
  int prof_head = 0;
  bool ev_ok = false, ms_valid = false;
  std::string err;
};

#define MCHECK(m, call)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      (m)->err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
      return ORBX_E_HIP;                                                                      \
    }                                                                                         \
  } while (0)

#if defined(RESOLVE_STAMPS) || defined(WALK_STAMPS) || defined(SCAN_STAMPS) || defined(MF_STAMPS) || defined(WIDE_STAMPS)
static void *getenv_ptr(const char *name) { const char *e = getenv(name); return e ? (void *)strtoull(e, nullptr, 0) : nullptr; }
#endif

extern "C" {

orbm_t *orbm_create(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return nullptr;
  if (hipSetDevice(device) != hipSuccess || raise_lds_limits(device) != hipSuccess) return nullptr;
  orbm_handle *m = new orbm_handle();
  m->device = device;
  if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) { delete m; return nullptr; }
  if (const char *e = getenv("ORBM_SCAN_MODE")) { const int v = atoi(e); if (v >= SCAN_AUTO && v <= SCAN_WALK) m->scan_mode = v; }   // measurements only; see orbm_set_scan_mode
  if (const char *e = getenv("ORBM_HAMMING_ENGINE")) { const int v = atoi(e); if (v >= 0 && v <= 2) m->hamming_engine = v; }                                               // measurements only; see orbm_set_hamming_engine
  return m;
}

void orbm_destroy(orbm_t *m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  DevBuf *bufs[] = {&m->d_kp, &m->d_desc, &m->d_ur, &m->d_qdesc, &m->d_qf[0], &m->d_qf[1], &m->d_qf[2], &m->d_qf[3], &m->d_qi[0], &m->d_qi[1],
                    &m->d_qfl, &m->d_slot, &m->d_sobs, &m->d_moq, &m->d_bd, &m->d_nm, &m->d_a, &m->d_b, &m->d_c, &m->d_topk, &m->d_partner, &m->d_qside, &m->d_qany, &m->scratch[0], &m->scratch[1], &m->scratch[2], &m->scratch[3],
                    &m->scratch[4], &m->scratch[5], &m->scratch[6], &m->scratch[7], &m->d_block, &m->d_lfq, &m->d_rank};
  for (DevBuf *b : bufs) b->release();
  if (m->pin) (void)hipHostFree(m->pin);
  if (m->ev_ok)
    for (auto &set : m->ev)
      for (auto &e : set) (void)hipEventDestroy(e);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
}

const char *orbm_last_error(const orbm_t *m) { return m ? m->err.c_str() : "null handle"; }

void orbm_set_profiling(orbm_t *m, int enable) {
  if (!m) return;
  m->profiling = enable != 0;
  m->prof_head = 0;
  m->ms_valid = false;
  if (m->profiling && !m->ev_ok) {
    (void)hipSetDevice(m->device);
    bool ok = true;
    for (auto &set : m->ev)
      for (auto &e : set) ok = ok && hipEventCreate(&e) == hipSuccess;
    m->ev_ok = ok;
  }
}

// 0 = decided per frame pair on the device (default), 1 = always k_match_scan, 2 = always k_match_walk (frames of at most 2048
// keypoints; larger ones are scanned whatever the mode).  Results do not depend on the mode.
int orbm_set_scan_mode(orbm_t *m, int mode) {
  if (!m || mode < SCAN_AUTO || mode > SCAN_WALK) return ORBX_E_ARG;
  m->scan_mode = mode;
  return 0;
}

// Where open-window searches compute their Hamming distances: 0 = vector ALU (k_match_scan) like every other block, 1 = the
// all-keypoints scan of open-window query blocks on the matrix pipe (k_match_scan_mfma), 2 (default) = as 1, and frame pairs ALL of
// whose queries are open build their lists inside k_match_resolve (fused form: claimed keypoints masked per 512-query super-chunk).
// Results do not depend on it.
int orbm_set_hamming_engine(orbm_t *m, int engine) {
  if (!m || engine < 0 || engine > 2) return ORBX_E_ARG;
  m->hamming_engine = engine;
  return 0;
}

float orbm_get_last_ms(orbm_t *m) {
  if (!m || !m->profiling || !m->ev_ok || !m->ms_valid || m->prof_head <= 0) return -1.f;
  float t = -1.f;
  hipEvent_t *e = m->ev[(m->prof_head - 1) % PROF_DEPTH];
  if (hipEventSynchronize(e[2]) != hipSuccess) return -1.f;
  if (hipEventElapsedTime(&t, e[0], e[2]) != hipSuccess) return -1.f;
  return t;
}

// {scan, resolve} averaged over the searches launched since profiling was switched on (the PROF_DEPTH most recent ones).
int orbm_get_stage_ms(orbm_t *m, float *ms, int cap) {
  if (!m || !ms || cap < 2 || !m->profiling || !m->ev_ok || !m->ms_valid || m->prof_head <= 0) return 0;
  const int nb = std::min(m->prof_head, PROF_DEPTH);
  if (hipEventSynchronize(m->ev[(m->prof_head - 1) % PROF_DEPTH][2]) != hipSuccess) return 0;
  ms[0] = ms[1] = 0.f;
  for (int b = 0; b < nb; b++) {
    hipEvent_t *e = m->ev[(m->prof_head - 1 - b) % PROF_DEPTH];
    float t0 = 0, t1 = 0;
    if (hipEventElapsedTime(&t0, e[0], e[1]) != hipSuccess || hipEventElapsedTime(&t1, e[1], e[2]) != hipSuccess) return 0;
    ms[0] += t0 / (float)nb;
    ms[1] += t1 / (float)nb;
  }
  return 2;
}

// ORBmatcher.cc:2463-2483 (the bit trick there computes exactly popcount)
int orbm_descriptor_distance(const uint8_t *a, const uint8_t *b) {
  if (!a || !b) return ORBX_E_ARG;
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t pa, pb;
    memcpy(&pa, a + 4 * i, 4);
    memcpy(&pb, b + 4 * i, 4);
    dist += __builtin_popcount(pa ^ pb);
  }
  return dist;
}

void orbm_three_maxima(const int *histo, int L, int *ind1, int *ind2, int *ind3) {  // ORBmatcher.cc:2416-2458
  if (!ind1 || !ind2 || !ind3) return;
  int max1 = 0, max2 = 0, max3 = 0;
  *ind1 = *ind2 = *ind3 = -1;
  if (!histo) return;
  for (int i = 0; i < L; i++) {
    const int s = histo[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
    else if (s > max3) { max3 = s; *ind3 = i; }
  }
  if ((float)max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
  else if ((float)max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

float orbm_radius_by_viewing_cos(float viewCos) { return ((double)viewCos > 0.998) ? 2.5f : 4.0f; }  // ORBmatcher.cc:216-222

// cv::undistortPoints with R = I, P = K (Frame.cc:856, :883): 5 fixed-point iterations in double; one source
// for the host functions below and k_undistort (orb_project_kernels.h)
static void undistort_one(double u, double v, const float *K, const float *D, int nD, float *ou, float *ov) { undistort_point(u, v, K, D, nD, ou, ov); }

void orbm_undistort_keypoints(int n, const orbx_keypoint_t *keys, const float *K, const float *D, int nD, orbx_keypoint_t *keys_un) {
  if (!keys || !keys_un || !K || !D) return;
  for (int i = 0; i < n; i++) {
    orbx_keypoint_t k = keys[i];
    if (D[0] != 0.0f) undistort_one((double)keys[i].x, (double)keys[i].y, K, D, nD, &k.x, &k.y);
    keys_un[i] = k;
  }
}

void orbm_image_bounds(int cols, int rows, const float *K, const float *D, int nD, float *min_x, float *max_x, float *min_y, float *max_y) {
  if (!K || !D || !min_x || !max_x || !min_y || !max_y) return;
  if (D[0] != 0.0f) {
    const float c[4][2] = {{0.f, 0.f}, {(float)cols, 0.f}, {0.f, (float)rows}, {(float)cols, (float)rows}};
    float o[4][2];
    for (int i = 0; i < 4; i++) undistort_one((double)c[i][0], (double)c[i][1], K, D, nD, &o[i][0], &o[i][1]);
    *min_x = std::min(o[0][0], o[2][0]);
    *max_x = std::max(o[1][0], o[3][0]);
    *min_y = std::min(o[0][1], o[1][1]);
    *max_y = std::max(o[2][1], o[3][1]);
  } else {
    *min_x = 0.0f; *max_x = (float)cols; *min_y = 0.0f; *max_y = (float)rows;
  }
}

int orbm_undistort_keypoints_batch_device(orbm_t *m, const orbx_keypoint_t *d_keys, int key_stride, const int32_t *d_counts, int count_stride,
                                          int n_const, int nframes, const float *K, const float *D, int nD, orbx_keypoint_t *d_keys_un, void *stream_) {
  if (!m || !d_keys || !d_keys_un || !K || !D || nframes <= 0 || key_stride <= 0 || (nD != 4 && nD != 5)) return ORBX_E_ARG;
  if (!d_counts && (n_const < 0 || n_const > key_stride)) return ORBX_E_ARG;
  MCHECK(m, hipSetDevice(m->device));
  UndistortParams U;
  memset(&U, 0, sizeof(U));
  U.keys = reinterpret_cast<const float *>(d_keys); U.keys_un = reinterpret_cast<float *>(d_keys_un); U.key_stride = key_stride;
  U.counts = d_counts; U.count_stride = count_stride; U.count_const = n_const;
  for (int i = 0; i < 4; i++) U.K[i] = K[i];
  for (int i = 0; i < nD; i++) U.D[i] = D[i];
  U.nD = nD;
  const int maxn = d_counts ? key_stride : n_const;
  if (maxn == 0) return 0;
  hipLaunchKernelGGL(k_undistort, dim3((maxn + 255) / 256, nframes), dim3(256), 0, (hipStream_t)stream_, U);
  MCHECK(m, hipGetLastError());
  return 0;
}

void orbm_project(int cam_type, const float *p, float X, float Y, float Z, float *u, float *v) {
  if (!p || !u || !v) return;
  if (cam_type == 0) {  // Pinhole.cpp:46-49
    *u = p[0] * X / Z + p[2];
    *v = p[1] * Y / Z + p[3];
  } else {  // KannalaBrandt8.cpp:29-45
    const float x2_plus_y2 = X * X + Y * Y;
    const float theta = orbat::ref_atan2f(sqrtf(x2_plus_y2), Z);
    const float psi = orbat::ref_atan2f(Y, X);
    const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2,
                theta9 = theta7 * theta2;
    const float r = theta + p[4] * theta3 + p[5] * theta5 + p[6] * theta7 + p[7] * theta9;
    // cos / sin on a float resolve to the float overloads (cosf / sinf) once <math.h> is in the translation unit, which
    // opencv2/opencv.hpp brings (libm choices) - the same assumption MapPoint::PredictScale's log(float) rests on;
    // evaluated through the bit-exact glibc replicas the device uses (tests/test_libm_replicas.py: equal to the host libm)
#ifdef ORB_KB8_DOUBLE_TRIG   // build switch for reference builds in which cos(psi) / sin(psi) bind ::cos(double) (GCC 5, or no <math.h> wrapper)
    *u = (float)((double)(p[0] * r) * ::cos((double)psi) + (double)p[2]);
    *v = (float)((double)(p[1] * r) * ::sin((double)psi) + (double)p[3]);
#else
    *u = p[0] * r * orbsc::ref_cosf(psi) + p[2];
    *v = p[1] * r * orbsc::ref_sinf(psi) + p[3];
#endif
  }
}

int orbm_search_by_projection_batch_device(orbm_t *m, const orbm_frame_t *f, int frame_stride, const int32_t *d_frame_n,
                                           int frame_n_stride, const orbm_queries_t *q, int query_stride,
                                           const int32_t *d_query_n, int query_n_stride, int npairs, float nnratio,
                                           int th_dist, int use_second, int32_t *d_slot, uint8_t *d_slot_obs,
                                           int32_t *d_moq, int32_t *d_bd, int32_t *d_nm, void *stream_) {
  if (!m || !f || !q || npairs <= 0 || !d_slot || !d_slot_obs) return ORBX_E_ARG;
  if (!f->keys_un || !f->descriptors || !q->descriptors || !q->u || !q->v || !q->radius || !q->min_level || !q->max_level) return ORBX_E_ARG;
  if (!(f->max_x > f->min_x) || !(f->max_y > f->min_y)) return ORBX_E_ARG;
  // bestDist starts at 256 in the reference: a threshold >= 256 would "accept" a query without any candidate (index -1 there)
  if (th_dist < 0 || th_dist > 255) { m->err = "th_dist must be in [0, 255]"; return ORBX_E_ARG; }
  MCHECK(m, hipSetDevice(m->device));
  hipStream_t s = (hipStream_t)stream_;  // verbatim: NULL is the device's default stream
  MatchProblemSet M;
  memset(&M, 0, sizeof(M));
  M.kp = reinterpret_cast<const float *>(f->keys_un);
  M.desc = f->descriptors;
  M.u_right = f->u_right;
  M.frame_stride = frame_stride;
  M.frame_n = d_frame_n; M.frame_n_stride = frame_n_stride; M.frame_n_const = f->n;
  M.min_x = f->min_x; M.min_y = f->min_y;
  M.inv_w = (float)ORBM_GRID_COLS / (f->max_x - f->min_x);   // Frame.cc:379
  M.inv_h = (float)ORBM_GRID_ROWS / (f->max_y - f->min_y);   // Frame.cc:380
  M.qdesc = q->descriptors; M.qu = q->u; M.qv = q->v; M.qr = q->radius; M.qur = q->u_r;
  M.qminl = q->min_level; M.qmaxl = q->max_level; M.qflags = q->flags;
  M.query_stride = query_stride;
  M.query_n = d_query_n; M.query_n_stride = query_n_stride; M.query_n_const = q->nq;
  M.nnratio = nnratio; M.th_dist = th_dist; M.use_second = use_second;
  M.slot = d_slot; M.slot_obs = d_slot_obs; M.match_of_query = d_moq; M.best_dist = d_bd; M.nmatches = d_nm;
#if defined(RESOLVE_STAMPS) || defined(WALK_STAMPS) || defined(SCAN_STAMPS) || defined(MF_STAMPS) || defined(WIDE_STAMPS)
  M.dbg = (long long *)getenv_ptr("ORBHIP_DBG_PTR");
#endif
  M.nleft = m->ext.nleft; M.partner = m->ext.partner; M.qside = m->ext.qside; M.couple = m->ext.couple;
  M.serial = m->ext.serial; M.qany = m->ext.qany;
  const int init_th_low = m->ext.init_th_low;  // >= 0: SearchForInitialization's resolve instead of the claim loop
  const bool fuse = m->ext.fuse_inv_sigma2 != nullptr;
  if (fuse) for (int i = 0; i < 16; i++) M.inv_sigma2[i] = m->ext.fuse_inv_sigma2[i];
  m->ext = {};
  const int maxn = d_frame_n ? frame_stride : f->n;
  const int maxq = d_query_n ? query_stride : q->nq;
  if (maxn > ORBM_MAX_KEYPOINTS) { m->err = "more than 15360 keypoints per frame not supported by the search kernels"; return ORBX_E_ARG; }
  if (maxn <= 0 || maxq <= 0) return ORBX_E_ARG;
  // scratch: TOPK keys per query (grows on demand; not on the steady-state path)
  const bool k32 = maxn <= 2048;  // Key32 holds an 11-bit keypoint index
  // Latency mode: with few problems in flight the scan's candidate loop is the critical path (one workgroup per 256 queries walks
  // every candidate), so the candidate chunks are split over up to 8 workgroups per query block and their top-8 lists merged.
  const int qblocks = (maxq + MATCH_NT - 1) / MATCH_NT;
  int nslices = 1;
  if ((long long)npairs * qblocks <= 32) nslices = std::max(1, std::min(8, (maxn + MATCH_CH - 1) / MATCH_CH));
  const size_t slice_stride = MATCH_TOPK * ((size_t)(npairs - 1) * query_stride + maxq);   // keys per slice
  const size_t need = (k32 ? sizeof(uint32_t) : sizeof(unsigned long long)) * slice_stride * (size_t)nslices;
  if (need > m->d_topk.bytes) {
    MCHECK(m, hipStreamSynchronize(s));
    MCHECK(m, m->d_topk.reserve(need));
  }
  // matrix-pipe scan of the open-window query blocks (monocular Key32 problems, batch mode): rank tables for it
  const int force0 = !k32 ? SCAN_DENSE : (m->next_scan_mode >= 0 ? m->next_scan_mode : m->scan_mode);
  const bool mfma = m->hamming_engine >= 1 && k32 && !fuse && !M.qside && !M.partner && !M.u_right && nslices == 1 && force0 != SCAN_WALK;
  const size_t nrank = (size_t)(npairs - 1) * frame_stride + maxn;
  if (mfma && sizeof(uint32_t) * (2 * nrank + (size_t)npairs) > m->d_rank.bytes) {
    MCHECK(m, hipStreamSynchronize(s));
    MCHECK(m, m->d_rank.reserve(sizeof(uint32_t) * (2 * nrank + (size_t)npairs)));
  }
  const bool prof = m->profiling && m->ev_ok;
  hipEvent_t *pev = m->ev[m->prof_head % PROF_DEPTH];
  if (prof) MCHECK(m, hipEventRecord(pev[0], s));
  M.npairs = npairs; M.scan_qblocks = qblocks;
  const dim3 sgrid(8 * qblocks * ((npairs + 7) / 8), 1, nslices);   // XCD-aware 1-D order, see k_match_scan
  if (nslices > 1 && M.qany) MCHECK(m, hipMemsetAsync(M.qany, 0, (size_t)(npairs - 1) * query_stride + maxq, s));
  // resolve LDS: owner words (n + 1 dummy), slot words, column-sorted keypoint list (u16), partner list, octave bytes; see k_match_resolve
  const size_t small = sizeof(uint32_t) * (size_t)((maxn + 1) + maxn + (maxn + 1) / 2 + (M.partner ? (maxn + 1) / 2 + 1 : 0) + 2);
  const size_t big = small + sizeof(uint32_t) * (size_t)((maxn + 3) / 4) + 48 * (size_t)maxn;   // + octave bytes, records, descriptors
  const bool ldscand = big <= 136 * 1024;   // + the kernel's static LDS (Key32: wide list array 16 KiB, chunk lists, requests)
  // fused form: + one accumulator seed per keypoint; serial / coupled problems never take the wide form
  const bool fused = mfma && m->hamming_engine >= 2 && init_th_low < 0 && !M.serial && M.couple == 0 && big + sizeof(uint32_t) * (size_t)maxn <= 138 * 1024;
  const size_t lds = fused ? big + sizeof(uint32_t) * (size_t)maxn : ldscand ? big : small;
  if (lds > 152 * 1024) { m->err = "too many keypoints per frame for the search kernels' LDS state (fisheye-stereo frames: at most 13000)"; return ORBX_E_ARG; }
  const dim3 rblock(64 * RESOLVE_NW_OF(fused));
  // Window walk (k_match_walk) or full scan (k_match_scan): decided per pair on the device unless a mode is forced; frames beyond
  // 2048 keypoints (Key64) are always scanned.  The walk's workgroups come first: they are the short ones.
  const int force = !k32 ? SCAN_DENSE : (m->next_scan_mode >= 0 ? m->next_scan_mode : m->scan_mode);
  m->next_scan_mode = -1;
  const int capn = std::min((maxn + 7) & ~7, WALK_MAX_N);
  // few pairs in flight: four lanes per query and four times the workgroups (see k_match_walk)
  const int lpq = (long long)npairs * qblocks <= 32 ? 4 : 1;
  const int wqblocks = (maxq + MATCH_NT / lpq - 1) / (MATCH_NT / lpq);
  const size_t wlds = sizeof(uint32_t) * (GRID_CELLS + 4) + (12 + (fuse || M.u_right ? 4 : 0)) * (size_t)capn + 2 * WALK_LIST * MATCH_NT +
                      (lpq > 1 ? sizeof(uint32_t) * MATCH_TOPK * MATCH_NT : 0);
  const dim3 wgrid(8 * wqblocks * ((npairs + 7) / 8));
#define LAUNCH_WALK(MODE)                                                                                                                                    \
  do {                                                                                                                                                        \
    if (lpq > 1) hipLaunchKernelGGL((k_match_walk<Key32, MODE, 4>), wgrid, dim3(MATCH_NT), wlds, s, M, (Key32::T *)m->d_topk.p, force, capn, wqblocks);      \
    else hipLaunchKernelGGL((k_match_walk<Key32, MODE, 1>), wgrid, dim3(MATCH_NT), wlds, s, M, (Key32::T *)m->d_topk.p, force, capn, wqblocks);               \
  } while (0)
  if (force != SCAN_DENSE) {
    if (fuse) LAUNCH_WALK(SCAN_FUSE); else if (M.qside) LAUNCH_WALK(SCAN_FISHEYE); else if (M.u_right) LAUNCH_WALK(SCAN_UR); else LAUNCH_WALK(SCAN_PLAIN);
  }
#undef LAUNCH_WALK
  uint32_t *rec = mfma ? (uint32_t *)m->d_rank.p : nullptr, *keyrec = mfma ? rec + nrank : nullptr, *pairflag = mfma ? keyrec + nrank : nullptr;
  // Engine 2 (fused): the pairs k_match_rank flags make their lists inside k_match_resolve; what is left (pairs with windowed queries)
  // goes to the walk / the vector-ALU scan.  k_match_scan_mfma is not launched then: a launch whose workgroups only read a flag and
  // leave still queues for the kernel's registers and LDS on every CU (0.32 ms average residence beside three other pipelines).
  if (mfma) {
    hipLaunchKernelGGL(k_match_rank, dim3(npairs), dim3(MF_NT), 0, s, M, rec, keyrec, pairflag, fused ? 1 : 0);
    if (!fused)
      hipLaunchKernelGGL(k_match_scan_mfma, dim3(sgrid.x), dim3(MF_NT), 0, s, M, (uint32_t *)m->d_topk.p, (const uint32_t *)rec, (const uint32_t *)keyrec, (const uint32_t *)pairflag);
  }
  const int mf = mfma && !fused ? 1 : 0;
#define LAUNCH_MATCH(KT, LC)                                                                                              \
  do {                                                                                                                    \
    if (force == SCAN_WALK) {}                                                                                            \
    else if (fuse) hipLaunchKernelGGL((k_match_scan<KT, SCAN_FUSE>), sgrid, dim3(MATCH_NT), 0, s, M, (KT::T *)m->d_topk.p, slice_stride, force, 0, (const uint32_t *)nullptr);             \
    else if (M.qside) hipLaunchKernelGGL((k_match_scan<KT, SCAN_FISHEYE>), sgrid, dim3(MATCH_NT), 0, s, M, (KT::T *)m->d_topk.p, slice_stride, force, 0, (const uint32_t *)nullptr);  \
    else if (M.u_right) hipLaunchKernelGGL((k_match_scan<KT, SCAN_UR>), sgrid, dim3(MATCH_NT), 0, s, M, (KT::T *)m->d_topk.p, slice_stride, force, 0, (const uint32_t *)nullptr);     \
    else hipLaunchKernelGGL((k_match_scan<KT, SCAN_PLAIN>), sgrid, dim3(MATCH_NT), 0, s, M, (KT::T *)m->d_topk.p, slice_stride, force, mf, (const uint32_t *)pairflag);                 \
    if (nslices > 1 && force != SCAN_WALK) hipLaunchKernelGGL((k_topk_merge<KT>), dim3(qblocks, npairs), dim3(MATCH_NT), 0, s, M, (KT::T *)m->d_topk.p, slice_stride, nslices, force); \
    if (prof) MCHECK(m, hipEventRecord(pev[1], s));                                                                     \
    if (init_th_low >= 0)                                                                                                 \
      hipLaunchKernelGGL((k_init_resolve<KT>), dim3(npairs), dim3(64), 2 * (size_t)maxn + 16, s, M, (const KT::T *)m->d_topk.p, init_th_low); \
    else                                                                                                                  \
      hipLaunchKernelGGL((k_match_resolve<KT, LC>), dim3(npairs), rblock, lds, s, M, (const KT::T *)m->d_topk.p, maxn, force, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr);   \
  } while (0)
  if (fused) {   // Key32, candidates in LDS: the scan launches as above (they return at once for fused pairs), then the fused resolve
    hipLaunchKernelGGL((k_match_scan<Key32, SCAN_PLAIN>), sgrid, dim3(MATCH_NT), 0, s, M, (Key32::T *)m->d_topk.p, slice_stride, force, mf, (const uint32_t *)pairflag);
    if (prof) MCHECK(m, hipEventRecord(pev[1], s));
    hipLaunchKernelGGL((k_match_resolve<Key32, true, true>), dim3(npairs), rblock, lds, s, M, (const Key32::T *)m->d_topk.p, maxn, force, (const uint32_t *)rec,
                       (const uint32_t *)keyrec, (const uint32_t *)pairflag);
  } else if (k32) { if (ldscand) LAUNCH_MATCH(Key32, true); else LAUNCH_MATCH(Key32, false); }
  else     { if (ldscand) LAUNCH_MATCH(Key64, true); else LAUNCH_MATCH(Key64, false); }
#undef LAUNCH_MATCH
  if (prof) { MCHECK(m, hipEventRecord(pev[2], s)); m->prof_head++; m->ms_valid = true; }
  MCHECK(m, hipGetLastError());
  return 0;
}

// host-pointer fisheye-stereo options of one search (see MatchProblemSet)
struct StereoExt { int nleft; const int32_t *partner; const uint8_t *qside; int couple; int serial; int init_th_low = -1; const float *fuse_inv_sigma2 = nullptr; };

static int search_host(orbm_t *m, const orbm_frame_t *f, const orbm_queries_t *q, float nnratio, int th_dist, int use_second,
                       int32_t *slot, uint8_t *slot_obs, int32_t *match_of_query, int32_t *best_dist, const StereoExt *ext);

int orbm_search_by_projection(orbm_t *m, const orbm_frame_t *f, const orbm_queries_t *q, float nnratio, int th_dist,
                              int use_second, int32_t *slot, uint8_t *slot_obs, int32_t *match_of_query, int32_t *best_dist) {
  return search_host(m, f, q, nnratio, th_dist, use_second, slot, slot_obs, match_of_query, best_dist, nullptr);
}

static int search_host(orbm_t *m, const orbm_frame_t *f, const orbm_queries_t *q, float nnratio, int th_dist, int use_second,
                       int32_t *slot, uint8_t *slot_obs, int32_t *match_of_query, int32_t *best_dist, const StereoExt *ext) {
  if (!m || !f || !q || !slot || !slot_obs) return ORBX_E_ARG;
  const int n = f->n, nq = q->nq;
  if (n < 0 || nq < 0) return ORBX_E_ARG;
  if (n == 0 || nq == 0) {
    for (int i = 0; i < nq; i++) { if (match_of_query) match_of_query[i] = -1; if (best_dist) best_dist[i] = 256; }
    return 0;
  }
  MCHECK(m, hipSetDevice(m->device));
  hipStream_t s = m->stream;
  // All inputs travel as ONE pinned block -> one H2D copy, all outputs as one block -> one D2H copy, one synchronisation
  // (a dozen pageable copies would each stage and synchronise on their own).
  struct Part { const void *src; size_t bytes, off; };
  size_t total = 0;
  auto part = [&](const void *src, size_t bytes) { Part p{src, bytes, total}; total += (bytes + 255) & ~(size_t)255; return p; };
  const Part pKp = part(f->keys_un, sizeof(orbx_keypoint_t) * (size_t)n), pDesc = part(f->descriptors, 32 * (size_t)n);
  const Part pUr = part(f->u_right, f->u_right ? sizeof(float) * (size_t)n : 0);
  const Part pQd = part(q->descriptors, 32 * (size_t)nq), pQu = part(q->u, sizeof(float) * (size_t)nq), pQv = part(q->v, sizeof(float) * (size_t)nq);
  const Part pQr = part(q->radius, sizeof(float) * (size_t)nq), pQur = part(q->u_r, q->u_r ? sizeof(float) * (size_t)nq : 0);
  const Part pMinl = part(q->min_level, sizeof(int32_t) * (size_t)nq), pMaxl = part(q->max_level, sizeof(int32_t) * (size_t)nq);
  const Part pFl = part(q->flags, q->flags ? (size_t)nq : 0);
  const Part pPartner = part(ext ? ext->partner : nullptr, ext && ext->partner ? sizeof(int32_t) * (size_t)n : 0);
  const Part pSide = part(ext ? ext->qside : nullptr, ext && ext->qside ? (size_t)nq : 0);
  const size_t in_total = total;
  // in/out and out-only arrays behind the inputs (same device block; downloaded as one range)
  const size_t out_off = total;
  const Part pSlot = part(slot, sizeof(int32_t) * (size_t)n), pSobs = part(slot_obs, (size_t)n);
  const size_t upload_total = total;
  const Part pMoq = part(nullptr, sizeof(int32_t) * (size_t)nq), pBd = part(nullptr, sizeof(int32_t) * (size_t)nq), pNm = part(nullptr, sizeof(int32_t));
  const Part pAny = part(nullptr, ext && ext->couple == 2 ? (size_t)nq : 0);
  (void)in_total;
  if (m->pin_bytes < total) {
    if (m->pin) (void)hipHostFree(m->pin);
    m->pin = nullptr; m->pin_bytes = 0;
    MCHECK(m, hipHostMalloc(&m->pin, 2 * total, hipHostMallocDefault));   // grown rarely: a re-allocation costs tens of milliseconds
    m->pin_bytes = 2 * total;
  }
  MCHECK(m, m->d_block.reserve(m->d_block.bytes >= total ? total : 2 * total));
  uint8_t *hp = (uint8_t *)m->pin, *dp = (uint8_t *)m->d_block.p;
  for (const Part *p : {&pKp, &pDesc, &pUr, &pQd, &pQu, &pQv, &pQr, &pQur, &pMinl, &pMaxl, &pFl, &pPartner, &pSide, &pSlot, &pSobs})
    if (p->bytes) memcpy(hp + p->off, p->src, p->bytes);
  MCHECK(m, hipMemcpyAsync(dp, hp, upload_total, hipMemcpyHostToDevice, s));
  m->ext = {};
  if (ext) {
    m->ext.nleft = ext->nleft;
    m->ext.partner = ext->partner ? (const int32_t *)(dp + pPartner.off) : nullptr;
    m->ext.qside = ext->qside ? (const uint8_t *)(dp + pSide.off) : nullptr;
    m->ext.couple = ext->couple;
    m->ext.serial = ext->serial;
    m->ext.qany = ext->couple == 2 ? (uint8_t *)(dp + pAny.off) : nullptr;
    m->ext.init_th_low = ext->init_th_low;
    m->ext.fuse_inv_sigma2 = ext->fuse_inv_sigma2;   // host array of 16 floats, copied into the kernel arguments at launch
  }
  orbm_frame_t df = *f;
  df.keys_un = (const orbx_keypoint_t *)(dp + pKp.off);
  df.descriptors = dp + pDesc.off;
  df.u_right = f->u_right ? (const float *)(dp + pUr.off) : nullptr;
  orbm_queries_t dq = *q;
  dq.descriptors = dp + pQd.off;
  dq.u = (const float *)(dp + pQu.off);
  dq.v = (const float *)(dp + pQv.off);
  dq.radius = (const float *)(dp + pQr.off);
  dq.u_r = q->u_r ? (const float *)(dp + pQur.off) : nullptr;
  dq.min_level = (const int32_t *)(dp + pMinl.off);
  dq.max_level = (const int32_t *)(dp + pMaxl.off);
  dq.flags = q->flags ? (const uint8_t *)(dp + pFl.off) : nullptr;
  // The radii are on the host here: the walk-or-scan decision (pair_walks) is taken now and only the chosen kernels are launched.
  if (m->scan_mode == SCAN_AUTO) {
    const float iw = (float)ORBM_GRID_COLS / (f->max_x - f->min_x), ih = (float)ORBM_GRID_ROWS / (f->max_y - f->min_y);
    bool big = n > WALK_MAX_N;
    for (int i = 0; i < nq && !big; i++) {
      if (q->flags && !(q->flags[i] & 1)) continue;
      const float u = q->u[i], v = q->v[i], r = q->radius[i];
      const int cx0 = std::max(0, (int)floorf((u - f->min_x - r) * iw)), cx1 = std::min(63, (int)ceilf((u - f->min_x + r) * iw));
      const int cy0 = std::max(0, (int)floorf((v - f->min_y - r) * ih)), cy1 = std::min(47, (int)ceilf((v - f->min_y + r) * ih));
      if (cx0 < 64 && cx1 >= 0 && cy0 < 48 && cy1 >= 0 && (cx1 - cx0 + 1) * (cy1 - cy0 + 1) > WALK_MAX_CELLS) big = true;
    }
    m->next_scan_mode = big ? SCAN_DENSE : SCAN_WALK;
  }
  int rc = orbm_search_by_projection_batch_device(m, &df, n, nullptr, 0, &dq, nq, nullptr, 0, 1, nnratio, th_dist, use_second,
                                                  (int32_t *)(dp + pSlot.off), dp + pSobs.off, (int32_t *)(dp + pMoq.off),
                                                  (int32_t *)(dp + pBd.off), (int32_t *)(dp + pNm.off), s);
  if (rc < 0) return rc;
  MCHECK(m, hipMemcpyAsync(hp + out_off, dp + out_off, pNm.off + sizeof(int32_t) - out_off, hipMemcpyDeviceToHost, s));
  MCHECK(m, hipStreamSynchronize(s));
  memcpy(slot, hp + pSlot.off, pSlot.bytes);
  memcpy(slot_obs, hp + pSobs.off, pSobs.bytes);
  if (match_of_query) memcpy(match_of_query, hp + pMoq.off, pMoq.bytes);
  if (best_dist) memcpy(best_dist, hp + pBd.off, pBd.bytes);
  int32_t nm = 0;
  memcpy(&nm, hp + pNm.off, sizeof(nm));
  return nm;
}

// cv::Mat products of ORBmatcher.cc:2038-2047, :2072 restated: a 3x3 * 3x1 `A*B + C` MatExpr is a single
// cv::gemm whose small-matrix float path forms a0*b0 + a1*b1 + a2*b2 in float, then adds C; `-A.t()*B` takes the
// generic path that accumulates in double.  [OPENCV-UNVERIFIED], identical in the test oracle.
static void mat3_mul_add(const float *R, const float *x, const float *t, float *out) {  // R: row-major, row stride 4
  for (int i = 0; i < 3; i++) {
    float t0 = R[i * 4 + 0] * x[0] + R[i * 4 + 1] * x[1] + R[i * 4 + 2] * x[2];
    out[i] = (float)((double)t0 + (double)t[i]);
  }
}

int orbm_search_by_projection_last_frame_batch_device(orbm_t *m, const orbm_frame_t *cur0, int frame_stride, const int32_t *d_frame_n,
                                                      int frame_n_stride, const orbm_last_frame_t *last0, int last_stride,
                                                      const int32_t *d_last_n, int last_n_stride, int npairs, const float *sf, int nlevels,
                                                      int cam_type, const float *cam_params, float mb, float mbf, float th, int bMono,
                                                      int checkOri, int32_t *d_slot, uint8_t *d_slot_obs, int32_t *d_moq,
                                                      int32_t *d_nmatches, void *stream_) {
  if (!m || !cur0 || !last0 || !sf || !cam_params || npairs <= 0 || !d_slot || !d_slot_obs || !d_nmatches) return ORBX_E_ARG;
  if (nlevels < 1 || nlevels > 16 || (cam_type != 0 && cam_type != 1)) return ORBX_E_ARG;
  if (!last0->has_mp || !last0->Xw || !last0->mpdesc || !last0->last_keys || !last0->Tcw || !last0->Tlw) return ORBX_E_ARG;
  const int maxq = d_last_n ? last_stride : last0->n;
  if (maxq <= 0 || last_stride < maxq) return ORBX_E_ARG;
  MCHECK(m, hipSetDevice(m->device));
  hipStream_t s = (hipStream_t)stream_;
  // scratch: the query arrays the projection kernel writes (29 B per query) - grows on demand, not on the steady-state path
  const size_t nqa = (size_t)(npairs - 1) * last_stride + maxq, nq4 = (nqa + 63) & ~(size_t)63;
  const size_t need = nq4 * (4 * sizeof(float) + 3 * sizeof(int32_t) + 1);
  if (need > m->d_lfq.bytes) {
    MCHECK(m, hipStreamSynchronize(s));
    MCHECK(m, m->d_lfq.reserve(need + (need >> 2)));
  }
  float *qf = (float *)m->d_lfq.p;
  int32_t *qi = (int32_t *)(qf + 4 * nq4);
  uint8_t *qfl = (uint8_t *)(qi + 3 * nq4);
  int32_t *moq = d_moq ? d_moq : qi + 2 * nq4;
  LastFrameParams P;
  memset(&P, 0, sizeof(P));
  P.has_mp = last0->has_mp; P.Xw = last0->Xw; P.last_kp = reinterpret_cast<const float *>(last0->last_keys); P.obs = last0->obs;
  P.Tcw = last0->Tcw; P.Tlw = last0->Tlw;
  P.last_stride = last_stride; P.last_n = d_last_n; P.last_n_stride = last_n_stride; P.last_n_const = last0->n;
  P.min_x = cur0->min_x; P.max_x = cur0->max_x; P.min_y = cur0->min_y; P.max_y = cur0->max_y;
  for (int l = 0; l < nlevels; l++) P.sf[l] = sf[l];
  P.nlevels = nlevels;
  P.cam_type = cam_type;
  for (int k = 0; k < (cam_type == 0 ? 4 : 8); k++) P.cam[k] = cam_params[k];
  P.mb = mb; P.mbf = mbf; P.th = th; P.bMono = bMono ? 1 : 0;
  P.qu = qf; P.qv = qf + nq4; P.qr = qf + 2 * nq4; P.qur = qf + 3 * nq4;
  P.qminl = qi; P.qmaxl = qi + nq4; P.qflags = qfl;
  hipLaunchKernelGGL(k_lastframe_project, dim3((last_stride + 255) / 256, npairs), dim3(256), 0, s, P);
  orbm_queries_t q;
  q.nq = last0->n; q.descriptors = last0->mpdesc; q.u = P.qu; q.v = P.qv; q.radius = P.qr;
  q.min_level = P.qminl; q.max_level = P.qmaxl; q.u_r = cur0->u_right ? P.qur : nullptr; q.flags = P.qflags;
  int rc = orbm_search_by_projection_batch_device(m, cur0, frame_stride, d_frame_n, frame_n_stride, &q, last_stride, d_last_n, last_n_stride, npairs,
                                                  0.f, ORBM_TH_HIGH, 0, d_slot, d_slot_obs, moq, nullptr, d_nmatches, s);   // :2148-2162
  if (rc < 0) return rc;
  if (checkOri) {                                                                                                            // :2177-2185, :2263-2286
    RotPruneParams R;
    memset(&R, 0, sizeof(R));
    R.last_kp = reinterpret_cast<const float *>(last0->last_keys);
    R.last_stride = last_stride; R.last_n = d_last_n; R.last_n_stride = last_n_stride; R.last_n_const = last0->n;
    R.cur_kp = reinterpret_cast<const float *>(cur0->keys_un); R.frame_stride = frame_stride;
    R.moq = moq; R.slot = d_slot; R.slot_obs = d_slot_obs; R.nmatches = d_nmatches;
    hipLaunchKernelGGL(k_rot_prune, dim3(npairs), dim3(256), 0, s, R);
  }
  MCHECK(m, hipGetLastError());
  return 0;
}

int orbm_search_by_projection_last_frame(orbm_t *m, const orbm_frame_t *cur, const float *sf, int nlevels, int nLast,
                                         const uint8_t *has_mp, const float *Xw, const uint8_t *mpdesc,
                                         const orbx_keypoint_t *last_keys, const uint8_t *obs, const float *Tcw,
                                         const float *Tlw, int cam_type, const float *cam_params, float mb, float mbf,
                                         float th, int bMono, int checkOri, int32_t *slot, uint8_t *slot_obs) {
  if (!m || !cur || !sf || nLast < 0 || !Tcw || !Tlw || !cam_params || !slot || !slot_obs) return ORBX_E_ARG;
  if (nLast > 0 && (!has_mp || !Xw || !mpdesc || !last_keys)) return ORBX_E_ARG;
  if (nlevels < 1 || nlevels > 16) return ORBX_E_ARG;
  const int n = cur->n;
  if (n < 0) return ORBX_E_ARG;
  if (n == 0 || nLast == 0) return 0;
  if (!cur->keys_un || !cur->descriptors) return ORBX_E_ARG;
  for (int i = 0; i < nLast; i++)
    if (has_mp[i] && (last_keys[i].octave < 0 || last_keys[i].octave >= nlevels)) return ORBX_E_ARG;
  MCHECK(m, hipSetDevice(m->device));
  hipStream_t s = m->stream;
  // one pinned block up, one block down, one synchronisation (as search_host); projection, search and pruning on the device
  struct Part { const void *src; size_t bytes, off; };
  size_t total = 0;
  auto part = [&](const void *src, size_t bytes) { Part p{src, bytes, total}; total += (bytes + 255) & ~(size_t)255; return p; };
  const Part pKp = part(cur->keys_un, sizeof(orbx_keypoint_t) * (size_t)n), pDesc = part(cur->descriptors, 32 * (size_t)n);
  const Part pUr = part(cur->u_right, cur->u_right ? sizeof(float) * (size_t)n : 0);
  const Part pHas = part(has_mp, (size_t)nLast), pXw = part(Xw, 3 * sizeof(float) * (size_t)nLast), pMd = part(mpdesc, 32 * (size_t)nLast);
  const Part pLk = part(last_keys, sizeof(orbx_keypoint_t) * (size_t)nLast), pObs = part(obs, obs ? (size_t)nLast : 0);
  const Part pTc = part(Tcw, 16 * sizeof(float)), pTl = part(Tlw, 16 * sizeof(float));
  const size_t out_off = total;
  const Part pSlot = part(slot, sizeof(int32_t) * (size_t)n), pSobs = part(slot_obs, (size_t)n);
  const size_t upload_total = total;
  const Part pNm = part(nullptr, sizeof(int32_t));
  if (m->pin_bytes < total) {
    if (m->pin) (void)hipHostFree(m->pin);
    m->pin = nullptr; m->pin_bytes = 0;
    MCHECK(m, hipHostMalloc(&m->pin, 2 * total, hipHostMallocDefault));   // grown rarely: a re-allocation costs tens of milliseconds
    m->pin_bytes = 2 * total;
  }
  MCHECK(m, m->d_block.reserve(m->d_block.bytes >= total ? total : 2 * total));
  uint8_t *hp = (uint8_t *)m->pin, *dp = (uint8_t *)m->d_block.p;
  for (const Part *p : {&pKp, &pDesc, &pUr, &pHas, &pXw, &pMd, &pLk, &pObs, &pTc, &pTl, &pSlot, &pSobs})
    if (p->bytes) memcpy(hp + p->off, p->src, p->bytes);
  MCHECK(m, hipMemcpyAsync(dp, hp, upload_total, hipMemcpyHostToDevice, s));
  orbm_frame_t df = *cur;
  df.keys_un = (const orbx_keypoint_t *)(dp + pKp.off);
  df.descriptors = dp + pDesc.off;
  df.u_right = cur->u_right ? (const float *)(dp + pUr.off) : nullptr;
  orbm_last_frame_t dl;
  memset(&dl, 0, sizeof(dl));
  dl.n = nLast;
  dl.has_mp = dp + pHas.off; dl.Xw = (const float *)(dp + pXw.off); dl.mpdesc = dp + pMd.off;
  dl.last_keys = (const orbx_keypoint_t *)(dp + pLk.off);
  dl.obs = obs ? dp + pObs.off : nullptr;
  dl.Tcw = (const float *)(dp + pTc.off); dl.Tlw = (const float *)(dp + pTl.off);
  m->ext = {};
  // The windows are th * scale factor of the last keypoint's octave (ORBmatcher.cc:2109): known here, so the walk-or-scan decision
  // is taken on the host from the largest one (an upper bound of every query's cell window) and only the chosen kernels are launched
  if (m->scan_mode == SCAN_AUTO && n <= WALK_MAX_N && cur->max_x > cur->min_x && cur->max_y > cur->min_y) {
    float rmax = 0.f;
    for (int i = 0; i < nLast; i++)
      if (has_mp[i]) rmax = std::max(rmax, th * sf[last_keys[i].octave]);
    const float iw = (float)ORBM_GRID_COLS / (cur->max_x - cur->min_x), ih = (float)ORBM_GRID_ROWS / (cur->max_y - cur->min_y);
    const long long cx = std::min<long long>(ORBM_GRID_COLS, (long long)ceilf(2.f * rmax * iw) + 2), cy = std::min<long long>(ORBM_GRID_ROWS, (long long)ceilf(2.f * rmax * ih) + 2);
    if (cx * cy <= WALK_MAX_CELLS) m->next_scan_mode = SCAN_WALK;
  }
  int rc = orbm_search_by_projection_last_frame_batch_device(m, &df, n, nullptr, 0, &dl, nLast, nullptr, 0, 1, sf, nlevels, cam_type, cam_params, mb, mbf,
                                                             th, bMono, checkOri, (int32_t *)(dp + pSlot.off), dp + pSobs.off, nullptr,
                                                             (int32_t *)(dp + pNm.off), s);
  if (rc < 0) return rc;
  MCHECK(m, hipMemcpyAsync(hp + out_off, dp + out_off, pNm.off + sizeof(int32_t) - out_off, hipMemcpyDeviceToHost, s));
  MCHECK(m, hipStreamSynchronize(s));
  memcpy(slot, hp + pSlot.off, pSlot.bytes);
  memcpy(slot_obs, hp + pSobs.off, pSobs.bytes);
  int32_t nm = 0;
  memcpy(&nm, hp + pNm.off, sizeof(nm));
  return nm;
}

static int prune_by_rotation(int nq, const int32_t *moq, const float *query_angle, const orbx_keypoint_t *keys, int32_t *slot,
                             uint8_t *slot_obs, int nmatches);

// ---- fisheye-stereo frames (Nleft != -1) -------------------------------------------------------------------------
// One problem over the concatenated keypoints [left ; right] with two queries per map point (even = left image,
// odd = right image): the device core handles the image restriction, the stereo-partner writes and the pair coupling.
static int build_partner(int n, int n_left, const int32_t *l2r, const int32_t *r2l, std::vector<int32_t> &partner) {
  partner.assign((size_t)n, -1);
  bool any = false;
  for (int i = 0; i < n_left; i++)
    if (l2r && l2r[i] >= 0) { if (n_left + l2r[i] >= n) return ORBX_E_ARG; partner[i] = n_left + l2r[i]; any = true; }
  for (int j = 0; j < n - n_left; j++)
    if (r2l && r2l[j] >= 0) { if (r2l[j] >= n_left) return ORBX_E_ARG; partner[n_left + j] = r2l[j]; any = true; }
  return any ? 1 : 0;
}

int orbm_search_by_projection_fisheye(orbm_t *m, const orbm_frame_t *f, int n_left, const int32_t *left_to_right,
                                      const int32_t *right_to_left, const orbm_queries_t *q, float nnratio, int th_dist,
                                      int32_t *slot, uint8_t *slot_obs, int32_t *match_of_query, int32_t *best_dist) {
  if (!m || !f || !q || n_left < 0 || n_left > f->n || (q->nq & 1)) return ORBX_E_ARG;
  std::vector<int32_t> partner;
  const int anyp = build_partner(f->n, n_left, left_to_right, right_to_left, partner);
  if (anyp < 0) return anyp;
  std::vector<uint8_t> side((size_t)q->nq);
  bool release = false;  // a taking query without observations can release a claim through a partner write
  for (int i = 0; i < q->nq; i++) {
    side[i] = (uint8_t)(i & 1);
    const uint8_t fl = q->flags ? q->flags[i] : (uint8_t)3;
    if ((fl & 1) && !(fl & 2)) release = true;
  }
  StereoExt ext{n_left, anyp ? partner.data() : nullptr, side.data(), 1, (anyp && release) ? 1 : 0};
  return search_host(m, f, q, nnratio, th_dist, 1, slot, slot_obs, match_of_query, best_dist, &ext);
}

int orbm_search_by_projection_last_frame_fisheye(orbm_t *m, const orbm_frame_t *cur, int n_left, const float *sf, int nlevels,
                                                 int nLast, const uint8_t *has_mp, const float *Xw, const uint8_t *mpdesc,
                                                 const orbx_keypoint_t *last_keys, const uint8_t *obs, const float *Tcw,
                                                 const float *Tlw, const float *Trl, int cam_type, const float *cam_params,
                                                 float mb, float th, int bMono, int checkOri, int32_t *slot, uint8_t *slot_obs) {
  if (!m || !cur || !sf || nLast < 0 || !Tcw || !Tlw || !Trl || !cam_params || !slot || !slot_obs) return ORBX_E_ARG;
  if (n_left < 0 || n_left > cur->n) return ORBX_E_ARG;
  if (nLast > 0 && (!has_mp || !Xw || !mpdesc || !last_keys)) return ORBX_E_ARG;
  const float tcw[3] = {Tcw[3], Tcw[7], Tcw[11]}, tlw[3] = {Tlw[3], Tlw[7], Tlw[11]}, trl[3] = {Trl[3], Trl[7], Trl[11]};
  float twc[3], tlc[3];
  for (int i = 0; i < 3; i++) {  // twc = -Rcw.t()*tcw, :2041
    double s = 0;
    for (int k = 0; k < 3; k++) s += (double)Tcw[k * 4 + i] * (double)tcw[k];
    twc[i] = (float)(s * -1.0);
  }
  mat3_mul_add(Tlw, twc, tlw, tlc);
  const bool bForward = tlc[2] > mb && !bMono, bBackward = -tlc[2] > mb && !bMono;
  const int nq = 2 * nLast;
  std::vector<float> u(nq, 0.f), v(nq, 0.f), rad(nq, 0.f), qangle(nq, 0.f);
  std::vector<int32_t> minl(nq, -1), maxl(nq, -1), moq(nq, -1);
  std::vector<uint8_t> flags(nq, 0), side(nq), qd((size_t)nq * 32);
  for (int i = 0; i < nLast; i++) {
    side[2 * i] = 0; side[2 * i + 1] = 1;
    memcpy(&qd[(size_t)(2 * i) * 32], mpdesc + (size_t)i * 32, 32);
    memcpy(&qd[(size_t)(2 * i + 1) * 32], mpdesc + (size_t)i * 32, 32);
    qangle[2 * i] = qangle[2 * i + 1] = last_keys[i].angle;
    if (!has_mp[i]) continue;
    float x3Dc[3], x3Dr[3];
    mat3_mul_add(Tcw, Xw + 3 * i, tcw, x3Dc);                  // :2072
    const float invzc = (float)(1.0 / (double)x3Dc[2]);        // :2076
    if (invzc < 0) continue;
    float ux, vy;
    orbm_project(cam_type, cam_params, x3Dc[0], x3Dc[1], x3Dc[2], &ux, &vy);
    if (ux < cur->min_x || ux > cur->max_x) continue;          // :2094-2097
    if (vy < cur->min_y || vy > cur->max_y) continue;
    const int nLastOctave = last_keys[i].octave;
    if (nLastOctave < 0 || nLastOctave >= nlevels) return ORBX_E_ARG;
    mat3_mul_add(Trl, x3Dc, trl, x3Dr);                        // :2190
    float uxr, vyr;
    orbm_project(cam_type, cam_params, x3Dr[0], x3Dr[1], x3Dr[2], &uxr, &vyr);
    const uint8_t fl = (uint8_t)(1u | ((obs ? (obs[i] & 1u) : 1u) << 1));
    for (int h = 0; h < 2; h++) {
      const int j = 2 * i + h;
      u[j] = h ? uxr : ux; v[j] = h ? vyr : vy;
      rad[j] = th * sf[nLastOctave];                           // :2105, :2197
      if (bForward) { minl[j] = nLastOctave; maxl[j] = -1; }   // :2113-2118, :2201-2206
      else if (bBackward) { minl[j] = 0; maxl[j] = nLastOctave; }
      else { minl[j] = nLastOctave - 1; maxl[j] = nLastOctave + 1; }
      flags[j] = fl;
    }
  }
  orbm_queries_t q;
  q.nq = nq; q.descriptors = qd.data(); q.u = u.data(); q.v = v.data(); q.radius = rad.data();
  q.min_level = minl.data(); q.max_level = maxl.data(); q.u_r = nullptr; q.flags = flags.data();
  orbm_frame_t f = *cur;
  f.u_right = nullptr;  // Nleft != -1: no mvuRight test (:2139)
  StereoExt ext{n_left, nullptr, side.data(), 2, 0};
  int nmatches = search_host(m, &f, &q, 0.f, ORBM_TH_HIGH, 0, slot, slot_obs, moq.data(), nullptr, &ext);
  if (nmatches < 0) return nmatches;
  for (int j = 0; j < nq; j++)  // the device stored query indices; the caller's ids are last-frame indices (in order: last writer wins)
    if (moq[j] >= 0) slot[moq[j]] = j >> 1;
  if (!checkOri) return nmatches;
  return prune_by_rotation(nq, moq.data(), qangle.data(), cur->keys_un, slot, slot_obs, nmatches);
}

// Rotation-consistency pruning shared by the projection searches (e.g. ORBmatcher.cc:2177-2185 + :2263-2286).
static int prune_by_rotation(int nq, const int32_t *moq, const float *query_angle, const orbx_keypoint_t *keys, int32_t *slot,
                             uint8_t *slot_obs, int nmatches) {
  std::vector<std::vector<int>> rotHist(ORBM_HISTO_LENGTH);
  const float factor = 1.0f / ORBM_HISTO_LENGTH;
  for (int i = 0; i < nq; i++) {
    if (moq[i] < 0) continue;
    float rot = query_angle[i] - keys[moq[i]].angle;
    if ((double)rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == ORBM_HISTO_LENGTH) bin = 0;
    if (bin >= 0 && bin < ORBM_HISTO_LENGTH) rotHist[bin].push_back(moq[i]);
  }
  int sizes[ORBM_HISTO_LENGTH], ind1, ind2, ind3;
  for (int i = 0; i < ORBM_HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
  orbm_three_maxima(sizes, ORBM_HISTO_LENGTH, &ind1, &ind2, &ind3);
  for (int i = 0; i < ORBM_HISTO_LENGTH; i++) {
    if (i == ind1 || i == ind2 || i == ind3) continue;
    for (int idx : rotHist[i]) { slot[idx] = -1; slot_obs[idx] = 0; nmatches--; }
  }
  return nmatches;
}

int orbm_search_by_projection_keyframe(orbm_t *m, const orbm_frame_t *cur, const float *sf, int nlevels, float logScaleFactor, int nKF,
                                       const uint8_t *valid, const float *Xw, const uint8_t *mpdesc, const float *kf_angle,
                                       const float *max_dist, const float *min_dist, const float *Tcw, int cam_type,
                                       const float *cam_params, float th, int ORBdist, int checkOri, int32_t *slot, uint8_t *slot_obs) {
  if (!m || !cur || !sf || nlevels < 1 || nKF < 0 || !Tcw || !cam_params || !slot || !slot_obs) return ORBX_E_ARG;
  if (nKF > 0 && (!valid || !Xw || !mpdesc || !kf_angle || !max_dist || !min_dist)) return ORBX_E_ARG;
  const float tcw[3] = {Tcw[3], Tcw[7], Tcw[11]};
  float Ow[3];  // Ow = -Rcw.t()*tcw, :2297
  for (int i = 0; i < 3; i++) {
    double acc = 0;
    for (int k = 0; k < 3; k++) acc += (double)Tcw[4 * k + i] * (double)tcw[k];
    Ow[i] = (float)(-acc);
  }
  std::vector<float> u(nKF, 0.f), v(nKF, 0.f), rad(nKF, 0.f);
  std::vector<int32_t> minl(nKF, -1), maxl(nKF, -1), moq(nKF, -1);
  std::vector<uint8_t> flags(nKF, 0);
  for (int i = 0; i < nKF; i++) {
    if (!valid[i]) continue;
    const float *x3Dw = Xw + 3 * i;
    float x3Dc[3];
    mat3_mul_add(Tcw, x3Dw, tcw, x3Dc);                                        // :2317
    float ux, vy;
    orbm_project(cam_type, cam_params, x3Dc[0], x3Dc[1], x3Dc[2], &ux, &vy);     // :2319
    if (ux < cur->min_x || ux > cur->max_x) continue;                           // :2321-2324
    if (vy < cur->min_y || vy > cur->max_y) continue;
    double n2 = 0;                                                               // cv::norm(x3Dw-Ow), :2327-2328
    for (int k = 0; k < 3; k++) { const float po = x3Dw[k] - Ow[k]; n2 += (double)po * (double)po; }
    const float dist3D = (float)sqrt(n2);
    const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
    if (dist3D < minDistance || dist3D > maxDistance) continue;                 // :2334-2335
    const float ratio = max_dist[i] / dist3D;                                    // MapPoint::PredictScale, MapPoint.cc:587-602
    int lvl = (int)ceilf(logf(ratio) / logScaleFactor);
    lvl = lvl < 0 ? 0 : (lvl >= nlevels ? nlevels - 1 : lvl);
    u[i] = ux; v[i] = vy;
    rad[i] = th * sf[lvl];                                                       // :2340
    minl[i] = lvl - 1; maxl[i] = lvl + 1;                                        // :2342
    flags[i] = 3;
  }
  orbm_queries_t q;
  q.nq = nKF; q.descriptors = mpdesc; q.u = u.data(); q.v = v.data(); q.radius = rad.data();
  q.min_level = minl.data(); q.max_level = maxl.data(); q.u_r = nullptr; q.flags = flags.data();
  orbm_frame_t f = *cur;
  f.u_right = nullptr;  // no stereo gate in this overload
  int nmatches = orbm_search_by_projection(m, &f, &q, 0.f, ORBdist, 0, slot, slot_obs, moq.data(), nullptr);
  if (nmatches < 0 || !checkOri) return nmatches;
  return prune_by_rotation(nKF, moq.data(), kf_angle, cur->keys_un, slot, slot_obs, nmatches);
}

int orbm_search_by_projection_sim3(orbm_t *m, const orbm_frame_t *kf, const float *sf, int nlevels, float logScaleFactor, int nP,
                                   const uint8_t *valid, const float *Xw, const float *normal, const uint8_t *mpdesc,
                                   const float *max_dist, const float *min_dist, const float *Scw, const float *cam, int th,
                                   float ratioHamming, int32_t *slot, uint8_t *slot_obs) {
  return orbm_search_by_projection_sim3_cam(m, kf, sf, nlevels, logScaleFactor, nP, valid, Xw, normal, mpdesc, max_dist, min_dist, Scw, 0, cam, th,
                                            ratioHamming, slot, slot_obs);
}

int orbm_search_by_projection_sim3_cam(orbm_t *m, const orbm_frame_t *kf, const float *sf, int nlevels, float logScaleFactor, int nP,
                                       const uint8_t *valid, const float *Xw, const float *normal, const uint8_t *mpdesc,
                                       const float *max_dist, const float *min_dist, const float *Scw, int cam_type, const float *cam, int th,
                                       float ratioHamming, int32_t *slot, uint8_t *slot_obs) {
  if (!m || !kf || !sf || nlevels < 1 || nP < 0 || !Scw || !cam || !slot || !slot_obs || (cam_type != 0 && cam_type != 1)) return ORBX_E_ARG;
  if (nP > 0 && (!valid || !Xw || !normal || !mpdesc || !max_dist || !min_dist)) return ORBX_E_ARG;
  // Decompose Scw, :498-503: scw = sqrt(row0 . row0) (Mat::dot accumulates in double); Rcw = sRcw/scw, tcw = t/scw
  double dot = 0;
  for (int k = 0; k < 3; k++) dot += (double)Scw[k] * (double)Scw[k];
  const float scw = (float)sqrt(dot);
  const double inv = 1. / (double)scw;
  float T[16] = {0};  // row-major [Rcw | tcw] so that the shared 3x3*3x1+t helper applies
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) T[4 * i + j] = (float)((double)Scw[4 * i + j] * inv);
    T[4 * i + 3] = (float)((double)Scw[4 * i + 3] * inv);
  }
  const float tcw[3] = {T[3], T[7], T[11]};
  float Ow[3];
  for (int i = 0; i < 3; i++) {
    double acc = 0;
    for (int k = 0; k < 3; k++) acc += (double)T[4 * k + i] * (double)tcw[k];
    Ow[i] = (float)(-acc);
  }
  std::vector<float> u(nP, 0.f), v(nP, 0.f), rad(nP, 0.f);
  std::vector<int32_t> minl(nP, -1), maxl(nP, -1);
  std::vector<uint8_t> flags(nP, 0);
  for (int i = 0; i < nP; i++) {
    if (!valid[i]) continue;
    const float *p3Dw = Xw + 3 * i;
    float p3Dc[3];
    mat3_mul_add(T, p3Dw, tcw, p3Dc);                                         // :523
    if ((double)p3Dc[2] < 0.0) continue;                                      // :526
    float ux, vy;
    orbm_project(cam_type, cam, p3Dc[0], p3Dc[1], p3Dc[2], &ux, &vy);         // :534, pKF->mpCamera->project
    if (!(ux >= kf->min_x && ux < kf->max_x && vy >= kf->min_y && vy < kf->max_y)) continue;  // KeyFrame::IsInImage, KeyFrame.cc:844-847
    float PO[3];
    double n2 = 0, pd = 0;
    for (int k = 0; k < 3; k++) { PO[k] = p3Dw[k] - Ow[k]; n2 += (double)PO[k] * (double)PO[k]; }
    const float dist = (float)sqrt(n2);                                       // cv::norm, :544
    if (dist < 0.8f * min_dist[i] || dist > 1.2f * max_dist[i]) continue;     // :546
    for (int k = 0; k < 3; k++) pd += (double)PO[k] * (double)normal[3 * i + k];
    if (pd < 0.5 * (double)dist) continue;                                    // :552
    const float ratio = max_dist[i] / dist;                                   // MapPoint::PredictScale(dist, pKF), MapPoint.cc:570-585
    int lvl = (int)ceilf(logf(ratio) / logScaleFactor);
    lvl = lvl < 0 ? 0 : (lvl >= nlevels ? nlevels - 1 : lvl);
    u[i] = ux; v[i] = vy;
    rad[i] = (float)th * sf[lvl];                                             // :558
    minl[i] = lvl - 1; maxl[i] = lvl;                                         // :579-580
    flags[i] = 3;
  }
  orbm_queries_t q;
  q.nq = nP; q.descriptors = mpdesc; q.u = u.data(); q.v = v.data(); q.radius = rad.data();
  q.min_level = minl.data(); q.max_level = maxl.data(); q.u_r = nullptr; q.flags = flags.data();
  orbm_frame_t f = *kf;
  f.u_right = nullptr;
  // bestDist <= TH_LOW*ratioHamming with an int on the left: same as bestDist <= floor(50.f*ratioHamming)
  const int th_dist = (int)floorf((float)ORBM_TH_LOW * ratioHamming);
  return orbm_search_by_projection(m, &f, &q, 0.f, th_dist, 0, slot, slot_obs, nullptr, nullptr);
}

// ---- Fuse (ORBmatcher.cc:1425-1658 and :1660-1786) ------------------------------------------------------------------------
// The search of one map point does not depend on the others (no claim: a keypoint that already holds a map point stays a
// candidate, :1622-1640), so the device returns (bestIdx, bestDist) per map point and the caller applies the
// Replace / AddObservation bookkeeping on its objects in order.  Projection and gates on the host in fp32 as written.
static int fuse_core(orbm_t *m, const orbm_frame_t *kf, const float *sf, const float *inv_sigma2, int nlevels, float logScaleFactor, int nP,
                     const uint8_t *valid, const float *Xw, const float *normal, const uint8_t *mpdesc, const float *max_dist,
                     const float *min_dist, const float *T /* row-major 4x4 [Rcw | tcw] */, const float *Ow, int cam_type,
                     const float *cam, float bf, float th, bool chi2, int32_t *best_idx, int32_t *best_dist) {
  const float tcw[3] = {T[3], T[7], T[11]};
  std::vector<float> u(nP, 0.f), v(nP, 0.f), rad(nP, 0.f), ur(nP, 0.f);
  std::vector<int32_t> minl(nP, -1), maxl(nP, -1);
  std::vector<uint8_t> flags(nP, 0);
  for (int i = 0; i < nP; i++) {
    best_idx[i] = -1; best_dist[i] = 256;
    if (!valid[i]) continue;
    const float *p3Dw = Xw + 3 * i;
    float p3Dc[3];
    mat3_mul_add(T, p3Dw, tcw, p3Dc);                                         // :1472 / :1694
    if (p3Dc[2] < 0.0f) continue;                                             // :1475 / :1697
    const float invz = 1 / p3Dc[2];                                           // :1481
    float ux, vy;
    orbm_project(cam_type, cam, p3Dc[0], p3Dc[1], p3Dc[2], &ux, &vy);         // :1487 / :1704
    if (!(ux >= kf->min_x && ux < kf->max_x && vy >= kf->min_y && vy < kf->max_y)) continue;  // KeyFrame::IsInImage, KeyFrame.cc:844-847
    float PO[3];
    double n2 = 0, pd = 0;
    for (int k = 0; k < 3; k++) { PO[k] = p3Dw[k] - Ow[k]; n2 += (double)PO[k] * (double)PO[k]; }
    const float dist3D = (float)sqrt(n2);                                     // cv::norm, :1502 / :1715
    if (dist3D < 0.8f * min_dist[i] || dist3D > 1.2f * max_dist[i]) continue; // MapPoint.cc:552-563, :1505 / :1718
    for (int k = 0; k < 3; k++) pd += (double)PO[k] * (double)normal[3 * i + k];
    if (pd < 0.5 * (double)dist3D) continue;                                  // :1514 / :1724
    const float ratio = max_dist[i] / dist3D;                                 // MapPoint::PredictScale, MapPoint.cc:570-585
    int lvl = (int)ceilf(logf(ratio) / logScaleFactor);
    lvl = lvl < 0 ? 0 : (lvl >= nlevels ? nlevels - 1 : lvl);
    u[i] = ux; v[i] = vy; ur[i] = ux - bf * invz;                             // :1495
    rad[i] = th * sf[lvl];                                                    // :1524 / :1731
    minl[i] = lvl - 1; maxl[i] = lvl;                                         // :1555 / :1752
    flags[i] = 1;                                                             // takes part, never claims
  }
  orbm_queries_t q;
  q.nq = nP; q.descriptors = mpdesc; q.u = u.data(); q.v = v.data(); q.radius = rad.data();
  q.min_level = minl.data(); q.max_level = maxl.data(); q.u_r = ur.data(); q.flags = flags.data();
  orbm_frame_t f = *kf;
  std::vector<float> no_ur;
  if (chi2 && !f.u_right) { no_ur.assign((size_t)std::max(kf->n, 1), -1.0f); f.u_right = no_ur.data(); }
  if (!chi2) f.u_right = nullptr;
  std::vector<int32_t> slot((size_t)std::max(kf->n, 1), -1);
  std::vector<uint8_t> sobs((size_t)std::max(kf->n, 1), 0);
  float is2[16] = {0};
  StereoExt ext{};
  ext.nleft = kf->n;
  if (chi2) { for (int l = 0; l < nlevels && l < 16; l++) is2[l] = inv_sigma2[l]; ext.fuse_inv_sigma2 = is2; }
  const int rc = search_host(m, &f, &q, 0.f, ORBM_TH_LOW, 0, slot.data(), sobs.data(), best_idx, best_dist, &ext);
  if (rc < 0) return rc;
  int nFused = 0;
  for (int i = 0; i < nP; i++) nFused += best_idx[i] >= 0 ? 1 : 0;           // bestDist <= TH_LOW, :1622 / :1767
  return nFused;
}

int orbm_fuse(orbm_t *m, const orbm_frame_t *kf, const float *scale_factors, const float *inv_level_sigma2, int nlevels,
              float log_scale_factor, int nP, const uint8_t *valid, const float *Xw, const float *normal, const uint8_t *mpdesc,
              const float *max_dist, const float *min_dist, const float *Tcw, const float *Ow, int cam_type, const float *cam_params,
              float bf, float th, int32_t *best_idx, int32_t *best_dist) {
  if (!m || !kf || !scale_factors || !inv_level_sigma2 || nlevels < 1 || nlevels > 16 || nP < 0 || !Tcw || !Ow || !cam_params || !best_idx || !best_dist) return ORBX_E_ARG;
  if (nP > 0 && (!valid || !Xw || !normal || !mpdesc || !max_dist || !min_dist)) return ORBX_E_ARG;
  if (nP == 0) return 0;
  return fuse_core(m, kf, scale_factors, inv_level_sigma2, nlevels, log_scale_factor, nP, valid, Xw, normal, mpdesc, max_dist, min_dist, Tcw, Ow,
                   cam_type, cam_params, bf, th, true, best_idx, best_dist);
}

int orbm_fuse_sim3(orbm_t *m, const orbm_frame_t *kf, const float *scale_factors, int nlevels, float log_scale_factor, int nP,
                   const uint8_t *valid, const float *Xw, const float *normal, const uint8_t *mpdesc, const float *max_dist,
                   const float *min_dist, const float *Scw, const float *cam, float th, int32_t *best_idx, int32_t *best_dist) {
  return orbm_fuse_sim3_cam(m, kf, scale_factors, nlevels, log_scale_factor, nP, valid, Xw, normal, mpdesc, max_dist, min_dist, Scw, 0, cam, th,
                            best_idx, best_dist);
}

int orbm_fuse_sim3_cam(orbm_t *m, const orbm_frame_t *kf, const float *scale_factors, int nlevels, float log_scale_factor, int nP,
                       const uint8_t *valid, const float *Xw, const float *normal, const uint8_t *mpdesc, const float *max_dist,
                       const float *min_dist, const float *Scw, int cam_type, const float *cam, float th, int32_t *best_idx, int32_t *best_dist) {
  if (!m || !kf || !scale_factors || nlevels < 1 || nP < 0 || !Scw || !cam || !best_idx || !best_dist || (cam_type != 0 && cam_type != 1)) return ORBX_E_ARG;
  if (nP > 0 && (!valid || !Xw || !normal || !mpdesc || !max_dist || !min_dist)) return ORBX_E_ARG;
  if (nP == 0) return 0;
  // Decompose Scw, :1668-1673 (as in orbm_search_by_projection_sim3)
  double dot = 0;
  for (int k = 0; k < 3; k++) dot += (double)Scw[k] * (double)Scw[k];
  const float scw = (float)sqrt(dot);
  const double inv = 1. / (double)scw;
  float T[16] = {0};
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) T[4 * i + j] = (float)((double)Scw[4 * i + j] * inv);
    T[4 * i + 3] = (float)((double)Scw[4 * i + 3] * inv);
  }
  const float tcw[3] = {T[3], T[7], T[11]};
  float Ow[3];
  for (int i = 0; i < 3; i++) {
    double acc = 0;
    for (int k = 0; k < 3; k++) acc += (double)T[4 * k + i] * (double)tcw[k];
    Ow[i] = (float)(-acc);
  }
  return fuse_core(m, kf, scale_factors, nullptr, nlevels, log_scale_factor, nP, valid, Xw, normal, mpdesc, max_dist, min_dist, T, Ow, cam_type, cam,
                   0.f, th, false, best_idx, best_dist);
}

// ---- SearchBySim3 (ORBmatcher.cc:1788-2012) ------------------------------------------------------------------------------
// Two independent no-claim searches (map points of KF1 in KF2 and back) and a mutual-consistency pass.  cv::Mat algebra per
// SURVEY.md A.8 [OPENCV-UNVERIFIED]: scalar * Mat scales in double; a product with a transposed or scaled operand
// accumulates in double; plain 3x3 * 3x1 + 3x1 takes the small-matrix float path (mat3_mul_add).
namespace {
struct Sim3Side {  // queries of one direction
  std::vector<float> u, v, rad;
  std::vector<int32_t> minl, maxl, best;
  std::vector<uint8_t> flags;
};
// Points of keyframe A (world Xw, pose RAw/tAw) into keyframe B through p_B = sRBA * p_A + tBA.
void sim3_project(const orbm_frame_t *kfB, const float *sfB, int nlevelsB, float logSfB, int nA, const uint8_t *valid, const float *Xw,
                  const float *max_dist, const float *min_dist, const float *RAw, const float *tAw, const float *sRBA, const float *tBA,
                  const float *cam, float th, Sim3Side &S) {
  S.u.assign(nA, 0.f); S.v.assign(nA, 0.f); S.rad.assign(nA, 0.f);
  S.minl.assign(nA, -1); S.maxl.assign(nA, -1); S.best.assign(nA, -1); S.flags.assign(nA, 0);
  float TA[16] = {0}, TB[16] = {0};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { TA[4 * i + j] = RAw[3 * i + j]; TB[4 * i + j] = sRBA[3 * i + j]; }
  for (int i = 0; i < nA; i++) {
    if (!valid[i]) continue;                                                   // :1828-1832
    float pA[3], pB[3];
    mat3_mul_add(TA, Xw + 3 * i, tAw, pA);                                     // :1835
    mat3_mul_add(TB, pA, tBA, pB);                                             // :1836
    if ((double)pB[2] < 0.0) continue;                                         // :1839
    const float invz = (float)(1.0 / (double)pB[2]);                           // :1842
    const float x = pB[0] * invz, y = pB[1] * invz;
    const float u = cam[0] * x + cam[2], v = cam[1] * y + cam[3];              // :1846-1847, pKF1's intrinsics in both directions
    if (!(u >= kfB->min_x && u < kfB->max_x && v >= kfB->min_y && v < kfB->max_y)) continue;  // KeyFrame::IsInImage
    double n2 = 0;
    for (int k = 0; k < 3; k++) n2 += (double)pB[k] * (double)pB[k];
    const float dist3D = (float)sqrt(n2);                                      // cv::norm, :1855
    if (dist3D < 0.8f * min_dist[i] || dist3D > 1.2f * max_dist[i]) continue;  // :1858
    const float ratio = max_dist[i] / dist3D;                                  // MapPoint::PredictScale, MapPoint.cc:570-585
    int lvl = (int)ceilf(logf(ratio) / logSfB);
    lvl = lvl < 0 ? 0 : (lvl >= nlevelsB ? nlevelsB - 1 : lvl);
    S.u[i] = u; S.v[i] = v; S.rad[i] = th * sfB[lvl];                          // :1865
    S.minl[i] = lvl - 1; S.maxl[i] = lvl;                                      // :1884
    S.flags[i] = 1;
  }
}
}  // namespace

int orbm_search_by_sim3(orbm_t *m, const orbm_frame_t *kf1, const float *sf1, int nlevels1, float log_sf1, const uint8_t *valid1,
                        const float *Xw1, const uint8_t *mpdesc1, const float *max_dist1, const float *min_dist1, const float *R1w,
                        const float *t1w, const orbm_frame_t *kf2, const float *sf2, int nlevels2, float log_sf2, const uint8_t *valid2,
                        const float *Xw2, const uint8_t *mpdesc2, const float *max_dist2, const float *min_dist2, const float *R2w,
                        const float *t2w, float s12, const float *R12, const float *t12, const float *cam1, float th, int32_t *matches12) {
  if (!m || !kf1 || !kf2 || !sf1 || !sf2 || nlevels1 < 1 || nlevels2 < 1 || !R1w || !t1w || !R2w || !t2w || !R12 || !t12 || !cam1 || !matches12)
    return ORBX_E_ARG;
  const int N1 = kf1->n, N2 = kf2->n;
  if (N1 < 0 || N2 < 0) return ORBX_E_ARG;
  for (int i = 0; i < N1; i++) matches12[i] = -1;
  if (N1 == 0 || N2 == 0) return 0;
  if (!valid1 || !Xw1 || !mpdesc1 || !max_dist1 || !min_dist1 || !valid2 || !Xw2 || !mpdesc2 || !max_dist2 || !min_dist2) return ORBX_E_ARG;
  float sR12[9], sR21[9], t21[3];
  const double a21 = 1.0 / (double)s12;                                        // :1806
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      sR12[3 * i + j] = (float)((double)R12[3 * i + j] * (double)s12);         // :1805
      sR21[3 * i + j] = (float)((double)R12[3 * j + i] * a21);
    }
  for (int i = 0; i < 3; i++) {                                                // t21 = -sR21*t12, :1807
    double acc = 0;
    for (int k = 0; k < 3; k++) acc += (double)sR21[3 * i + k] * (double)t12[k];
    t21[i] = (float)(acc * -1.0);
  }
  Sim3Side A, B;
  sim3_project(kf2, sf2, nlevels2, log_sf2, N1, valid1, Xw1, max_dist1, min_dist1, R1w, t1w, sR21, t21, cam1, th, A);
  sim3_project(kf1, sf1, nlevels1, log_sf1, N2, valid2, Xw2, max_dist2, min_dist2, R2w, t2w, sR12, t12, cam1, th, B);
  struct Run { const orbm_frame_t *kf; Sim3Side *S; const uint8_t *desc; int nq; } runs[2] = {{kf2, &A, mpdesc1, N1}, {kf1, &B, mpdesc2, N2}};
  for (const Run &r : runs) {
    orbm_queries_t q;
    q.nq = r.nq; q.descriptors = r.desc; q.u = r.S->u.data(); q.v = r.S->v.data(); q.radius = r.S->rad.data();
    q.min_level = r.S->minl.data(); q.max_level = r.S->maxl.data(); q.u_r = nullptr; q.flags = r.S->flags.data();
    orbm_frame_t f = *r.kf;
    f.u_right = nullptr;
    std::vector<int32_t> slot((size_t)r.kf->n, -1);
    std::vector<uint8_t> sobs((size_t)r.kf->n, 0);
    const int rc = search_host(m, &f, &q, 0.f, ORBM_TH_HIGH, 0, slot.data(), sobs.data(), r.S->best.data(), nullptr, nullptr);  // :1895, :1967
    if (rc < 0) return rc;
  }
  int nFound = 0;                                                              // :1973-1987
  for (int i1 = 0; i1 < N1; i1++) {
    const int idx2 = A.best[i1];
    if (idx2 >= 0 && B.best[idx2] == i1) { matches12[i1] = idx2; nFound++; }
  }
  return nFound;
}

// ---- SearchForTriangulation (ORBmatcher.cc:981-1222), Pinhole / Pinhole, no second camera --------------------------------
namespace {
// cv::Mat algebra of ORBmatcher.cc:988-1010 and Pinhole.cpp:143-148 restated ([OPENCV-UNVERIFIED]):
// products without flags take cv::gemm's small-matrix float path, transposed / scaled operands the generic path with
// double accumulation, cv::invert(3x3 CV_32F) evaluates the cofactor formula in double.
void mul33(const float *A, const float *B, float *D) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) D[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
void inv33(const float *S, float *D) {
  const double s00 = S[0], s01 = S[1], s02 = S[2], s10 = S[3], s11 = S[4], s12 = S[5], s20 = S[6], s21 = S[7], s22 = S[8];
  double det = s00 * (s11 * s22 - s12 * s21) - s01 * (s10 * s22 - s12 * s20) + s02 * (s10 * s21 - s11 * s20);
  if (det == 0.) { for (int i = 0; i < 9; i++) D[i] = 0.f; return; }
  det = 1. / det;
  D[0] = (float)((s11 * s22 - s12 * s21) * det); D[1] = (float)((s02 * s21 - s01 * s22) * det); D[2] = (float)((s01 * s12 - s02 * s11) * det);
  D[3] = (float)((s12 * s20 - s10 * s22) * det); D[4] = (float)((s00 * s22 - s02 * s20) * det); D[5] = (float)((s02 * s10 - s00 * s12) * det);
  D[6] = (float)((s10 * s21 - s11 * s20) * det); D[7] = (float)((s01 * s20 - s00 * s21) * det); D[8] = (float)((s00 * s11 - s01 * s10) * det);
}
}  // namespace

// merge-walk of the two feature vectors (ORBmatcher.cc:1036-1188): one work item per unmatched keypoint of KF1 in a shared node
static int triangulation_items(orbm_t *m, const orbm_keyframe_t *k1, const orbm_keyframe_t *k2, int bOnlyStereo, std::vector<TriItem> &items) {
  int f1 = 0, f2 = 0;
  while (f1 < k1->n_nodes && f2 < k2->n_nodes) {
    if (k1->node_id[f1] == k2->node_id[f2]) {
      const int s2 = k2->node_start[f2], l2 = k2->node_start[f2 + 1] - s2;
      if (l2 > 0xffff) { m->err = "vocabulary node with more than 65535 keypoints"; return ORBX_E_ARG; }
      for (int i1 = k1->node_start[f1]; i1 < k1->node_start[f1 + 1]; i1++) {
        const int idx1 = k1->node_idx[i1];
        if (idx1 < 0 || idx1 >= k1->n) return ORBX_E_ARG;
        if (k1->has_mappoint[idx1]) continue;                          // :1050-1055
        if (bOnlyStereo && !(k1->u_right[idx1] >= 0)) continue;        // :1057-1061
        if (l2 > 0) items.push_back(TriItem{idx1, s2, l2});
      }
      f1++; f2++;
    } else if (k1->node_id[f1] < k2->node_id[f2]) {
      while (f1 < k1->n_nodes && k1->node_id[f1] < k2->node_id[f2]) f1++;
    } else {
      while (f2 < k2->n_nodes && k2->node_id[f2] < k1->node_id[f1]) f2++;
    }
  }
  return 0;
}

// Candidate lists of SearchForTriangulation, per keypoint of KF1 in CSR form: start[k1->n + 1]; per candidate the keypoint of KF2 and
// the distance, ordered (distance ascending, position in the node descending) = the order in which the reference's running best
// would prefer them ("dist > bestDist -> skip", update on <=: the LAST minimum among the candidates its predicate accepts).
static int triangulation_candidates(orbm_t *m, const orbm_keyframe_t *k1, const orbm_keyframe_t *k2, float ep_x, float ep_y, int epipole_gate,
                                    int bOnlyStereo, std::vector<int32_t> &start, std::vector<int32_t> &idx2, std::vector<int32_t> &dist) {
  start.assign((size_t)k1->n + 1, 0);
  idx2.clear(); dist.clear();
  if (k1->n == 0 || k2->n == 0 || k1->n_nodes == 0 || k2->n_nodes == 0) return 0;
  std::vector<TriItem> items;
  const int rc = triangulation_items(m, k1, k2, bOnlyStereo, items);
  if (rc < 0) return rc;
  if (items.empty()) return 0;
  MCHECK(m, hipSetDevice(m->device));
  hipStream_t s = m->stream;
  const size_t nidx2 = (size_t)k2->node_start[k2->n_nodes];
  DevBuf *B[] = {&m->d_desc, &m->d_ur, &m->d_qdesc, &m->d_qf[0], &m->d_qf[1], &m->d_qfl, &m->d_qi[0], &m->d_qi[1]};
  const void *src[] = {k1->descriptors, k1->u_right, k2->descriptors, k2->keys_un, k2->u_right, k2->has_mappoint, k2->node_idx, items.data()};
  const size_t bytes[] = {32 * (size_t)k1->n, sizeof(float) * (size_t)k1->n, 32 * (size_t)k2->n, sizeof(orbx_keypoint_t) * (size_t)k2->n,
                          sizeof(float) * (size_t)k2->n, (size_t)k2->n, sizeof(int32_t) * nidx2, sizeof(TriItem) * items.size()};
  for (int i = 0; i < 8; i++) {
    MCHECK(m, B[i]->reserve(std::max<size_t>(bytes[i], 4)));
    MCHECK(m, hipMemcpyAsync(B[i]->p, src[i], bytes[i], hipMemcpyHostToDevice, s));
  }
  TriCandParams T;
  memset(&T, 0, sizeof(T));
  T.desc1 = (const uint32_t *)m->d_desc.p; T.ur1 = (const float *)m->d_ur.p;
  T.desc2 = (const uint32_t *)m->d_qdesc.p; T.kp2 = (const float *)m->d_qf[0].p; T.ur2 = (const float *)m->d_qf[1].p;
  T.hasmp2 = (const uint8_t *)m->d_qfl.p; T.node_idx2 = (const int32_t *)m->d_qi[0].p;
  T.items = (const TriItem *)m->d_qi[1].p; T.nitems = (int)items.size();
  for (int l = 0; l < ORBX_MAX_LEVELS; l++) T.sf2[l] = l < k2->nlevels ? k2->scale_factors[l] : 0.f;
  T.epx = ep_x; T.epy = ep_y; T.epipole_gate = epipole_gate; T.bOnlyStereo = bOnlyStereo;
  const size_t ni = items.size();
  std::vector<int32_t> off(ni), cnt(ni);
  std::vector<uint32_t> keys;
  size_t cap = std::max<size_t>(16 * ni, 1024);
  for (int attempt = 0; attempt < 2; attempt++) {   // the second attempt knows the exact total
    MCHECK(m, m->scratch[0].reserve(sizeof(int32_t) * (2 * ni + 1)));
    MCHECK(m, m->scratch[1].reserve(sizeof(uint32_t) * cap));
    T.item_off = (int32_t *)m->scratch[0].p; T.item_cnt = T.item_off + ni; T.total = T.item_cnt + ni;
    T.keys = (uint32_t *)m->scratch[1].p; T.cap = (int)std::min<size_t>(cap, 0x7fffffff);
    MCHECK(m, hipMemsetAsync(T.total, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k_triangulation_candidates, dim3((T.nitems + 3) / 4), dim3(256), 0, s, T);
    MCHECK(m, hipGetLastError());
    int32_t total = 0;
    MCHECK(m, hipMemcpyAsync(&total, T.total, sizeof(total), hipMemcpyDeviceToHost, s));
    MCHECK(m, hipMemcpyAsync(off.data(), T.item_off, sizeof(int32_t) * ni, hipMemcpyDeviceToHost, s));
    MCHECK(m, hipMemcpyAsync(cnt.data(), T.item_cnt, sizeof(int32_t) * ni, hipMemcpyDeviceToHost, s));
    MCHECK(m, hipStreamSynchronize(s));
    if ((size_t)total <= cap) {
      keys.resize((size_t)total);
      if (total > 0) MCHECK(m, hipMemcpy(keys.data(), T.keys, sizeof(uint32_t) * (size_t)total, hipMemcpyDeviceToHost));
      break;
    }
    if (attempt == 1) { m->err = "triangulation candidates: buffer too small twice"; return ORBX_E_HIP; }
    cap = (size_t)total;
  }
  // CSR over the keypoints of KF1 (a keypoint belongs to one node, i.e. to at most one item), every list ordered by key
  for (size_t i = 0; i < ni; i++) start[(size_t)items[i].idx1 + 1] += cnt[i];
  for (int i = 0; i < k1->n; i++) start[(size_t)i + 1] += start[(size_t)i];
  idx2.resize(keys.size()); dist.resize(keys.size());
  for (size_t i = 0; i < ni; i++) {
    if (!cnt[i]) continue;
    uint32_t *kb = keys.data() + off[i];
    std::sort(kb, kb + cnt[i]);
    const int o = start[(size_t)items[i].idx1];
    for (int c = 0; c < cnt[i]; c++) {
      idx2[(size_t)o + c] = k2->node_idx[items[i].start2 + (0xffff - (int)(kb[c] & 0xffffu))];
      dist[(size_t)o + c] = (int32_t)(kb[c] >> 16);
    }
  }
  return (int)keys.size();
}

int orbm_triangulation_candidates(orbm_t *m, const orbm_keyframe_t *k1, const orbm_keyframe_t *k2, float ep_x, float ep_y, int epipole_gate,
                                  int bOnlyStereo, int32_t *cand_start, int32_t *cand_idx2, int32_t *cand_dist, int cap) {
  if (!m || !k1 || !k2 || !cand_start || cap < 0 || (cap > 0 && (!cand_idx2 || !cand_dist))) return ORBX_E_ARG;
  if (k1->n < 0 || k2->n < 0 || k2->nlevels < 1 || k2->nlevels > ORBX_MAX_LEVELS) return ORBX_E_ARG;
  if ((k1->n > 0 && (!k1->u_right || !k1->has_mappoint)) || (k2->n > 0 && (!k2->u_right || !k2->has_mappoint))) return ORBX_E_ARG;
  std::vector<int32_t> start, idx2, dist;
  const int total = triangulation_candidates(m, k1, k2, ep_x, ep_y, epipole_gate, bOnlyStereo, start, idx2, dist);
  if (total < 0) return total;
  memcpy(cand_start, start.data(), sizeof(int32_t) * start.size());
  if (total <= cap && total > 0) {
    memcpy(cand_idx2, idx2.data(), sizeof(int32_t) * (size_t)total);
    memcpy(cand_dist, dist.data(), sizeof(int32_t) * (size_t)total);
  }
  return total;   // > cap: nothing but cand_start was written; call again with that capacity
}

static int prune_pairs_by_rotation(const orbm_keyframe_t *k1, const orbm_keyframe_t *k2, int32_t *matches12, int nmatches);

int orbm_search_for_triangulation_pred(orbm_t *m, const orbm_keyframe_t *k1, const orbm_keyframe_t *k2, float ep_x, float ep_y, int epipole_gate,
                                       int bOnlyStereo, int bCoarse, int checkOri, orbm_pair_predicate_t pred, void *user, int32_t *matches12) {
  if (!m || !k1 || !k2 || !matches12 || (!pred && !bCoarse)) return ORBX_E_ARG;
  if (k1->n < 0 || k2->n < 0 || k2->nlevels < 1 || k2->nlevels > ORBX_MAX_LEVELS) return ORBX_E_ARG;
  if ((k1->n > 0 && (!k1->u_right || !k1->has_mappoint)) || (k2->n > 0 && (!k2->u_right || !k2->has_mappoint))) return ORBX_E_ARG;
  for (int i = 0; i < k1->n; i++) matches12[i] = -1;
  std::vector<int32_t> start, idx2, dist;
  const int total = triangulation_candidates(m, k1, k2, ep_x, ep_y, epipole_gate, bOnlyStereo, start, idx2, dist);
  if (total < 0) return total;
  int nmatches = 0;
  for (int i = 0; i < k1->n; i++)
    for (int c = start[(size_t)i]; c < start[(size_t)i + 1]; c++)
      if (bCoarse || pred(user, i, idx2[(size_t)c])) { matches12[i] = idx2[(size_t)c]; nmatches++; break; }   // :1148-1153
  if (checkOri && nmatches > 0) nmatches = prune_pairs_by_rotation(k1, k2, matches12, nmatches);
  return nmatches;
}

int orbm_search_for_triangulation(orbm_t *m, const orbm_keyframe_t *k1, const orbm_keyframe_t *k2, const float *R1w, const float *t1w,
                                  const float *R2w, const float *t2w, const float *Cw1, const float *cam1, const float *cam2,
                                  int bOnlyStereo, int bCoarse, int checkOri, int32_t *matches12) {
  if (!m || !k1 || !k2 || !R1w || !t1w || !R2w || !t2w || !Cw1 || !cam1 || !cam2 || !matches12) return ORBX_E_ARG;
  if (k1->n < 0 || k2->n < 0 || k2->nlevels < 1 || k2->nlevels > ORBX_MAX_LEVELS) return ORBX_E_ARG;
  for (int i = 0; i < k1->n; i++) matches12[i] = -1;
  if (k1->n == 0 || k2->n == 0 || k1->n_nodes == 0 || k2->n_nodes == 0) return 0;
  if (!k1->u_right || !k2->u_right || !k1->has_mappoint || !k2->has_mappoint) return ORBX_E_ARG;
  TriParams T;
  memset(&T, 0, sizeof(T));
  // epipole in image 2: C2 = R2w*Cw+t2w, ep = pCamera2->project(C2), :988-994
  float C2[3];
  for (int i = 0; i < 3; i++) {
    const float t0 = R2w[3 * i] * Cw1[0] + R2w[3 * i + 1] * Cw1[1] + R2w[3 * i + 2] * Cw1[2];
    C2[i] = (float)((double)t0 + (double)t2w[i]);
  }
  orbm_project(0, cam2, C2[0], C2[1], C2[2], &T.epx, &T.epy);
  // R12 = R1w*R2w.t(); t12 = -R1w*R2w.t()*t2w+t1w, :1008-1010
  float R12[9], nR[9], t12[3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double acc = 0;
      for (int k = 0; k < 3; k++) acc += (double)R1w[3 * i + k] * (double)R2w[3 * j + k];
      R12[3 * i + j] = (float)acc;
      nR[3 * i + j] = (float)(-acc);
    }
  for (int i = 0; i < 3; i++) {
    const float t0 = nR[3 * i] * t2w[0] + nR[3 * i + 1] * t2w[1] + nR[3 * i + 2] * t2w[2];
    t12[i] = (float)((double)t0 + (double)t1w[i]);
  }
  {  // F12 = K1.t().inv()*t12x*R12*K2.inv(), Pinhole.cpp:145-148
    const float tx[9] = {0, -t12[2], t12[1], t12[2], 0, -t12[0], -t12[1], t12[0], 0};
    const float K1t[9] = {cam1[0], 0.f, 0.f, 0.f, cam1[1], 0.f, cam1[2], cam1[3], 1.f};
    const float K2[9] = {cam2[0], 0.f, cam2[2], 0.f, cam2[1], cam2[3], 0.f, 0.f, 1.f};
    float K1ti[9], K2i[9], A[9], B[9];
    inv33(K1t, K1ti);
    inv33(K2, K2i);
    mul33(K1ti, tx, A);
    mul33(A, R12, B);
    mul33(B, K2i, T.F12);
  }
  std::vector<TriItem> items;
  { const int rc = triangulation_items(m, k1, k2, bOnlyStereo, items); if (rc < 0) return rc; }
  int nmatches = 0;
  if (!items.empty()) {
    MCHECK(m, hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const size_t nidx2 = (size_t)k2->node_start[k2->n_nodes];
    DevBuf *B[] = {&m->d_kp, &m->d_desc, &m->d_ur, &m->d_qdesc, &m->d_qf[0], &m->d_qf[1], &m->d_qfl, &m->d_qi[0], &m->d_qi[1], &m->d_moq};
    const void *src[] = {k1->keys_un, k1->descriptors, k1->u_right, k2->descriptors, k2->keys_un, k2->u_right, k2->has_mappoint, k2->node_idx, items.data(), nullptr};
    const size_t bytes[] = {sizeof(orbx_keypoint_t) * (size_t)k1->n, 32 * (size_t)k1->n, sizeof(float) * (size_t)k1->n, 32 * (size_t)k2->n,
                            sizeof(orbx_keypoint_t) * (size_t)k2->n, sizeof(float) * (size_t)k2->n, (size_t)k2->n, sizeof(int32_t) * nidx2,
                            sizeof(TriItem) * items.size(), sizeof(int32_t) * (size_t)k1->n};
    for (int i = 0; i < 10; i++) {
      MCHECK(m, B[i]->reserve(std::max<size_t>(bytes[i], 4)));
      if (src[i]) MCHECK(m, hipMemcpyAsync(B[i]->p, src[i], bytes[i], hipMemcpyHostToDevice, s));
    }
    MCHECK(m, hipMemsetAsync(m->d_moq.p, 0xff, bytes[9], s));
    T.kp1 = (const float *)m->d_kp.p; T.desc1 = (const uint32_t *)m->d_desc.p; T.ur1 = (const float *)m->d_ur.p;
    T.desc2 = (const uint32_t *)m->d_qdesc.p; T.kp2 = (const float *)m->d_qf[0].p; T.ur2 = (const float *)m->d_qf[1].p;
    T.hasmp2 = (const uint8_t *)m->d_qfl.p; T.node_idx2 = (const int32_t *)m->d_qi[0].p;
    T.items = (const TriItem *)m->d_qi[1].p; T.nitems = (int)items.size();
    for (int l = 0; l < k2->nlevels; l++) { T.sf2[l] = k2->scale_factors[l]; T.sigma2_2[l] = k2->level_sigma2[l]; }
    T.bOnlyStereo = bOnlyStereo; T.bCoarse = bCoarse;
    T.matches12 = (int32_t *)m->d_moq.p;
    hipLaunchKernelGGL(k_triangulation_match, dim3((T.nitems + 3) / 4), dim3(256), 0, s, T);
    MCHECK(m, hipGetLastError());
    MCHECK(m, hipMemcpyAsync(matches12, m->d_moq.p, bytes[9], hipMemcpyDeviceToHost, s));
    MCHECK(m, hipStreamSynchronize(s));
    for (int i = 0; i < k1->n; i++) nmatches += matches12[i] >= 0;
  }
  if (checkOri && nmatches > 0) nmatches = prune_pairs_by_rotation(k1, k2, matches12, nmatches);
  return nmatches;
}

// rotation-histogram pruning of SearchForTriangulation's pairs, ORBmatcher.cc:1162-1172, :1191-1207
static int prune_pairs_by_rotation(const orbm_keyframe_t *k1, const orbm_keyframe_t *k2, int32_t *matches12, int nmatches) {
  std::vector<std::vector<int>> rotHist(ORBM_HISTO_LENGTH);
  const float factor = 1.0f / ORBM_HISTO_LENGTH;
  // the reference fills the histogram in merge-walk order; only the bin sizes matter afterwards
  for (int i = 0; i < k1->n; i++) {
    if (matches12[i] < 0) continue;
    float rot = k1->keys_un[i].angle - k2->keys_un[matches12[i]].angle;
    if ((double)rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == ORBM_HISTO_LENGTH) bin = 0;
    if (bin >= 0 && bin < ORBM_HISTO_LENGTH) rotHist[bin].push_back(i);
  }
  int sizes[ORBM_HISTO_LENGTH], ind1, ind2, ind3;
  for (int i = 0; i < ORBM_HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
  orbm_three_maxima(sizes, ORBM_HISTO_LENGTH, &ind1, &ind2, &ind3);
  for (int i = 0; i < ORBM_HISTO_LENGTH; i++) {
    if (i == ind1 || i == ind2 || i == ind3) continue;
    for (int idx : rotHist[i]) { matches12[idx] = -1; nmatches--; }
  }
  return nmatches;
}

int orbm_search_for_initialization(orbm_t *m, const orbm_frame_t *f1, const orbm_frame_t *f2, float *prev_matched, int window_size,
                                   float nnratio, int checkOri, int32_t *matches12) {
  if (!m || !f1 || !f2 || !matches12 || f1->n < 0 || f2->n < 0) return ORBX_E_ARG;
  if (f1->n > 0 && (!f1->keys_un || !f1->descriptors || !prev_matched)) return ORBX_E_ARG;
  const int n1 = f1->n;
  for (int i = 0; i < n1; i++) matches12[i] = -1;                 // :725
  // queries: the level-0 keypoints of F1 in index order (:737-739), window centre = vbPrevMatched (:741)
  std::vector<int32_t> qi;
  for (int i = 0; i < n1; i++)
    if (f1->keys_un[i].octave <= 0) qi.push_back(i);
  const int nq = (int)qi.size();
  if (nq == 0 || f2->n == 0) return 0;
  std::vector<float> u(nq), v(nq), rad(nq, (float)window_size);
  std::vector<int32_t> lvl(nq, 0), acc(nq, -1);
  std::vector<uint8_t> qd((size_t)nq * 32);
  for (int k = 0; k < nq; k++) {
    u[k] = prev_matched[2 * qi[k]]; v[k] = prev_matched[2 * qi[k] + 1];
    memcpy(&qd[(size_t)k * 32], f1->descriptors + (size_t)qi[k] * 32, 32);
  }
  orbm_queries_t q;
  q.nq = nq; q.descriptors = qd.data(); q.u = u.data(); q.v = v.data(); q.radius = rad.data();
  q.min_level = lvl.data(); q.max_level = lvl.data();             // GetFeaturesInArea(..., level1, level1), level1 == 0
  q.u_r = nullptr; q.flags = nullptr;
  orbm_frame_t f = *f2;
  f.u_right = nullptr;
  std::vector<int32_t> slot((size_t)f2->n, -1);
  std::vector<uint8_t> sobs((size_t)f2->n, 0);
  StereoExt ext{};
  ext.nleft = f2->n; ext.init_th_low = ORBM_TH_LOW;
  const int rc = search_host(m, &f, &q, nnratio, ORBM_TH_LOW, 1, slot.data(), sobs.data(), acc.data(), nullptr, &ext);
  if (rc < 0) return rc;
  // replay of the bookkeeping the device leaves to the host: steals (:781-785) and the rotation histogram (:791-801)
  int nmatches = 0;
  std::vector<int32_t> m21((size_t)f2->n, -1);
  std::vector<std::vector<int>> rotHist(ORBM_HISTO_LENGTH);
  const float factor = 1.0f / ORBM_HISTO_LENGTH;
  for (int k = 0; k < nq; k++) {
    const int i2 = acc[k];
    if (i2 < 0) continue;
    const int i1 = qi[k];
    if (m21[i2] >= 0) { matches12[m21[i2]] = -1; nmatches--; }
    matches12[i1] = i2; m21[i2] = i1; nmatches++;
    if (checkOri) {
      float rot = f1->keys_un[i1].angle - f2->keys_un[i2].angle;
      if ((double)rot < 0.0) rot += 360.0f;
      int bin = (int)roundf(rot * factor);
      if (bin == ORBM_HISTO_LENGTH) bin = 0;
      if (bin >= 0 && bin < ORBM_HISTO_LENGTH) rotHist[bin].push_back(i1);
    }
  }
  if (checkOri) {
    int sizes[ORBM_HISTO_LENGTH], ind1, ind2, ind3;
    for (int i = 0; i < ORBM_HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
    orbm_three_maxima(sizes, ORBM_HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int i = 0; i < ORBM_HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx1 : rotHist[i])
        if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }   // :815-819
    }
  }
  for (int i = 0; i < n1; i++)                                     // :826-828
    if (matches12[i] >= 0) { prev_matched[2 * i] = f2->keys_un[matches12[i]].x; prev_matched[2 * i + 1] = f2->keys_un[matches12[i]].y; }
  return nmatches;
}

int orbx_compute_stereo_matches(orbx_t *hl, int frame_l, orbx_t *hr, int frame_r, int nL, const orbx_keypoint_t *keysL,
                                const uint8_t *descL, int nR, const orbx_keypoint_t *keysR, const uint8_t *descR, float mb, float mbf,
                                float *uRight, float *depth) {
  if (!hl || !hr || nL < 0 || nR < 0 || !uRight || !depth) return ORBX_E_ARG;
  if (!hl->have_last || !hr->have_last) { hl->err = "orbx_compute_stereo_matches: extract both images first"; return ORBX_E_ARG; }
  if (frame_l < 0 || frame_l >= hl->last.nframes || frame_r < 0 || frame_r >= hr->last.nframes) return ORBX_E_ARG;
  if (hl->device != hr->device || hl->rows != hr->rows || hl->cols != hr->cols || hl->nlevels != hr->nlevels) {
    hl->err = "orbx_compute_stereo_matches: the two extractors must share device, image size and pyramid";
    return ORBX_E_ARG;
  }
  for (int i = 0; i < nL; i++) { uRight[i] = -1.0f; depth[i] = -1.0f; }   // Frame.cc:903-904
  if (nL == 0 || nR == 0) return 0;
  if (!keysL || !descL || !keysR || !descR || nR > 65535) return ORBX_E_ARG;
  if (!(mb > 0.f)) return ORBX_E_ARG;
  XCHECK(hl, hipSetDevice(hl->device));
  hipStream_t s = hl->stream;
  // the two extractions may have run on any stream (the handles' own, or the caller's with orbx_extract_batch_device): the
  // search is ordered behind the last kernel of both; level 0 is read in place from the caller's images (orbhip.h)
  if (hl->last_pending) XCHECK(hl, hipStreamWaitEvent(s, hl->last_done, 0));
  if (hr->last_pending) XCHECK(hl, hipStreamWaitEvent(s, hr->last_done, 0));
  DevBuf *bufs = hl->stereo;
  const size_t sz[7] = {sizeof(orbx_keypoint_t) * (size_t)nL, sizeof(orbx_keypoint_t) * (size_t)nR, 32 * (size_t)nL, 32 * (size_t)nR,
                        sizeof(float) * (size_t)nL, sizeof(float) * (size_t)nL, sizeof(int32_t) * (size_t)nL};
  const void *src[4] = {keysL, keysR, descL, descR};
  for (int i = 0; i < 7; i++) XCHECK(hl, bufs[i].reserve(sz[i]));
  for (int i = 0; i < 4; i++) XCHECK(hl, hipMemcpyAsync(bufs[i].p, src[i], sz[i], hipMemcpyHostToDevice, s));
  StereoParams S;
  memset(&S, 0, sizeof(S));
  S.imgL0 = hl->last.img0 + (size_t)frame_l * hl->last.img0_frame_stride; S.strideL0 = hl->last.img0_stride;
  S.imgR0 = hr->last.img0 + (size_t)frame_r * hr->last.img0_frame_stride; S.strideR0 = hr->last.img0_stride;
  S.pyrL = hl->last.pyr + (size_t)frame_l * hl->last.pyr_fs;
  S.pyrR = hr->last.pyr + (size_t)frame_r * hr->last.pyr_fs;
  for (int l = 0; l < hl->nlevels; l++) {
    const LevelGeom &G = hl->geom[l];
    S.w[l] = G.w; S.h[l] = G.h; S.pitch[l] = G.pitch; S.off[l] = G.off;
    S.sf[l] = hl->mvScaleFactor[l]; S.invsf[l] = hl->mvInvScaleFactor[l];
  }
  S.nlevels = hl->nlevels; S.rows = hl->rows;
  S.kpL = (const float *)bufs[0].p; S.kpR = (const float *)bufs[1].p;
  S.descL = (const uint32_t *)bufs[2].p; S.descR = (const uint32_t *)bufs[3].p;
  S.nL = nL; S.nR = nR; S.mb = mb; S.mbf = mbf;
  S.uRight = (float *)bufs[4].p; S.depth = (float *)bufs[5].p; S.sad = (int32_t *)bufs[6].p;
  hipLaunchKernelGGL(k_stereo_match, dim3((nL + 3) / 4), dim3(256), 0, s, S);
  XCHECK(hl, hipGetLastError());
  std::vector<int32_t> sad((size_t)nL);
  XCHECK(hl, hipMemcpyAsync(uRight, bufs[4].p, sz[4], hipMemcpyDeviceToHost, s));
  XCHECK(hl, hipMemcpyAsync(depth, bufs[5].p, sz[5], hipMemcpyDeviceToHost, s));
  XCHECK(hl, hipMemcpyAsync(sad.data(), bufs[6].p, sz[6], hipMemcpyDeviceToHost, s));
  XCHECK(hl, hipStreamSynchronize(s));
  // median filter over the accepted matches, Frame.cc:1060-1073 (an empty set - where the reference indexes an empty
  // vector - is defined as "nothing to do")
  std::vector<std::pair<int, int>> vDistIdx;
  for (int i = 0; i < nL; i++)
    if (sad[i] >= 0) vDistIdx.push_back(std::make_pair(sad[i], i));
  if (!vDistIdx.empty()) {
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
      if ((float)vDistIdx[i].first < thDist) break;
      uRight[vDistIdx[i].second] = -1;
      depth[vDistIdx[i].second] = -1;
    }
  }
  return 0;
}

// Shared by the two SearchByBoW overloads.  kf_kf: second operand is a keyframe (candidates need a good map point, strict
// threshold, result indexed by the first keyframe's keypoints: out[kf->n]); otherwise out[f->n] is indexed by frame keypoint.
static int bow_core(orbm_t *m, const orbm_keyframe_t *kf, const orbm_keyframe_t *f, float nnratio, int checkOri, bool kf_kf, int nleftF, int32_t *out) {
  if (!m || !kf || !f || !out || kf->n < 0 || f->n < 0) return ORBX_E_ARG;
  const int nout = kf_kf ? kf->n : f->n;
  for (int i = 0; i < nout; i++) out[i] = -1;                     // :277 / :852
  if (kf->n == 0 || f->n == 0 || kf->n_nodes <= 0 || f->n_nodes <= 0) return 0;
  if (!kf->descriptors || !f->descriptors || !kf->has_mappoint || !kf->node_id || !f->node_id || !kf->node_start || !f->node_start ||
      !kf->node_idx || !f->node_idx || !kf->keys_un || !f->keys_un || (kf_kf && !f->has_mappoint)) return ORBX_E_ARG;
  // merge-walk of the two feature vectors (:292-296, :440-447): one work item per shared node
  std::vector<BowItem> items;
  int a = 0, b = 0;
  while (a < kf->n_nodes && b < f->n_nodes) {
    if (kf->node_id[a] == f->node_id[b]) {
      const int lk = kf->node_start[a + 1] - kf->node_start[a], lf = f->node_start[b + 1] - f->node_start[b];
      if (lf > 2048) { m->err = "SearchByBoW: more than 2048 keypoints of the second operand in one vocabulary node"; return ORBX_E_ARG; }
      if (lk > 0 && lf > 0) items.push_back(BowItem{kf->node_start[a], lk, f->node_start[b], lf});
      a++; b++;
    } else if (kf->node_id[a] < f->node_id[b]) {
      while (a < kf->n_nodes && kf->node_id[a] < f->node_id[b]) a++;   // lower_bound
    } else {
      while (b < f->n_nodes && f->node_id[b] < kf->node_id[a]) b++;
    }
  }
  if (items.empty()) return 0;
  MCHECK(m, hipSetDevice(m->device));
  hipStream_t s = m->stream;
  const size_t nidxKF = (size_t)kf->node_start[kf->n_nodes], nidxF = (size_t)f->node_start[f->n_nodes];
  DevBuf *bufs = m->scratch;
  const size_t sz[8] = {32 * (size_t)kf->n, 32 * (size_t)f->n, (size_t)kf->n, sizeof(int32_t) * nidxKF, sizeof(int32_t) * nidxF,
                        sizeof(BowItem) * items.size(), sizeof(int32_t) * (size_t)nout, kf_kf ? (size_t)f->n : 0};
  const void *src[8] = {kf->descriptors, f->descriptors, kf->has_mappoint, kf->node_idx, f->node_idx, items.data(), out, kf_kf ? f->has_mappoint : nullptr};
  for (int i = 0; i < 8; i++) {
    MCHECK(m, bufs[i].reserve(std::max<size_t>(sz[i], 4)));
    if (sz[i]) MCHECK(m, hipMemcpyAsync(bufs[i].p, src[i], sz[i], hipMemcpyHostToDevice, s));
  }
  BowParams B;
  memset(&B, 0, sizeof(B));
  B.descKF = (const uint32_t *)bufs[0].p; B.descF = (const uint32_t *)bufs[1].p; B.hasmpKF = (const uint8_t *)bufs[2].p;
  B.node_idxKF = (const int32_t *)bufs[3].p; B.node_idxF = (const int32_t *)bufs[4].p;
  B.items = (const BowItem *)bufs[5].p; B.nitems = (int)items.size();
  B.nnratio = nnratio;
  B.nleftF = nleftF >= 0 ? nleftF : 0x7fffffff;
  if (kf_kf) { B.match12 = (int32_t *)bufs[6].p; B.hasmpF = (const uint8_t *)bufs[7].p; B.strict = 1; }
  else B.matchF = (int32_t *)bufs[6].p;
  hipLaunchKernelGGL(k_bow_match, dim3((B.nitems + 3) / 4), dim3(256), 0, s, B);
  MCHECK(m, hipGetLastError());
  MCHECK(m, hipMemcpyAsync(out, bufs[6].p, sz[6], hipMemcpyDeviceToHost, s));
  MCHECK(m, hipStreamSynchronize(s));
  int nmatches = 0;
  for (int i = 0; i < nout; i++) nmatches += out[i] >= 0 ? 1 : 0;
  if (!checkOri) return nmatches;
  // rotation histogram (:391-404, :451-466 resp. :929-939, :960-975): bins hold the index `out` is addressed with
  std::vector<std::vector<int>> rotHist(ORBM_HISTO_LENGTH);
  const float factor = 1.0f / ORBM_HISTO_LENGTH;
  for (int i = 0; i < nout; i++) {
    if (out[i] < 0) continue;
    const int idxKF = kf_kf ? i : out[i], idxF = kf_kf ? out[i] : i;
    float rot = kf->keys_un[idxKF].angle - f->keys_un[idxF].angle;
    if ((double)rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == ORBM_HISTO_LENGTH) bin = 0;
    if (bin >= 0 && bin < ORBM_HISTO_LENGTH) rotHist[bin].push_back(i);
  }
  int sizes[ORBM_HISTO_LENGTH], ind1, ind2, ind3;
  for (int i = 0; i < ORBM_HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
  orbm_three_maxima(sizes, ORBM_HISTO_LENGTH, &ind1, &ind2, &ind3);
  for (int i = 0; i < ORBM_HISTO_LENGTH; i++) {
    if (i == ind1 || i == ind2 || i == ind3) continue;
    for (int idx : rotHist[i]) { out[idx] = -1; nmatches--; }
  }
  return nmatches;
}

int orbm_search_by_bow(orbm_t *m, const orbm_keyframe_t *kf, const orbm_keyframe_t *f, float nnratio, int checkOri, int32_t *matchF) {
  return bow_core(m, kf, f, nnratio, checkOri, false, -1, matchF);
}

int orbm_search_by_bow_fisheye(orbm_t *m, const orbm_keyframe_t *kf, const orbm_keyframe_t *f, int n_left_f, float nnratio, int checkOri,
                               int32_t *matchF) {
  if (n_left_f < 0 || (f && n_left_f > f->n)) return ORBX_E_ARG;
  return bow_core(m, kf, f, nnratio, checkOri, false, n_left_f, matchF);
}

int orbm_search_by_bow_keyframes(orbm_t *m, const orbm_keyframe_t *kf1, const orbm_keyframe_t *kf2, float nnratio, int checkOri,
                                 int32_t *matches12) {
  return bow_core(m, kf1, kf2, nnratio, checkOri, true, -1, matches12);
}

int orbm_distinctive_descriptors(orbm_t *m, int nmp, const int32_t *start, const uint8_t *desc, int32_t *best) {
  if (!m || nmp < 0 || (nmp > 0 && (!start || !best))) return ORBX_E_ARG;
  if (nmp == 0) return 0;
  const int total = start[nmp];
  for (int i = 0; i < nmp; i++) {
    const int n = start[i + 1] - start[i];
    if (n < 0 || n > 64 * DISTINCT_MAXT) { m->err = "ComputeDistinctiveDescriptors: more than 1024 observations of one map point"; return ORBX_E_ARG; }
  }
  if (total > 0 && !desc) return ORBX_E_ARG;
  MCHECK(m, hipSetDevice(m->device));
  hipStream_t s = m->stream;
  MCHECK(m, m->d_a.reserve(std::max<size_t>(32 * (size_t)total, 4)));
  MCHECK(m, m->d_b.reserve(sizeof(int32_t) * (size_t)(nmp + 1)));
  MCHECK(m, m->d_c.reserve(sizeof(int32_t) * (size_t)nmp));
  if (total > 0) MCHECK(m, hipMemcpyAsync(m->d_a.p, desc, 32 * (size_t)total, hipMemcpyHostToDevice, s));
  MCHECK(m, hipMemcpyAsync(m->d_b.p, start, sizeof(int32_t) * (size_t)(nmp + 1), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_distinctive, dim3((nmp + 3) / 4), dim3(256), 0, s, (const uint32_t *)m->d_a.p, (const int32_t *)m->d_b.p, nmp, (int32_t *)m->d_c.p);
  MCHECK(m, hipGetLastError());
  MCHECK(m, hipMemcpyAsync(best, m->d_c.p, sizeof(int32_t) * (size_t)nmp, hipMemcpyDeviceToHost, s));
  MCHECK(m, hipStreamSynchronize(s));
  return 0;
}

int orbm_knn_match2(orbm_t *m, const uint8_t *q, int nq, const uint8_t *c, int nc, int32_t *idx2, int32_t *dist2) {
  if (!m || nq < 0 || nc < 0 || (nq > 0 && (!q || !idx2 || !dist2)) || (nc > 0 && !c)) return ORBX_E_ARG;
  if (nq == 0) return 0;
  if (nc == 0) { for (int i = 0; i < 2 * nq; i++) { idx2[i] = -1; dist2[i] = -1; } return 0; }
  MCHECK(m, hipSetDevice(m->device));
  hipStream_t s = m->stream;
  MCHECK(m, m->d_a.reserve(32 * (size_t)nq));
  MCHECK(m, m->d_b.reserve(32 * (size_t)nc));
  MCHECK(m, m->d_c.reserve(sizeof(int32_t) * 4 * (size_t)nq));
  MCHECK(m, hipMemcpyAsync(m->d_a.p, q, 32 * (size_t)nq, hipMemcpyHostToDevice, s));
  MCHECK(m, hipMemcpyAsync(m->d_b.p, c, 32 * (size_t)nc, hipMemcpyHostToDevice, s));
  int32_t *d_idx = (int32_t *)m->d_c.p, *d_dist = d_idx + 2 * (size_t)nq;
  // matrix-pipe form (orb_match_mfma.h) unless the vector-ALU engine is selected; its keys carry the train index in 20 bits
  if (m->hamming_engine >= 1 && nc < (1 << 20))
    hipLaunchKernelGGL(k_knn2_mfma, dim3((nq + MF_NT - 1) / MF_NT), dim3(MF_NT), 0, s, (const uint32_t *)m->d_a.p, nq, (const uint32_t *)m->d_b.p, nc, d_idx, d_dist);
  else
    hipLaunchKernelGGL(k_knn2, dim3((nq + 3) / 4), dim3(256), 0, s, (const uint32_t *)m->d_a.p, nq, (const uint32_t *)m->d_b.p, nc, d_idx, d_dist);
  MCHECK(m, hipGetLastError());
  MCHECK(m, hipMemcpyAsync(idx2, d_idx, sizeof(int32_t) * 2 * (size_t)nq, hipMemcpyDeviceToHost, s));
  MCHECK(m, hipMemcpyAsync(dist2, d_dist, sizeof(int32_t) * 2 * (size_t)nq, hipMemcpyDeviceToHost, s));
  MCHECK(m, hipStreamSynchronize(s));
  return 0;
}

int orbm_hamming_matrix(orbm_t *m, const uint8_t *q, int nq, const uint8_t *c, int nc, uint16_t *dist) {
  if (!m || !q || !c || !dist || nq <= 0 || nc <= 0) return ORBX_E_ARG;
  MCHECK(m, hipSetDevice(m->device));
  hipStream_t s = m->stream;
  MCHECK(m, m->d_a.reserve(32 * (size_t)nq));
  MCHECK(m, m->d_b.reserve(32 * (size_t)nc));
  MCHECK(m, m->d_c.reserve(sizeof(uint16_t) * (size_t)nq * nc));
  MCHECK(m, hipMemcpyAsync(m->d_a.p, q, 32 * (size_t)nq, hipMemcpyHostToDevice, s));
  MCHECK(m, hipMemcpyAsync(m->d_b.p, c, 32 * (size_t)nc, hipMemcpyHostToDevice, s));
  constexpr int QSLAB = 128;   // query rows per workgroup of the matrix-pipe form: four tiles behind one expansion of its 256 candidates
  if (m->hamming_engine >= 1 && (nq + QSLAB - 1) / QSLAB <= 65535)
    hipLaunchKernelGGL(k_hamming_matrix_mfma, dim3((nc + MF_NT - 1) / MF_NT, (nq + QSLAB - 1) / QSLAB), dim3(MF_NT), 0, s, (const uint32_t *)m->d_a.p, nq,
                       (const uint32_t *)m->d_b.p, nc, (uint16_t *)m->d_c.p, QSLAB);
  else
    hipLaunchKernelGGL(k_hamming_matrix, dim3((nq + 255) / 256), dim3(256), 0, s, (const uint32_t *)m->d_a.p, nq, (const uint32_t *)m->d_b.p, nc,
                       (uint16_t *)m->d_c.p);
  MCHECK(m, hipGetLastError());
  MCHECK(m, hipMemcpyAsync(dist, m->d_c.p, sizeof(uint16_t) * (size_t)nq * nc, hipMemcpyDeviceToHost, s));
  MCHECK(m, hipStreamSynchronize(s));
  return 0;
}

}  // extern "C"
