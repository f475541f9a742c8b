// orb_common.h -- shared host/device structures of liborbhip (not part of the public ABI).
#pragma once
#include <stdint.h>
#include <stddef.h>

#define ORB_EDGE_THRESHOLD 19  // ORBextractor.cc:72
#define ORB_PATCH_SIZE 31      // ORBextractor.cc:70
#define ORB_HALF_PATCH 15      // ORBextractor.cc:71
#define ORB_MIN_BORDER 16      // EDGE_THRESHOLD-3, ORBextractor.cc:771
#define ORB_DISC_PIXELS 749    // pixels of the radius-15 disc given by umax (ORBextractor.cc:452-467)
#define ORB_MAXL 16

// FAST cell tile in LDS: interior <= 59x59 (wCell = ceil(width/floor(width/30)) < 60), +3 px ring.  The pitch is five
// 16-byte chunks: the tile is filled by global_load_lds_dwordx4, whose LDS image is lane-linear (no padding between rows).
#define FAST_TILE_PITCH 80
#define FAST_TILE_COLS 65      // widest cell window (interior 59 + 6)
#define FAST_TILE_ROWS 76      // tallest group window: two cells of up to 35 rows + 6 (a single cell: up to 59 + 6)
#define FAST_S_PITCH 64  // score plane incl. 1-px zero ring: <= 61 columns

// Geometry of one pyramid level; filled on the host (orbx_configure), read by every kernel.
struct LevelGeom {
  int w, h;          // level image size (ORBextractor.cc:1192-1193)
  int pitch;         // bytes per row of the level plane in the workspace
  int bpitch;        // bytes per row of the blurred plane
  size_t off;        // byte offset of the level plane inside one frame's pyramid block (level 0: unused)
  size_t boff;       // byte offset of the blurred plane inside one frame's blur block
  // FAST cell grid (ORBextractor.cc:771-785)
  int maxBorderX, maxBorderY;
  int nCols, nRows, wCell, hCell;
  int cellCap;       // slots per cell = ceil(wCell/2)*ceil(hCell/2): no two 8-neighbours are both strict maxima
  int cellBase;      // index of this level's first cell in a frame's cell table
  int slotBase;      // index of this level's first slot in a frame's slot array
  // octree (ORBextractor.cc:537-761)
  int N;             // mnFeaturesPerLevel[level]
  int nIni;          // round(width/height), :541
  float hX;          // :543
  int candBase;      // first dense candidate of this level in a frame's candidate arrays
  int candCap;
  int kpBase, kpCap; // level keypoint arrays (octree output)
  float scale;       // mvScaleFactor[level]
  float kpsize;      // (float)(int)(PATCH_SIZE*scale), :862
  // resize tables (level >= 1): index into xtab/ytab
  int xtabBase, ytabBase;
  int resizeRows;    // output rows per k_resize workgroup of this level: 16, or 8 when the 16-row LDS stage would not fit (host)
  int resizeSrcRows; // most source rows any resizeRows-row tile of this level reads (host: sizes k_resize's LDS stage)
  // blur tiles
  int tilesX, tilesY, tileBase;
  uint32_t rowTileMagic;  // floor(2^32 / row tiles of k_resize), see xcd_map
};

// k_pyramid_chain (single-frame calls): one record per 32x32 output tile of a level L >= 1.  rect[l] = the rectangle of level l
// (l < L: everything the tile needs of that level, derived on the host from the resize tables; l = L: the tile itself).
#define CHAIN_TILE 32
#define CHAIN_NT 1024
struct ChainTile {
  uint16_t level, pad;
  uint16_t x[ORB_MAXL], y[ORB_MAXL], w[ORB_MAXL], h[ORB_MAXL];
};

struct FrameParams {
  LevelGeom geom[ORB_MAXL];  // by value: lives in the kernarg segment, so every geometry access is a scalar load
  int nlevels;
  int nframes;
  int iniTh, minTh;
  int lap0, lap1;
  // level 0 = caller's images
  const uint8_t *img0;
  size_t img0_stride, img0_frame_stride;
  // workspace (per-frame strides in elements of the pointed type)
  uint8_t *pyr;    size_t pyr_fs;
  uint8_t *blur;   size_t blur_fs;
  uint32_t *cellCnt; int cell_fs;
  uint32_t *slots;   size_t slot_fs;
  uint32_t *cand;    int cand_fs;   // packed (S<<24 | y<<12 | x), detection-rectangle coordinates
  uint16_t *knode;                  // node id per candidate (same indexing as cand)
  uint32_t *lkp;     int lkp_fs;    // octree output, packed like cand, list order
  uint16_t *lrank;                  // bit15 = inside lapping area, bits0..14 = rank among same class in the level
  int32_t *lcnt;                    // [frame][nlevels][2] = {count, lapped count}
  int32_t *candCnt;                 // [frame][nlevels] dense candidate count
  const int2 *xtab, *ytab;          // resize tables {src index, a0 | a1<<16}
  const int8_t *disc;               // ORB_DISC_PIXELS x (u, v)
  const uint32_t *cells;            // FAST cell records, 8 words each (orbx_configure), same for every frame
  const uint32_t *tiles;            // blur tile records, 8 words each: x0|y0<<16, level, w|h<<16, pitch|bpitch<<16, off, boff
  uint32_t magicCells, magicTiles, magicKpBlk;  // floor(2^32 / items per frame) of k_fast, k_blur, k_describe (xcd_map)
  int totalTiles;                   // blur tiles per frame
  int totalCells;                   // FAST cells per frame
  int totalGroups;                  // k_fast workgroups per frame: groups of up to 2 x 2 neighbouring cells
  uint32_t magicGroups;             // floor(2^32 / totalGroups), xcd_map
  const uint32_t *groups;           // 4 words per group: first cell | gx, gy, cell-row stride | tile origin | tile size, level, valid
  int totalKp;                      // sum of kpCap
  int octCap;                       // node capacity of the octree kernel
  const ChainTile *chain;           // k_pyramid_chain's tile records
  int chainBuf0, chainBuf1;         // bytes of its two LDS level buffers (even / odd levels)
  // outputs
  void *out_kps;       // orbx_keypoint_t [nframes][cap]
  uint8_t *out_desc;   // [nframes][cap][32]
  int32_t *out_counts; // [nframes][2]
  int cap;
};
