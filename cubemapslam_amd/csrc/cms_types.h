// cms_types.h -- plain structs shared by the host side of the C-ABI library and the gfx950 kernels.
#ifndef CMS_TYPES_H
#define CMS_TYPES_H
#include <stddef.h>
#include <stdint.h>

#define CMS_MAX_LEVELS 12
#define CMS_EDGE 19        /* EDGE_THRESHOLD, ORBExtractor.cpp:45 */
#define CMS_MINB 16        /* EDGE_THRESHOLD - 3, ORBExtractor.cpp:747 */

struct CmsLevel {
  int w, h;             // level size: cvRound(W * invScale[l]) (ORBExtractor.cpp:932-933)
  int stride;           // bytes per row in the device pyramid (multiple of 128)
  size_t off;           // byte offset of the level inside one frame's pyramid
  int nCols, nRows, wCell, hCell;  // FAST cell grid (ORBExtractor.cpp:759-762)
  int cell0;            // index of this level's first cell in the flattened (all levels) cell list
  int quota;            // mnFeaturesPerLevel
  int cand_cap;         // capacity of the candidate list of this level
  size_t cand_off;      // offset (entries) of the level's candidate list inside one frame's candidate buffer
  int kp_off;           // offset of the level inside one frame's distributed-key-point list
  float scale;          // mvScaleFactor[l]
  float patch_size;     // (int)(31 * scale) as float (ORBExtractor.cpp:810)
  size_t tab_off;       // offset (entries) of this level's resize coefficient table (x then y)
  int zlo, zhi;         // pixels [0, zlo) and [w - zhi, w) (rows and columns alike) depend only on the corner blocks of the
                        // cross, i.e. are exactly 0 at this level whenever the canvas came from k_remap
};

struct CmsGeom {
  int nlevels, W, F;
  int ini_th, min_th;
  int total_cells;
  int kp_cap;                 // per-frame capacity of the key-point lists: sum(quota + 3)
  int qt_maxn;                // power of two >= max quota + 3
  size_t pyr_bytes;           // bytes per frame pyramid
  size_t cand_total;          // candidate entries per frame
  int tile_h, tile_stride;    // FAST LDS tile geometry (rows, bytes per row, multiple of 4)
  int sc_stride, sc_h;        // FAST LDS score tile
  int list_cap;               // FAST LDS corner list capacity
  int cell_cap;               // capacity of one FAST cell's candidate slot
  int fast_cell_lds;          // LDS bytes of one FAST cell (one wavefront); a workgroup holds CMS_FAST_WPB of them
  int skip_zero_cells;        // set per launch: the canvas was produced by k_remap, FAST cells inside the zero corners are skipped
  int dbg_stop;               // developer switch (env CMS_DBG_FAST_STOP): cut k_fast_cells short after phase N, 0 = off
  int gauss_column_mode;      // 0: integer column pass of the 7x7 Gaussian; 1: SSE2 float column pass of an x86 OpenCV <= 3.2 build (cms_set_gaussian_mode)
  CmsLevel lv[CMS_MAX_LEVELS];
};

// resize coefficient entry: source index + the two 11-bit weights (cv::resize INTER_LINEAR fixed point)
struct CmsResizeTab { short s; short a0; short a1; short pad; };

struct CmsKeyPoint { float x, y, size, angle, response; int octave; };

#endif
