// mcorb_common.h -- structures shared by the host engine and the gfx950 kernels.
#pragma once
#include <stdint.h>
#include <stddef.h>

namespace mcorb {

constexpr int kMaxLevels = 16;
constexpr int kEdge = 19;          // EDGE_THRESHOLD, ORBextractor.cpp:72
constexpr int kMinBorder = 16;     // EDGE_THRESHOLD-3, ORBextractor.cpp:788
constexpr int kCellW = 35;         // W, ORBextractor.cpp:784
constexpr int kMaxRoi = 76;        // wCell < 70 by construction, +6 overlap
constexpr int kTilePitch = 80;     // LDS pitch of a FAST cell tile (3 phase bytes + 76, multiple of 4)

// Packed FAST candidate: y (12 bits, relative to minBorder) | x (12 bits) | response (8 bits).
__host__ __device__ inline uint32_t pack_cand(int x, int y, int resp) { return ((uint32_t)y << 20) | ((uint32_t)x << 8) | (uint32_t)resp; }
__host__ __device__ inline int cand_x(uint32_t p) { return (int)((p >> 8) & 0xfffu); }
__host__ __device__ inline int cand_y(uint32_t p) { return (int)(p >> 20); }
__host__ __device__ inline int cand_resp(uint32_t p) { return (int)(p & 0xffu); }

// Packed selected keypoint handed back to the device: level (4) | y (14) | x (14), level coordinates.
__host__ __device__ inline uint32_t pack_sel(int level, int x, int y) { return ((uint32_t)level << 28) | ((uint32_t)y << 14) | (uint32_t)x; }

struct LevelGeom {
    int w, h;            // level size (ORBextractor.cpp:1177-1178)
    int pitch;           // bytes per row in HBM (multiple of 64)
    uint32_t off;        // byte offset of the plane inside one image's pyramid block
    int nCols, nRows;    // cell grid (ORBextractor.cpp:799-800)
    int wCell, hCell;    // (:801-802)
    int cell0;           // index of this level's first cell inside one image
    int tile0;           // index of this level's first 64x16 blur tile inside one image
    int tilesX, tilesY;
    int maxBorderX, maxBorderY;   // (:790-791)
    uint32_t xtab, ytab; // element offsets into the resize tables (levels >= 1)
};

struct Geom {
    int nlevels;
    int cells;            // cells per image, all levels
    int tiles;            // blur tiles per image, all levels
    int cellCap;          // keypoint slots per cell in the scratch array
    uint32_t imgBytes;    // bytes of one image's pyramid block
    int kcap;             // keypoint capacity per image
    int candCap;          // candidate capacity per image
    LevelGeom lv[kMaxLevels];
};

// resize table entries (built on the host exactly as cv::resize builds xofs/ialpha, yofs/ibeta)
struct ResizeTap { uint16_t s0, s1; int16_t c0, c1; };

}  // namespace mcorb
