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
constexpr int kResizeTileH = 32;              // output rows per resize workgroup (kResizeRows in the kernel)
constexpr int kBlurTW = 128, kBlurTH = 32;   // blur output tile per workgroup
constexpr int kTilePitch = 80;     // LDS pitch of a FAST cell tile (3 phase bytes + 76, multiple of 4)

// Packed FAST candidate: y (12 bits, relative to minBorder) | x (12 bits) | response (8 bits).
__host__ __device__ inline uint32_t pack_cand(int x, int y, int resp) { return ((uint32_t)y << 20) | ((uint32_t)x << 8) | (uint32_t)resp; }
__host__ __device__ inline int cand_x(uint32_t p) { return (int)((p >> 8) & 0xfffu); }
__host__ __device__ inline int cand_y(uint32_t p) { return (int)(p >> 20); }
__host__ __device__ inline int cand_resp(uint32_t p) { return (int)(p & 0xffu); }

// Winner of the reference's final pick inside one path-code bucket (ORBextractor.cpp:757-775):
// key = response << 23 | (kPickOrderMask - position in vToDistributeKeys order), larger key wins;
// pos = index of that candidate in the level's bucket-sorted list.  key == 0: empty bucket.
// val = that candidate itself (packed y | x | response), so the host can build the keypoint without the list.
constexpr int kPickOrderMask = (1 << 23) - 1;
struct BucketBest { uint32_t key, pos, val; };   // host statement (test hook): also the winner's position
struct BucketWin { uint32_t key, val; };         // what k_compact ships per bucket: 8 bytes

// Per-image table block k_compact fills in DEVICE memory; one DMA per batch brings the blocks to the host:
// ints [0, 17) level offsets into the image's candidate list (+ the total), [17, 33) `shipped` flag per level,
// [kTblHead, kTblHead + bucketTotal) bucket start offsets, then -- from the next EVEN int on: the records are stored and read
// 8 bytes at a time, and bucketTotal (a sum of nBuckets + 1 over the levels) can be odd -- bucketTotal BucketWin records.
constexpr int kTblLvlOff = 0, kTblShipped = kMaxLevels + 1, kTblHead = 48;
__host__ __device__ inline int tbl_win_off(int bucketTotal) { return kTblHead + ((bucketTotal + 1) & ~1); }
__host__ __device__ inline int tbl_ints(int bucketTotal) { return (tbl_win_off(bucketTotal) + 2 * bucketTotal + 15) & ~15; }

// result of the vocabulary descent of one descriptor: word id and weight of the leaf it reached (word < 0 never
// happens for a well-formed tree), and the node id `levelsup` levels above the leaves (FeatureVector key)
struct BowRes { int32_t word, nodeup; double weight; };

// Packed selected keypoint handed back to the device: level (4) | y (14) | x (14), level coordinates.
__host__ __device__ inline uint32_t pack_sel(int level, int x, int y) { return ((uint32_t)level << 28) | ((uint32_t)y << 14) | (uint32_t)x; }

// The BLURRED planes are stored in tiles of 16 px x 8 rows = 128 bytes (one fabric line): k_describe stages a
// 27 x 27 neighbourhood per keypoint, which is 27 lines of a row-major plane but 12-15 tiles.  Tile (tr, tc) of a
// level sits at (tr * pitch / 16 + tc) * 128, pixel (x, y) at byte (y & 7) * 16 + (x & 15) of its tile.  Only k_blur
// writes this layout and only k_describe* / mcorb_rig_get_blurred read it; the un-blurred pyramid stays row-major.
constexpr int kBlurTileRows = 8, kBlurTileCols = 16, kBlurTileBytes = 128;
#if defined(__HIPCC__)
__host__ __device__
#endif
inline size_t blur_tiled_offset(int pitch, int x, int y)
{
    return ((size_t)(y >> 3) * (size_t)(pitch >> 4) + (size_t)(x >> 4)) * kBlurTileBytes + (size_t)((y & 7) << 4) + (size_t)(x & 15);
}

struct LevelGeom {
    int w, h;            // level size (ORBextractor.cpp:1177-1178)
    int pitch;           // bytes per row in HBM (multiple of 64)
    uint32_t off;        // byte offset of the plane inside one image's pyramid block
    int nCols, nRows;    // cell grid (ORBextractor.cpp:799-800)
    int wCell, hCell;    // (:801-802)
    int cell0;           // index of this level's first cell inside one image
    int tile0;           // index of this level's first blur tile inside one image
    int tilesX, tilesY;
    int maxBorderX, maxBorderY;   // (:790-791)
    uint32_t xtab, ytab; // element offsets into the resize tables (levels >= 1)
    // quad-tree bucketing of the FAST candidates (DistributeOctTree's first `depth` splits)
    int nIni;            // root nodes, round(width/height) (ORBextractor.cpp:558)
    float hX;            // root width (:560)
    int depth;           // path-code depth D: buckets = nIni * 4^D
    int nBuckets;
    int bucket0;         // index of this level's first bucket-start entry inside one image (nBuckets+1 entries)
    int quota;           // mnFeaturesPerLevel[level] (DistributeOctTree's N): candidates are shipped to the host only when
                         // fewer than `quota` buckets are non-empty, i.e. when the tree can go deeper than the bucketing
    // path-code tables (u16 units inside the rig's table array): code(x, y) = lut[lutx + x] | lut[luty + y], see path_code_tables()
    uint32_t lutx, luty;
};

struct Geom {
    int nlevels;
    int cells;            // cells per image, all levels
    int tiles;            // blur tiles per image, all levels
    int cellCap;          // keypoint slots per cell in the scratch array
    uint32_t imgBytes;    // bytes of one image's pyramid block
    int kcap;             // keypoint capacity per image
    int candCap;          // candidates per image in the device list (worst case: cells * cellCap)
    int hostCandCap;      // capacity of the host copy per image (mcorb_params.cand_cap)
    int bucketTotal;      // bucket-start entries per image, all levels
    LevelGeom lv[kMaxLevels];
};

// Quad-tree path of a candidate: root node, then `depth` DivideNode splits
// (ExtractorNode::DivideNode, ORBextractor.cpp:479-535: halfX = ceil((UR.x-UL.x)/2), children
// n1..n4 = (x<split,y<split), (x>=,y<), (x<,y>=), (x>=,y>=)).  Candidates with equal codes are
// exactly the key set of one tree node at that depth, whatever order the tree is expanded in.
__host__ __device__ inline uint32_t path_code(int x, int y, int W0, int H0, int nIni, float hX, int depth)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int r = (int)__fdiv_rn((float)x, hX);                  // vpIniNodes[kp.pt.x/hX] (:584)
    if (r >= nIni) r = nIni - 1;
    int x0 = (int)__fmul_rn(hX, (float)r), x1 = (int)__fmul_rn(hX, (float)(r + 1));   // (:570-571)
#else
    int r = (int)((float)x / hX);
    if (r >= nIni) r = nIni - 1;
    int x0 = (int)(hX * (float)r), x1 = (int)(hX * (float)(r + 1));
#endif
    int y0 = 0, y1 = H0;
    (void)W0;
    uint32_t code = (uint32_t)r;
    for (int d = 0; d < depth; d++) {
        const int sx = x0 + ((x1 - x0 + 1) >> 1), sy = y0 + ((y1 - y0 + 1) >> 1);
        const int qx = x >= sx, qy = y >= sy;
        if (qx) x0 = sx; else x1 = sx;
        if (qy) y0 = sy; else y1 = sy;
        code = (code << 2) | (uint32_t)(qx + 2 * qy);
    }
    return code;
}

// path_code() decides the x half (root node + one bit per split) from x alone and the y half from y alone, so
//   path_code(x, y) == tx[x] | ty[y]   with   tx[x] = root << 2*depth | x-bits on the even positions, ty[y] = y-bits on the odd ones.
// The tables are built on the host with path_code() itself (x in [0, W0), y in [0, H0)): k_compact reads two table entries
// instead of a float division and `depth` split iterations per candidate.
inline void path_code_tables(int W0, int H0, int nIni, float hX, int depth, uint16_t *tx, uint16_t *ty)
{
    uint32_t ymask = 0;
    for (int d = 0; d < depth; d++) ymask |= 2u << (2 * d);
    for (int x = 0; x < W0; x++) tx[x] = (uint16_t)(path_code(x, 0, W0, H0, nIni, hX, depth) & ~ymask);
    for (int y = 0; y < H0; y++) ty[y] = (uint16_t)(path_code(0, y, W0, H0, nIni, hX, depth) & ymask);
}

// resize table entries (built on the host exactly as cv::resize builds xofs/ialpha, yofs/ibeta)
struct ResizeTap { uint16_t s0, s1; int16_t c0, c1; };

}  // namespace mcorb
