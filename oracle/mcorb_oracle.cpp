/*
 * mcorb_oracle.cpp -- CPU restatement (test oracle / timed CPU baseline).
 * See mcorb_oracle.h for the usage rules ("test infrastructure", "parity unpinned").
 *
 * Written to be read side by side with the reference: every routine keeps the
 * reference's (or OpenCV's) evaluation order and numeric types, because the
 * contract is bit-exactness.  C++ (not C) only because DistributeOctTree's
 * result depends on std::list ordering and on std::sort's handling of
 * equivalent elements, which cannot be restated independently of libstdc++.
 * Build: see oracle/Makefile (-O2 as the reference's CMakeLists.txt:34,
 * -ffp-contract=off so no float expression is fused).
 */
#include "mcorb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#include <algorithm>
#include <cmath>
#include <list>
#include <utility>
#include <vector>

/* ------------------------------------------------------------------------- */
/* OpenCV scalar helpers                                         */
/* ------------------------------------------------------------------------- */
extern "C" int orc_cv_round_d(double v) { return (int)lrint(v); }   /* cvRound: cvtsd2si, half-even */
extern "C" int orc_cv_round_f(float v) { return (int)lrintf(v); }
extern "C" int orc_cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
extern "C" int orc_cv_ceil_f(float v) { int i = (int)v; return i + (i < v); }
static inline int cvFloorD(double v) { int i = (int)v; return i - (i > v); }
static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) (A.5) */
static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

static const int PATCH_SIZE = 31;        /* ORBextractor.cpp:70 */
static const int HALF_PATCH_SIZE = 15;   /* :71 */
static const int EDGE_THRESHOLD = 19;    /* :72 */

static const int8_t brief_pattern[1024] = {
#include "../mc-slam_amd/csrc/brief_pattern_31.inc"
};

/* ------------------------------------------------------------------------- */
/* cv::resize INTER_LINEAR, CV_8UC1 (A.3)                                     */
/* ------------------------------------------------------------------------- */
/* The per-axis table loop of cv::resize(): scale = ssize/(double)dsize;
 * fx=(float)((d+0.5)*scale-0.5); s=cvFloor(fx); fx-=s.  For the x axis OpenCV
 * also clamps (s<0 -> s=0,fx=0; s>=ssize-1 -> s=ssize-1,fx=0); the y axis is
 * NOT clamped here but row indices are clipped at use.  This routine returns
 * the unclamped pair; callers apply the axis-specific rule. */
static void axis_table(int ssize, int dsize, int clamp_x, int *ofs, int16_t *coef)
{
    double scale = (double)ssize / dsize;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = orc_cv_floor_f(f);
        f -= s;
        if (clamp_x) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        ofs[d] = s;
        float c0 = 1.f - f, c1 = f;
        coef[2 * d + 0] = sat_short(orc_cv_round_f(c0 * 2048.f));   /* INTER_RESIZE_COEF_SCALE */
        coef[2 * d + 1] = sat_short(orc_cv_round_f(c1 * 2048.f));
    }
}
extern "C" void orc_resize_tables(int ssize, int dsize, int *ofs, int16_t *coef2)
{
    axis_table(ssize, dsize, 1, ofs, coef2);
}

extern "C" void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride,
                                     uint8_t *dst, int dw, int dh, int dstride)
{
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<int16_t> ialpha(2 * dw), ibeta(2 * dh);
    axis_table(sw, dw, 1, xofs.data(), ialpha.data());
    axis_table(sh, dh, 0, yofs.data(), ibeta.data());
    /* xmax: first dx whose sx+1 falls outside the source (HResizeLinear tail) */
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++)
        if (xofs[dx] + 1 >= sw) { xmax = dx < xmax ? dx : xmax; }
    std::vector<int> R0(dw), R1(dw);
    for (int dy = 0; dy < dh; dy++) {
        int sy = yofs[dy];
        /* resizeGeneric_Invoker: rows clip(sy0 - ksize2 + 1 + k, 0, ssize.height), ksize2 = 1 */
        int y0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
        int y1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
        const uint8_t *S0 = src + (size_t)y0 * sstride, *S1 = src + (size_t)y1 * sstride;
        for (int dx = 0; dx < xmax; dx++) {   /* HResizeLinear */
            int sx = xofs[dx];
            int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
            R0[dx] = S0[sx] * a0 + S0[sx + 1] * a1;
            R1[dx] = S1[sx] * a0 + S1[sx + 1] * a1;
        }
        for (int dx = xmax; dx < dw; dx++) {
            int sx = xofs[dx];
            R0[dx] = S0[sx] * 2048;
            R1[dx] = S1[sx] * 2048;
        }
        int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw; x++)   /* VResizeLinear<uchar,int,short,...> */
            D[x] = (uint8_t)((((b0 * (R0[x] >> 4)) >> 16) + ((b1 * (R1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

/* cv::copyMakeBorder BORDER_REFLECT_101 (A.5) */
extern "C" void orc_copy_make_border_101(const uint8_t *src, int w, int h, int sstride,
                                         uint8_t *dst, int dstride, int border)
{
    for (int y = -border; y < h + border; y++) {
        const uint8_t *S = src + (size_t)reflect101(y, h) * sstride;
        uint8_t *D = dst + (size_t)(y + border) * dstride;
        for (int x = -border; x < w + border; x++) D[x + border] = S[reflect101(x, w)];
    }
}

/* ------------------------------------------------------------------------- */
/* cv::FAST TYPE_9_16 (A.2): FAST_t<16> + cornerScore<16>                      */
/* ------------------------------------------------------------------------- */
static const int fast_off16[16][2] = {   /* makeOffsets, patternSize 16: (dx,dy) */
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static void make_offsets(int pixel[25], int stride)
{
    for (int k = 0; k < 16; k++) pixel[k] = fast_off16[k][0] + fast_off16[k][1] * stride;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
}

static int corner_score16(const uint8_t *ptr, const int pixel[25], int threshold)
{
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[N];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);

    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    threshold = -b0 - 1;
    return threshold;
}

extern "C" int orc_fast_corner_score(const uint8_t *img, int stride, int x, int y, int threshold)
{
    int pixel[25];
    make_offsets(pixel, stride);
    return corner_score16(img + (size_t)y * stride + x, pixel, threshold);
}

extern "C" int orc_fast_9_16(const uint8_t *img, int stride, int cols, int rows, int threshold,
                             int nonmax, int *xs, int *ys, int *score, int cap)
{
    const int K = 8, N = 16 + K + 1;
    int i, j, k, pixel[25];
    make_offsets(pixel, stride);
    int nout = 0;

    threshold = std::min(std::max(threshold, 0), 255);
    uint8_t threshold_tab[512];
    for (i = -255; i <= 255; i++)
        threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);

    if (cols <= 0 || rows <= 0) return 0;
    std::vector<uint8_t> bufv((size_t)cols * 3, 0);
    uint8_t *buf[3] = {bufv.data(), bufv.data() + cols, bufv.data() + 2 * cols};
    std::vector<int> cpv((size_t)(cols + 1) * 3, 0);
    int *cpbuf[3] = {cpv.data() + 1, cpv.data() + 1 + (cols + 1), cpv.data() + 1 + 2 * (cols + 1)};

    for (i = 3; i < rows - 2; i++) {
        const uint8_t *ptr = img + (size_t)i * stride + 3;
        uint8_t *curr = buf[(i - 3) % 3];
        int *cornerpos = cpbuf[(i - 3) % 3];
        memset(curr, 0, cols);
        int ncorners = 0;

        if (i < rows - 3) {
            for (j = 3; j < cols - 3; j++, ptr++) {
                int v = ptr[0];
                const uint8_t *tab = &threshold_tab[0] - v + 255;
                int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
                d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
                d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
                d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
                d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
                d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];

                if (d & 1) {
                    int vt = v - threshold, count = 0;
                    for (k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x < vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                if (nonmax) curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                                break;
                            }
                        } else
                            count = 0;
                    }
                }
                if (d & 2) {
                    int vt = v + threshold, count = 0;
                    for (k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x > vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                if (nonmax) curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                                break;
                            }
                        } else
                            count = 0;
                    }
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;

        const uint8_t *prev = buf[(i - 4 + 3) % 3];
        const uint8_t *pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3];
        ncorners = cornerpos[-1];
        for (k = 0; k < ncorners; k++) {
            j = cornerpos[k];
            int sc = prev[j];
            if (!nonmax ||
                (sc > prev[j + 1] && sc > prev[j - 1] &&
                 sc > pprev[j - 1] && sc > pprev[j] && sc > pprev[j + 1] &&
                 sc > curr[j - 1] && sc > curr[j] && sc > curr[j + 1])) {
                if (nout < cap) { xs[nout] = j; ys[nout] = i - 1; score[nout] = sc; }
                nout++;
            }
        }
    }
    return nout;
}

/* ------------------------------------------------------------------------- */
/* cv::GaussianBlur 7x7 sigma 2, CV_8U fixed point (A.4)                       */
/* ------------------------------------------------------------------------- */
/* getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED: exp(-x^2/(2s^2))
 * normalised in double, converted to 8.8 with error diffusion from the outside
 * in, the centre tap taking 256 - sum. */
extern "C" void orc_gaussian_kernel_q8(int taps7[7])
{
    const int n = 7;
    const double sigma = 2.0;
    double k[7], sum = 0;
    double scale2X = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        k[i] = exp(scale2X * x * x);
        sum += k[i];
    }
    for (int i = 0; i < n; i++) k[i] /= sum;
    double err = 0;
    long long s = 0;
    for (int i = 0; i < n / 2; i++) {
        double adj = k[i] * 256.0 + err;
        long long v0 = orc_cv_round_d(adj);
        err = adj - (double)v0;
        taps7[i] = (int)v0;
        taps7[n - 1 - i] = (int)v0;
        s += v0;
    }
    taps7[n / 2] = (int)(256 - 2 * s);
}

extern "C" void orc_gaussian_blur_7x7_s2(const uint8_t *src, int w, int h, int sstride,
                                         uint8_t *dst, int dstride)
{
    int kq[7];
    orc_gaussian_kernel_q8(kq);
    /* horizontal: ufixedpoint16 (8.8) = sum k*src ; vertical: ufixedpoint32 (16.16).
     * Rows/columns are extended by borderInterpolate(REFLECT_101) once, then filtered. */
    std::vector<uint16_t> H((size_t)w * h);
    std::vector<uint8_t> row((size_t)w + 6);
    for (int y = 0; y < h; y++) {
        const uint8_t *S = src + (size_t)y * sstride;
        for (int x = -3; x < w + 3; x++) row[x + 3] = S[(x >= 0 && x < w) ? x : reflect101(x, w)];
        uint16_t *Hr = &H[(size_t)y * w];
        const uint8_t *r = row.data();
        for (int x = 0; x < w; x++) {
            uint32_t acc = (uint32_t)kq[0] * r[x] + (uint32_t)kq[1] * r[x + 1] + (uint32_t)kq[2] * r[x + 2] +
                           (uint32_t)kq[3] * r[x + 3] + (uint32_t)kq[4] * r[x + 4] + (uint32_t)kq[5] * r[x + 5] +
                           (uint32_t)kq[6] * r[x + 6];
            Hr[x] = (uint16_t)acc;   /* <= 65280, never saturates */
        }
    }
    std::vector<uint8_t> out((size_t)w * h);   /* src may alias dst (in-place call in the reference) */
    for (int y = 0; y < h; y++) {
        const uint16_t *R[7];
        for (int t = -3; t <= 3; t++) R[t + 3] = &H[(size_t)reflect101(y + t, h) * w];
        uint8_t *O = &out[(size_t)y * w];
        for (int x = 0; x < w; x++) {
            uint32_t acc = (uint32_t)kq[0] * R[0][x] + (uint32_t)kq[1] * R[1][x] + (uint32_t)kq[2] * R[2][x] +
                           (uint32_t)kq[3] * R[3][x] + (uint32_t)kq[4] * R[4][x] + (uint32_t)kq[5] * R[5][x] +
                           (uint32_t)kq[6] * R[6][x];
            O[x] = sat_u8((int)((acc + 32768u) >> 16));
        }
    }
    for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * dstride, &out[(size_t)y * w], w);
}

/* cv::fastAtan2 (A.8): scalar path of fastAtan32f_ */
extern "C" float orc_fast_atan2(float y, float x)
{
    const float atan2_p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float atan2_p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float atan2_p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float atan2_p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    } else {
        c = ax / (ay + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* MultiCameraFrame::setData hand-off, MultiCameraFrame.cpp:108-116 (A.6) */
extern "C" void orc_stage_f32(const float *img, int w, int h, int stride_bytes, int channels,
                              uint8_t *gray, int gstride)
{
    for (int y = 0; y < h; y++) {
        const float *S = (const float *)((const char *)img + (size_t)y * stride_bytes);
        uint8_t *D = gray + (size_t)y * gstride;
        for (int x = 0; x < w; x++) {
            if (channels == 1) {
                float m = S[x] * 255.f;                       /* multiply(img,255,img) */
                D[x] = sat_u8(orc_cv_round_f(m));             /* convertTo(CV_8U) */
            } else {
                int b = sat_u8(orc_cv_round_f(S[3 * x + 0] * 255.f));
                int g = sat_u8(orc_cv_round_f(S[3 * x + 1] * 255.f));
                int r = sat_u8(orc_cv_round_f(S[3 * x + 2] * 255.f));
                /* cvtColor BGR2GRAY, 8U: (B*1868 + G*9617 + R*4899 + (1<<13)) >> 14 */
                D[x] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14);
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* DistributeOctTree (ORBextractor.cpp:479-778)                               */
/* ------------------------------------------------------------------------- */
namespace {
struct OKey { float x, y, response; int idx; };
struct P2i { int x, y; };
struct ONode {
    ONode() : bNoMore(false) {}
    void DivideNode(ONode &n1, ONode &n2, ONode &n3, ONode &n4);
    std::vector<OKey> vKeys;
    P2i UL, UR, BL, BR;
    std::list<ONode>::iterator lit;
    bool bNoMore;
};

/* ExtractorNode::DivideNode, :479-535 */
void ONode::DivideNode(ONode &n1, ONode &n2, ONode &n3, ONode &n4)
{
    const int halfX = ceil(static_cast<float>(UR.x - UL.x) / 2);
    const int halfY = ceil(static_cast<float>(BR.y - UL.y) / 2);

    n1.UL = UL;
    n1.UR = P2i{UL.x + halfX, UL.y};
    n1.BL = P2i{UL.x, UL.y + halfY};
    n1.BR = P2i{UL.x + halfX, UL.y + halfY};
    n1.vKeys.reserve(vKeys.size());

    n2.UL = n1.UR;
    n2.UR = UR;
    n2.BL = n1.BR;
    n2.BR = P2i{UR.x, UL.y + halfY};
    n2.vKeys.reserve(vKeys.size());

    n3.UL = n1.BL;
    n3.UR = n1.BR;
    n3.BL = BL;
    n3.BR = P2i{n1.BR.x, BL.y};
    n3.vKeys.reserve(vKeys.size());

    n4.UL = n3.UR;
    n4.UR = n2.BR;
    n4.BL = n3.BR;
    n4.BR = BR;
    n4.vKeys.reserve(vKeys.size());

    for (size_t i = 0; i < vKeys.size(); i++) {
        const OKey &kp = vKeys[i];
        if (kp.x < n1.UR.x) {
            if (kp.y < n1.BR.y) n1.vKeys.push_back(kp);
            else n3.vKeys.push_back(kp);
        } else if (kp.y < n1.BR.y)
            n2.vKeys.push_back(kp);
        else
            n4.vKeys.push_back(kp);
    }
    if (n1.vKeys.size() == 1) n1.bNoMore = true;
    if (n2.vKeys.size() == 1) n2.bNoMore = true;
    if (n3.vKeys.size() == 1) n3.bNoMore = true;
    if (n4.vKeys.size() == 1) n4.bNoMore = true;
}

/* compareNodes, :537-552 */
bool compareNodes(std::pair<int, ONode *> &e1, std::pair<int, ONode *> &e2)
{
    if (e1.first < e2.first) return true;
    else if (e1.first > e2.first) return false;
    else {
        if (e1.second->UL.x < e2.second->UL.x) return true;
        else return false;
    }
}

/* DistributeOctTree, :554-778.  Returns retained keys in list order. */
int distribute_octree(const std::vector<OKey> &vToDistributeKeys, int minX, int maxX, int minY,
                      int maxY, int N, std::vector<OKey> &vResultKeys)
{
    vResultKeys.clear();
    const int nIni = round(static_cast<float>(maxX - minX) / (maxY - minY));
    if (nIni < 1) return -2;   /* reference: division by zero / out-of-range root index */
    const float hX = static_cast<float>(maxX - minX) / nIni;

    std::list<ONode> lNodes;
    std::vector<ONode *> vpIniNodes;
    vpIniNodes.resize(nIni);

    for (int i = 0; i < nIni; i++) {
        ONode ni;
        ni.UL = P2i{(int)(hX * static_cast<float>(i)), 0};
        ni.UR = P2i{(int)(hX * static_cast<float>(i + 1)), 0};
        ni.BL = P2i{ni.UL.x, maxY - minY};
        ni.BR = P2i{ni.UR.x, maxY - minY};
        ni.vKeys.reserve(vToDistributeKeys.size());
        lNodes.push_back(ni);
        vpIniNodes[i] = &lNodes.back();
    }
    for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
        const OKey &kp = vToDistributeKeys[i];
        vpIniNodes[(int)(kp.x / hX)]->vKeys.push_back(kp);
    }

    std::list<ONode>::iterator lit = lNodes.begin();
    while (lit != lNodes.end()) {
        if (lit->vKeys.size() == 1) { lit->bNoMore = true; lit++; }
        else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
        else lit++;
    }

    bool bFinish = false;
    int iteration = 0;
    std::vector<std::pair<int, ONode *> > vSizeAndPointerToNode;
    vSizeAndPointerToNode.reserve(lNodes.size() * 4);

    while (!bFinish) {
        iteration++;
        int prevSize = lNodes.size();
        lit = lNodes.begin();
        int nToExpand = 0;
        vSizeAndPointerToNode.clear();

        while (lit != lNodes.end()) {
            if (lit->bNoMore) { lit++; continue; }
            else {
                ONode n1, n2, n3, n4;
                lit->DivideNode(n1, n2, n3, n4);
                ONode *ch[4] = {&n1, &n2, &n3, &n4};
                for (int c = 0; c < 4; c++) {
                    if (ch[c]->vKeys.size() > 0) {
                        lNodes.push_front(*ch[c]);
                        if (ch[c]->vKeys.size() > 1) {
                            nToExpand++;
                            vSizeAndPointerToNode.push_back(std::make_pair((int)ch[c]->vKeys.size(), &lNodes.front()));
                            lNodes.front().lit = lNodes.begin();
                        }
                    }
                }
                lit = lNodes.erase(lit);
                continue;
            }
        }

        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
            bFinish = true;
        } else if (((int)lNodes.size() + nToExpand * 3) > N) {
            while (!bFinish) {
                prevSize = lNodes.size();
                std::vector<std::pair<int, ONode *> > vPrevSizeAndPointerToNode = vSizeAndPointerToNode;
                vSizeAndPointerToNode.clear();

                std::sort(vPrevSizeAndPointerToNode.begin(), vPrevSizeAndPointerToNode.end(), compareNodes);
                for (int j = vPrevSizeAndPointerToNode.size() - 1; j >= 0; j--) {
                    ONode n1, n2, n3, n4;
                    vPrevSizeAndPointerToNode[j].second->DivideNode(n1, n2, n3, n4);
                    ONode *ch[4] = {&n1, &n2, &n3, &n4};
                    for (int c = 0; c < 4; c++) {
                        if (ch[c]->vKeys.size() > 0) {
                            lNodes.push_front(*ch[c]);
                            if (ch[c]->vKeys.size() > 1) {
                                vSizeAndPointerToNode.push_back(std::make_pair((int)ch[c]->vKeys.size(), &lNodes.front()));
                                lNodes.front().lit = lNodes.begin();
                            }
                        }
                    }
                    lNodes.erase(vPrevSizeAndPointerToNode[j].second->lit);
                    if ((int)lNodes.size() >= N) break;
                }
                if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
            }
        }
    }

    /* retain the best point in each node (first max wins) */
    for (std::list<ONode>::iterator it = lNodes.begin(); it != lNodes.end(); it++) {
        std::vector<OKey> &vNodeKeys = it->vKeys;
        OKey *pKP = &vNodeKeys[0];
        float maxResponse = pKP->response;
        for (size_t k = 1; k < vNodeKeys.size(); k++) {
            if (vNodeKeys[k].response > maxResponse) {
                pKP = &vNodeKeys[k];
                maxResponse = vNodeKeys[k].response;
            }
        }
        vResultKeys.push_back(*pKP);
    }
    return (int)vResultKeys.size();
}
}   // namespace

extern "C" int orc_distribute_octree(const float *x, const float *y, const float *resp, int n,
                                     int minX, int maxX, int minY, int maxY, int N, int *out_idx, int cap)
{
    std::vector<OKey> in(n), out;
    for (int i = 0; i < n; i++) in[i] = OKey{x[i], y[i], resp[i], i};
    int r = distribute_octree(in, minX, maxX, minY, maxY, N, out);
    if (r < 0) return r;
    for (int i = 0; i < r && i < cap; i++) out_idx[i] = out[i].idx;
    return r;
}

/* ------------------------------------------------------------------------- */
/* extractor object                                                           */
/* ------------------------------------------------------------------------- */
struct orc_extractor {
    int nfeatures;
    double scaleFactor;          /* ORBextractor.h:103 -- a double member fed from a float argument */
    int nlevels, iniThFAST, minThFAST, orientation;
    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    /* state of the last call */
    struct Plane { int w = 0, h = 0; std::vector<uint8_t> bordered, blurred; bool has_blur = false; };
    std::vector<Plane> pyr;
    std::vector<std::vector<OKey> > cand;
    std::vector<std::vector<orc_keypoint> > lvlkps;
};

extern "C" orc_extractor *orc_create(int _nfeatures, float _scaleFactor, int _nlevels,
                                     int _iniThFAST, int _minThFAST, int orientation)
{
    if (_nlevels < 1 || _nlevels > ORC_MAX_LEVELS) return NULL;
    orc_extractor *e = new orc_extractor;
    e->nfeatures = _nfeatures; e->scaleFactor = _scaleFactor; e->nlevels = _nlevels;
    e->iniThFAST = _iniThFAST; e->minThFAST = _minThFAST; e->orientation = orientation;
    const int nlevels = _nlevels;
    /* ORBextractor.cpp:413-429 */
    e->mvScaleFactor.resize(nlevels);
    e->mvLevelSigma2.resize(nlevels);
    e->mvScaleFactor[0] = 1.0f;
    e->mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        e->mvScaleFactor[i] = e->mvScaleFactor[i - 1] * e->scaleFactor;   /* float*double -> float */
        e->mvLevelSigma2[i] = e->mvScaleFactor[i] * e->mvScaleFactor[i];
    }
    e->mvInvScaleFactor.resize(nlevels);
    e->mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
        e->mvInvScaleFactor[i] = 1.0f / e->mvScaleFactor[i];
        e->mvInvLevelSigma2[i] = 1.0f / e->mvLevelSigma2[i];
    }
    /* :433-444 */
    e->mnFeaturesPerLevel.resize(nlevels);
    float factor = 1.0f / e->scaleFactor;
    float nDesiredFeaturesPerScale =
        e->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        e->mnFeaturesPerLevel[level] = orc_cv_round_f(nDesiredFeaturesPerScale);
        sumFeatures += e->mnFeaturesPerLevel[level];
        nDesiredFeaturesPerScale *= factor;
    }
    e->mnFeaturesPerLevel[nlevels - 1] = std::max(e->nfeatures - sumFeatures, 0);
    /* :450-467 */
    e->umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = orc_cv_floor_f(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1);
    int vmin = orc_cv_ceil_f(HALF_PATCH_SIZE * sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) e->umax[v] = orc_cv_round_d(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    return e;
}
extern "C" void orc_destroy(orc_extractor *e) { delete e; }

extern "C" void orc_get_tables(const orc_extractor *e, float *scale, float *inv_scale,
                               float *sigma2, float *inv_sigma2, int *quota, int *umax16)
{
    for (int i = 0; i < e->nlevels; i++) {
        if (scale) scale[i] = e->mvScaleFactor[i];
        if (inv_scale) inv_scale[i] = e->mvInvScaleFactor[i];
        if (sigma2) sigma2[i] = e->mvLevelSigma2[i];
        if (inv_sigma2) inv_sigma2[i] = e->mvInvLevelSigma2[i];
        if (quota) quota[i] = e->mnFeaturesPerLevel[i];
    }
    if (umax16) for (int i = 0; i <= HALF_PATCH_SIZE; i++) umax16[i] = e->umax[i];
}

extern "C" void orc_level_size(const orc_extractor *e, int level, int w, int h, int *lw, int *lh)
{
    float scale = e->mvInvScaleFactor[level];
    *lw = orc_cv_round_f((float)w * scale);
    *lh = orc_cv_round_f((float)h * scale);
}

/* IC_Angle, ORBextractor.cpp:75-102 (dead in the reference; orientation==1 only) */
static float ic_angle(const uint8_t *img, int step, float ptx, float pty, const std::vector<int> &u_max)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = img + (size_t)orc_cv_round_f(pty) * step + orc_cv_round_f(ptx);
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = u_max[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* computeOrbDescriptor, ORBextractor.cpp:105-145 */
static const float factorPI = (float)(M_PI / 180.f);
static void compute_orb_descriptor(float kx, float ky, float kangle, const uint8_t *img, int step,
                                   uint8_t *desc)
{
    float angle = (float)kangle * factorPI;
    float a = (float)cosf(angle), b = (float)sinf(angle);   /* reference: cos(float) under `using namespace std` resolves to the float overload */
    const uint8_t *center = img + (size_t)orc_cv_round_f(ky) * step + orc_cv_round_f(kx);
    const int8_t *pattern = brief_pattern;
#define GET_VALUE(idx)                                                                         \
    center[orc_cv_round_f(pattern[2 * (idx)] * b + pattern[2 * (idx) + 1] * a) * step +       \
           orc_cv_round_f(pattern[2 * (idx)] * a - pattern[2 * (idx) + 1] * b)]
    for (int i = 0; i < 32; ++i, pattern += 32) {
        int t0, t1, val;
        t0 = GET_VALUE(0); t1 = GET_VALUE(1); val = t0 < t1;
        t0 = GET_VALUE(2); t1 = GET_VALUE(3); val |= (t0 < t1) << 1;
        t0 = GET_VALUE(4); t1 = GET_VALUE(5); val |= (t0 < t1) << 2;
        t0 = GET_VALUE(6); t1 = GET_VALUE(7); val |= (t0 < t1) << 3;
        t0 = GET_VALUE(8); t1 = GET_VALUE(9); val |= (t0 < t1) << 4;
        t0 = GET_VALUE(10); t1 = GET_VALUE(11); val |= (t0 < t1) << 5;
        t0 = GET_VALUE(12); t1 = GET_VALUE(13); val |= (t0 < t1) << 6;
        t0 = GET_VALUE(14); t1 = GET_VALUE(15); val |= (t0 < t1) << 7;
        desc[i] = (uint8_t)val;
    }
#undef GET_VALUE
}

extern "C" int orc_extract(orc_extractor *e, const uint8_t *gray, int w, int h, int stride,
                           int lap_x0, int lap_x1, orc_keypoint *okps, uint8_t *odesc, int cap,
                           int *n_out)
{
    if (n_out) *n_out = 0;
    if (!gray || w <= 0 || h <= 0) return -1;   /* _image.empty(), :1090-1091 */
    const int nlevels = e->nlevels;
    const int B = EDGE_THRESHOLD;

    /* --- ComputePyramid, :1173-1198 --- */
    e->pyr.assign(nlevels, orc_extractor::Plane());
    for (int level = 0; level < nlevels; ++level) {
        int lw, lh;
        orc_level_size(e, level, w, h, &lw, &lh);
        if (lw - 2 * B + 6 < 35 || lh - 2 * B + 6 < 35) return -2;   /* nCols/nRows would be 0 (:799-802) */
        orc_extractor::Plane &P = e->pyr[level];
        P.w = lw; P.h = lh;
        const int bs = lw + 2 * B;
        P.bordered.assign((size_t)bs * (lh + 2 * B), 0);
        uint8_t *interior = P.bordered.data() + (size_t)B * bs + B;
        if (level != 0) {
            const orc_extractor::Plane &Q = e->pyr[level - 1];
            const int qs = Q.w + 2 * B;
            orc_resize_linear_u8(Q.bordered.data() + (size_t)B * qs + B, Q.w, Q.h, qs, interior, lw, lh, bs);
            std::vector<uint8_t> tmp((size_t)lw * lh);
            for (int y = 0; y < lh; y++) memcpy(&tmp[(size_t)y * lw], interior + (size_t)y * bs, lw);
            orc_copy_make_border_101(tmp.data(), lw, lh, lw, P.bordered.data(), bs, B);
        } else {
            orc_copy_make_border_101(gray, w, h, stride, P.bordered.data(), bs, B);
        }
    }

    /* --- ComputeKeyPointsOctTree, :780-895 --- */
    e->cand.assign(nlevels, std::vector<OKey>());
    e->lvlkps.assign(nlevels, std::vector<orc_keypoint>());
    const float W = 35;
    std::vector<int> fx(8192), fy(8192), fs(8192);
    for (int level = 0; level < nlevels; ++level) {
        const orc_extractor::Plane &P = e->pyr[level];
        const int bs = P.w + 2 * B;
        const uint8_t *img = P.bordered.data() + (size_t)B * bs + B;   /* mvImagePyramid[level] */
        const int minBorderX = EDGE_THRESHOLD - 3;
        const int minBorderY = minBorderX;
        const int maxBorderX = P.w - EDGE_THRESHOLD + 3;
        const int maxBorderY = P.h - EDGE_THRESHOLD + 3;

        std::vector<OKey> &vToDistributeKeys = e->cand[level];
        vToDistributeKeys.reserve(e->nfeatures * 10);

        const float width = (maxBorderX - minBorderX);
        const float height = (maxBorderY - minBorderY);
        const int nCols = width / W;
        const int nRows = height / W;
        const int wCell = ceil(width / nCols);
        const int hCell = ceil(height / nRows);

        for (int i = 0; i < nRows; i++) {
            const float iniY = minBorderY + i * hCell;
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = minBorderX + j * wCell;
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = maxBorderX;

                const int r0 = (int)iniY, r1 = (int)maxY, c0 = (int)iniX, c1 = (int)maxX;
                const uint8_t *roi = img + (size_t)r0 * bs + c0;
                if ((size_t)(c1 - c0) * (r1 - r0) > fx.size()) {
                    fx.resize((size_t)(c1 - c0) * (r1 - r0)); fy.resize(fx.size()); fs.resize(fx.size());
                }
                int nk = orc_fast_9_16(roi, bs, c1 - c0, r1 - r0, e->iniThFAST, 1, fx.data(), fy.data(),
                                       fs.data(), (int)fx.size());
                if (nk == 0)
                    nk = orc_fast_9_16(roi, bs, c1 - c0, r1 - r0, e->minThFAST, 1, fx.data(), fy.data(),
                                       fs.data(), (int)fx.size());
                for (int k = 0; k < nk; k++) {
                    OKey kp;
                    kp.x = (float)fx[k]; kp.y = (float)fy[k]; kp.response = (float)fs[k];
                    kp.x += j * wCell;
                    kp.y += i * hCell;
                    kp.idx = (int)vToDistributeKeys.size();
                    vToDistributeKeys.push_back(kp);
                }
            }
        }

        std::vector<OKey> sel;
        int r = distribute_octree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                  e->mnFeaturesPerLevel[level], sel);
        if (r < 0) return -2;
        const int scaledPatchSize = PATCH_SIZE * e->mvScaleFactor[level];
        std::vector<orc_keypoint> &keypoints = e->lvlkps[level];
        keypoints.resize(sel.size());
        for (size_t i = 0; i < sel.size(); i++) {
            orc_keypoint &k = keypoints[i];
            k.x = sel[i].x; k.y = sel[i].y;
            k.size = 7.f; k.angle = -1.f; k.response = sel[i].response; k.octave = 0; k.class_id = -1;
            k.x += minBorderX;
            k.y += minBorderY;
            k.octave = level;
            k.size = scaledPatchSize;
        }
    }
    /* computeOrientation, :470-477 / :893-894 */
    for (int level = 0; level < nlevels; ++level) {
        const orc_extractor::Plane &P = e->pyr[level];
        const int bs = P.w + 2 * B;
        const uint8_t *img = P.bordered.data() + (size_t)B * bs + B;
        for (orc_keypoint &k : e->lvlkps[level])
            k.angle = e->orientation ? ic_angle(img, bs, k.x, k.y, e->umax) : 0.0f;
    }

    /* --- operator() assembly, :1103-1170 --- */
    int nkeypoints = 0;
    for (int level = 0; level < nlevels; ++level) nkeypoints += (int)e->lvlkps[level].size();
    if (n_out) *n_out = nkeypoints;
    if (nkeypoints > cap) return -3;

    int monoIndex = 0, stereoIndex = nkeypoints - 1;
    for (int level = 0; level < nlevels; ++level) {
        std::vector<orc_keypoint> keypoints = e->lvlkps[level];   /* copy: scaling below must not touch the stored level coords */
        int nkeypointsLevel = (int)keypoints.size();
        if (nkeypointsLevel == 0) continue;

        orc_extractor::Plane &P = e->pyr[level];
        const int bs = P.w + 2 * B;
        /* workingMat = mvImagePyramid[level].clone(); GaussianBlur(7x7, 2, 2, REFLECT_101) */
        P.blurred.assign((size_t)P.w * P.h, 0);
        orc_gaussian_blur_7x7_s2(P.bordered.data() + (size_t)B * bs + B, P.w, P.h, bs, P.blurred.data(), P.w);
        P.has_blur = true;

        std::vector<uint8_t> desc((size_t)nkeypointsLevel * 32);
        for (int i = 0; i < nkeypointsLevel; i++)
            compute_orb_descriptor(keypoints[i].x, keypoints[i].y, keypoints[i].angle, P.blurred.data(),
                                   P.w, &desc[(size_t)i * 32]);

        float scale = e->mvScaleFactor[level];
        for (int i = 0; i < nkeypointsLevel; i++) {
            orc_keypoint kp = keypoints[i];
            if (level != 0) { kp.x *= scale; kp.y *= scale; }
            if (kp.x >= lap_x0 && kp.x <= lap_x1) {
                okps[stereoIndex] = kp;
                memcpy(odesc + (size_t)stereoIndex * 32, &desc[(size_t)i * 32], 32);
                stereoIndex--;
            } else {
                okps[monoIndex] = kp;
                memcpy(odesc + (size_t)monoIndex * 32, &desc[(size_t)i * 32], 32);
                monoIndex++;
            }
        }
    }
    return monoIndex;
}

extern "C" const uint8_t *orc_last_level_bordered(const orc_extractor *e, int level, int *w, int *h, int *stride)
{
    if (level < 0 || level >= (int)e->pyr.size()) return NULL;
    const orc_extractor::Plane &P = e->pyr[level];
    *w = P.w + 2 * EDGE_THRESHOLD; *h = P.h + 2 * EDGE_THRESHOLD; *stride = *w;
    return P.bordered.data();
}
extern "C" const uint8_t *orc_last_level(const orc_extractor *e, int level, int *w, int *h, int *stride)
{
    if (level < 0 || level >= (int)e->pyr.size()) return NULL;
    const orc_extractor::Plane &P = e->pyr[level];
    *w = P.w; *h = P.h; *stride = P.w + 2 * EDGE_THRESHOLD;
    return P.bordered.data() + (size_t)EDGE_THRESHOLD * (*stride) + EDGE_THRESHOLD;
}
extern "C" const uint8_t *orc_last_blurred(const orc_extractor *e, int level, int *w, int *h, int *stride)
{
    if (level < 0 || level >= (int)e->pyr.size() || !e->pyr[level].has_blur) return NULL;
    const orc_extractor::Plane &P = e->pyr[level];
    *w = P.w; *h = P.h; *stride = P.w;
    return P.blurred.data();
}
extern "C" int orc_last_candidates(const orc_extractor *e, int level, float *x, float *y, float *resp, int cap)
{
    if (level < 0 || level >= (int)e->cand.size()) return -1;
    const std::vector<OKey> &c = e->cand[level];
    for (int i = 0; i < (int)c.size() && i < cap; i++) { x[i] = c[i].x; y[i] = c[i].y; resp[i] = c[i].response; }
    return (int)c.size();
}
extern "C" int orc_last_level_keypoints(const orc_extractor *e, int level, orc_keypoint *kps, int cap)
{
    if (level < 0 || level >= (int)e->lvlkps.size()) return -1;
    const std::vector<orc_keypoint> &c = e->lvlkps[level];
    for (int i = 0; i < (int)c.size() && i < cap; i++) kps[i] = c[i];
    return (int)c.size();
}

/* ------------------------------------------------------------------------- */
/* descriptors / matching                                                     */
/* ------------------------------------------------------------------------- */
/* ORBextractor::DescriptorDistance, :1202-1218 */
extern "C" int orc_descriptor_distance(const uint8_t a[32], const uint8_t b[32])
{
    int32_t pa[8], pb[8];
    memcpy(pa, a, 32); memcpy(pb, b, 32);
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        unsigned int v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* ORBextractor::getMatches_distRatio, :1228-1290 (TH_LOW = 75, ORBextractor.h:27) */
extern "C" int orc_get_matches_dist_ratio(const uint8_t *A, const uint32_t *iA, int nA,
                                          const uint8_t *B, const uint32_t *iB, int nB,
                                          double max_neighbor_ratio,
                                          uint32_t *mA, uint32_t *mB, int *bookK)
{
    const int TH_LOW = 75;
    std::vector<unsigned int> i_match_A, i_match_B;
    int BookK = bookK ? *bookK : 0;
    for (int ai = 0; ai < nA; ai++) {
        int best_j_now = -1;
        double best_dist_1 = 1e9;
        double best_dist_2 = 1e9;
        for (int j = 0; j < nB; j++) {
            double d = orc_descriptor_distance(A + (size_t)iA[ai] * 32, B + (size_t)iB[j] * 32);
            BookK++;
            if (d < best_dist_1) {
                best_j_now = j;
                best_dist_2 = best_dist_1;
                best_dist_1 = d;
            } else if (d < best_dist_2) {
                best_dist_2 = d;
            }
        }
        if (best_dist_1 <= TH_LOW) {
            if (best_dist_1 / best_dist_2 <= max_neighbor_ratio) {
                unsigned int idx_B = iB[best_j_now];
                std::vector<unsigned int>::iterator bit = std::find(i_match_B.begin(), i_match_B.end(), idx_B);
                if (bit == i_match_B.end()) {
                    i_match_B.push_back(idx_B);
                    i_match_A.push_back(iA[ai]);
                } else {
                    unsigned int idx_A = i_match_A[bit - i_match_B.begin()];
                    double d = orc_descriptor_distance(A + (size_t)idx_A * 32, B + (size_t)idx_B * 32);
                    BookK++;
                    if (best_dist_1 < d) i_match_A[bit - i_match_B.begin()] = iA[ai];
                }
            }
        }
    }
    if (bookK) *bookK = BookK;
    for (size_t k = 0; k < i_match_A.size(); k++) { mA[k] = i_match_A[k]; mB[k] = i_match_B[k]; }
    return (int)i_match_A.size();
}

/* BFMatcher(NORM_HAMMING)::knnMatch k=2 == cv::batchDistance K=2 insertion (A.7) */
extern "C" void orc_knn2(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx, int32_t *dist)
{
    const int K = 2;
    for (int i = 0; i < nq; i++) {
        int32_t *nidxptr = idx + (size_t)i * K, *distptr = dist + (size_t)i * K;
        for (int k = 0; k < K; k++) { nidxptr[k] = -1; distptr[k] = INT_MAX; }
        for (int j = 0; j < nt; j++) {
            /* normHamming over 32 bytes (OpenCV uses hardware popcount where present) */
            uint64_t a[4], b[4];
            memcpy(a, q + (size_t)i * 32, 32);
            memcpy(b, t + (size_t)j * 32, 32);
            int d = __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) +
                    __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
            if (d < distptr[K - 1]) {
                int k;
                for (k = K - 2; k >= 0 && distptr[k] > d; k--) {
                    nidxptr[k + 1] = nidxptr[k];
                    distptr[k + 1] = distptr[k];
                }
                nidxptr[k + 1] = j;
                distptr[k + 1] = d;
            }
        }
        for (int k = 0; k < K; k++) if (nidxptr[k] < 0) distptr[k] = -1;
    }
}

/* MultiCameraFrame::BruteForceMatch, MultiCameraFrame.cpp:1024-1086 */
extern "C" int orc_bruteforce_match(const uint8_t *q, int nq, const uint8_t *t, int nt,
                                    float dist_thresh, float neigh_ratio,
                                    uint32_t *idx1, uint32_t *idx2, int cap)
{
    std::vector<int32_t> idx((size_t)nq * 2), dist((size_t)nq * 2);
    orc_knn2(q, nq, t, nt, idx.data(), dist.data());
    int n = 0;
    for (int i = 0; i < nq; i++) {
        if (idx[2 * i] < 0 || idx[2 * i + 1] < 0) continue;   /* reference indexes m[1] unguarded (:1061) */
        float d0 = (float)dist[2 * i], d1 = (float)dist[2 * i + 1];   /* DMatch::distance is float */
        if (d0 < neigh_ratio * d1) {
            if (d0 > dist_thresh) continue;
            if (n < cap) { idx1[n] = (uint32_t)i; idx2[n] = (uint32_t)idx[2 * i]; }
            n++;
        }
    }
    return n;
}

/* MultiCameraFrame::computeIntraMatches(matches, old), :1100-1288.  old=true when F != NULL: F holds one
 * row-major 3x3 per pair (the Mat built at :1126-1142 -- that cv::Mat algebra is outside this restatement),
 * kps[c] are image_kps_undist[c], sigma2 = GetScaleSigmaSquares(). */
extern "C" int orc_intra_matches_epi(const uint8_t *const *desc, const int *n, int ncams,
                                     float dist_thresh, float neigh_ratio, const double *Fall,
                                     const orc_keypoint *const *kps, const float *sigma2,
                                     int32_t *tracks, int cap_tracks, int *mergeable)
{
    const bool old = Fall != NULL;
    int pair = 0;
    std::vector<std::vector<int> > match_inv_idx;
    for (int i = 0; i < ncams; i++) match_inv_idx.push_back(std::vector<int>(n[i], -1));
    std::vector<std::vector<int> > matches;   /* IntraMatch::matchIndex, widened to ncams */
    int intramatches = 0, cnt_mergable_matches = 0;
    for (int i = 0; i < ncams - 1; i++) {
        for (int j = i + 1; j < ncams; j++) {
            std::vector<uint32_t> indices1(n[i] + 1), indices2(n[i] + 1);
            int nm = orc_bruteforce_match(desc[i], n[i], desc[j], n[j], dist_thresh, neigh_ratio,
                                          indices1.data(), indices2.data(), n[i]);
            const double *F = old ? Fall + 9 * pair : NULL;   /* F.at<double>(r,c) = F[3*r+c] */
            pair++;
            for (int k = 0; k < nm; k++) {
                int cami_feat = indices1[k];
                int camj_feat = indices2[k];
                int match_idx = match_inv_idx[i][cami_feat];
                int match_idx_2 = match_inv_idx[j][camj_feat];
                if (old) {   /* :1178-1207 */
                    const orc_keypoint &kp1 = kps[i][cami_feat], &kp2 = kps[j][camj_feat];
                    bool res;
                    float a = kp2.x * F[0] + kp2.y * F[3] + F[6];
                    float b = kp2.x * F[1] + kp2.y * F[4] + F[7];
                    float c = kp2.x * F[2] + kp2.y * F[5] + F[8];
                    float den = a * a + b * b;
                    den = den ? 1. / std::sqrt(den) : 1.;
                    a *= den; b *= den; c *= den;
                    den = a * a + b * b;
                    float num = a * kp1.x + b * kp1.y + c;
                    float dsqr;
                    float check_thresh;
                    if (den == 0)
                        res = false;
                    else {
                        dsqr = num * num / den;
                        check_thresh = 3.84 * sigma2[kp1.octave];
                        res = dsqr < check_thresh ? true : false;
                    }
                    if (!res) continue;
                }
                if (match_idx == -1 && match_idx_2 == -1) {
                    std::vector<int> temp(ncams, -1);
                    temp[i] = cami_feat;
                    temp[j] = camj_feat;
                    matches.push_back(temp);
                    match_inv_idx[i][cami_feat] = intramatches;
                    match_inv_idx[j][camj_feat] = intramatches;
                    intramatches++;
                } else {
                    if (match_idx == -1 && match_idx_2 != -1) {
                        if (matches[match_idx_2][i] == -1) {
                            matches[match_idx_2][i] = cami_feat;
                            match_inv_idx[i][cami_feat] = match_idx_2;
                        }
                    }
                    if (match_idx != -1 && match_idx_2 != -1) {
                        if (match_idx != match_idx_2) cnt_mergable_matches++;
                    }
                    if (match_idx != -1 && match_idx_2 == -1) {
                        matches[match_idx][j] = camj_feat;
                        match_inv_idx[j][camj_feat] = match_idx;
                    }
                }
            }
        }
    }
    if (mergeable) *mergeable = cnt_mergable_matches;
    for (size_t m = 0; m < matches.size() && (int)m < cap_tracks; m++)
        for (int c = 0; c < ncams; c++) tracks[m * ncams + c] = matches[m][c];
    return (int)matches.size();
}

extern "C" int orc_intra_matches(const uint8_t *const *desc, const int *n, int ncams,
                                 float dist_thresh, float neigh_ratio,
                                 int32_t *tracks, int cap_tracks, int *mergeable)
{
    return orc_intra_matches_epi(desc, n, ncams, dist_thresh, neigh_ratio, NULL, NULL, NULL, tracks, cap_tracks, mergeable);
}

/* MultiCameraFrame::computeRepresentativeDesc, :530-567 */
extern "C" int orc_representative_desc(const uint8_t *descs, int n)
{
    const size_t N = n;
    std::vector<float> Distances(N * N);
    for (size_t i = 0; i < N; i++) {
        Distances[i * N + i] = 0;
        for (size_t j = i + 1; j < N; j++) {
            int distij = orc_descriptor_distance(descs + i * 32, descs + j * 32);
            Distances[i * N + j] = distij;
            Distances[j * N + i] = distij;
        }
    }
    int BestMedian = INT_MAX;
    int BestIdx = 0;
    for (size_t i = 0; i < N; i++) {
        std::vector<int> vDists(Distances.begin() + i * N, Distances.begin() + i * N + N);
        std::sort(vDists.begin(), vDists.end());
        int median = vDists[0.5 * (N - 1)];
        if (median < BestMedian) { BestMedian = median; BestIdx = i; }
    }
    return BestIdx;
}

/* argsorte(data, ascen) (MCSlam/include/MCSlam/utils.h:21-30): std::sort of the index sequence 0..n-1 by data; equal
 * entries land where libstdc++'s introsort puts them -- that placement is part of the reference's result (the mono
 * fill of FrontEnd::obtainLfFeatures, FrontEnd.cpp:514, sorts integer-valued FAST responses: ties are the rule) */
#include <numeric>
extern "C" void orc_argsorte(const float *data_in, int n, int ascen, int *indices_out)
{
    std::vector<float> data(data_in, data_in + n);
    std::vector<int> indices(n);
    std::iota(indices.begin(), indices.end(), 0);
    if (ascen)
        std::sort(indices.begin(), indices.end(), [&data](int i, int j) -> bool { return data[i] < data[j]; });
    else
        std::sort(indices.begin(), indices.end(), [&data](int i, int j) -> bool { return data[i] > data[j]; });
    for (int i = 0; i < n; i++) indices_out[i] = indices[i];
}

/* ------------------------------------------------------------------------- */
/* DBoW2 TemplatedVocabulary<FORB>::transform (un-vendored dependency of the        */
/* reference, restated from the published algorithm -- call sites MultiCameraFrame.cpp:257,      */
/* FrontEnd.cpp:525,929; vocabulary layout as loadFromTextFile builds it, FrontEnd.h:137-138)    */
/* ------------------------------------------------------------------------- */
#include <map>
namespace {
struct VNode {
    int id = 0, parent = 0, word_id = 0;
    double weight = 0;
    std::vector<int> children;
    uint8_t descriptor[32];
    bool isLeaf() const { return children.empty(); }
};
}   // namespace

extern "C" int orc_bow_transform(int k, int L, int scoring, int weighting, const int32_t *parent, const uint8_t *is_leaf,
                                 const uint8_t *ndesc, const double *nweight, int nnodes, const uint8_t *feats, int nf,
                                 int levelsup, uint32_t *bow_ids, double *bow_vals, int *nbow, uint32_t *fv_nodes,
                                 int32_t *fv_offsets, int *nfv, int32_t *fv_feats)
{
    (void)k;
    /* loadFromTextFile: node ids in file order, children in file order, words numbered in file order */
    std::vector<VNode> m_nodes(1);
    m_nodes[0].id = 0;
    int nwords = 0;
    for (int i = 0; i < nnodes; i++) {
        int nid = (int)m_nodes.size();
        m_nodes.resize(m_nodes.size() + 1);
        m_nodes[nid].id = nid;
        int pid = parent[i];
        m_nodes[nid].parent = pid;
        m_nodes[pid].children.push_back(nid);
        memcpy(m_nodes[nid].descriptor, ndesc + (size_t)i * 32, 32);
        m_nodes[nid].weight = nweight[i];
        if (is_leaf[i] > 0) m_nodes[nid].word_id = nwords++;
    }
    std::map<unsigned int, double> v;                       /* BowVector */
    std::map<unsigned int, std::vector<unsigned int> > fv;  /* FeatureVector */
    const bool must = scoring != 5;                         /* DotProductScoring::mustNormalize is false */
    const int norm_l2 = scoring == 1;
    const int m_L = L;
    for (int i_feature = 0; i_feature < nf; i_feature++) {
        const uint8_t *feature = feats + (size_t)i_feature * 32;
        /* transform(feature, word_id, weight, &nid, levelsup) */
        unsigned int nid = 0;
        int final_id = 0;
        const int nid_level = m_L - levelsup;
        if (nid_level <= 0) nid = 0;
        int current_level = 0;
        do {
            ++current_level;
            const std::vector<int> &nodes = m_nodes[final_id].children;
            final_id = nodes[0];
            double best_d = orc_descriptor_distance(feature, m_nodes[final_id].descriptor);
            for (size_t c = 1; c < nodes.size(); ++c) {
                int id = nodes[c];
                double d = orc_descriptor_distance(feature, m_nodes[id].descriptor);
                if (d < best_d) { best_d = d; final_id = id; }
            }
            if (current_level == nid_level) nid = final_id;
        } while (!m_nodes[final_id].isLeaf());
        const unsigned int id = m_nodes[final_id].word_id;
        const double w = m_nodes[final_id].weight;
        if (w > 0) {
            std::map<unsigned int, double>::iterator vit = v.lower_bound(id);
            if (weighting == 0 || weighting == 1) {   /* TF_IDF, TF: addWeight */
                if (vit != v.end() && !(v.key_comp()(id, vit->first))) vit->second += w;
                else v.insert(vit, std::map<unsigned int, double>::value_type(id, w));
            } else {                                   /* IDF, BINARY: addIfNotExist */
                if (vit == v.end() || (v.key_comp()(id, vit->first))) v.insert(vit, std::map<unsigned int, double>::value_type(id, w));
            }
            fv[nid].push_back(i_feature);             /* addFeature */
        }
    }
    if ((weighting == 0 || weighting == 1) && !v.empty() && !must) {
        const double nd = v.size();
        for (std::map<unsigned int, double>::iterator vit = v.begin(); vit != v.end(); vit++) vit->second /= nd;
    }
    if (must) {   /* BowVector::normalize */
        double norm = 0.0;
        if (!norm_l2) { for (std::map<unsigned int, double>::iterator it = v.begin(); it != v.end(); ++it) norm += fabs(it->second); }
        else { for (std::map<unsigned int, double>::iterator it = v.begin(); it != v.end(); ++it) norm += it->second * it->second; norm = sqrt(norm); }
        if (norm > 0.0) for (std::map<unsigned int, double>::iterator it = v.begin(); it != v.end(); ++it) it->second /= norm;
    }
    int i = 0;
    for (std::map<unsigned int, double>::iterator it = v.begin(); it != v.end(); ++it, ++i) { bow_ids[i] = it->first; bow_vals[i] = it->second; }
    *nbow = i;
    i = 0;
    int off = 0;
    for (std::map<unsigned int, std::vector<unsigned int> >::iterator it = fv.begin(); it != fv.end(); ++it, ++i) {
        fv_nodes[i] = it->first;
        fv_offsets[i] = off;
        for (size_t f = 0; f < it->second.size(); f++) fv_feats[off++] = (int32_t)it->second[f];
    }
    fv_offsets[i] = off;
    *nfv = i;
    return 0;
}

/* ------------------------------------------------------------------------- */
/* MultiCameraFrame::computeIntraMatches(matches, words_) -- the BoW-guided (live) variant,       */
/* MCSlam/src/MultiCameraFrame.cpp:586-943, with checkItersEnd (:569-575).  FeatureVectors are     */
/* passed as arrays (node ids ascending, offsets, feature indices) and rebuilt as std::map.        */
/* matchIndex is widened from array<int,5> to ncams entries.                                       */
/* ------------------------------------------------------------------------- */
extern "C" int orc_intra_matches_bow(const uint8_t *const *desc, const float *const *kp_y, int ncams,
                                     const uint32_t *const *fv_nodes, const int32_t *const *fv_offsets,
                                     const int32_t *const *fv_feats, const int *nfv, double max_neighbor_ratio,
                                     int32_t *tracks, int32_t *n_rays_out, int cap_tracks, uint32_t *words_out, int cap_words,
                                     int *nwords_out)
{
    typedef std::map<unsigned int, std::vector<unsigned int> > FeatureVector;
    const int TH_LOW = 75;
    const int num_cams_ = ncams;
    struct IM { std::vector<int> matchIndex; int n_rays; };
    std::vector<IM> matches;
    std::vector<unsigned int> words_;
    if (nwords_out) *nwords_out = 0;

    std::vector<FeatureVector> featVecs(num_cams_);
    std::vector<FeatureVector::const_iterator> featVecIters(num_cams_), featVecEnds(num_cams_);
    for (int i = 0; i < num_cams_; i++) {
        for (int k = 0; k < nfv[i]; k++) {
            std::vector<unsigned int> &v = featVecs[i][fv_nodes[i][k]];
            for (int f = fv_offsets[i][k]; f < fv_offsets[i][k + 1]; f++) v.push_back((unsigned int)fv_feats[i][f]);
        }
        if (featVecs[i].size() == 0) return 0;                     /* :602-603 */
        featVecIters[i] = featVecs[i].begin();
        featVecEnds[i] = std::prev(featVecs[i].end());             /* :606 */
    }
    int intraMatchInd = 0;
    for (;;) {
        /* checkItersEnd, :569-575 */
        bool res = true;
        for (int i = 0; i < num_cams_; i++) res = res && featVecIters[i]->first >= featVecEnds[i]->first;
        if (res) break;

        std::vector<unsigned int> wordIds;
        for (int i = 0; i < num_cams_; i++) {
            if (featVecIters[i] == featVecEnds[i]) wordIds.push_back(INT_MAX);
            else wordIds.push_back(featVecIters[i]->first);
        }
        int min_val = INT_MAX - 1;
        int second_min_val = INT_MAX - 1;
        std::vector<int> selected_cams;
        std::vector<std::vector<int> > matchedFlags;
        for (int i = 0; i < num_cams_; i++) {
            if (wordIds[i] < (unsigned int)min_val) {
                second_min_val = min_val;
                min_val = wordIds[i];
                selected_cams.clear();
                matchedFlags.clear();
                selected_cams.push_back(i);
                matchedFlags.push_back(std::vector<int>(featVecIters[i]->second.size(), -1));
            } else if (wordIds[i] == (unsigned int)min_val) {
                selected_cams.push_back(i);
                matchedFlags.push_back(std::vector<int>(featVecIters[i]->second.size(), -1));
            } else if (wordIds[i] < (unsigned int)second_min_val)
                second_min_val = wordIds[i];
        }
        if (selected_cams.size() >= 2) {
            for (int i = 0; i < (int)selected_cams.size() - 1; i++) {
                int cam1 = selected_cams[i];
                std::vector<unsigned int> feat_cam1 = featVecIters[cam1]->second;
                for (int cam1_feat_ind = 0; cam1_feat_ind < (int)feat_cam1.size(); cam1_feat_ind++) {
                    bool foundMatch = false;
                    int cam1_existing_intramatch = matchedFlags[i][cam1_feat_ind];
                    if (cam1_existing_intramatch != -1) continue;
                    IM temp;
                    temp.matchIndex.assign(num_cams_, -1);
                    temp.matchIndex[cam1] = feat_cam1[cam1_feat_ind];
                    temp.n_rays = 1;
                    matches.push_back(temp);
                    matchedFlags[i][cam1_feat_ind] = intraMatchInd;
                    bool updateOnce = true;
                    for (int j = i + 1; j < (int)selected_cams.size(); j++) {
                        int cam2 = selected_cams[j];
                        std::vector<unsigned int> feat_cam2 = featVecIters[cam2]->second;
                        int best_j_now = -1;
                        double best_dist_1 = 1e9;
                        double best_dist_2 = 1e9;
                        for (int cam2_feat_ind = 0; cam2_feat_ind < (int)feat_cam2.size(); cam2_feat_ind++) {
                            if (fabsf(kp_y[cam1][feat_cam1[cam1_feat_ind]] - kp_y[cam2][feat_cam2[cam2_feat_ind]]) >= 50) continue;
                            double d = orc_descriptor_distance(desc[cam1] + (size_t)feat_cam1[cam1_feat_ind] * 32,
                                                               desc[cam2] + (size_t)feat_cam2[cam2_feat_ind] * 32);
                            if (d < best_dist_1) {
                                best_j_now = cam2_feat_ind;
                                best_dist_2 = best_dist_1;
                                best_dist_1 = d;
                            } else if (d < best_dist_2) {
                                best_dist_2 = d;
                            }
                        }
                        if (best_dist_1 <= TH_LOW && best_dist_1 / best_dist_2 <= max_neighbor_ratio) {
                            int existing_intramatch = matchedFlags[j][best_j_now];
                            if (existing_intramatch == intraMatchInd) continue;
                            if (existing_intramatch == -1) {
                                matches[intraMatchInd].matchIndex[cam2] = feat_cam2[best_j_now];
                                matches[intraMatchInd].n_rays++;
                                matchedFlags[j][best_j_now] = intraMatchInd;
                                foundMatch = true;
                            } else {
                                int old_cam1_match_ind = matches[existing_intramatch].matchIndex[cam1];
                                if (old_cam1_match_ind == -1) {
                                    if (updateOnce) updateOnce = false;
                                    else continue;
                                    bool update_match = true;
                                    std::vector<int> tempmatchIndex = matches[existing_intramatch].matchIndex;
                                    int tmp_nrays_inc = 0;
                                    for (int tt1 = 0; tt1 < num_cams_; tt1++) {
                                        if (matches[intraMatchInd].matchIndex[tt1] != -1) {
                                            if (matches[existing_intramatch].matchIndex[tt1] != -1) { update_match = false; break; }
                                            tempmatchIndex[tt1] = matches[intraMatchInd].matchIndex[tt1];
                                            tmp_nrays_inc++;
                                        }
                                    }
                                    if (update_match) {
                                        matches[existing_intramatch].matchIndex = tempmatchIndex;
                                        matches[existing_intramatch].n_rays += tmp_nrays_inc;
                                        matchedFlags[i][cam1_feat_ind] = existing_intramatch;
                                        intraMatchInd = existing_intramatch;
                                        foundMatch = true;
                                        matches.pop_back();
                                    }
                                    continue;
                                }
                                double d = orc_descriptor_distance(desc[cam1] + (size_t)old_cam1_match_ind * 32,
                                                                   desc[cam2] + (size_t)feat_cam2[best_j_now] * 32);
                                if (best_dist_1 < d) {
                                    matches[existing_intramatch].matchIndex[cam2] = -1;
                                    matches[existing_intramatch].n_rays--;
                                    matches[intraMatchInd].matchIndex[cam2] = feat_cam2[best_j_now];
                                    matches[intraMatchInd].n_rays++;
                                    matchedFlags[j][best_j_now] = intraMatchInd;
                                    foundMatch = true;
                                }
                            }
                        }
                    }
                    if (foundMatch) {
                        words_.push_back(featVecIters[selected_cams[0]]->first);
                        intraMatchInd = (int)matches.size();
                    } else {
                        matches.pop_back();
                        matchedFlags[i][cam1_feat_ind] = -1;
                    }
                }
            }
        }
        for (int s_word = 0; s_word < (int)selected_cams.size(); s_word++) ++featVecIters[selected_cams[s_word]];
    }
    for (size_t m = 0; m < matches.size() && (int)m < cap_tracks; m++) {
        for (int c = 0; c < ncams; c++) tracks[m * ncams + c] = matches[m].matchIndex[c];
        if (n_rays_out) n_rays_out[m] = matches[m].n_rays;
    }
    for (size_t w = 0; w < words_.size() && (int)w < cap_words; w++) words_out[w] = words_[w];
    if (nwords_out) *nwords_out = (int)words_.size();
    return (int)matches.size();
}
