// cvmock -- a stand-in for the handful of OpenCV declarations include/mcorb_adapter.hpp touches under -DMCORB_WITH_OPENCV.
//
// PURPOSE: SYNTAX ONLY.  OpenCV is not in this image, so the flavour of the adapter a maintainer would actually compile
// had never met a compiler.  This header lets tests/test_host_logic.py run `g++ -fsyntax-only` (and a tiny link + run of
// the matrix algebra) over it.  It pins NOTHING about OpenCV's behaviour: every class below is this repository's own
// minimal code (a dense row-major double / byte matrix), not OpenCV's, and no parity claim may cite it.
#pragma once
#include <stdint.h>
#include <string.h>

#include <cassert>
#include <cmath>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_64F 6
#define CV_8UC1 CV_8U
#define CV_Assert(x) assert(x)
#define CV_VERSION "cvmock (declarations only)"

inline int cvRound(double v) { return (int)std::lrint(v); }   // (global, as in OpenCV; tools/crosscheck only: syntax)

namespace cv {

struct Range {
    int start, end;
    Range(int s, int e) : start(s), end(e) {}
};

struct Point2f {
    float x = 0, y = 0;
};
struct Size {
    int width, height;
    Size(int w, int h) : width(w), height(h) {}
};
enum { INTER_LINEAR = 1, BORDER_REFLECT_101 = 4, NORM_HAMMING = 6 };

struct KeyPoint {   // field order of cv::KeyPoint: pt, size, angle, response, octave, class_id
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};

class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;          // bytes per row
    uint8_t *data = nullptr;
    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void *ext) : rows(r), cols(c), step((size_t)c), data((uint8_t *)ext), type_(type) {}   // header over caller memory
    Mat clone() const { Mat m(rows, cols, type_); copyTo(m); return m; }
    bool isContinuous() const { return step == (size_t)cols * esz(); }
    uint8_t *ptr(int r) { return data + (size_t)r * step; }
    const uint8_t *ptr(int r) const { return data + (size_t)r * step; }
    Mat rowRange(int a, int b) const { return (*this)(Range(a, b), Range(0, cols)); }
    Mat colRange(int a, int b) const { return (*this)(Range(0, rows), Range(a, b)); }
    void create(int r, int c, int type)
    {
        rows = r; cols = c; type_ = type;
        step = (size_t)c * esz();
        buf_ = std::shared_ptr<std::vector<uint8_t>>(new std::vector<uint8_t>((size_t)r * step, 0));
        data = buf_->data();
    }
    void release() { *this = Mat(); }
    bool empty() const { return rows == 0 || cols == 0; }
    int type() const { return type_; }
    static Mat eye(int r, int c, int type)
    {
        Mat m(r, c, type);
        for (int i = 0; i < (r < c ? r : c); i++) m.at<double>(i, i) = 1.0;
        return m;
    }
    template <typename T> T &at(int r, int c) { return *reinterpret_cast<T *>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T> const T &at(int r, int c) const { return *reinterpret_cast<const T *>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    Mat operator()(const Range &rr, const Range &cr) const   // view sharing the storage
    {
        Mat v;
        v.buf_ = buf_; v.type_ = type_; v.step = step;
        v.rows = rr.end - rr.start; v.cols = cr.end - cr.start;
        v.data = data + (size_t)rr.start * step + (size_t)cr.start * esz();
        return v;
    }
    void copyTo(Mat dst) const   // dst is a view: same size
    {
        assert(dst.rows == rows && dst.cols == cols);
        for (int r = 0; r < rows; r++) memcpy(dst.data + (size_t)r * dst.step, data + (size_t)r * step, (size_t)cols * esz());
    }
    Mat t() const
    {
        Mat m(cols, rows, type_);
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++) m.at<double>(c, r) = at<double>(r, c);
        return m;
    }
    Mat inv() const   // Gauss-Jordan with partial pivoting, square CV_64F
    {
        assert(rows == cols && type_ == CV_64F);
        const int n = rows;
        std::vector<double> a((size_t)n * 2 * n, 0.0);
        for (int r = 0; r < n; r++) {
            for (int c = 0; c < n; c++) a[(size_t)r * 2 * n + c] = at<double>(r, c);
            a[(size_t)r * 2 * n + n + r] = 1.0;
        }
        for (int i = 0; i < n; i++) {
            int p = i;
            for (int r = i + 1; r < n; r++) if (std::fabs(a[(size_t)r * 2 * n + i]) > std::fabs(a[(size_t)p * 2 * n + i])) p = r;
            for (int c = 0; c < 2 * n; c++) std::swap(a[(size_t)i * 2 * n + c], a[(size_t)p * 2 * n + c]);
            const double d = a[(size_t)i * 2 * n + i];
            for (int c = 0; c < 2 * n; c++) a[(size_t)i * 2 * n + c] /= d;
            for (int r = 0; r < n; r++) {
                if (r == i) continue;
                const double f = a[(size_t)r * 2 * n + i];
                for (int c = 0; c < 2 * n; c++) a[(size_t)r * 2 * n + c] -= f * a[(size_t)i * 2 * n + c];
            }
        }
        Mat m(n, n, CV_64F);
        for (int r = 0; r < n; r++)
            for (int c = 0; c < n; c++) m.at<double>(r, c) = a[(size_t)r * 2 * n + n + c];
        return m;
    }
    friend Mat operator*(const Mat &a, const Mat &b)
    {
        assert(a.cols == b.rows);
        Mat m(a.rows, b.cols, CV_64F);
        for (int r = 0; r < a.rows; r++)
            for (int c = 0; c < b.cols; c++) {
                double s = 0;
                for (int k = 0; k < a.cols; k++) s += a.at<double>(r, k) * b.at<double>(k, c);
                m.at<double>(r, c) = s;
            }
        return m;
    }

protected:
    size_t esz() const { return type_ == CV_64F ? 8 : 1; }
    int type_ = CV_8U;
    std::shared_ptr<std::vector<uint8_t>> buf_;
};

template <typename T> class Mat_ : public Mat {   // only the comma initialiser the adapter uses: (Mat_<double>(3, 3) << a, b, ...)
public:
    Mat_(int r, int c) : Mat(r, c, CV_64F) {}
    struct Init {
        Mat_ *m;
        int i;
        Init operator,(T v) { m->template at<T>(i / m->cols, i % m->cols) = v; return Init{m, i + 1}; }
        operator Mat() const { return *m; }
    };
    Init operator<<(T v) { this->template at<T>(0, 0) = v; return Init{this, 1}; }
};

class _InputArray {
public:
    _InputArray(const Mat &m) : m_(m) {}
    bool empty() const { return m_.empty(); }
    Mat getMat() const { return m_; }

protected:
    Mat m_;
};
class _OutputArray {
public:
    _OutputArray(Mat &m) : p_(&m) {}
    void release() const { p_->release(); }
    void create(int r, int c, int type) const { p_->create(r, c, type); }
    Mat getMat() const { return *p_; }

protected:
    Mat *p_;
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;

}  // namespace cv
