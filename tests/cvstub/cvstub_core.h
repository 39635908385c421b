// tests/cvstub/cvstub_core.h -- the few OpenCV types the adaptor (adapter/*.cc) touches, just complete enough to
// COMPILE and to run tests/adapter_driver.cc.  Test infrastructure only: it exists because OpenCV is not installed in
// the build container; it is not an OpenCV replacement (no reference build uses it) and never ships in liborbx.
#ifndef CVSTUB_CORE_H
#define CVSTUB_CORE_H

#include <assert.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_32FC1 5

typedef unsigned char uchar;

namespace cv
{

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T _x, T _y) : x(_x), y(_y) {}
};
typedef Point_<float> Point2f;
typedef Point_<int> Point2i;
typedef Point2i Point;

struct KeyPoint {           // same layout as the real cv::KeyPoint: 28 bytes
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
};
typedef char cvstub_keypoint_28_bytes[sizeof(KeyPoint) == 28 ? 1 : -1];

class Mat;
class _InputArray;
class _OutputArray;

class Mat
{
public:
    int rows, cols;
    uchar *data;
    size_t step;

    Mat() : rows(0), cols(0), data(NULL), step(0), type_(CV_8U) {}
    Mat(int r, int c, int t) : rows(0), cols(0), data(NULL), step(0), type_(t) { create(r, c, t); }
    void create(int r, int c, int t)
    {
        if (r == rows && c == cols && t == type_ && data) return;
        type_ = t; rows = r; cols = c; step = (size_t)c * elemSize();
        buf_.reset(new std::vector<uchar>((size_t)r * step + 1));
        data = &(*buf_)[0];
    }
    void release() { buf_.reset(); data = NULL; rows = cols = 0; step = 0; }
    bool empty() const { return data == NULL || rows == 0 || cols == 0; }
    int type() const { return type_; }
    size_t elemSize() const { return type_ == CV_32F ? 4 : 1; }
    size_t total() const { return (size_t)rows * cols; }
    bool isContinuous() const { return rows <= 1 || step == (size_t)cols * elemSize(); }
    Mat rowRange(int a, int b) const { Mat m(*this); m.data = data + (size_t)a * step; m.rows = b - a; return m; }
    Mat colRange(int a, int b) const { Mat m(*this); m.data = data + (size_t)a * elemSize(); m.cols = b - a; return m; }   // a view: step unchanged
    Mat row(int r) const { return rowRange(r, r + 1); }
    Mat col(int c) const { return colRange(c, c + 1); }
    Mat t() const
    {
        assert(type_ == CV_32F);
        Mat m(cols, rows, CV_32F);
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++) *reinterpret_cast<float *>(m.data + (size_t)c * m.step + (size_t)r * 4) = *reinterpret_cast<const float *>(data + (size_t)r * step + (size_t)c * 4);
        return m;
    }
    Mat clone() const { Mat m(rows, cols, type_); for (int r = 0; r < rows; r++) memcpy(m.data + r * m.step, data + r * step, (size_t)cols * elemSize()); return m; }
    void copyTo(const _OutputArray &dst) const;
    double dot(const Mat &o) const
    {
        assert(type_ == CV_32F && o.type_ == CV_32F && total() == o.total());
        double acc = 0;
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++) {
                const size_t k = (size_t)r * cols + c;     // the other operand may be a row against a column of the same length
                acc += (double)*reinterpret_cast<const float *>(data + (size_t)r * step + (size_t)c * 4) *
                       *reinterpret_cast<const float *>(o.data + (k / o.cols) * o.step + (k % o.cols) * 4);
            }
        return acc;
    }
    template <typename T> T &at(int r, int c) { return *reinterpret_cast<T *>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T> const T &at(int r, int c) const { return *reinterpret_cast<const T *>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T> T &at(int i) { return cols == 1 ? at<T>(i, 0) : at<T>(0, i); }
    template <typename T> const T &at(int i) const { return cols == 1 ? at<T>(i, 0) : at<T>(0, i); }
    template <typename T> T *ptr(int r = 0) { return reinterpret_cast<T *>(data + (size_t)r * step); }
    template <typename T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + (size_t)r * step); }

private:
    int type_;
    std::shared_ptr<std::vector<uchar> > buf_;
};

// CV_32F matrix product / sum (cv::gemm accumulates float products in double)
inline Mat operator*(const Mat &a, const Mat &b)
{
    assert(a.type() == CV_32F && b.type() == CV_32F && a.cols == b.rows);
    Mat c(a.rows, b.cols, CV_32F);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < b.cols; j++) {
            double acc = 0;
            for (int k = 0; k < a.cols; k++) acc += (double)a.at<float>(i, k) * b.at<float>(k, j);
            c.at<float>(i, j) = (float)acc;
        }
    return c;
}
inline Mat operator+(const Mat &a, const Mat &b)
{
    assert(a.type() == CV_32F && a.rows == b.rows && a.cols == b.cols);
    Mat c(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) c.at<float>(i, j) = a.at<float>(i, j) + b.at<float>(i, j);
    return c;
}

inline Mat operator-(const Mat &a, const Mat &b)
{
    assert(a.type() == CV_32F && a.rows == b.rows && a.cols == b.cols);
    Mat c(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) c.at<float>(i, j) = a.at<float>(i, j) - b.at<float>(i, j);
    return c;
}
inline Mat operator*(double k, const Mat &a)
{
    assert(a.type() == CV_32F);
    Mat c(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) c.at<float>(i, j) = (float)(a.at<float>(i, j) * k);
    return c;
}
inline Mat operator/(const Mat &a, double k) { return (1.0 / k) * a; }
inline double norm(const Mat &a) { return sqrt(a.dot(a)); }

inline Mat operator-(const Mat &a)
{
    assert(a.type() == CV_32F);
    Mat c(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) c.at<float>(i, j) = -a.at<float>(i, j);
    return c;
}

class _InputArray
{
public:
    _InputArray() : m_(NULL) {}
    _InputArray(const Mat &m) : m_(&m) {}
    bool empty() const { return !m_ || m_->empty(); }
    Mat getMat() const { return m_ ? *m_ : Mat(); }
private:
    const Mat *m_;
};
class _OutputArray
{
public:
    _OutputArray(Mat &m) : m_(&m) {}
    void release() const { m_->release(); }
    Mat &getMatRef() const { return *m_; }
private:
    Mat *m_;
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;

inline void Mat::copyTo(const _OutputArray &dst) const
{
    Mat &d = dst.getMatRef();
    d.create(rows, cols, type_);
    for (int r = 0; r < rows; r++) memcpy(d.data + (size_t)r * d.step, data + (size_t)r * step, (size_t)cols * elemSize());
}

} // namespace cv

#endif
