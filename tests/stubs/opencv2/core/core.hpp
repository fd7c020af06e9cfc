// TEST-ONLY stand-in for <opencv2/core/core.hpp>, written for this repository (no OpenCV text): the cv:: types that
// pli_slam_amd/adapters/*.hpp touch, WITH storage and behaviour, so that the adapters can be compiled AND RUN on a machine
// without OpenCV (tests/cpp/dropin_harness.cpp, tests/test_cpp_dropin.py).  It pins nothing about OpenCV: the float matrix
// arithmetic below is this file's own definition (products exact in double, one sum in double, one rounding to float per
// operator; real OpenCV fuses `A*x + t` into one gemm), and the Python side of the test mirrors exactly this definition.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_32FC1 5
namespace cv {
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float a, float b) : x(a), y(b) {} };
struct KeyPoint {
  Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
  KeyPoint() {}
  KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};

class MatExpr;
class Mat {
 public:
  int rows = 0, cols = 0;
  unsigned char* data = nullptr;
  size_t step = 0;                       // bytes per row

  Mat() {}
  Mat(int r, int c, int t) { create(r, c, t); }
  int type() const { return type_; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  size_t elemSize() const { return type_ == CV_32F ? 4 : 1; }
  void create(int r, int c, int t) {
    if (data && r == rows && c == cols && t == type_ && step == (size_t)c * elemSizeOf(t)) return;   // cv::Mat::create keeps a fitting buffer
    type_ = t; rows = r; cols = c; step = (size_t)c * elemSizeOf(t);
    buf_ = std::make_shared<std::vector<unsigned char>>((size_t)r * step + 16, 0);
    data = buf_->data();
  }
  void release() { buf_.reset(); data = nullptr; rows = cols = 0; step = 0; }
  Mat clone() const {
    Mat m;
    if (empty()) return m;
    m.create(rows, cols, type_);
    for (int r = 0; r < rows; ++r) std::memcpy(m.ptr(r), ptr(r), (size_t)cols * elemSize());
    return m;
  }
  static Mat zeros(int r, int c, int t) { return Mat(r, c, t); }
  static Mat eye(int r, int c, int t) {
    Mat m(r, c, t);
    if (t != CV_32F) throw std::logic_error("stub cv::Mat::eye: CV_32F only");
    for (int i = 0; i < r && i < c; ++i) m.at<float>(i, i) = 1.0f;
    return m;
  }
  Mat rowRange(int a, int b) const { Mat m = *this; m.data = data + (size_t)a * step; m.rows = b - a; return m; }
  Mat colRange(int a, int b) const { Mat m = *this; m.data = data + (size_t)a * elemSize(); m.cols = b - a; return m; }
  Mat col(int c) const { return colRange(c, c + 1); }
  Mat row(int r) const { return rowRange(r, r + 1); }
  MatExpr t() const;                     // defined below MatExpr
  unsigned char* ptr(int r = 0) { return data + (size_t)r * step; }
  const unsigned char* ptr(int r = 0) const { return data + (size_t)r * step; }
  template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
  template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
  template <class T> T& at(int i, int j) { return ptr<T>(i)[j]; }
  template <class T> const T& at(int i, int j) const { return ptr<T>(i)[j]; }
  // single index: element i of a row or column vector (cv::Mat::at(int) does the same)
  template <class T> T& at(int i) { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }
  template <class T> const T& at(int i) const { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }
  void needFloat(const char* who) const {
    if (type_ != CV_32F) throw std::logic_error(std::string("stub cv::Mat::") + who + ": CV_32F only");
  }

 private:
  static size_t elemSizeOf(int t) { return t == CV_32F ? 4 : 1; }
  int type_ = CV_8U;
  std::shared_ptr<std::vector<unsigned char>> buf_;
};
// A*B, -A.t()*B, A*B + C: evaluated as ONE gemm when converted to a Mat — products and sum in double, then
// (float)(alpha * sum + c): the convention oracle/match_oracle.hpp (cvmatDot3) states for OpenCV's CV_32F gemm.
class MatExpr {
 public:
  Mat a, b, c;
  double alpha = 1.0;
  bool transA = false, hasB = false, hasC = false;
  MatExpr(const Mat& m) : a(m) { a.needFloat("MatExpr"); }
  float A(int i, int k) const { return transA ? a.at<float>(k, i) : a.at<float>(i, k); }
  int arows() const { return transA ? a.cols : a.rows; }
  int acols() const { return transA ? a.rows : a.cols; }
  operator Mat() const {
    const int R = arows(), K = acols(), Cn = hasB ? b.cols : K;
    if (hasB && b.rows != K) throw std::logic_error("stub cv::MatExpr: size mismatch in A*B");
    if (hasC && (c.rows != R || c.cols != Cn)) throw std::logic_error("stub cv::MatExpr: size mismatch in + C");
    Mat m(R, Cn, CV_32F);
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < Cn; ++j) {
        double d;
        if (hasB) {
          d = 0.0;
          for (int k = 0; k < K; ++k) d += (double)A(i, k) * (double)b.at<float>(k, j);
        } else {
          d = (double)A(i, j);
        }
        m.at<float>(i, j) = (float)(alpha * d + (hasC ? (double)c.at<float>(i, j) : 0.0));
      }
    return m;
  }
};
inline MatExpr Mat::t() const { MatExpr e(*this); e.transA = true; return e; }
inline MatExpr operator-(const MatExpr& x) {
  if (x.hasC) throw std::logic_error("stub cv::MatExpr: -(A*B + C) is not provided");
  MatExpr e = x; e.alpha = -e.alpha; return e;
}
inline MatExpr operator-(const Mat& x) { MatExpr e(x); e.alpha = -1.0; return e; }
inline MatExpr operator*(const MatExpr& x, const Mat& y) {
  if (x.hasB || x.hasC) throw std::logic_error("stub cv::MatExpr: only (alpha * op(A)) * B is provided");
  y.needFloat("operator*");
  MatExpr e = x; e.b = y; e.hasB = true; return e;
}
inline MatExpr operator*(const Mat& x, const Mat& y) { return MatExpr(x) * y; }
inline MatExpr operator+(const MatExpr& x, const Mat& y) {
  if (x.hasC) throw std::logic_error("stub cv::MatExpr: only one + C is provided");
  y.needFloat("operator+");
  MatExpr e = x; e.c = y; e.hasC = true; return e;
}
inline MatExpr operator+(const Mat& x, const Mat& y) { return MatExpr(x) + y; }

class _InputArray {
 public:
  _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)) {}
  bool empty() const { return m_->empty(); }
  Mat getMat() const { return *m_; }
 protected:
  Mat* m_;
};
class _OutputArray : public _InputArray {
 public:
  _OutputArray(Mat& m) : _InputArray(m) {}
  void create(int r, int c, int t) const { m_->create(r, c, t); }
  void release() const { m_->release(); }
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}  // namespace cv
