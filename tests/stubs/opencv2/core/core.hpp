// TEST-ONLY stand-in for <opencv2/core/core.hpp>: just enough declarations of the cv:: types that
// pli_slam_amd/adapters/orbslam_adapters.hpp touches, so that the adapter can be SYNTAX-CHECKED on a machine without
// OpenCV (tests/test_cpp_host.py, g++ -fsyntax-only).  Nothing here is implemented and nothing is pinned by it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
namespace cv {
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float a, float b) : x(a), y(b) {} };
struct KeyPoint {
  Point2f pt; float size, angle, response; int octave, class_id;
  KeyPoint() {}
  KeyPoint(float x, float y, float s, float a, float r, int o, int c) : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};
class MatExpr;
class Mat {
 public:
  int rows = 0, cols = 0;
  unsigned char* data = nullptr;
  size_t step = 0;
  Mat();
  Mat(const MatExpr&);
  int type() const;
  bool empty() const;
  void create(int r, int c, int t);
  void release();
  Mat rowRange(int a, int b) const;
  Mat colRange(int a, int b) const;
  Mat col(int c) const;
  Mat row(int r) const;
  MatExpr t() const;
  unsigned char* ptr(int r = 0);
  const unsigned char* ptr(int r = 0) const;
  template <class T> T* ptr(int r = 0);
  template <class T> const T* ptr(int r = 0) const;
  template <class T> T& at(int i);
  template <class T> const T& at(int i) const;
};
class MatExpr {
 public:
  MatExpr();
  MatExpr(const Mat&);
};
MatExpr operator*(const MatExpr&, const MatExpr&);
MatExpr operator+(const MatExpr&, const MatExpr&);
MatExpr operator-(const MatExpr&);
MatExpr operator*(const Mat&, const Mat&);
MatExpr operator+(const Mat&, const Mat&);
MatExpr operator-(const Mat&);
MatExpr operator*(const MatExpr&, const Mat&);
MatExpr operator+(const MatExpr&, const Mat&);
class _InputArray { public: _InputArray(const Mat&); bool empty() const; Mat getMat() const; };
class _OutputArray : public _InputArray { public: _OutputArray(Mat&); void create(int r, int c, int t) const; void release() const; };
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}  // namespace cv
