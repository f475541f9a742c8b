// NOT OpenCV.  Test double for ONE purpose: `g++ -fsyntax-only` of THIS repository's adapter
// (3_orb_slam3_selfnote_amd/csrc/adapter/ORBextractor_hip.cc) in an image that has no OpenCV.  It declares just the
// cv:: names that the adapter and the class declaration it implements mention, with the signatures OpenCV 3.x gives
// them; nothing here has a body worth running and no reference SOURCE file is ever compiled against it
// (tests/test_adapter_typecheck.py only passes -fsyntax-only and only the adapter .cc).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
namespace cv {
typedef unsigned char uchar;
template <typename T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T a, T b) : x(a), y(b) {} };
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;
struct Rect { int x, y, width, height; Rect(int a, int b, int c, int d) : x(a), y(b), width(c), height(d) {} };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
struct Mat {
  int rows, cols;
  uchar *data;
  size_t step;
  Mat();
  Mat(int r, int c, int type);
  Mat(int r, int c, int type, void *data, size_t step);
  int type() const;
  bool empty() const;
  Mat rowRange(int a, int b) const;
  Mat operator()(const Rect &r) const;
  void copyTo(Mat m) const;
};
struct _InputArray { _InputArray(const Mat &m); bool empty() const; Mat getMat() const; };
struct _OutputArray { _OutputArray(Mat &m); void create(int r, int c, int type) const; void release() const; Mat getMat() const; };
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
}  // namespace cv
