// NOT OpenCV.  Declaration-only test double for ONE purpose: `g++ -fsyntax-only` of THIS repository's adapters
// (3_orb_slam3_selfnote_amd/csrc/adapter/*.cc) against the reference's unmodified headers in an image that has no OpenCV.
// It declares the cv:: names those headers (and the in-tree DBoW2 headers they include) and the adapters mention, with the
// signatures OpenCV 3.x gives them.  Nothing here has a body and no reference SOURCE file is ever compiled against it
// (tests/test_adapter_typecheck.py passes -fsyntax-only and only the adapter .cc files).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#include <math.h>
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_64F 6
namespace cv {
typedef unsigned char uchar;
typedef std::string String;
template <typename T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T a, T b) : x(a), y(b) {} };
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;
template <typename T> struct Point3_ { T x, y, z; Point3_() : x(0), y(0), z(0) {} Point3_(T a, T b, T c) : x(a), y(b), z(c) {} };
typedef Point3_<float> Point3f;
typedef Point3_<double> Point3d;
template <typename T> struct Size_ { T width, height; Size_() : width(0), height(0) {} Size_(T a, T b) : width(a), height(b) {} };
typedef Size_<int> Size;
struct Rect { int x, y, width, height; Rect(int a, int b, int c, int d) : x(a), y(b), width(c), height(d) {} };
struct Scalar { double val[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0); };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
template <typename T, int M, int N> struct Matx { T val[M * N]; };
struct MatExpr;
struct Mat {
  int flags, dims, rows, cols;
  uchar *data;
  size_t step;
  Mat();
  Mat(int r, int c, int type);
  Mat(const Mat &);
  Mat(const MatExpr &);
  Mat &operator=(const Mat &);
  Mat &operator=(const MatExpr &);
  int type() const;
  bool empty() const;
  Mat clone() const;
  Mat row(int i) const;
  Mat col(int i) const;
  Mat rowRange(int a, int b) const;
  Mat colRange(int a, int b) const;
  Mat operator()(const Rect &r) const;
  Mat t() const;
  Mat inv() const;
  double dot(const Mat &) const;
  void copyTo(Mat m) const;
  uchar *ptr(int r = 0);
  const uchar *ptr(int r = 0) const;
  template <typename T> T *ptr(int r = 0);
  template <typename T> const T *ptr(int r = 0) const;
  bool isContinuous() const;
  void create(int r, int c, int type);
  size_t elemSize() const;
  size_t total() const;
  Size size() const;
  int channels() const;
  void release();
  template <typename T> T &at(int r);
  template <typename T> const T &at(int r) const;
  template <typename T> T &at(int r, int c);
  template <typename T> const T &at(int r, int c) const;
  static Mat zeros(int r, int c, int type);
  static Mat zeros(Size s, int type);
  static Mat eye(int r, int c, int type);
};
template <typename T> struct Mat_ : Mat { Mat_(); Mat_(int r, int c); T &operator()(int r, int c); };
struct MatExpr { operator Mat() const; };
MatExpr operator*(const Mat &, const Mat &);
MatExpr operator*(double, const Mat &);
MatExpr operator*(const Mat &, double);
MatExpr operator+(const Mat &, const Mat &);
MatExpr operator-(const Mat &, const Mat &);
MatExpr operator-(const Mat &);
MatExpr operator/(const Mat &, double);
double norm(const Mat &);
struct _InputArray { _InputArray(const Mat &m); bool empty() const; Mat getMat() const; };
struct _OutputArray { _OutputArray(Mat &m); void create(int r, int c, int type) const; void release() const; Mat getMat() const; };
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
struct DMatch { int queryIdx, trainIdx, imgIdx; float distance; };
enum { NORM_HAMMING = 6 };
struct BFMatcher { BFMatcher(int normType = 4, bool crossCheck = false); void knnMatch(const Mat &, const Mat &, std::vector<std::vector<DMatch> > &, int) const; };
struct FileNode;
struct FileNodeIterator { FileNodeIterator &operator++(); FileNode operator*() const; bool operator!=(const FileNodeIterator &) const; };
struct FileNode {
  FileNode operator[](const char *) const; FileNode operator[](const std::string &) const; FileNode operator[](int) const;
  operator int() const; operator float() const; operator double() const; operator std::string() const;
  bool empty() const; size_t size() const; int type() const; bool isSeq() const;
  FileNodeIterator begin() const; FileNodeIterator end() const;
  enum { SEQ = 5, MAP = 6 };
};
struct FileStorage {
  enum { READ = 0, WRITE = 1 };
  FileStorage(); FileStorage(const std::string &, int);
  bool isOpened() const; void release();
  FileNode operator[](const char *) const; FileNode operator[](const std::string &) const;
};
template <typename T> FileStorage &operator<<(FileStorage &, const T &);
template <typename T> void operator>>(const FileNode &, T &);
}  // namespace cv
