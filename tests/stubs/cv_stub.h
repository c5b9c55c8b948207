// cv_stub.h -- DECLARATIONS-ONLY stand-in for the handful of OpenCV types the reference's headers and integration/*.cpp name.
// Purpose: `g++ -fsyntax-only` over integration/CubemapHipBridge.cpp + OrbExtractorHip.h against the reference's real headers
// (tests/test_integration_syntax.py), so that a typo in the bridge does not ship.  Nothing here has a body that computes anything; nothing is
// ever linked or run; it pins NOTHING about parity.
#ifndef CMS_TEST_CV_STUB_H
#define CMS_TEST_CV_STUB_H
#include <cassert>
#include <cmath>
#include <cstddef>
#include <iostream>
#include <sstream>
#include <fstream>
#include <string>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32F 5
#define CV_32FC1 5
#define CV_64F 6
#define CV_64FC1 6
#define CV_PI 3.1415926535897932384626433832795
typedef unsigned char uchar;
namespace cv {
template <class T> struct Point_ { T x, y; Point_(); Point_(T, T); template <class U> Point_(const Point_<U>&); Point_ operator*(double) const; Point_ operator+(const Point_&) const; Point_ operator-(const Point_&) const; };
typedef Point_<float> Point2f; typedef Point_<double> Point2d; typedef Point_<int> Point2i; typedef Point2i Point;
template <class T> struct Point3_ { T x, y, z; Point3_(); Point3_(T, T, T); template <class U> Point3_(const Point3_<U>&); };
typedef Point3_<float> Point3f; typedef Point3_<double> Point3d;
template <class T> struct Size_ { T width, height; Size_(); Size_(T, T); };
typedef Size_<int> Size;
template <class T> struct Rect_ { T x, y, width, height; Rect_(); Rect_(T, T, T, T); };
typedef Rect_<int> Rect;
template <class T, int N> struct Vec { T val[N]; Vec(); Vec(T); Vec(T, T); Vec(T, T, T); Vec(T, T, T, T); T& operator()(int); const T& operator()(int) const; T& operator[](int); const T& operator[](int) const;
  Vec operator+(const Vec&) const; Vec operator-(const Vec&) const; Vec operator*(double) const; Vec operator/(double) const; T dot(const Vec&) const; Vec cross(const Vec&) const; template <class U> operator Vec<U, N>() const; };
template <class T, int N> Vec<T, N> operator*(double, const Vec<T, N>&);
template <class T, int N> Vec<T, N> operator-(const Vec<T, N>&);
typedef Vec<float, 2> Vec2f; typedef Vec<float, 3> Vec3f; typedef Vec<float, 4> Vec4f; typedef Vec<double, 2> Vec2d; typedef Vec<double, 3> Vec3d; typedef Vec<double, 4> Vec4d; typedef Vec<uchar, 3> Vec3b;
template <class T, int N> double norm(const Vec<T, N>&);
template <class T> double norm(const Point_<T>&);
template <class T> double norm(const Point3_<T>&);
template <class T> T sqrt(T);
struct Scalar { double val[4]; Scalar(); Scalar(double); Scalar(double, double, double, double = 0); };
struct Range { int start, end; Range(); Range(int, int); static Range all(); };
struct MatExpr;
struct Mat {
  int flags, dims, rows, cols; uchar* data; struct Step { size_t p[2]; operator size_t() const; size_t operator[](int) const; } step;
  Mat(); Mat(int, int, int); Mat(int, int, int, const Scalar&); Mat(Size, int); Mat(int, int, int, void*, size_t = 0); Mat(const Mat&); Mat(const MatExpr&);
  template <class T> explicit Mat(const std::vector<T>&); template <class T, int N> explicit Mat(const Vec<T, N>&); template <class T> explicit Mat(const Point3_<T>&);
  ~Mat(); Mat& operator=(const Mat&); Mat& operator=(const MatExpr&); Mat& operator=(const Scalar&);
  Mat clone() const; void copyTo(Mat&) const; void copyTo(struct _OutputArray const&) const; void convertTo(Mat&, int, double = 1, double = 0) const; void create(int, int, int); void release();
  Mat row(int) const; Mat col(int) const; Mat rowRange(int, int) const; Mat colRange(int, int) const; Mat operator()(const Rect&) const; Mat operator()(Range, Range) const;
  MatExpr t() const; MatExpr inv(int = 0) const; MatExpr mul(const Mat&, double = 1) const; double dot(const Mat&) const; Mat cross(const Mat&) const; Mat reshape(int, int = 0) const;
  bool empty() const; int type() const; int channels() const; int depth() const; size_t total() const; size_t elemSize() const; Size size() const; bool isContinuous() const;
  template <class T> T& at(int); template <class T> const T& at(int) const; template <class T> T& at(int, int); template <class T> const T& at(int, int) const; template <class T> T& at(Point); template <class T> T* ptr(int = 0); template <class T> const T* ptr(int = 0) const; uchar* ptr(int = 0); const uchar* ptr(int = 0) const;
  static MatExpr zeros(int, int, int); static MatExpr ones(int, int, int); static MatExpr eye(int, int, int); static MatExpr zeros(Size, int);
  void setTo(const Scalar&); void push_back(const Mat&);
};
struct MatExpr { MatExpr(); MatExpr(const Mat&); operator Mat() const; MatExpr t() const; MatExpr inv(int = 0) const; Mat row(int) const; Mat col(int) const; Mat rowRange(int, int) const; Mat colRange(int, int) const; template <class T> T& at(int, int); double dot(const Mat&) const; };
MatExpr operator+(const Mat&, const Mat&); MatExpr operator-(const Mat&, const Mat&); MatExpr operator*(const Mat&, const Mat&); MatExpr operator*(const Mat&, double); MatExpr operator*(double, const Mat&); MatExpr operator/(const Mat&, double); MatExpr operator-(const Mat&);
MatExpr operator+(const MatExpr&, const Mat&); MatExpr operator+(const Mat&, const MatExpr&); MatExpr operator+(const MatExpr&, const MatExpr&); MatExpr operator-(const MatExpr&, const Mat&); MatExpr operator-(const Mat&, const MatExpr&); MatExpr operator-(const MatExpr&, const MatExpr&);
MatExpr operator*(const MatExpr&, const Mat&); MatExpr operator*(const Mat&, const MatExpr&); MatExpr operator*(const MatExpr&, const MatExpr&); MatExpr operator*(const MatExpr&, double); MatExpr operator*(double, const MatExpr&); MatExpr operator/(const MatExpr&, double); MatExpr operator-(const MatExpr&);
template <class T> struct Mat_ : Mat { Mat_(); Mat_(int, int); Mat_(int, int, const T&); Mat_(const Mat&); Mat_(const MatExpr&); Mat_& operator=(const Mat&); Mat_& operator=(const MatExpr&); T& operator()(int, int); const T& operator()(int, int) const; T& operator()(int); const T& operator()(int) const; Mat_& operator<<(const T&); Mat_& operator,(const T&); Mat_ clone() const; };
template <class T, int M, int N> struct Matx { T val[M * N]; Matx(); T& operator()(int, int); const T& operator()(int, int) const; };
typedef Matx<double, 3, 3> Matx33d; typedef Matx<float, 3, 3> Matx33f;
double norm(const Mat&, int = 4); double norm(const MatExpr&, int = 4); double norm(const Mat&, const Mat&, int = 4);
std::ostream& operator<<(std::ostream&, const Mat&);
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; KeyPoint(); KeyPoint(Point2f, float, float = -1, float = 0, int = 0, int = -1); KeyPoint(float, float, float, float = -1, float = 0, int = 0, int = -1); };
struct _InputArray { _InputArray(); _InputArray(const Mat&); _InputArray(const MatExpr&); template <class T> _InputArray(const std::vector<T>&); Mat getMat(int = -1) const; bool empty() const; };
struct _OutputArray : _InputArray { _OutputArray(); _OutputArray(Mat&); template <class T> _OutputArray(std::vector<T>&); void create(int, int, int) const; void release() const; Mat& getMatRef(int = -1) const; };
typedef const _InputArray& InputArray; typedef const _OutputArray& OutputArray; typedef const _OutputArray& InputOutputArray;
const _OutputArray& noArray();
struct FileNode { operator int() const; operator float() const; operator double() const; operator std::string() const; bool empty() const; };
struct FileStorage { enum { READ = 0, WRITE = 1 }; FileStorage(); FileStorage(const std::string&, int); bool isOpened() const; FileNode operator[](const std::string&) const; FileNode operator[](const char*) const; void release(); };
struct SVD { enum { MODIFY_A = 1, NO_UV = 2, FULL_UV = 4 }; Mat u, w, vt; SVD(); SVD(InputArray, int = 0); static void compute(InputArray, OutputArray, OutputArray, OutputArray, int = 0); };
void eigen(InputArray, OutputArray, OutputArray = noArray());
double determinant(InputArray); void hconcat(InputArray, InputArray, OutputArray); void vconcat(InputArray, InputArray, OutputArray);
void Rodrigues(InputArray, OutputArray, OutputArray = noArray());
template <class E, class M> void cv2eigen(const Mat&, M&); template <class M> void eigen2cv(const M&, Mat&);
float fastAtan2(float, float);
int cvRound(double); int cvFloor(double); int cvCeil(double);
}  // namespace cv
using cv::cvRound; using cv::cvFloor; using cv::cvCeil;
#endif
