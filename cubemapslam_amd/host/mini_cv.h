// mini_cv.h -- the few OpenCV value types the reference's hot-path signatures mention, as plain structs.
// OpenCV is not available in this image (and is not needed by the product): the host-side mirror classes in this
// directory keep the reference's names and argument meaning so that Tracking / LocalMapping style callers compile
// unchanged against either cv:: or this header.  Only what ORBextractor::operator(), ORBMatcher, Optimizer and the
// remap entry point touch is provided.
#ifndef CMS_MINI_CV_H
#define CMS_MINI_CV_H
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace cv {

enum { CV_8U = 0, CV_8UC1 = 0, CV_32F = 5, CV_64F = 6 };
enum { INTER_LINEAR = 1, BORDER_CONSTANT = 0 };

template <class T> struct Point_ { T x = 0, y = 0; Point_() {} Point_(T a, T b) : x(a), y(b) {} };
typedef Point_<float> Point2f;
template <class T, int N> struct Vec {
  T v[N];
  T& operator()(int i) { return v[i]; }
  const T& operator()(int i) const { return v[i]; }
};
typedef Vec<float, 3> Vec3f;
struct Scalar { double v[4] = {0, 0, 0, 0}; };

struct KeyPoint {
  Point2f pt;
  float size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
};

// Reference-counted 2-D matrix of one of {u8, f32, f64}; row-major with a byte step (like cv::Mat for these uses).
class Mat {
 public:
  int rows = 0, cols = 0, depth = CV_8U;
  size_t step = 0;
  uint8_t* data = nullptr;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void* ext, size_t ext_step) : rows(r), cols(c), depth(type), step(ext_step), data((uint8_t*)ext) {}
  void create(int r, int c, int type) {
    if (r == rows && c == cols && type == depth && data) return;
    rows = r; cols = c; depth = type; step = (size_t)c * elemSize();
    owner_.reset(new std::vector<uint8_t>((size_t)r * step));
    data = owner_->data();
  }
  static Mat zeros(int r, int c, int type) { Mat m(r, c, type); std::memset(m.data, 0, (size_t)r * m.step); return m; }
  void release() { owner_.reset(); data = nullptr; rows = cols = 0; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return depth; }
  size_t elemSize() const { return depth == CV_8U ? 1 : depth == CV_32F ? 4 : 8; }
  template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
  template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
  template <class T> T& at(int r, int c) { return ptr<T>(r)[c]; }
  template <class T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
  Mat row(int r) const { Mat m; m.rows = 1; m.cols = cols; m.depth = depth; m.step = step; m.data = data + (size_t)r * step; m.owner_ = owner_; return m; }
  void setTo(uint8_t v) { for (int r = 0; r < rows; ++r) std::memset(data + (size_t)r * step, v, (size_t)cols * elemSize()); }

 private:
  std::shared_ptr<std::vector<uint8_t>> owner_;
};
typedef const Mat& InputArray;
typedef Mat& OutputArray;

}  // namespace cv
#endif
