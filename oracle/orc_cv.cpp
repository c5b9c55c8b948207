// oracle/orc_cv.cpp -- CPU oracle (TEST INFRASTRUCTURE ONLY, parity unpinned: see orc_api.h).
// Restatement of the OpenCV 2.4.11 / 3.2 primitives the reference calls on its hot path.  OpenCV is a
// third-party, un-vendored dependency (README.md:59); the reference call sites are
//   cv::remap            src/System.cpp:350-354
//   cv::resize, copyMakeBorder   src/ORBExtractor.cpp:941-949
//   cv::FAST             src/ORBExtractor.cpp:783-789
//   cv::GaussianBlur     src/ORBExtractor.cpp:908
//   cv::fastAtan2        src/ORBExtractor.cpp:74
//   cvRound              src/ORBExtractor.cpp:53,94-96,411,436,932
// Semantics follow SURVEY.md Appendix C (plain C++ / integer paths of those OpenCV versions).
#include "orc_api.h"
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}
inline short sat_short(float v) {
  int iv = (int)lrint((double)v);
  return (short)(iv < -32768 ? -32768 : iv > 32767 ? 32767 : iv);
}
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
}  // namespace

extern "C" {

int orc_cv_round(double v) { return (int)lrint(v); }  // round-half-to-even (cvtsd2si)

float orc_fast_atan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
  const float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// remap(INTER_LINEAR, BORDER_CONSTANT 0), 8UC1, two CV_32FC1 maps: 5-bit fractional fixed point,
// 15-bit weights, dst = (sum w*p + 16384) >> 15.
void orc_remap_bilinear(const uint8_t* src, int sw, int sh, int sstride, const float* map1, const float* map2,
                        int mstride, uint8_t* dst, int dw, int dh, int dstride) {
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x) {
      const int sx = orc_cv_round((double)(map1[(size_t)y * mstride + x] * 32.f));  // float product, as cvRound(sX[x]*INTER_TAB_SIZE)
      const int sy = orc_cv_round((double)(map2[(size_t)y * mstride + x] * 32.f));
      int X = sx >> 5, Y = sy >> 5;
      X = X < -32768 ? -32768 : X > 32767 ? 32767 : X;  // saturate_cast<short>
      Y = Y < -32768 ? -32768 : Y > 32767 ? 32767 : Y;
      const int ax = sx & 31, ay = sy & 31;
      int w[4] = {(32 - ax) * (32 - ay) * 32, ax * (32 - ay) * 32, (32 - ax) * ay * 32, ax * ay * 32};
      if (ax == 0 && ay == 0) { w[0] = 32767; w[3] = 1; }  // table entry after saturate_cast<short> + sum fix-up
      int v;
      if ((unsigned)X < (unsigned)(sw - 1) && (unsigned)Y < (unsigned)(sh - 1)) {
        const uint8_t* S = src + (size_t)Y * sstride + X;
        v = S[0] * w[0] + S[1] * w[1] + S[sstride] * w[2] + S[sstride + 1] * w[3];
      } else if (X >= sw || X + 1 < 0 || Y >= sh || Y + 1 < 0) {
        dst[(size_t)y * dstride + x] = 0;
        continue;
      } else {
        auto at = [&](int yy, int xx) -> int {
          return ((unsigned)xx < (unsigned)sw && (unsigned)yy < (unsigned)sh) ? src[(size_t)yy * sstride + xx] : 0;
        };
        v = at(Y, X) * w[0] + at(Y, X + 1) * w[1] + at(Y + 1, X) * w[2] + at(Y + 1, X + 1) * w[3];
      }
      dst[(size_t)y * dstride + x] = sat_u8((v + 16384) >> 15);
    }
}

void orc_fisheye_to_cubemap(const orc_camera* cam, const float* map1, const float* map2, const uint8_t* fisheye,
                            int fstride, uint8_t* cubemap, int cstride) {
  const int F = cam->face, W = 3 * F;  // System.cpp:327-355; corner blocks are left untouched
  const int fx0[5] = {F, 0, 2 * F, F, F};  // front, left, right, upper, lower
  const int fy0[5] = {F, F, F, 0, 2 * F};
  for (int f = 0; f < 5; ++f) {
    const size_t mo = (size_t)fy0[f] * W + fx0[f];
    orc_remap_bilinear(fisheye, cam->Iw, cam->Ih, fstride, map1 + mo, map2 + mo, W,
                       cubemap + (size_t)fy0[f] * cstride + fx0[f], F, F, cstride);
  }
}

// resize(INTER_LINEAR) 8UC1: 11-bit coefficients, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2
void orc_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(2 * dw), ibeta(2 * dh);
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)std::floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    ialpha[2 * dx] = sat_short((1.f - fx) * 2048);
    ialpha[2 * dx + 1] = sat_short(fx * 2048);
  }
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)std::floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[2 * dy] = sat_short((1.f - fy) * 2048);
    ibeta[2 * dy + 1] = sat_short(fy * 2048);
  }
  auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
  std::vector<int> row0(dw), row1(dw);
  for (int dy = 0; dy < dh; ++dy) {
    const int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
    const uint8_t* S0 = src + (size_t)sy0 * sstride;
    const uint8_t* S1 = src + (size_t)sy1 * sstride;
    for (int dx = 0; dx < dw; ++dx) {
      const int sx = xofs[dx];
      const int sx1 = sx + 1 < sw ? sx + 1 : sx;  // weight is 0 there (xmax path multiplies S[sx] by 2048)
      row0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx1] * ialpha[2 * dx + 1];
      row1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx1] * ialpha[2 * dx + 1];
    }
    const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    for (int dx = 0; dx < dw; ++dx)
      dst[(size_t)dy * dstride + dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
  }
}

// GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) 8UC1, OpenCV <= 3.2 fixed-point path:
// integer kernel cvRound(k*256), int32 row pass, column pass (sum + 32768) >> 16, saturated.
//
// column_mode 1 = what an x86 (SSE2) build of OpenCV 2.4 / 3.2 computes: the column functor SymmColumnVec_32s8u
// converts the int32 row sums to float, multiplies by the kernel as float(k / 65536), accumulates in float in the order
// S[0] k0 + (S[1] + S[-1]) k1 + (S[2] + S[-2]) k2 + (S[3] + S[-3]) k3, and converts with cvtps2dq (round half to EVEN) + saturating packs,
// for the columns its 16- and 4-wide loops cover (x < width & ~3); the last width % 4 columns take the scalar integer formula.
// Every product and partial sum is exact in binary32 while the sum is below 256, so the two modes differ exactly where the integer sum
// sits on a tie (sum mod 65536 == 32768) whose quotient is even -- restated here literally in float arithmetic, not as that rule.
static void gaussian_blur7_mode(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, int column_mode);
void orc_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) { gaussian_blur7_mode(src, w, h, sstride, dst, dstride, 0); }
void orc_gaussian_blur7_mode(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, int column_mode) {
  gaussian_blur7_mode(src, w, h, sstride, dst, dstride, column_mode);
}
static void gaussian_blur7_mode(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, int column_mode) {
  int k[7];
  {
    const double sigma = 2.0, scale2X = -0.5 / (sigma * sigma);
    float cf[7];
    double sum = 0;
    for (int i = 0; i < 7; ++i) {
      const double x = i - 3, t = std::exp(scale2X * x * x);
      cf[i] = (float)t;
      sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; ++i) k[i] = orc_cv_round((double)((float)(cf[i] * sum)) * 256.0);
  }
  // row pass into an int buffer (source row copied once into a REFLECT_101-padded line so the inner loop is branch free
  // and vectorisable -- the CPU baseline should not be slower than it has to be), then the column pass over row pointers
  std::vector<int> tmp((size_t)w * h);
  std::vector<uint8_t> line((size_t)w + 6);
  for (int y = 0; y < h; ++y) {
    for (int x = -3; x < w + 3; ++x) line[x + 3] = src[(size_t)y * sstride + reflect101(x, w)];
    int* trow = &tmp[(size_t)y * w];
    const uint8_t* p = line.data();
    for (int x = 0; x < w; ++x)
      trow[x] = k[0] * p[x] + k[1] * p[x + 1] + k[2] * p[x + 2] + k[3] * p[x + 3] + k[4] * p[x + 4] + k[5] * p[x + 5] + k[6] * p[x + 6];
  }
  for (int y = 0; y < h; ++y) {
    const int* r[7];
    for (int t = -3; t <= 3; ++t) r[t + 3] = &tmp[(size_t)reflect101(y + t, h) * w];
    uint8_t* drow = dst + (size_t)y * dstride;
    for (int x = 0; x < w; ++x) {
      if (column_mode == 1 && x < (w & ~3)) {
        const float kf[4] = {(float)(k[3] * (1.0 / 65536.0)), (float)(k[2] * (1.0 / 65536.0)), (float)(k[1] * (1.0 / 65536.0)), (float)(k[0] * (1.0 / 65536.0))};
        float acc = (float)r[3][x] * kf[0] + 0.0f;                         // delta = 0
        acc = acc + (float)(r[4][x] + r[2][x]) * kf[1];
        acc = acc + (float)(r[5][x] + r[1][x]) * kf[2];
        acc = acc + (float)(r[6][x] + r[0][x]) * kf[3];
        drow[x] = sat_u8((int)std::nearbyintf(acc));                      // cvtps2dq under the default rounding mode: half to even
        continue;
      }
      const int s = k[0] * r[0][x] + k[1] * r[1][x] + k[2] * r[2][x] + k[3] * r[3][x] + k[4] * r[4][x] + k[5] * r[5][x] + k[6] * r[6][x];
      drow[x] = sat_u8((s + 32768) >> 16);
    }
  }
}

// cv::FAST(img, kps, threshold, true) == FAST_t<16>: literal restatement incl. the 3-row score ring buffer
// and the one-row-late strict 8-neighbour non-max suppression; cornerScore<16> as in fast_score.cpp.
static int corner_score16(const uint8_t* ptr, const int pixel[25], int threshold) {
  const int K = 8, N = K * 3 + 1;
  int v = ptr[0];
  short d[N];
  for (int k = 0; k < N; ++k) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
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
  for (int k = 0; k < 16; k += 2) {
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
  return -b0 - 1;
}

int orc_fast(const uint8_t* img, int w, int h, int stride, int threshold, int* out_xys, int cap) {
  static const int offs[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                  {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
  const int K = 8, N = 16 + K + 1;
  int pixel[25];
  for (int k = 0; k < 16; ++k) pixel[k] = offs[k][0] + offs[k][1] * stride;
  for (int k = 16; k < 25; ++k) pixel[k] = pixel[k - 16];
  threshold = std::min(std::max(threshold, 0), 255);
  uint8_t tab[512];
  for (int i = -255; i <= 255; ++i) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
  static thread_local std::vector<uint8_t> sbuf;   // reused across the thousands of per-cell calls of one frame
  static thread_local std::vector<int> cbuf;
  sbuf.assign((size_t)3 * w, 0);
  cbuf.assign((size_t)3 * (w + 1), 0);
  uint8_t* buf[3] = {sbuf.data(), sbuf.data() + w, sbuf.data() + 2 * w};
  int* cpbuf[3] = {cbuf.data() + 1, cbuf.data() + (w + 1) + 1, cbuf.data() + 2 * (w + 1) + 1};
  int n = 0;
  for (int i = 3; i < h - 2; ++i) {
    const uint8_t* ptr = img + (size_t)i * stride + 3;
    uint8_t* curr = buf[(i - 3) % 3];
    int* cornerpos = cpbuf[(i - 3) % 3];
    memset(curr, 0, w);
    int ncorners = 0;
    if (i < h - 3) {
      for (int j = 3; j < w - 3; ++j, ++ptr) {
        const int v = ptr[0];
        const uint8_t* t = &tab[0] - v + 255;
        int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
        if (d == 0) continue;
        d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
        d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
        d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
        if (d == 0) continue;
        d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
        d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
        d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
        d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];
        if (d & 1) {
          const int vt = v - threshold;
          int count = 0;
          for (int k = 0; k < N; ++k) {
            if (ptr[pixel[k]] < vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
        if (d & 2) {
          const int vt = v + threshold;
          int count = 0;
          for (int k = 0; k < N; ++k) {
            if (ptr[pixel[k]] > vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
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
    const uint8_t* prev = buf[(i - 4 + 3) % 3];
    const uint8_t* pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3];
    ncorners = cornerpos[-1];
    for (int k = 0; k < ncorners; ++k) {
      const int j = cornerpos[k];
      const int score = prev[j];
      if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
          score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1]) {
        if (n < cap) { out_xys[3 * n] = j; out_xys[3 * n + 1] = i - 1; out_xys[3 * n + 2] = score; }
        ++n;
      }
    }
  }
  return n;
}

}  // extern "C"
