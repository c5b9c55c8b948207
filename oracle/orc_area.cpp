// oracle/orc_area.cpp -- CPU oracle (TEST INFRASTRUCTURE ONLY, parity unpinned: see orc_api.h).
// Restatement of Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cpp:158-176, 728-744) and Frame::GetFeaturesInArea with its
// static helper AddCells (src/Frame.cpp:36-72, 251-716): the per-face 50 x 50 grid of key-point indices and the window query
// with cross-face unfolding.  The query's case analysis is written out branch by branch like the reference (including its
// quirks: a few ranges that are supersets of the window, `CUBEFACE_GRID_ROWS` used as an inclusive bound, the UPPER face with
// the window above it searching the LOWER face); AddCells clamps every range to [0, 49] and applies the level and the canvas
// distance test, so out-of-range or far-away cells only cost time there.
#include <cmath>
#include <cstdint>
#include <vector>
#include "orc_api.h"

namespace {
const int G = 50;   // CUBEFACE_GRID_ROWS == CUBEFACE_GRID_COLS (Frame.h:43-44)
enum { FRONT = 0, LEFT = 1, RIGHT = 2, UPPER = 3, LOWER = 4 };

struct Grid {
  std::vector<std::vector<int>> cell;   // [face][ix][iy] flattened
  std::vector<int>& at(int f, int ix, int iy) { return cell[(size_t)(f * G + ix) * G + iy]; }
};
}  // namespace

extern "C" int orc_features_in_area(const orc_camera* cam, int n, const float* kx, const float* ky, const int* koct, int nq,
                                    const float* qx, const float* qy, const float* qr, const int* qmin, const int* qmax,
                                    int* off, int* idx, int cap) {
  const int F = cam->face;
  const float W = (float)(3 * F);                                    // imGray.cols (Frame.cpp:145-149)
  const float inv = (float)(3 * G) / (W - 0.0f);                     // mfGridElementLengthInv
  Grid g;
  g.cell.resize((size_t)5 * G * G);
  for (int i = 0; i < n; ++i) {                                      // AssignFeaturesToGrid + PosInGrid
    const int f = orc_face_in_cubemap(cam, kx[i], ky[i]);
    if (f < 0) continue;
    int px = (int)((kx[i] - 0.0f) * inv), py = (int)((ky[i] - 0.0f) * inv);
    px %= G; py %= G;
    g.at(f, px, py).push_back(i);
  }
  int total = 0;
  off[0] = 0;
  for (int q = 0; q < nq; ++q) {
    const float x = qx[q], y = qy[q], r = qr[q];
    const int minLevel = qmin[q], maxLevel = qmax[q];
    const bool check = (minLevel > 0) || (maxLevel >= 0);
    auto add = [&](int face, int x0, int x1, int y0, int y1) {      // AddCells
      const int ax = x0 < 0 ? 0 : x0, bx = x1 > G - 1 ? G - 1 : x1, ay = y0 < 0 ? 0 : y0, by = y1 > G - 1 ? G - 1 : y1;
      for (int ix = ax; ix <= bx; ++ix)
        for (int iy = ay; iy <= by; ++iy)
          for (int j : g.at(face, ix, iy)) {
            if (check) {
              if (koct[j] < minLevel) continue;
              if (maxLevel >= 0 && koct[j] > maxLevel) continue;
            }
            const float dx = kx[j] - x, dy = ky[j] - y;
            if (std::fabs(dx) < r && std::fabs(dy) < r) { if (total < cap) idx[total] = j; ++total; }
          }
    };
    // FaceInCubemap<float>(x, y): the quotients stay float
    int face = -1;
    {
      const float i = x / (float)F, j = y / (float)F;
      if (i >= 0 && i < 1 && j >= 1 && j < 2) face = LEFT;
      else if (i >= 1 && i < 2 && j >= 0 && j < 1) face = UPPER;
      else if (i >= 1 && i < 2 && j >= 1 && j < 2) face = FRONT;
      else if (i >= 1 && i < 2 && j >= 2 && j < 3) face = LOWER;
      else if (i >= 2 && i < 3 && j >= 1 && j < 2) face = RIGHT;
    }
    if (face >= 0) {
      const int cornerX = (int)x / F * F, cornerY = (int)y / F * F;
      const float xin = x - (float)cornerX, yin = y - (float)cornerY;
      const float xs = xin - r, xe = xin + r, ys = yin - r, ye = yin + r;
      const bool xu = xs < 0, xo = xe > (float)(F - 1), yu = ys < 0, yo = ye > (float)(F - 1);
      const bool xinf = !xo && !xu, yinf = !yo && !yu;
      auto fl = [&](float v) { return (int)std::floor(v * inv); };
      const float Ff = (float)F;
      if (xinf && yinf) {
        add(face, fl(xs), fl(xe), fl(ys), fl(ye));
      } else if (xinf && !yinf) {
        const int a = fl(xs), b = fl(xe);
        switch (face) {
          case FRONT:
            if (yo) { add(FRONT, a, b, fl(ys), G - 1); add(LOWER, a, b, 0, fl(ye - Ff)); }
            else { add(UPPER, a, b, fl(ys + Ff), G - 1); add(FRONT, a, b, 0, fl(ye)); }
            break;
          case LEFT:
            if (yo) { add(LEFT, a, b, fl(ys), G - 1); add(LOWER, 0, fl(ye - Ff), G - b - 1, G - a - 1); }
            else { add(UPPER, 0, fl(-ys), a, b); add(LEFT, a, b, 0, fl(ye)); }
            break;
          case RIGHT:
            if (yo) { add(RIGHT, a, b, fl(ys), G - 1); add(LOWER, G - fl(ye - Ff) - 1, G - 1, a, b); }
            else { add(UPPER, fl(ys + Ff), G - 1, G - b - 1, G - a - 1); add(RIGHT, a, b, 0, fl(ye)); }
            break;
          case UPPER:
            if (yo) { add(UPPER, a, b, fl(ys), G - 1); add(FRONT, a, b, 0, fl(ye - Ff)); }
            else { add(LOWER, a, b, 0, fl(ye)); }                  // the reference names LOWER_FACE here (Frame.cpp:373)
            break;
          case LOWER:
            if (yo) { add(LOWER, a, b, fl(ys), G - 1); }
            else { add(FRONT, a, b, fl(ys + Ff), G - 1); add(LOWER, a, b, 0, fl(ye)); }
            break;
        }
      } else if (!xinf && yinf) {
        const int c = fl(ys), d = fl(ye);
        switch (face) {
          case FRONT:
            if (xo) { add(FRONT, fl(xs), G - 1, c, d); add(RIGHT, 0, fl(xe - Ff), c, d); }
            else { add(LEFT, fl(xs + Ff), G - 1, c, d); add(FRONT, 0, fl(xe), c, d); }
            break;
          case LEFT:
            if (xo) { add(FRONT, 0, fl(xe - Ff), c, d); add(LEFT, fl(xs), G - 1, c, d); }
            else { add(LEFT, 0, fl(xe), c, d); }
            break;
          case RIGHT:
            if (xo) { add(RIGHT, fl(xs), G - 1, c, d); }
            else { add(FRONT, fl(xs + Ff), G - 1, c, d); add(RIGHT, 0, fl(xe), c, d); }
            break;
          case UPPER:
            if (xo) { add(UPPER, fl(xs), G - 1, c, d); add(RIGHT, G - d - 1, G - c - 1, 0, fl(xe - Ff)); }
            else { add(LEFT, c, d, 0, fl(-xs)); add(UPPER, 0, fl(xe), c, d); }
            break;
          case LOWER:
            if (xo) { add(LOWER, fl(xs), G - 1, c, d); add(RIGHT, c, d, G - fl(xe - Ff) - 1, G); }
            else { add(LEFT, G - d - 1, G - c - 1, fl(xs + Ff), G - 1); add(LOWER, 0, fl(xe), c, d); }
            break;
        }
      } else {
        switch (face) {
          case FRONT:
            if (xo && yo) { const int a = fl(xs), c = fl(ys); add(FRONT, a, G - 1, c, G - 1); add(RIGHT, 0, fl(xe - Ff), c, G - 1); add(LOWER, a, G - 1, 0, fl(ye - Ff)); }
            else if (xu && yo) { const int b = fl(xe), c = fl(ys); add(FRONT, 0, b, c, G - 1); add(LEFT, fl(xs + Ff), G - 1, c, G - 1); add(LOWER, 0, b, 0, fl(ye - Ff)); }
            else if (xo && yu) { const int a = fl(xs), d = fl(ye); add(FRONT, a, G - 1, 0, d); add(RIGHT, 0, fl(xe - Ff), 0, d); add(UPPER, a, G - 1, G - fl(ys + Ff) - 1, G - 1); }
            else if (xu && yu) { const int b = fl(xe), d = fl(ye); add(FRONT, 0, b, 0, d); add(LEFT, G - fl(xs + Ff) - 1, G - 1, 0, d); add(UPPER, 0, b, G - fl(ys + Ff) - 1, G - 1); }
            break;
          case LEFT:
            if (xo && yo) { const int a = fl(xs), c = fl(ys); add(LEFT, a, G - 1, c, G - 1); add(FRONT, 0, fl(xe - Ff), c, G - 1); add(LOWER, 0, fl(ye - Ff), 0, G - a - 1); }
            else if (xu && yo) { const int b = fl(xe), c = fl(ys); add(LEFT, 0, b, c, G - 1); add(LOWER, 0, fl(ye - Ff), G - b - 1, G - 1); }
            else if (xo && yu) { const int a = fl(xs), d = fl(ye); add(LEFT, a, G - 1, 0, d); add(FRONT, 0, fl(xe - Ff), 0, d); add(UPPER, 0, fl(-ys), a, G - 1); }
            else if (xu && yu) { const int b = fl(xe), d = fl(ye); add(LEFT, 0, b, 0, d); add(UPPER, 0, fl(-ys), 0, d); }
            break;
          case RIGHT:
            if (xo && yo) { const int a = fl(xs), c = fl(ys); add(RIGHT, a, G - 1, c, G - 1); add(LOWER, G - fl(ye - Ff) - 1, G - 1, a, G - 1); }
            else if (xu && yo) { const int b = fl(xe), c = fl(ys); add(RIGHT, 0, b, c, G - 1); add(FRONT, G - fl(-xs) - 1, G - 1, c, G - 1); add(LOWER, G - fl(ye - Ff) - 1, G - 1, 0, b); }
            else if (xo && yu) { const int a = fl(xs), d = fl(ye); add(RIGHT, a, G - 1, 0, d); add(UPPER, G - fl(-ys) - 1, G - 1, 0, G - a - 1); }
            else if (xu && yu) { const int b = fl(xe), d = fl(ye); add(RIGHT, 0, b, 0, d); add(FRONT, G - fl(-xs) - 1, G - 1, 0, d); add(UPPER, G - fl(-ys) - 1, G - 1, G - b - 1, G - 1); }
            break;
          case UPPER:
            if (xo && yo) { const int a = fl(xs), c = fl(ys); add(UPPER, a, G - 1, c, G - 1); add(RIGHT, 0, G - c - 1, 0, fl(xe - Ff)); add(FRONT, a, G - 1, 0, fl(ye - Ff)); }
            else if (xu && yo) { const int b = fl(xe), c = fl(ys); add(UPPER, 0, b, c, G - 1); add(LEFT, c, G - 1, 0, fl(-xs)); add(FRONT, 0, b, 0, fl(ye - Ff)); }
            else if (xo && yu) { const int a = fl(xs), d = fl(ye); add(UPPER, a, G - 1, 0, d); add(RIGHT, G - d - 1, G - 1, 0, d); }
            else if (xu && yu) { const int b = fl(xe), d = fl(ye); add(UPPER, 0, b, 0, d); add(LEFT, 0, d, 0, fl(-xs)); }
            break;
          case LOWER:
            if (xo && yo) { const int a = fl(xs), c = fl(ys); add(LOWER, a, G - 1, c, G - 1); add(RIGHT, a, G - 1, G - fl(xe - Ff) - 1, G - 1); }
            else if (xu && yo) { const int b = fl(xe), c = fl(ys); add(LOWER, 0, b, c, G - 1); add(LEFT, 0, G - c - 1, G - fl(xs + Ff) - 1, G - 1); }
            else if (xo && yu) { const int a = fl(xs), d = fl(ye); add(LOWER, a, G - 1, 0, d); add(RIGHT, 0, d, G - fl(xe - Ff) - 1, G); add(FRONT, a, G - 1, G - fl(-ys) - 1, G - 1); }
            else if (xu && yu) { const int b = fl(xe), d = fl(ye); const int an = fl(-xs); add(LOWER, 0, b, 0, d); add(LEFT, G - an - 1, G - 1, G - an - 1, G); add(FRONT, 0, b, G - fl(-ys) - 1, G - 1); }
            break;
        }
      }
    }
    off[q + 1] = total;
  }
  return total;
}
