// Host emulation of the product's window-unfolding table (cubemapslam_amd/csrc/cms_area_table.h) + a plain AddCells walk, so that
// the table can be checked against the oracle's independent branch-by-branch transcription on the CPU.
#define CMS_AREA_HOST_EMU
#include "../../cubemapslam_amd/csrc/cms_area_table.h"
#include <cmath>
#include <vector>

extern "C" int area_emu(int F, int n, const float* kx, const float* ky, const int* koct, int nq, const float* qx, const float* qy,
                        const float* qr, const int* qmin, const int* qmax, int* off, int* idx, int cap) {
  const int G = CMS_AREA_G;
  const float inv = (float)(3 * G) / (float)(3 * F);
  std::vector<std::vector<int>> cell((size_t)5 * G * G);
  for (int i = 0; i < n; ++i) {
    CmsAreaRectI tmp[3];
    // face of the key point: same float test as the query's (FaceInCubemap(cv::Point2f) widens to double after a float division)
    const double fi = kx[i] / (float)F, fj = ky[i] / (float)F;
    int f = -1;
    if (fi >= 0 && fi < 1 && fj >= 1 && fj < 2) f = 1;
    else if (fi >= 1 && fi < 2 && fj >= 0 && fj < 1) f = 3;
    else if (fi >= 1 && fi < 2 && fj >= 1 && fj < 2) f = 0;
    else if (fi >= 1 && fi < 2 && fj >= 2 && fj < 3) f = 4;
    else if (fi >= 2 && fi < 3 && fj >= 1 && fj < 2) f = 2;
    (void)tmp;
    if (f < 0) continue;
    const int px = (int)(kx[i] * inv) % G, py = (int)(ky[i] * inv) % G;
    cell[(size_t)(f * G + px) * G + py].push_back(i);
  }
  int total = 0;
  off[0] = 0;
  for (int q = 0; q < nq; ++q) {
    CmsAreaRectI rc[3];
    const int nr = cms_area_rects(qx[q], qy[q], qr[q], F, inv, rc);
    const bool check = qmin[q] > 0 || qmax[q] >= 0;
    for (int k = 0; k < nr; ++k) {
      const int x0 = rc[k].x0 < 0 ? 0 : rc[k].x0, x1 = rc[k].x1 > G - 1 ? G - 1 : rc[k].x1;
      const int y0 = rc[k].y0 < 0 ? 0 : rc[k].y0, y1 = rc[k].y1 > G - 1 ? G - 1 : rc[k].y1;
      for (int ix = x0; ix <= x1; ++ix)
        for (int iy = y0; iy <= y1; ++iy)
          for (int j : cell[(size_t)(rc[k].face * G + ix) * G + iy]) {
            if (check && (koct[j] < qmin[q] || (qmax[q] >= 0 && koct[j] > qmax[q]))) continue;
            if (std::fabs(kx[j] - qx[q]) < qr[q] && std::fabs(ky[j] - qy[q]) < qr[q]) { if (total < cap) idx[total] = j; ++total; }
          }
    }
    off[q + 1] = total;
  }
  return total;
}
