// oracle/orc_orb.cpp -- CPU oracle (TEST INFRASTRUCTURE ONLY, parity unpinned: see orc_api.h).
// Literal restatement of the reference extractor, src/ORBExtractor.cpp:
//   ctor tables 381-442 | ComputePyramid 928-953 | ComputeKeyPointsOctTree 739-827 |
//   DistributeOctTree 511-737 + ExtractorNode::DivideNode 453-509 | IC_Angle 48-75 |
//   operator() 838-926 (face / mask cull 887-904, GaussianBlur 907-908, descriptors 79-118, 829-836)
// Documented definitions where the reference is undefined / non-terminating:
//   * DistributeOctTree sorts pair<int,ExtractorNode*> (ORBExtractor.cpp:658): ties on node size are
//     broken by heap address.  The oracle breaks them by node CREATION ORDER (ascending), which is what a
//     monotonic allocator yields.
//   * The reference never leaves its while(!bFinish) loop when a level has fewer than N/100 candidate
//     corners (all nodes final, ORBExtractor.cpp:641-643).  The oracle stops when no node can be split.
#include "orc_api.h"
#include "orc_pattern.inc"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <list>
#include <vector>

namespace {
const int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19;

struct KP { float x, y, size, angle, response; int octave; };
struct Img {
  int w = 0, h = 0;
  std::vector<uint8_t> d;  // tight
  uint8_t at(int y, int x) const { return d[(size_t)y * w + x]; }
};

struct Node {
  int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
  std::vector<KP> keys;
  std::list<Node>::iterator lit;
  bool noMore = false;
  int seq = 0;
};

void divide(const Node& p, Node& n1, Node& n2, Node& n3, Node& n4) {
  const int halfX = (int)std::ceil(static_cast<float>(p.URx - p.ULx) / 2);
  const int halfY = (int)std::ceil(static_cast<float>(p.BRy - p.ULy) / 2);
  n1.ULx = p.ULx; n1.ULy = p.ULy; n1.URx = p.ULx + halfX; n1.URy = p.ULy;
  n1.BLx = p.ULx; n1.BLy = p.ULy + halfY; n1.BRx = p.ULx + halfX; n1.BRy = p.ULy + halfY;
  n2.ULx = n1.URx; n2.ULy = n1.URy; n2.URx = p.URx; n2.URy = p.URy;
  n2.BLx = n1.BRx; n2.BLy = n1.BRy; n2.BRx = p.URx; n2.BRy = p.ULy + halfY;
  n3.ULx = n1.BLx; n3.ULy = n1.BLy; n3.URx = n1.BRx; n3.URy = n1.BRy;
  n3.BLx = p.BLx; n3.BLy = p.BLy; n3.BRx = n1.BRx; n3.BRy = p.BLy;
  n4.ULx = n3.URx; n4.ULy = n3.URy; n4.URx = n2.BRx; n4.URy = n2.BRy;
  n4.BLx = n3.BRx; n4.BLy = n3.BRy; n4.BRx = p.BRx; n4.BRy = p.BRy;
  for (const KP& kp : p.keys) {
    if (kp.x < n1.URx) {
      if (kp.y < n1.BRy) n1.keys.push_back(kp); else n3.keys.push_back(kp);
    } else if (kp.y < n1.BRy) n2.keys.push_back(kp);
    else n4.keys.push_back(kp);
  }
  if (n1.keys.size() == 1) n1.noMore = true;
  if (n2.keys.size() == 1) n2.noMore = true;
  if (n3.keys.size() == 1) n3.noMore = true;
  if (n4.keys.size() == 1) n4.noMore = true;
}

std::vector<KP> distribute(const std::vector<KP>& in, int minX, int maxX, int minY, int maxY, int N) {
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  std::vector<KP> result;
  if (nIni < 1) return result;
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::list<Node> nodes;
  std::vector<Node*> ini(nIni);
  int seq = 0;
  for (int i = 0; i < nIni; ++i) {
    Node ni;
    ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
    ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
    ni.BLx = ni.ULx; ni.BLy = maxY - minY;
    ni.BRx = ni.URx; ni.BRy = maxY - minY;
    ni.seq = seq++;
    nodes.push_back(ni);
    ini[i] = &nodes.back();
  }
  for (const KP& kp : in) {
    int idx = (int)(kp.x / hX);
    if (idx < 0) idx = 0;
    if (idx >= nIni) idx = nIni - 1;
    ini[idx]->keys.push_back(kp);
  }
  for (auto lit = nodes.begin(); lit != nodes.end();) {
    if (lit->keys.size() == 1) { lit->noMore = true; ++lit; }
    else if (lit->keys.empty()) lit = nodes.erase(lit);
    else ++lit;
  }
  bool finish = false;
  typedef std::pair<int, std::pair<int, Node*>> SizeSeqNode;  // (size, creation seq) ordering, see header note
  std::vector<SizeSeqNode> vSize;
  auto push_child = [&](Node& c, bool track, int* nToExpand) {
    if (c.keys.empty()) return;
    c.seq = seq++;
    nodes.push_front(c);
    if (c.keys.size() > 1) {
      if (nToExpand) ++*nToExpand;
      if (track) vSize.push_back(std::make_pair((int)c.keys.size(), std::make_pair(nodes.front().seq, &nodes.front())));
      nodes.front().lit = nodes.begin();
    }
  };
  while (!finish) {
    const int prevSize = (int)nodes.size();
    auto lit = nodes.begin();
    int nToExpand = 0;
    vSize.clear();
    bool anySplit = false;
    while (lit != nodes.end()) {
      if (lit->noMore) { ++lit; continue; }
      Node n1, n2, n3, n4;
      divide(*lit, n1, n2, n3, n4);
      push_child(n1, true, &nToExpand);
      push_child(n2, true, &nToExpand);
      push_child(n3, true, &nToExpand);
      push_child(n4, true, &nToExpand);
      lit = nodes.erase(lit);
      anySplit = true;
    }
    if ((int)nodes.size() >= N || ((int)nodes.size() == prevSize && (int)nodes.size() >= N / 100)) {
      finish = true;
    } else if (((int)nodes.size() + nToExpand * 3) > N) {
      while (!finish) {
        const int prevSize2 = (int)nodes.size();
        std::vector<SizeSeqNode> vPrev = vSize;
        vSize.clear();
        std::sort(vPrev.begin(), vPrev.end(),
                  [](const SizeSeqNode& a, const SizeSeqNode& b) {
                    return a.first != b.first ? a.first < b.first : a.second.first < b.second.first;
                  });
        for (int j = (int)vPrev.size() - 1; j >= 0; --j) {
          Node n1, n2, n3, n4;
          Node* p = vPrev[j].second.second;
          divide(*p, n1, n2, n3, n4);
          push_child(n1, true, nullptr);
          push_child(n2, true, nullptr);
          push_child(n3, true, nullptr);
          push_child(n4, true, nullptr);
          nodes.erase(p->lit);
          if ((int)nodes.size() >= N) break;
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize2) finish = true;
      }
    } else if (!anySplit) {
      finish = true;  // reference would spin forever here (see header note)
    }
  }
  result.reserve(nodes.size());
  for (auto& nd : nodes) {
    const KP* best = &nd.keys[0];
    float maxResponse = best->response;
    for (size_t k = 1; k < nd.keys.size(); ++k)
      if (nd.keys[k].response > maxResponse) { best = &nd.keys[k]; maxResponse = nd.keys[k].response; }
    result.push_back(*best);
  }
  return result;
}
}  // namespace

struct orc_orb {
  orc_orb_params p;
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> quota;
  int umax[HALF_PATCH_SIZE + 1];
  std::vector<Img> pyr;
  std::vector<std::vector<KP>> cand;   // per level vToDistributeKeys
  std::vector<std::vector<KP>> dist;   // per level, after octree + orientation (level coords)
  int gaussian_column_mode = 0;        // 0: integer column pass (the definition of record), 1: SSE2 float column pass (orc_cv.cpp)
};

extern "C" {

void orc_orb_set_gaussian_mode(orc_orb* o, int column_mode) { o->gaussian_column_mode = column_mode; }

orc_orb* orc_orb_create(const orc_orb_params* pp) {
  orc_orb* o = new orc_orb;
  o->p = *pp;
  const int L = pp->nlevels;
  const double scaleFactor = pp->scale_factor;  // member is double (ORBExtractor.h:103)
  o->scale.resize(L); o->sigma2.resize(L); o->inv_scale.resize(L); o->inv_sigma2.resize(L); o->quota.resize(L);
  o->scale[0] = 1.0f; o->sigma2[0] = 1.0f;
  for (int i = 1; i < L; ++i) {
    o->scale[i] = (float)(o->scale[i - 1] * scaleFactor);
    o->sigma2[i] = o->scale[i] * o->scale[i];
  }
  for (int i = 0; i < L; ++i) { o->inv_scale[i] = 1.0f / o->scale[i]; o->inv_sigma2[i] = 1.0f / o->sigma2[i]; }
  float factor = (float)(1.0f / scaleFactor);
  float nDesired = pp->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
  int sum = 0;
  for (int l = 0; l < L - 1; ++l) {
    o->quota[l] = orc_cv_round(nDesired);
    sum += o->quota[l];
    nDesired *= factor;
  }
  o->quota[L - 1] = std::max(pp->nfeatures - sum, 0);
  int v, v0;
  const int vmax = (int)std::floor(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
  const int vmin = (int)std::ceil(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
  const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
  for (v = 0; v <= vmax; ++v) o->umax[v] = orc_cv_round(std::sqrt(hp2 - v * v));
  for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
    while (o->umax[v0] == o->umax[v0 + 1]) ++v0;
    o->umax[v] = v0;
    ++v0;
  }
  o->pyr.resize(L); o->cand.resize(L); o->dist.resize(L);
  return o;
}
void orc_orb_destroy(orc_orb* o) { delete o; }
int orc_orb_nlevels(const orc_orb* o) { return o->p.nlevels; }
void orc_orb_tables(const orc_orb* o, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int* features_per_level, int* umax16) {
  for (int i = 0; i < o->p.nlevels; ++i) {
    if (scale) scale[i] = o->scale[i];
    if (inv_scale) inv_scale[i] = o->inv_scale[i];
    if (sigma2) sigma2[i] = o->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = o->inv_sigma2[i];
    if (features_per_level) features_per_level[i] = o->quota[i];
  }
  if (umax16) for (int i = 0; i <= HALF_PATCH_SIZE; ++i) umax16[i] = o->umax[i];
}

int orc_distribute_octree(const int* xys, int n, int min_x, int max_x, int min_y, int max_y, int N, int* out_xys, int cap) {
  std::vector<KP> in(n);
  for (int i = 0; i < n; ++i) in[i] = KP{(float)xys[3 * i], (float)xys[3 * i + 1], 7.f, -1.f, (float)xys[3 * i + 2], 0};
  std::vector<KP> r = distribute(in, min_x, max_x, min_y, max_y, N);
  for (int i = 0; i < (int)r.size() && i < cap; ++i) {
    out_xys[3 * i] = (int)r[i].x; out_xys[3 * i + 1] = (int)r[i].y; out_xys[3 * i + 2] = (int)r[i].response;
  }
  return (int)r.size();
}

int orc_orb_extract(orc_orb* o, const orc_camera* cam, const uint8_t* image, int w, int h, int stride,
                    const uint8_t* mask, int mstride, orc_keypoint* kps, uint8_t* desc, int cap) {
  const int L = o->p.nlevels;
  if (!image || w <= 0 || h <= 0) return 0;
  // ---- ComputePyramid (928-953).  The 19-px REFLECT_101 frame around each level is never read by any later
  // stage (FAST reads [16,w-16), IC_Angle [4,w-4), descriptors work on a border-less clone), so it is not materialised.
  for (int l = 0; l < L; ++l) {
    Img& im = o->pyr[l];
    im.w = orc_cv_round((double)((float)w * o->inv_scale[l]));
    im.h = orc_cv_round((double)((float)h * o->inv_scale[l]));
    im.d.assign((size_t)im.w * im.h, 0);
    if (l == 0) {
      for (int y = 0; y < h; ++y) memcpy(&im.d[(size_t)y * w], image + (size_t)y * stride, w);
    } else {
      const Img& pv = o->pyr[l - 1];
      orc_resize_linear(pv.d.data(), pv.w, pv.h, pv.w, im.d.data(), im.w, im.h, im.w);
    }
  }
  // ---- ComputeKeyPointsOctTree (739-827)
  const float Wc = 30;
  for (int l = 0; l < L; ++l) {
    const Img& im = o->pyr[l];
    const int minBX = EDGE_THRESHOLD - 3, minBY = minBX;
    const int maxBX = im.w - EDGE_THRESHOLD + 3, maxBY = im.h - EDGE_THRESHOLD + 3;
    std::vector<KP>& vToDist = o->cand[l];
    vToDist.clear();
    o->dist[l].clear();
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const int nCols = (int)(width / Wc), nRows = (int)(height / Wc);
    if (nCols < 1 || nRows < 1) continue;
    const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
    std::vector<int> tmp(3 * (size_t)(wCell + 6) * (hCell + 6));
    for (int i = 0; i < nRows; ++i) {
      const float iniY = (float)(minBY + i * hCell);
      float maxY = iniY + hCell + 6;
      if (iniY >= maxBY - 3) continue;
      if (maxY > maxBY) maxY = (float)maxBY;
      for (int j = 0; j < nCols; ++j) {
        const float iniX = (float)(minBX + j * wCell);
        float maxX = iniX + wCell + 6;
        if (iniX >= maxBX - 6) continue;
        if (maxX > maxBX) maxX = (float)maxBX;
        const int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
        const uint8_t* roi = &im.d[(size_t)y0 * im.w + x0];
        int n = orc_fast(roi, x1 - x0, y1 - y0, im.w, o->p.ini_th_fast, tmp.data(), (int)tmp.size() / 3);
        if (n == 0) n = orc_fast(roi, x1 - x0, y1 - y0, im.w, o->p.min_th_fast, tmp.data(), (int)tmp.size() / 3);
        for (int k = 0; k < n; ++k)
          vToDist.push_back(KP{(float)(tmp[3 * k] + j * wCell), (float)(tmp[3 * k + 1] + i * hCell), 7.f, -1.f,
                               (float)tmp[3 * k + 2], 0});
      }
    }
    std::vector<KP> keys = distribute(vToDist, minBX, maxBX, minBY, maxBY, o->quota[l]);
    const int scaledPatchSize = (int)(PATCH_SIZE * o->scale[l]);
    for (KP& k : keys) { k.x += minBX; k.y += minBY; k.octave = l; k.size = (float)scaledPatchSize; }
    // computeOrientation / IC_Angle (48-75, 444-451)
    for (KP& k : keys) {
      int m_01 = 0, m_10 = 0;
      const int cy = orc_cv_round(k.y), cx = orc_cv_round(k.x);
      for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * im.at(cy, cx + u);
      for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        const int d = o->umax[v];
        for (int u = -d; u <= d; ++u) {
          const int val_plus = im.at(cy + v, cx + u), val_minus = im.at(cy - v, cx + u);
          v_sum += (val_plus - val_minus);
          m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
      }
      k.angle = orc_fast_atan2((float)m_01, (float)m_10);
    }
    o->dist[l] = keys;
  }
  // ---- operator() tail (858-925)
  int total = 0;
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  std::vector<uint8_t> blurred;
  for (int l = 0; l < L; ++l) {
    std::vector<KP> keys = o->dist[l];
    if (keys.empty()) continue;
    std::vector<KP> kept;
    std::vector<std::pair<float, float>> newpt;
    const float scale = o->scale[l];
    for (const KP& k : keys) {
      const float px = k.x * scale, py = k.y * scale;
      if (orc_face_in_cubemap(cam, px, py) == ORC_FACE_UNKNOWN) continue;
      if (px < 0 || (int)(px + 0.5f) >= w || py < 0 || (int)(py + 0.5f) >= h) continue;
      if (mask[(size_t)(int)(py + 0.5f) * mstride + (int)(px + 0.5f)] == 0) continue;
      kept.push_back(k);
      newpt.push_back(std::make_pair(px, py));
    }
    const Img& im = o->pyr[l];
    blurred.resize((size_t)im.w * im.h);
    orc_gaussian_blur7_mode(im.d.data(), im.w, im.h, im.w, blurred.data(), im.w, o->gaussian_column_mode);
    for (size_t i = 0; i < kept.size(); ++i) {
      const KP& k = kept[i];
      const float angle = (float)k.angle * factorPI;
      // reference: a = (float)cos(angle), b = (float)sin(angle) on floats (ORBExtractor.cpp:84) -> the platform's cosf/sinf.
      // libm float results differ between glibc versions in the last ulp (2.6 % of angles vs this image's glibc); the
      // oracle fixes the CORRECTLY ROUNDED value (what glibc >= 2.41 / CORE-MATH returns), evaluated via long double.
      const float a = (float)cosl((long double)angle), b = (float)sinl((long double)angle);
      const int cy = orc_cv_round(k.y), cx = orc_cv_round(k.x);
      uint8_t d32[32];
      const signed char* pat = kOrcPattern;
      auto tap = [&](int idx) -> int {
        const float px = (float)pat[2 * idx], py = (float)pat[2 * idx + 1];
        const int yy = cy + orc_cv_round((double)(px * b + py * a));
        const int xx = cx + orc_cv_round((double)(px * a - py * b));
        // In-bounds whenever the key point is >= 19 px from the level edge (always true: FAST interior);
        // the clamp only defines otherwise-undefined reads.
        const int yc = yy < 0 ? 0 : yy >= im.h ? im.h - 1 : yy, xc = xx < 0 ? 0 : xx >= im.w ? im.w - 1 : xx;
        return blurred[(size_t)yc * im.w + xc];
      };
      for (int i8 = 0; i8 < 32; ++i8, pat += 32) {
        int val = 0;
        for (int bit = 0; bit < 8; ++bit) val |= (tap(2 * bit) < tap(2 * bit + 1)) << bit;
        d32[i8] = (uint8_t)val;
      }
      if (total < cap) {
        kps[total] = orc_keypoint{newpt[i].first, newpt[i].second, k.size, k.angle, k.response, k.octave};
        memcpy(desc + (size_t)total * 32, d32, 32);
      }
      ++total;
    }
  }
  return total;
}

int orc_orb_level_size(const orc_orb* o, int level, int* w, int* h) {
  *w = o->pyr[level].w; *h = o->pyr[level].h; return 0;
}
void orc_orb_level_copy(const orc_orb* o, int level, uint8_t* dst, int dstride) {
  const Img& im = o->pyr[level];
  for (int y = 0; y < im.h; ++y) memcpy(dst + (size_t)y * dstride, &im.d[(size_t)y * im.w], im.w);
}
int orc_orb_level_candidates(const orc_orb* o, int level, int* xys, int cap) {
  const auto& c = o->cand[level];
  for (int i = 0; i < (int)c.size() && i < cap; ++i) {
    xys[3 * i] = (int)c[i].x; xys[3 * i + 1] = (int)c[i].y; xys[3 * i + 2] = (int)c[i].response;
  }
  return (int)c.size();
}
int orc_orb_level_distributed(const orc_orb* o, int level, orc_keypoint* kps, int cap) {
  const auto& c = o->dist[level];
  for (int i = 0; i < (int)c.size() && i < cap; ++i)
    kps[i] = orc_keypoint{c[i].x, c[i].y, c[i].size, c[i].angle, c[i].response, c[i].octave};
  return (int)c.size();
}

}  // extern "C"
