// oracle/orc_match.cpp -- CPU oracle (TEST INFRASTRUCTURE ONLY, parity unpinned: see orc_api.h).
// ORBMatcher::DescriptorDistance (src/ORBMatcher.cpp:951-967) and the best / second-best scan that every
// guided matcher runs over its candidate list (src/ORBMatcher.cpp:84-113 SearchByProjection(Frame, MapPoints);
// :186-205 SearchByProjection(Cur, Last) keeps only the best).
#include "orc_api.h"
#include <cstring>

extern "C" {

int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; ++i) {
    uint32_t pa, pb;
    memcpy(&pa, a + 4 * i, 4);
    memcpy(&pb, b + 4 * i, 4);
    unsigned int v = pa ^ pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

void orc_hamming_best2(const uint8_t* qdesc, int nq, const uint8_t* tdesc, const int* cand_off, const int* cand_idx,
                       const int* tlevel, int* best_idx, int* best_dist, int* best_level, int* second_dist,
                       int* second_level) {
  for (int q = 0; q < nq; ++q) {
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int c = cand_off[q]; c < cand_off[q + 1]; ++c) {
      const int idx = cand_idx[c];
      const int dist = orc_descriptor_distance(qdesc + (size_t)q * 32, tdesc + (size_t)idx * 32);
      if (dist < bestDist) {
        bestDist2 = bestDist; bestDist = dist;
        bestLevel2 = bestLevel; bestLevel = tlevel ? tlevel[idx] : 0;
        bestIdx = idx;
      } else if (dist < bestDist2) {
        bestLevel2 = tlevel ? tlevel[idx] : 0;
        bestDist2 = dist;
      }
    }
    best_idx[q] = bestIdx; best_dist[q] = bestDist; second_dist[q] = bestDist2;
    if (best_level) best_level[q] = bestLevel;
    if (second_level) second_level[q] = bestLevel2;
  }
}

void orc_hamming_matrix(const uint8_t* a, int na, const uint8_t* b, int nb, uint16_t* out) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j)
      out[(size_t)i * nb + j] = (uint16_t)orc_descriptor_distance(a + (size_t)i * 32, b + (size_t)j * 32);
}

}  // extern "C"
