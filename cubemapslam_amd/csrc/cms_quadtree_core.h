// cms_quadtree_core.h -- key-point distribution (ORBextractor::DistributeOctTree, ORBExtractor.cpp:511-737, and
// ExtractorNode::DivideNode, :453-509) re-designed as a single-workgroup data-parallel algorithm.
//
// The reference keeps a std::list of nodes, each owning a vector of key points, splits nodes one by one and pushes
// children to the list front.  Here a node is just a rectangle + a count, every candidate corner carries the list
// position of its node, and one "round" splits a whole set of nodes at once:
//   1. candidates vote their quadrant into per-node child counters (LDS atomics),
//   2. prefix sums give every child / surviving node its position in the NEW list
//      (children of the processed nodes, reversed, in front; untouched nodes behind, order kept),
//   3. candidates are re-labelled.
// The reference's "split the biggest nodes first until N is reached" tail becomes: sort the expandable nodes by
// (count, creation order) descending, prefix-sum the size deltas, cut at the first prefix that reaches N.
// The result (which key point survives in which list position) is identical to the sequential algorithm; ties on
// node size follow creation order ("reference non-determinism") and ties on response follow candidate order
// (cell row, cell column, y, x), reproduced with a packed 64-bit arg-max key.
//
// The same source compiles for the device (one workgroup per (frame, level)) and, with CMS_QT_HOST_EMU, for the
// host as a sequential emulation that tests/ use to check the logic against the oracle without a GPU.
#ifndef CMS_QUADTREE_CORE_H
#define CMS_QUADTREE_CORE_H
#include <stdint.h>

#ifdef CMS_QT_HOST_EMU
#include <algorithm>
#define QT_DEV inline
#define QT_PARFOR(i, n) for (int i = 0; i < (int)(n); ++i)
#define QT_SYNC() ((void)0)
#define QT_SINGLE if (true)
static inline uint32_t qt_atomic_add(uint32_t* p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
static inline void qt_atomic_max64(unsigned long long* p, unsigned long long v) { if (v > *p) *p = v; }
#define QT_TID 0
#define QT_NT 1
#else
#define QT_DEV __device__ __forceinline__
#define QT_PARFOR(i, n) for (int i = (int)threadIdx.x; i < (int)(n); i += (int)blockDim.x)
#define QT_SYNC() __syncthreads()
#define QT_SINGLE if (threadIdx.x == 0)
#define qt_atomic_add(p, v) atomicAdd((p), (v))
#define qt_atomic_max64(p, v) atomicMax((p), (v))
#define QT_TID ((int)threadIdx.x)
#define QT_NT ((int)blockDim.x)
#endif

struct QtRect { uint16_t x0, x1, y0, y1; };

struct QtParams {
  int n;        // number of candidates
  int N;        // quota for this level (mnFeaturesPerLevel)
  int width;    // maxBorderX - minBorderX
  int height;   // maxBorderY - minBorderY
  int minB;     // 16: candidates are stored in level pixel coordinates, the tree works relative to minB
  int wCell, hCell, nCols;  // FAST cell geometry (defines candidate order = arg-max tie-break)
};

// Workspace: all arrays have MAXN entries unless noted (MAXN = power of two >= N + 3).
struct QtWork {
  int maxn;
  QtRect* rect[2];
  uint32_t* cnt[2];
  uint32_t* childcnt;            // [4*MAXN]; aliased by `best` (u64[MAXN]) at the very end
  uint16_t* childpos;            // [4*MAXN]; [4p] doubles as "new position" of an untouched node
  uint8_t* flag;                 // node is in the processed set
  uint8_t* isex;                 // node is expandable and was created in the previous round
  uint32_t* s0; uint32_t* s1;    // scan temporaries
  uint32_t* skey;                // sort keys
  uint16_t* proc;                // processing order (list positions)
  uint32_t* part;                // [QT_NT] scan partials
  int* sc;                       // [16] scalars
};
enum { QT_CUR = 0, QT_S = 1, QT_M = 2, QT_TOTC = 3, QT_NKEEP = 4, QT_NEX = 5, QT_TMP = 6, QT_PREVTOTC = 7, QT_FIN = 8 };

QT_DEV void qt_exclusive_scan(uint32_t* a, int n, const QtWork& w, int* total) {
#ifdef CMS_QT_HOST_EMU
  uint32_t acc = 0;
  for (int i = 0; i < n; ++i) { uint32_t v = a[i]; a[i] = acc; acc += v; }
  *total = (int)acc;
#else
  // thread t owns a[t*per .. t*per+per); wave-level inclusive scan of the per-thread sums with shuffles, then the (at
  // most 16) wave totals are combined through LDS -- no serial pass over the elements
  const int T = QT_NT, t = QT_TID, lane = t & 63, wv = t >> 6, nw = T >> 6;
  const int per = (n + T - 1) / T;
  const int b = t * per, e = (b + per < n) ? b + per : n;
  uint32_t s = 0;
  for (int i = b; i < e; ++i) s += a[i];
  uint32_t incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
  if (lane == 63) w.part[wv] = incl;
  __syncthreads();
  uint32_t wbase = 0, tot = 0;
  for (int q = 0; q < nw; ++q) { const uint32_t v = w.part[q]; if (q < wv) wbase += v; tot += v; }
  uint32_t acc = wbase + incl - s;
  for (int i = b; i < e; ++i) { const uint32_t v = a[i]; a[i] = acc; acc += v; }
  if (t == 0) *total = (int)tot;
  __syncthreads();
#endif
}

// sort keys[0..n) descending (n <= maxn, maxn power of two)
QT_DEV void qt_sort_desc(uint32_t* keys, int n, const QtWork& w) {
#ifdef CMS_QT_HOST_EMU
  std::sort(keys, keys + n, [](uint32_t a, uint32_t b) { return a > b; });
#else
  // keys are unique (count << 12 | creation sequence), so the position of a key in descending order is the number of keys
  // greater than it: one pass of n broadcast LDS reads per thread and two barriers, instead of the ~45 barrier-separated
  // compare-exchange steps of a bitonic network on 512 keys.  w.s1 is free at this point and serves as the output buffer.
  QT_PARFOR(i, n) {
    const uint32_t k = keys[i];
    int r = 0;
    for (int q = 0; q < n; ++q) r += keys[q] > k ? 1 : 0;
    w.s1[r] = k;
  }
  __syncthreads();
  QT_PARFOR(i, n) keys[i] = w.s1[i];
  __syncthreads();
#endif
}

QT_DEV int qt_quadrant(const QtRect& r, int x, int y) {
  const int mx = r.x0 + ((r.x1 - r.x0 + 1) >> 1);  // UL.x + ceil((UR.x-UL.x)/2)  (ORBExtractor.cpp:455)
  const int my = r.y0 + ((r.y1 - r.y0 + 1) >> 1);
  return (x >= mx ? 1 : 0) + (y >= my ? 2 : 0);     // 0:n1 1:n2 2:n3 3:n4 (ORBExtractor.cpp:485-499)
}
QT_DEV QtRect qt_child_rect(const QtRect& r, int q) {
  const int mx = r.x0 + ((r.x1 - r.x0 + 1) >> 1);
  const int my = r.y0 + ((r.y1 - r.y0 + 1) >> 1);
  QtRect c;
  c.x0 = (uint16_t)((q & 1) ? mx : r.x0); c.x1 = (uint16_t)((q & 1) ? r.x1 : mx);
  c.y0 = (uint16_t)((q & 2) ? my : r.y0); c.y1 = (uint16_t)((q & 2) ? r.y1 : my);
  return c;
}
#define QT_CX(c) ((int)((c) & 0xFFFu))
#define QT_CY(c) ((int)(((c) >> 12) & 0xFFFu))
#define QT_CR(c) ((int)((c) >> 24))

// One split round.  On entry: w.flag[p] marks the nodes to split, w.proc[0..m) lists them in processing order,
// w.childcnt holds their child counts (already voted).  Builds the new list in the other ping-pong half,
// re-labels the candidates, refreshes isex / QT_NEX / QT_S / QT_PREVTOTC.
QT_DEV void qt_apply_round(const QtParams& P, const uint32_t* cand, uint16_t* node_of, const QtWork& w, int m) {
  const int cur = w.sc[QT_CUR], nxt = cur ^ 1, S = w.sc[QT_S];
  QtRect* rc = w.rect[cur]; QtRect* rn = w.rect[nxt];
  uint32_t* cc = w.cnt[cur]; uint32_t* cn = w.cnt[nxt];
  // children per processed node, in processing order
  QT_PARFOR(j, m) {
    const int p = w.proc[j];
    int nk = 0;
    for (int q = 0; q < 4; ++q) nk += w.childcnt[4 * p + q] > 0 ? 1 : 0;
    w.s0[j] = (uint32_t)nk;
  }
  QT_SYNC();
  qt_exclusive_scan(w.s0, m, w, &w.sc[QT_TOTC]);
  QT_PARFOR(p, S) w.s1[p] = w.flag[p] ? 0u : 1u;
  QT_SYNC();
  qt_exclusive_scan(w.s1, S, w, &w.sc[QT_NKEEP]);
  QT_SYNC();
  const int totC = w.sc[QT_TOTC], nkeep = w.sc[QT_NKEEP];
  QT_SINGLE { w.sc[QT_NEX] = 0; }
  QT_SYNC();
  QT_PARFOR(j, m) {
    const int p = w.proc[j];
    int r = 0, nex = 0;
    for (int q = 0; q < 4; ++q) {
      const uint32_t c = w.childcnt[4 * p + q];
      if (c == 0) continue;
      const int pos = totC - 1 - ((int)w.s0[j] + r);
      w.childpos[4 * p + q] = (uint16_t)pos;
      rn[pos] = qt_child_rect(rc[p], q);
      cn[pos] = c;
      w.isex[pos] = c > 1 ? 1 : 0;
      nex += c > 1 ? 1 : 0;
      ++r;
    }
    if (nex) qt_atomic_add((uint32_t*)&w.sc[QT_NEX], (uint32_t)nex);
  }
  QT_PARFOR(p, S) {
    if (!w.flag[p]) {
      const int pos = totC + (int)w.s1[p];
      w.childpos[4 * p] = (uint16_t)pos;
      rn[pos] = rc[p];
      cn[pos] = cc[p];
      w.isex[pos] = 0;
    }
  }
  QT_SYNC();
  // the child counters of this round are consumed; clear them for the NEW list, then re-label every candidate and let it vote
  // its quadrant inside its new node in the same pass (one walk over the candidates per round instead of two)
  QT_PARFOR(k, 4 * (totC + nkeep)) w.childcnt[k] = 0u;
  QT_SYNC();
  QT_PARFOR(i, P.n) {
    const int p = node_of[i];
    const uint32_t c = cand[i];
    const int x = QT_CX(c) - P.minB, y = QT_CY(c) - P.minB;
    const int np_ = w.flag[p] ? w.childpos[4 * p + qt_quadrant(rc[p], x, y)] : w.childpos[4 * p];
    node_of[i] = (uint16_t)np_;
    qt_atomic_add(&w.childcnt[4 * np_ + qt_quadrant(rn[np_], x, y)], 1u);
  }
  QT_SYNC();
  QT_SINGLE { w.sc[QT_CUR] = nxt; w.sc[QT_S] = totC + nkeep; w.sc[QT_PREVTOTC] = totC; }
  QT_SYNC();
}

// children vote: every candidate whose node is flagged adds 1 to its quadrant counter
QT_DEV void qt_vote(const QtParams& P, const uint32_t* cand, const uint16_t* node_of, const QtWork& w) {
  const int cur = w.sc[QT_CUR], S = w.sc[QT_S];
  QT_PARFOR(k, 4 * S) w.childcnt[k] = 0u;
  QT_SYNC();
  QT_PARFOR(i, P.n) {
    const int p = node_of[i];
    if (w.flag[p]) {
      const uint32_t c = cand[i];
      const int q = qt_quadrant(w.rect[cur][p], QT_CX(c) - P.minB, QT_CY(c) - P.minB);
      qt_atomic_add(&w.childcnt[4 * p + q], 1u);
    }
  }
  QT_SYNC();
}

// Returns the number of surviving key points; out[pos] = packed (x | y<<12 | response<<24) in list order.
QT_DEV int qt_run(const QtParams& P, const uint32_t* cand, uint16_t* node_of, const QtWork& w, uint32_t* out, int dbg = 0) {
  if (P.n <= 0) return 0;
  QT_SINGLE {
    w.sc[QT_CUR] = 0; w.sc[QT_S] = 1; w.sc[QT_NEX] = 0; w.sc[QT_FIN] = 0; w.sc[QT_PREVTOTC] = 0;
    QtRect r; r.x0 = 0; r.x1 = (uint16_t)P.width; r.y0 = 0; r.y1 = (uint16_t)P.height;
    w.rect[0][0] = r; w.cnt[0][0] = (uint32_t)P.n; w.isex[0] = 0;
  }
  QT_PARFOR(i, P.n) node_of[i] = 0;
  QT_SINGLE { w.flag[0] = 1; }
  QT_SYNC();
  qt_vote(P, cand, node_of, w);                  // root: every later vote happens inside qt_apply_round's re-labelling pass
  if (dbg == 1) return 0;                        // developer switch (CMS_DBG_FAST_STOP=31): timing of the phases
  for (int guard = 0; guard < 64 && !w.sc[QT_FIN] && !(dbg >= 10 && guard >= dbg - 10); ++guard) {
    // ---- main pass: split every node that holds more than one key point (ORBExtractor.cpp:565-636)
    const int prevS = w.sc[QT_S];
    {
      const int cur = w.sc[QT_CUR];
      QT_PARFOR(p, prevS) { const uint8_t f = w.cnt[cur][p] > 1 ? 1 : 0; w.flag[p] = f; w.s0[p] = f; }
      QT_SYNC();
      qt_exclusive_scan(w.s0, prevS, w, &w.sc[QT_M]);
      QT_SYNC();
      QT_PARFOR(p, prevS) if (w.flag[p]) w.proc[w.s0[p]] = (uint16_t)p;
      QT_SYNC();
    }
    const int m = w.sc[QT_M];
    qt_apply_round(P, cand, node_of, w, m);     // the votes were cast while the previous round re-labelled (or by the initial vote)
    int S = w.sc[QT_S];
    const int nEx = w.sc[QT_NEX];
    QT_SYNC();
    if (S >= P.N || (S == prevS && S >= P.N / 100)) {
      QT_SINGLE { w.sc[QT_FIN] = 1; }
    } else if (S + 3 * nEx > P.N) {
      // ---- final phase: split the largest expandable nodes first until N is reached (ORBExtractor.cpp:645-711)
      for (int g2 = 0; g2 < 64; ++g2) {
        const int prevS2 = w.sc[QT_S];
        const int totCprev = w.sc[QT_PREVTOTC];
        const int cur = w.sc[QT_CUR];
        // the expandable nodes created in the previous round sit at positions [0, totCprev), seq = totCprev-1-pos
        QT_PARFOR(p, prevS2) { w.flag[p] = w.isex[p]; }
        QT_PARFOR(p, totCprev) {
          w.s0[p] = w.isex[p] ? 1u : 0u;
        }
        QT_SYNC();
        qt_exclusive_scan(w.s0, totCprev, w, &w.sc[QT_M]);
        QT_SYNC();
        const int mall = w.sc[QT_M];
        QT_PARFOR(p, totCprev) if (w.isex[p]) w.skey[w.s0[p]] = (w.cnt[cur][p] << 12) | (uint32_t)(totCprev - 1 - p);
        QT_SYNC();
        qt_sort_desc(w.skey, mall, w);
        QT_SYNC();
        // size after each split, in processing order; cut at the first that reaches N
        QT_PARFOR(j, mall) {
          const int p = totCprev - 1 - (int)(w.skey[j] & 0xFFFu);
          int nk = 0;
          for (int q = 0; q < 4; ++q) nk += w.childcnt[4 * p + q] > 0 ? 1 : 0;
          w.s1[j] = (uint32_t)nk;   // delta + 1
        }
        QT_SYNC();
        qt_exclusive_scan(w.s1, mall, w, &w.sc[QT_TMP]);
        QT_SINGLE { w.sc[QT_M] = mall; }
        QT_SYNC();
        // inclusive size_j = prevS2 + (s1[j] + nk_j) - (j+1); find the smallest j with size_j >= N
        QT_PARFOR(j, mall) {
          const int p = totCprev - 1 - (int)(w.skey[j] & 0xFFFu);
          int nk = 0;
          for (int q = 0; q < 4; ++q) nk += w.childcnt[4 * p + q] > 0 ? 1 : 0;
          const int size_j = prevS2 + (int)w.s1[j] + nk - (j + 1);
          const int size_before = prevS2 + (int)w.s1[j] - j;
          if (size_j >= P.N && size_before < P.N) w.sc[QT_M] = j + 1;  // sizes are non-decreasing: unique writer
        }
        QT_SYNC();
        const int mproc = w.sc[QT_M];
        QT_PARFOR(p, prevS2) w.flag[p] = 0;
        QT_SYNC();
        QT_PARFOR(j, mproc) {
          const int p = totCprev - 1 - (int)(w.skey[j] & 0xFFFu);
          w.proc[j] = (uint16_t)p;
          w.flag[p] = 1;
        }
        QT_SYNC();
        qt_apply_round(P, cand, node_of, w, mproc);
        S = w.sc[QT_S];
        QT_SYNC();
        if (S >= P.N || S == prevS2) break;
      }
      QT_SINGLE { w.sc[QT_FIN] = 1; }
    } else if (m == 0) {
      QT_SINGLE { w.sc[QT_FIN] = 1; }  // nothing left to split (the reference would spin here)
    }
    QT_SYNC();
  }
  if (dbg == 2 || dbg >= 10) return 0;
  // ---- retain the best point of every node (ORBExtractor.cpp:713-734): max response, first in candidate order
  const int S = w.sc[QT_S];
  unsigned long long* best = (unsigned long long*)w.childcnt;
  QT_PARFOR(p, S) best[p] = 0ull;
  QT_SYNC();
  QT_PARFOR(i, P.n) {
    const uint32_t c = cand[i];
    const int x = QT_CX(c), y = QT_CY(c);
    const int cxi = (x - P.minB - 3) / P.wCell, cyi = (y - P.minB - 3) / P.hCell;
    const unsigned long long ord = ((unsigned long long)cyi << 32) | ((unsigned long long)cxi << 24) |
                                   ((unsigned long long)y << 12) | (unsigned long long)x;   // < 2^40
    const unsigned long long key = ((unsigned long long)QT_CR(c) << 40) | (0xFFFFFFFFFFull - ord);
    qt_atomic_max64(&best[node_of[i]], key);
  }
  QT_SYNC();
  QT_PARFOR(p, S) {
    const unsigned long long k = best[p];
    const unsigned long long ord = 0xFFFFFFFFFFull - (k & 0xFFFFFFFFFFull);
    out[p] = (uint32_t)(ord & 0xFFFFFFu) | ((uint32_t)(k >> 40) << 24);
  }
  QT_SYNC();
  return S;
}

#endif
