// cms_ba_schur_points.hip -- Schur complement of the local BA with every Hpl block read ONCE per Levenberg trial.
//
// k_ba_schur_chunks (cms_ba_kernels.hip) walks the co-visibility tuples pair by pair: deterministic and atomic-free, but each
// tuple re-reads its two 6x3 blocks and the point's Dinv from memory -- 360 B per tuple, 5.5x the unique bytes of a window
// (a point seen by k key frames has k(k+1)/2 tuples over k blocks).  With several windows in flight that kernel is the
// largest of the whole BA and bandwidth bound.  Here the work is point-major:
//
//   * the host cuts the (point-sorted) edge list into BATCHES of consecutive points whose edges fit LDS, and consecutive batches
//     into RANGES, one workgroup each;
//   * a workgroup stages a batch once -- B = Hpl[e], BD = B * Dinv[point], rb = B * (Dinv bl) per edge -- into LDS;
//   * every co-visible pose pair has one owner thread ("slot"; diagonal pairs, which get one tuple per edge, have several
//     helper slots) that walks ITS tuples of the batch (host-built list per (batch, slot), 16-bit local edge ids) and
//     accumulates BD[a1] B[a2]^T in registers across all batches of the range -- no atomics, fixed order;
//   * the helpers of a pair are added in LDS, the range writes one 42-vector per pair, k_ba_schur_reduce adds the ranges in
//     order into the "one chunk per pair" array the solve kernel already consumes.
//
// Global traffic per trial: Hpl + Dinv + db once (14 MB at E = 80k) + R x npairs x 336 B of partials, instead of 66 MB.
// (block_solver.hpp:367-437: Hschur -= Bi Dinv Bj^T, bschur -= Bi Dinv bl.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef BA_SP_MAXE
#define BA_SP_MAXE 208            /* edges per batch (< 256: 8-bit local ids): 208 x (18 + 18 + 6) doubles = 68 KB of LDS */
#define BA_SP_MAXT 512            /* tuples per batch: their 16-bit words (1 KB) and the per-slot offsets are staged in LDS too */
#endif
#define BA_SP_ROW 42              /* doubles per staged edge: B (18) | BD (18) | rb (6) */
#define BA_SP_MAX_THREADS 512     /* owner threads (slots) at most */
#define BA_SP_STAGERS 256         /* + four wavefronts that only stage the next batch (>= BA_SP_MAXE: one edge per thread) */
#ifndef BA_SP_RANGES
#define BA_SP_RANGES 64           /* workgroups per window (each walks nbat / 64 batches) */
#endif

struct BaSp {                      // device view of the host-built work lists (cms_api_ba.hip)
  int nbat, bpw, nslots, npairs, R;
  int nd_slots;                   // slots [0, nd_slots) are the helper slots of the diagonal pairs (summed in LDS), the rest own a pair alone
  const int* bat_e0;              // nbat + 1: first edge of every batch
  int off_stride;                 // u32 words per batch in `off32`: ceil((nslots + 1) / 2)
  const int* tup_base;            // nbat + 1: first u32 word of every batch in `tup32` (two 16-bit tuples a1 | a2 << 8 per word)
  const uint32_t* tup32;          // tuples of a batch, grouped by slot
  const uint32_t* off32;          // per batch nslots + 1 16-bit offsets into the batch's tuples
  const int* slot_pair;           // nslots: pair id of a slot
  const int* pair_slots;          // 2 x npairs: [first, end) slot of every pair (its helpers are consecutive)
  double* partial;                // R x npairs x 42
};

struct BaSpG {                     // BaSp with global-memory pointer types (see BaDevG)
  int nbat, bpw, nslots, npairs, R, nd_slots;
  const BA_AS1 int* bat_e0; int off_stride; const BA_AS1 int* tup_base; const BA_AS1 uint32_t* tup32; const BA_AS1 uint32_t* off32;
  const BA_AS1 int* slot_pair; const BA_AS1 int* pair_slots; BA_AS1 double* partial;
  __device__ __forceinline__ BaSpG() {}
  __device__ __forceinline__ BaSpG(const BaSp& s)
      : nbat(s.nbat), bpw(s.bpw), nslots(s.nslots), npairs(s.npairs), R(s.R), nd_slots(s.nd_slots), bat_e0(ba_g(s.bat_e0)), off_stride(s.off_stride),
        tup_base(ba_g(s.tup_base)), tup32(ba_g(s.tup32)), off32(ba_g(s.off32)), slot_pair(ba_g(s.slot_pair)), pair_slots(ba_g(s.pair_slots)), partial(ba_g(s.partial)) {}
};

__device__ __forceinline__ void ba_schur_points_body(int BX, BaDevG d, BaSpG sp, const double* __restrict__ Hpl, const double* __restrict__ Dinv,
                                                     const double* __restrict__ db, const double* __restrict__ Hll = nullptr,
                                                     const double* __restrict__ bl = nullptr, double lambda = 0.0,
                                                     const double* __restrict__ poses = nullptr, const double* __restrict__ pts = nullptr) {
  // poses != nullptr: the 6x3 block of an edge is rebuilt from the estimate and the stored weight (edge_block) instead of being read
  // Hll != nullptr: the staging thread inverts (Hll + lambda I) of its edge's point itself (same arithmetic as k_ba_dinv; a
  // point's k edges repeat it, which is cheaper than a separate kernel) and Dinv / db are not read
  extern __shared__ __align__(16) double sp_lds[];      // [BA_SP_MAXE][42], reused at the end for the helper sums
  // Two kinds of wavefronts: the first blockDim - BA_SP_STAGERS threads own pose pairs and multiply tuples; the last BA_SP_STAGERS
  // threads do nothing but fetch / rebuild the edges of the NEXT batch (their loads and Jacobians run while the owners multiply) and
  // copy them into LDS between the two barriers.
  const int tid = threadIdx.x;
  const int sid = tid - ((int)blockDim.x - BA_SP_STAGERS);     // >= 0: stager, lane sid of the staging team
  const bool stager = sid >= 0;
  const bool have = tid < sp.nslots;
  double acc[42];
#pragma unroll
  for (int i = 0; i < 42; ++i) acc[i] = 0;
  const int b0 = BX * sp.bpw, b1 = min(sp.nbat, b0 + sp.bpw);
  // staging is software pipelined: BA_SP_STAGERS >= BA_SP_MAXE, so a stager handles at most ONE edge per batch; batch b + 1 is fetched
  // (rebuilt) by the stagers while the owners multiply the tuples of batch b
  double Bv[18], Dv[9], dv[3];
  uint32_t tw = 0, ow = 0;
  bool mine = false;
  uint16_t* ltup = reinterpret_cast<uint16_t*>(sp_lds + (size_t)BA_SP_MAXE * BA_SP_ROW);
  uint16_t* loff = ltup + BA_SP_MAXT;
  auto fetch = [&](int bb) {
    const int tb = sp.tup_base[bb], ntw = sp.tup_base[bb + 1] - tb;
    tw = sid < ntw ? sp.tup32[tb + sid] : 0u;
    ow = sid < sp.off_stride ? sp.off32[(size_t)bb * sp.off_stride + sid] : 0u;
    const int e0 = sp.bat_e0[bb], ne = sp.bat_e0[bb + 1] - e0;
    mine = sid < ne;
    if (mine) {
      const int e = e0 + sid, p = d.e_point[e];
      if (poses) {
        edge_block(d, e, poses, pts, Bv);
      } else {
        const double2* B = reinterpret_cast<const double2*>(Hpl + 18 * (size_t)e);   // nine 16-byte loads
#pragma unroll
        for (int i = 0; i < 9; ++i) { const double2 u = B[i]; Bv[2 * i] = u.x; Bv[2 * i + 1] = u.y; }
      }
      if (Hll) {
        double D[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) D[i] = Hll[9 * (size_t)p + i] + ((i & 3) == 0 ? lambda : 0.0);
        const double b0 = bl[3 * (size_t)p], b1 = bl[3 * (size_t)p + 1], b2 = bl[3 * (size_t)p + 2];
        inv3(D, Dv);
        dv[0] = Dv[0] * b0 + Dv[1] * b1 + Dv[2] * b2;
        dv[1] = Dv[3] * b0 + Dv[4] * b1 + Dv[5] * b2;
        dv[2] = Dv[6] * b0 + Dv[7] * b1 + Dv[8] * b2;
      } else {
        const double* Di = Dinv + 9 * (size_t)p;
        const double* dbp = db + 3 * (size_t)p;
#pragma unroll
        for (int i = 0; i < 9; ++i) Dv[i] = Di[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) dv[i] = dbp[i];
      }
    }
  };
  // The two teams run their own loops (the register allocator then sees each role on its own: the owners' 42 accumulators and the
  // stagers' rebuilt blocks never have to be live together).  Both execute exactly two barriers per batch:
  //   A  the owners have finished the tuples of the previous batch     -> the stagers may overwrite the rows
  //   B  the rows, tuple words and offsets of this batch are in LDS     -> the owners may read them
  if (stager) {
    if (b0 < b1) fetch(b0);
    for (int b = b0; b < b1; ++b) {
      __syncthreads();                                   // A
      if (sid < BA_SP_MAXT / 2) reinterpret_cast<uint32_t*>(ltup)[sid] = tw;
      if (sid < sp.off_stride) reinterpret_cast<uint32_t*>(loff)[sid] = ow;
      if (mine) {
        double* row = sp_lds + (size_t)sid * BA_SP_ROW;
#pragma unroll
        for (int i = 0; i < 18; ++i) row[i] = Bv[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j)
            row[18 + 3 * i + j] = __builtin_fma(Bv[3 * i + 2], Dv[6 + j], __builtin_fma(Bv[3 * i + 1], Dv[3 + j], Bv[3 * i] * Dv[j]));
          row[36 + i] = __builtin_fma(Bv[3 * i + 2], dv[2], __builtin_fma(Bv[3 * i + 1], dv[1], Bv[3 * i] * dv[0]));
        }
      }
      __syncthreads();                                   // B
      if (b + 1 < b1) fetch(b + 1);                      // travels / is rebuilt while the owners multiply batch b
    }
  } else {
    for (int b = b0; b < b1; ++b) {
      __syncthreads();                                   // A
      __syncthreads();                                   // B
      const int l0 = have ? loff[tid] : 0, l1 = have ? loff[tid + 1] : 0;
      for (int t = l0; t < l1; ++t) {
        const int w = ltup[t];
        const int a1 = w & 0xFF, a2 = w >> 8;
        const double* r1 = sp_lds + (size_t)a1 * BA_SP_ROW + 18;     // BD of the first edge
        const double* r2 = sp_lds + (size_t)a2 * BA_SP_ROW;          // B of the second
        double bd[18], b2[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) { bd[i] = r1[i]; b2[i] = r2[i]; }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j)
            acc[6 * i + j] = __builtin_fma(bd[3 * i + 2], b2[3 * j + 2], __builtin_fma(bd[3 * i + 1], b2[3 * j + 1], __builtin_fma(bd[3 * i], b2[3 * j], acc[6 * i + j])));
        if (a1 == a2) {
#pragma unroll
          for (int i = 0; i < 6; ++i) acc[36 + i] += r1[18 + i];     // rb sits right behind BD
        }
      }
    }
  }
  // ---- this range's slice of `partial`: an off-diagonal pair has a single owner which writes its vector directly; the helper
  // slots of the diagonal pairs are added in LDS first (fixed order)
  __syncthreads();
  if (have && tid < sp.nd_slots) {
#pragma unroll
    for (int i = 0; i < 42; ++i) sp_lds[(size_t)tid * 42 + i] = acc[i];
  } else if (have) {
    BA_AS1 double* out = sp.partial + ((size_t)BX * sp.npairs + sp.slot_pair[tid]) * 42;
#pragma unroll
    for (int i = 0; i < 42; ++i) out[i] = acc[i];
  }
  __syncthreads();
  for (int o = tid; o < sp.npairs * 42; o += blockDim.x) {
    const int pr = o / 42, k = o - 42 * pr;
    const int s0 = sp.pair_slots[2 * pr], s1 = sp.pair_slots[2 * pr + 1];
    if (s0 >= sp.nd_slots) continue;
    double v = 0;
    for (int s = s0; s < s1; ++s) v += sp_lds[(size_t)s * 42 + k];
    sp.partial[((size_t)BX * sp.npairs + pr) * 42 + k] = v;
  }
}

// sum over the ranges, in order: one workgroup per pair; 4 x 64 threads = 4 interleaved range sub-sums x 42 values
__device__ __forceinline__ void ba_schur_reduce_body(int BX, BaSpG sp, double* __restrict__ pair_sum) {
  __shared__ double sh[4][42];
  const int k = threadIdx.x & 63, g = threadIdx.x >> 6;
  double v = 0;
  if (k < 42)
    for (int r = g; r < sp.R; r += 4) v += sp.partial[((size_t)r * sp.npairs + BX) * 42 + k];
  if (k < 42) sh[g][k] = v;
  __syncthreads();
  if (threadIdx.x < 42) pair_sum[(size_t)BX * 42 + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
