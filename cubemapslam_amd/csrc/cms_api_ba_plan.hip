// cms_api_ba_plan.hip -- the plan of a local-BA window with its observation-sized parts ON THE DEVICE (included by cms_api_ba.hip).
//
// What cms_ba_create has to decide for a window (Optimizer.cpp:246-357 assembles the graph; everything here is the product's own work
// list, no counterpart in the reference) used to be ~1.8 ms of one host thread per 80 k-observation window -- 2.5 ms next to 15 other such
// threads, 7 host cores per GPU at 23 k frames/s.  Almost all of it was passes over the E observations: two counting sorts for the
// point CSR, the signature of every point through an indirection, the sorted edge arrays and the per-edge words, the pose CSR, the
// look-ahead composition of the left-over chunks, the runs' operand / flush tables.  Here the host touches the observations ONCE
// (validation + per-point count + per-point key-frame set, a sequential pass) and then works on POINTS only:
//
//   host    signature groups (open addressing on the 64-bit key-frame set) -> runs -> internal point order (runs first, then the left-over
//           points in the caller's order) -> chunks (run chunks; left-over points packed greedily into chunks of <= 64 observations) ->
//           point CSR offsets, chunk descriptors, chunk costs, the left-over observations' diagonal copies.  P-sized, sequential, ~0.2 ms.
//   device  k_ba_expand_edges   one thread per observation: its position in the internal order (rank of its key frame among its point's),
//                               the sorted edge arrays, the per-edge word, and k_ba_gather's work (measurements into the internal order,
//                               initial estimate, cleared per-edge state)
//                               ... and, in the same launch, one thread per (run, lane): the MFMA body's operand table run_mf and flush
//                               table run_fl
// (Which copy of its key frame's diagonal block a LEFT-OVER observation adds to -- a bipartite matching lanes -> LDS banks per group of 16 lanes,
// BaDiagMatchHD -- is decided on the host after all: a point's key frames in ascending order ARE the set bits of its signature, so the matching
// needs no observation-sized data either, and as a device kernel it was a few hundred serial threads that held a hardware queue for 0.26 ms.)
//
// The look-ahead composition of the left-over chunks is gone from this path: round 4 measured it neutral for windows whose left-over points
// are a minority (profiles/r04_experiments.txt: look-ahead 1 / 4 / 8 = 88.7 / 89.6 / 86.9 us for the Schur launch); what matters is the
// matching of the diagonal copies, which stays.  Windows the fast path does not take (no runs or mostly left-over points, more than 64 key
// frames, a point seen twice by a key frame, any A/B knob that selects other kernels, cms_ba_linearize) go through ba_plan as before; for
// the windows both can plan, the two give byte-identical device arrays (tests/test_gpu_parity.py::test_ba_device_plan_equals_host_plan).
#include <stdint.h>

// (BaRunSig / ba_run_decode: cms_ba_schur_runwg.hip)
// run_mf[run * 64 + i] (cms_ba_schur_runs.hip): rows / columns of the signature's stacked matrix, the slots, the count
__host__ __device__ inline uint32_t ba_run_mf_word(const BaRunSig& rs, int i) {
  if (i < 48) {
    const int a = i / 6, rr = i - 6 * a;
    if (a < rs.kf && a < 8) return (uint32_t)(rs.fpos(a) * 18 + 3 * rr);
    if (i == 6 * rs.kf) return BA_RM_MF_RHS;
    return BA_RM_MF_NONE;
  }
  if (i < 56) return (i - 48 < rs.kf) ? (uint32_t)rs.fslot(i - 48) : BA_RM_MF_NONE;
  if (i == 56) return (uint32_t)rs.kf;
  return BA_RM_MF_NONE;
}
// run_fl[(run * 64 + lane) * 12 + w]: where accumulators 2 w and 2 w + 1 of the lane go (16-bit LDS offsets, 0xFFFF: nowhere)
__host__ __device__ inline uint32_t ba_run_fl_word(const BaRunSig& rs, int np, int l, int w) {
  const int n6 = 6 * rs.kf;
  uint32_t word = 0xFFFFFFFFu;
  if (n6 + 1 > 48) return word;
  const uint32_t dg_off = (uint32_t)((((np * (np + 1) / 2) - np) * BA_SE_SSTRIDE + 1) & ~1);
  for (int h = 0; h < 2; ++h) {
    const int idx = 2 * w + h, t = idx >> 2, g = idx & 3;
    const int ti = (0x210100 >> (4 * t)) & 15, tj = (0x222110 >> (4 * t)) & 15;      // tiles (0,0) (0,1) (1,1) (0,2) (1,2) (2,2)
    const int I = 16 * ti + (l >> 4) + 4 * g, N = 16 * tj + (l & 15);
    if (!(I < n6 && N <= n6 && (N == n6 || I <= N))) continue;
    const int a1 = I / 6, r1 = I % 6, s1 = rs.fslot(a1);
    uint32_t off;
    if (N == n6) off = dg_off + (uint32_t)((g * np + s1) * BA_SE_DSTRIDE + 21 + r1);
    else {
      const int a2 = N / 6, r2 = N % 6, s2 = rs.fslot(a2);
      if (a1 == a2) off = dg_off + (uint32_t)((g * np + s1) * BA_SE_DSTRIDE + (r1 * 6 - (r1 * (r1 - 1)) / 2 + (r2 - r1)));
      else off = (uint32_t)((s1 * np - (s1 * (s1 + 1)) / 2 + (s2 - s1 - 1)) * BA_SE_SSTRIDE + ba_se_off(r1, r2));
    }
    word = h ? ((word & 0x0000FFFFu) | (off << 16)) : ((word & 0xFFFF0000u) | (off & 0xFFFFu));
  }
  return word;
}

// ---- what the host hands to the expansion kernels
struct BaExpand {
  int K, P, E, np;
  const int* e_pose; const int* e_point; const int8_t* e_face;      // the caller's arrays
  const int* cedge;                                                 // caller's edge ids grouped by caller point (NULL: the caller's order IS grouped by point)
  const int* cpo;                                                   // P + 1: offsets of the caller's points in that grouped order
  const int* prank; const int* pinv; const int* pt_off;             // caller point -> internal point, back, and the internal CSR offsets
  const uint8_t* pcopy;                                             // P (internal): copy of the diagonal blocks for a run point's observations, 0xFF: left-over point
  const uint8_t* lo_copy; int e_lo0;                                // ... and per left-over observation (sorted position - e_lo0): the copy the host's matching chose
  const int* pose_slot;
  const double* raw_obs; const double* raw_inv; const double* raw_pts; const double* poses0;
  int* perm; int* iperm; int* s_pose; int* s_point; int8_t* s_face; uint32_t* info;      // iperm (may be NULL): caller's edge -> internal position, for the read-back
  double* e_obs; double* e_inv; double* pts0; double* poses; double* pts; uint8_t* level; double* err; uint8_t* flags;
  double* gsum; int n_gsum; double* gsum_bp; int n_gsum_bp;
  // tables
  const int* ce0; int n_rm, nchunks;                                // chunk first edges; chunks [n_rm, nchunks) are the left-over ones
  const uint64_t* run_sig; int n_runs; uint32_t* run_mf; uint32_t* run_fl; uint32_t* run_fg;
  const int* dcounts;                                               // non-NULL: the plan was made by k_ba_plan_many (cms_api_ba_devplan.hip) and n_rm, nchunks, n_runs, e_lo0, "grouped" are there
};

// one observation (position i of the caller's grouped order) / one table entry: the kernels' bodies, callable on the host too (cms_ba_debug_plan_fast
// runs the very same code over host arrays, so the CPU tests see what the device computes)
__host__ __device__ inline void ba_expand_edge_at(const BaExpand& x, int i) {
  const int e = x.cedge ? x.cedge[i] : i;
  const int q = x.e_point[e], k = x.e_pose[e];
  const int base = x.cpo[q], n = x.cpo[q + 1] - base;
  int a = 0;                                                      // rank of this observation among its point's: by key frame, then by position
  for (int j = 0; j < n; ++j) {
    const int ej = x.cedge ? x.cedge[base + j] : base + j;
    const int kj = x.e_pose[ej];
    a += (kj < k || (kj == k && base + j < i)) ? 1 : 0;
  }
  const int p = x.prank[q], pos = x.pt_off[p] + a;
  const int face = x.e_face[e];
  x.perm[pos] = e; x.s_pose[pos] = k; x.s_point[pos] = p; x.s_face[pos] = (int8_t)face;
  if (x.iperm) x.iperm[e] = pos;
  const uint32_t copy = x.pcopy[p] == 0xFF ? (uint32_t)x.lo_copy[pos - x.e_lo0] : (uint32_t)x.pcopy[p];
  x.info[pos] = (uint32_t)a | ((uint32_t)n << 5) | ((uint32_t)(x.pose_slot[k] + 1) << 10) | ((uint32_t)face << 16) | ((uint32_t)k << 19) | (copy << 27);
  if (x.e_obs) {
    x.e_obs[2 * (size_t)pos] = x.raw_obs[2 * (size_t)e]; x.e_obs[2 * (size_t)pos + 1] = x.raw_obs[2 * (size_t)e + 1];
    x.e_inv[pos] = x.raw_inv[e];
    x.err[2 * (size_t)pos] = 0.0; x.err[2 * (size_t)pos + 1] = 0.0;
    x.level[pos] = 0; x.flags[pos] = 0;
  }
}
// one (run, lane): the run's tables
__host__ __device__ inline void ba_expand_table_at(const BaExpand& x, int u) {
  const int r = u >> 6, l = u & 63;
  BaRunSig rs;
  ba_run_decode(x.run_sig[r], x.pose_slot, rs);
  x.run_mf[(size_t)r * 64 + l] = ba_run_mf_word(rs, l);
  for (int w = 0; w < 12; ++w) x.run_fl[((size_t)r * 64 + l) * 12 + w] = ba_run_fl_word(rs, x.np, l, w);
  if (x.run_fg)
    for (int i = 0; i < 24; ++i) x.run_fg[((size_t)r * 64 + l) * 24 + i] = ba_run_fg_word(rs, x.np, l, i);
}

// the windows of ONE cms_ba_create_many call expanded by one launch: blockIdx.y = window (a window group's sixteen set-ups were sixteen launches of
// ~19 us each that waited up to 1.8 ms for a queue slot behind the Levenberg rounds inside bench.py's step).  Eight descriptions fit the 4 KB of
// kernel arguments; a larger group is two launches
#define BA_EXPAND_BATCH 8
struct BaExpandBatch { BaExpand x[BA_EXPAND_BATCH]; };
static_assert(sizeof(BaExpandBatch) <= 4096, "k_ba_expand_edges_many: the batch must fit the kernel arguments");
__device__ __forceinline__ void ba_expand_body(const BaExpand& x, int t0, int gs);
extern "C" __global__ void __launch_bounds__(256) k_ba_expand_edges_many(BaExpandBatch b) {
  ba_expand_body(b.x[blockIdx.y], blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}
extern "C" __global__ void __launch_bounds__(256) k_ba_expand_edges(BaExpand x) {
  ba_expand_body(x, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}
__device__ __forceinline__ void ba_expand_body(const BaExpand& xin, int t0, int gs) {
  BaExpand x = xin;
  if (x.dcounts) {                                                  // (BA_DP_COUNTS: status, chunks, run chunks, class-0 run chunks, runs, points inside runs, lone points, grouped, first left-over edge)
    if (x.dcounts[0] != 1) return;                                  // the plan kernel gave the window up: the host plans it again
    x.nchunks = x.dcounts[1]; x.n_rm = x.dcounts[2]; x.n_runs = x.dcounts[4]; x.e_lo0 = x.dcounts[8];
    if (x.dcounts[7]) x.cedge = nullptr;
  }
  for (int i = t0; i < x.E; i += gs) ba_expand_edge_at(x, i);
  for (int u = t0; u < 64 * x.n_runs; u += gs) ba_expand_table_at(x, u);
  for (int i = t0; i < x.P; i += gs) {
    const int q = x.pinv[i];
#pragma unroll
    for (int j = 0; j < 3; ++j) { const double v = x.raw_pts[3 * (size_t)q + j]; x.pts0[3 * (size_t)i + j] = v; x.pts[3 * (size_t)i + j] = v; }
  }
  for (int i = t0; i < 7 * x.K; i += gs) x.poses[i] = x.poses0[i];
  for (int i = t0; i < x.n_gsum; i += gs) x.gsum[i] = 0.0;
  for (int i = t0; i < x.n_gsum_bp; i += gs) x.gsum_bp[i] = 0.0;
}
// cms_ba_read of a window whose permutations only the device holds: one kernel GATHERS poses, points and outlier flags in the caller's order and stores
// them straight into the window's pinned host block (coalesced stores over PCIe) -- no copy commands at all.  (A first version un-permuted into device
// buffers and copied those: three copies behind a kernel on the same stream, which the runtime then does with blit kernels, ~100 more launches per bench
// step at ~70 us each inside the step.)
extern "C" __global__ void __launch_bounds__(256)
k_ba_results_to_host(int K, int P, int E, const int* __restrict__ prank, const int* __restrict__ iperm, const double* __restrict__ poses, const double* __restrict__ pts,
                     const uint8_t* __restrict__ flags, double* __restrict__ out_poses, double* __restrict__ out_pts, uint8_t* __restrict__ out_flags) {
  const int gs = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  if (out_poses)
    for (int i = t0; i < 7 * K; i += gs) out_poses[i] = poses[i];
  if (out_pts)
    for (int i = t0; i < 3 * P; i += gs) { const int q = i / 3, j = i - 3 * q; out_pts[i] = pts[3 * (size_t)prank[q] + j]; }
  if (out_flags)
    for (int i4 = t0; 4 * i4 < E; i4 += gs) {                      // four flags per thread: dword stores
      uint32_t w = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int e = 4 * i4 + j; if (e < E) w |= (uint32_t)flags[iperm[e]] << (8 * j); }
      reinterpret_cast<uint32_t*>(out_flags)[i4] = w;
    }
}

// ... and the read-back of several windows in one launch (cms_ba_read_many): blockIdx.y = window
struct BaReadJob { int K, P, E; const int* prank; const int* iperm; const double* poses; const double* pts; const uint8_t* flags; double* out_poses; double* out_pts; uint8_t* out_flags; };
#define BA_READ_BATCH 16
struct BaReadBatch { BaReadJob j[BA_READ_BATCH]; };
extern "C" __global__ void __launch_bounds__(256) k_ba_results_to_host_many(BaReadBatch b) {
  const BaReadJob& q = b.j[blockIdx.y];
  const int gs = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  if (q.out_poses)
    for (int i = t0; i < 7 * q.K; i += gs) q.out_poses[i] = q.poses[i];
  if (q.out_pts)
    for (int i = t0; i < 3 * q.P; i += gs) { const int p = i / 3, j = i - 3 * p; q.out_pts[i] = q.pts[3 * (size_t)q.prank[p] + j]; }
  if (q.out_flags)
    for (int i4 = t0; 4 * i4 < q.E; i4 += gs) {
      uint32_t w = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int e = 4 * i4 + j; if (e < q.E) w |= (uint32_t)q.flags[q.iperm[e]] << (8 * j); }
      reinterpret_cast<uint32_t*>(q.out_flags)[i4] = w;
    }
}

// ---- host side: everything of the plan that is decided per POINT
struct BaFastPlan {
  std::vector<int> pose_slot, cnt, cpo, cedge, prank, pinv, pt_off, chunk_pt0, ce0, lone, pob, ident, pose_cnt;
  std::vector<uint8_t> pcopy, lo_copy;
  std::vector<uint64_t> sig, run_sig;
  std::vector<int4> rm_chunk;
  std::vector<uint32_t> rm_cost;
  std::vector<int> rm_cut;
  bool grouped = true;
  int n_rm = 0, n_rmA = 0, n_runs = 0, P_rm = 0, nchunks = 0;
};

// the knobs under which the fast plan's windows run exactly the kernels it prepares for (anything else: ba_plan)
static bool ba_fast_plan_allowed(const BaKnobs& kn) {
  static const bool host_plan = getenv("CMS_BA_HOST_PLAN") != nullptr;      // A/B: the host plan for every window
  return !host_plan && !kn.want_all_lists && !kn.no_fused && !kn.solve1 && !kn.trial_points && !kn.no_permute && kn.runs && !kn.rm_valu &&
         !kn.separate_first_pass && !kn.host_lm && !kn.single_host_lm && kn.compose_segments == 0 && kn.leftover_lookahead <= 1 && kn.run_min_chunks == 1 &&
         kn.run_min_pct == 100 && BA_SE_THREADS == 128 * BA_RM_PAIRS;
}

// returns 1: planned; 0: not a window for this path (the caller runs ba_plan); < 0: error (bad index)
template <class Tick>
static int ba_plan_fast(cms_ba* b, BaFastPlan& fp, int K, const uint8_t* fixed, int P, int E, const int* e_pose, const int* e_point, const int8_t* e_face, Tick&& tick) {
  const BaKnobs& kn = ba_knobs();
  if (K > 64 || b->det_points || !ba_fast_plan_allowed(kn)) return 0;
  // ---- the one pass over the observations: validation, observations per point, key-frame set per point, observations per key frame
  std::vector<int>&cnt = fp.cnt, &cpo = fp.cpo; std::vector<uint64_t>& sig = fp.sig;
  cnt.assign(P, 0); sig.assign(P, 0); fp.pose_cnt.assign(K, 0);
  bool grouped = true;                                              // every point's observations are consecutive in the caller's order, points ascending
  {
    int prev = -1;
    bool bad = false;
    for (int e = 0; e < E; ++e) {
      const int k = e_pose[e], p = e_point[e];
      if ((unsigned)k >= (unsigned)K || (unsigned)p >= (unsigned)P || (unsigned)e_face[e] > 4u) { bad = true; break; }
      ++cnt[p]; sig[p] |= 1ull << k; ++fp.pose_cnt[k];
      grouped = grouped && p >= prev; prev = p;
    }
    if (bad) return -1;
  }
  fp.grouped = grouped;
  std::vector<int>& pose_slot = fp.pose_slot;
  pose_slot.assign(K, -1);
  int np = 0;
  uint64_t free_mask = 0;
  for (int k = 0; k < K; ++k) if (!fixed[k]) { pose_slot[k] = np++; free_mask |= 1ull << k; }
  ba_plan_sizes(b, K, P, E, np);
  const int NP2 = np * (np + 1) / 2;
  const size_t se_fixed_lds = ba_se_fixed_lds(K, np), se_wave_lds = (size_t)64 * 18 * sizeof(double) + 64 * sizeof(int);
  int se_nw = BA_SE_THREADS / 64;
  while (se_nw > 2 && se_fixed_lds + se_nw * se_wave_lds > BA_LDS_CEILING) se_nw -= 2;
  const size_t rm_lds = se_fixed_lds + (size_t)BA_RM_PAIRS * 2 * BA_RM_BUF * sizeof(double);
  if (!(np >= 1 && se_fixed_lds + se_nw * se_wave_lds <= BA_LDS_CEILING && np <= 62 && b->solve_blk && b->solve_blk3 && rm_lds <= BA_LDS_CEILING)) return 0;
  for (int p = 0; p < P; ++p) if (cnt[p] > 31 || __builtin_popcountll(sig[p]) != cnt[p]) return 0;      // (a point seen twice by a key frame: the pair-owner kernel's case)
  cpo.resize(P + 1);
  cpo[0] = 0;
  for (int p = 0; p < P; ++p) cpo[p + 1] = cpo[p] + cnt[p];
  if (!grouped) {                                                   // the caller's edges grouped by point (stable): the device ranks them by key frame
    fp.cedge.resize(E);
    BA_TLV(int, fill); fill.assign(cpo.begin(), cpo.end() - 1);
    for (int e = 0; e < E; ++e) fp.cedge[fill[e_point[e]]++] = e;
  } else fp.cedge.clear();
  tick("pass");
  // ---- signature groups in order of first appearance (open addressing on the set itself), groups with enough points become runs
  struct Run { int k, first, npts, chunks, m, kf; };
  std::vector<Run> runs;
  BA_TLV(int, gid); gid.assign(P, -1);
  std::vector<int> gcount, gfirst;
  {
    int cap = 1024;
    while (cap < 4 * 1024 && cap < 2 * P) cap <<= 1;
    BA_TLV(uint64_t, hkey); BA_TLV(int, hval);
    hkey.assign(cap, 0); hval.assign(cap, -1);
    auto rehash = [&](int ncap) {
      std::vector<uint64_t> ok(hkey.begin(), hkey.end()); std::vector<int> ov(hval.begin(), hval.end());
      hkey.assign(ncap, 0); hval.assign(ncap, -1);
      for (size_t i = 0; i < ok.size(); ++i)
        if (ov[i] >= 0) { size_t j = (size_t)((ok[i] * 0x9E3779B97F4A7C15ull) >> 40) & (ncap - 1); while (hval[j] >= 0) j = (j + 1) & (ncap - 1); hkey[j] = ok[i]; hval[j] = ov[i]; }
      cap = ncap;
    };
    for (int p = 0; p < P; ++p) {
      const uint64_t key = sig[p];
      size_t j = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & (cap - 1);
      while (hval[j] >= 0 && hkey[j] != key) j = (j + 1) & (cap - 1);
      int g = hval[j];
      if (g < 0) {
        g = (int)gfirst.size(); gfirst.push_back(p); gcount.push_back(0); hkey[j] = key; hval[j] = g;
        if (2 * (int)gfirst.size() > cap) rehash(2 * cap);
      }
      gid[p] = g; ++gcount[g];
    }
  }
  const int ng = (int)gfirst.size();
  std::vector<int> run_of_group(ng, -1), take(ng, 0);
  for (int g = 0; g < ng; ++g) {
    const int q = gfirst[g], k = cnt[q];
    const int kf = __builtin_popcountll(sig[q] & free_mask);
    if (k < 1 || k > 9 || kf < 1 || kf * (kf + 1) / 2 > 64 || 6 * kf + 1 > 48) continue;
    const int m = std::min(64 / k, BA_RM_PTS);
    if (gcount[g] < m) continue;                                    // at least one full chunk (run_min_chunks 1, run_min_pct 100: the defaults this path requires)
    const int full = gcount[g] / m, tail = gcount[g] - full * m;
    const bool keep_tail = tail > 0 && 2 * tail >= m;
    run_of_group[g] = (int)runs.size();
    take[g] = full * m + (keep_tail ? tail : 0);
    runs.push_back({k, q, take[g], full + (keep_tail ? 1 : 0), m, kf});
  }
  ba_rw_order_runs(runs, run_of_group, [](const Run& r) { return r.kf; });      // signatures with two tile rows first (cms_ba_schur_runwg.hip)
  std::vector<int> run_pt0(runs.size() + 1, 0);
  for (size_t r = 0; r < runs.size(); ++r) run_pt0[r + 1] = run_pt0[r] + runs[r].npts;
  const int P_rm = run_pt0.back(), PL = P - P_rm;
  if (P_rm == 0 || 3 * PL > P) return 0;                            // no runs, or mostly left-over points: the look-ahead composition pays there (ba_plan)
  // ---- internal point order: the runs' points (run after run, caller's order inside a run), then the left-over points in the caller's order
  std::vector<int>&prank = fp.prank, &pinv = fp.pinv;
  prank.resize(P); pinv.resize(P);
  {
    std::vector<int> fill(run_pt0.begin(), run_pt0.end() - 1), seen(ng, 0);
    int nl = P_rm;
    for (int p = 0; p < P; ++p) {
      const int g = gid[p], r = run_of_group[g];
      int ip;
      if (r >= 0 && seen[g] < take[g]) { ip = fill[r]++; ++seen[g]; }
      else ip = nl++;
      prank[p] = ip; pinv[ip] = p;
    }
  }
  tick("runs");
  // ---- chunks: the runs' chunks (whole points of one signature), then the left-over points packed into chunks of <= 64 observations.  (The
  // left-over list is cut into the segments ba_compose_chunks uses -- a segment starts a chunk -- so that both planners give the same chunks.)
  std::vector<int>& chunk_pt0 = fp.chunk_pt0;
  chunk_pt0.clear();
  std::vector<int> rm_chunk_run;
  for (size_t r = 0; r < runs.size(); ++r)
    for (int c = 0; c < runs[r].chunks; ++c) { chunk_pt0.push_back(run_pt0[r] + c * runs[r].m); rm_chunk_run.push_back((int)r); }
  const int n_rm = (int)rm_chunk_run.size();
  int n_rmA = 0;
  while (n_rmA < n_rm && ba_rw_class(runs[rm_chunk_run[n_rmA]].kf) == 0) ++n_rmA;
  {
    const int nseg = std::max(1, std::min(8, PL / 512));
    for (int t = 0; t < nseg; ++t) {
      const int pb = (int)((long long)PL * t / nseg), pe = (int)((long long)PL * (t + 1) / nseg);
      if (pe <= pb) continue;
      chunk_pt0.push_back(P_rm + pb);
      int cur = 0;
      for (int i = pb; i < pe; ++i) {
        const int k = cnt[pinv[P_rm + i]];
        if (cur + k > 64 && cur > 0) { chunk_pt0.push_back(P_rm + i); cur = 0; }
        cur += k;
      }
    }
  }
  chunk_pt0.push_back(P);
  const int nchunks = (int)chunk_pt0.size() - 1;
  // ---- CSR offsets of the internal points, first edges of the chunks, the copies of the run points, points nobody observes
  std::vector<int>& pt_off = fp.pt_off;
  pt_off.resize(P + 1);
  pt_off[0] = 0;
  for (int p = 0; p < P; ++p) pt_off[p + 1] = pt_off[p] + cnt[pinv[p]];
  fp.ce0.resize(nchunks + 1);
  for (int c = 0; c <= nchunks; ++c) fp.ce0[c] = pt_off[chunk_pt0[c]];
  for (int c = 0; c < nchunks; ++c) if (fp.ce0[c + 1] - fp.ce0[c] > 64) return 0;      // (cannot happen with <= 31 observations per point)
  fp.pcopy.assign(P, 0xFF);
  fp.rm_chunk.resize(std::max(n_rm, 1));
  fp.rm_cost.assign((size_t)nchunks + 1, 0u);
  for (int c = 0; c < n_rm; ++c) {
    const Run& R = runs[rm_chunk_run[c]];
    const int p0 = chunk_pt0[c], p1 = std::min(chunk_pt0[c + 1], run_pt0[rm_chunk_run[c] + 1]);
    for (int p = p0; p < p1; ++p) fp.pcopy[p] = (uint8_t)((p - p0) & (BA_SE_DCOPIES - 1));
    fp.rm_chunk[c] = make_int4(pt_off[p0], (pt_off[p1] - pt_off[p0]) | (R.k << 8) | ((p1 - p0) << 16), rm_chunk_run[c], p0);
    fp.rm_cost[(size_t)c + 1] = fp.rm_cost[c] + ba_rm_chunk_cost(R.k, R.kf, p1 - p0);
  }
  for (int c = n_rm; c < nchunks; ++c) {
    int kmax = 1;
    for (int p = chunk_pt0[c]; p < chunk_pt0[c + 1]; ++p) kmax = std::max(kmax, cnt[pinv[p]]);
    fp.rm_cost[(size_t)c + 1] = fp.rm_cost[c] + (uint32_t)(kn.em_cost_a + kn.em_cost_b * (kmax / 2));
  }
  // ---- the left-over observations' copies of their key frames' diagonal blocks: per chunk and group of 16 lanes a matching lanes -> LDS banks
  // (every lane may use any of the BA_SE_DCOPIES copies = banks).  A point's observations in ascending key-frame order are the set bits of
  // its signature, so the lanes' slots follow from per-point data alone.
  {
    const int e_lo0 = pt_off[P_rm];
    fp.lo_copy.assign((size_t)std::max(E - e_lo0, 1), 0);
    for (int c = n_rm; c < nchunks; ++c) {
      BaDiagMatch M[4];                                             // (the host planner's matcher: same lanes in the same order, same choices)
      uint8_t lane_of[4][16];
      for (int g = 0; g < 4; ++g) M[g].n = 0;
      const int e0 = fp.ce0[c];
      int L = 0;
      for (int p = chunk_pt0[c]; p < chunk_pt0[c + 1]; ++p) {
        uint64_t bits = sig[pinv[p]];
        while (bits) {
          const int k = __builtin_ctzll(bits);
          bits &= bits - 1;
          const int sl = pose_slot[k], g = (L >> 4) & 3;
          if (sl >= 0) {
            BaDiagMatch& m = M[g];
            for (int r = 0; r < BA_SE_DCOPIES; ++r) m.bank[m.n][r] = (uint8_t)((BA_SE_DSTRIDE * (r * np + sl)) & 15);
            lane_of[g][m.n++] = (uint8_t)(L & 15);
          }
          ++L;
        }
      }
      for (int g = 0; g < 4; ++g) {
        if (M[g].n == 0) continue;
        M[g].run();
        for (int i = 0; i < M[g].n; ++i) fp.lo_copy[(size_t)(e0 - e_lo0) + 16 * g + lane_of[g][i]] = (uint8_t)M[g].choice[i];
      }
    }
  }
  tick("copies");
  fp.lone.clear();
  for (int p = 0; p < P; ++p) if (pt_off[p + 1] == pt_off[p]) fp.lone.push_back(p);
  fp.pob.assign((size_t)NP2, 0); fp.ident.resize((size_t)NP2 + 1);
  for (int I = 0; I < np; ++I)
    for (int Kc = 0; Kc <= I; ++Kc) fp.pob[(size_t)I * (I + 1) / 2 + Kc] = Kc * np - (Kc * (Kc - 1)) / 2 + (I - Kc);
  for (int i = 0; i <= NP2; ++i) fp.ident[i] = i;
  fp.run_sig.resize(std::max<size_t>(runs.size(), 1), 0);
  for (size_t r = 0; r < runs.size(); ++r) fp.run_sig[r] = sig[runs[r].first];
  fp.n_rm = n_rm; fp.n_rmA = n_rmA; fp.n_runs = (int)runs.size(); fp.P_rm = P_rm; fp.nchunks = nchunks;
  fp.rm_cut.clear();
  if (ba_want_rw_tables()) {
    fp.rm_cut.assign(2 * (BA_RW_CUTS + 1), 0);
    ba_rw_make_cuts(fp.rm_cost, 0, n_rmA, fp.rm_cut.data());
    ba_rw_make_cuts(fp.rm_cost, n_rmA, n_rm, fp.rm_cut.data() + (BA_RW_CUTS + 1));
  }
  BaSe& se = b->se;
  se.nchunks = nchunks; se.n_rm = n_rm; se.n_rmA = n_rmA; se.npairs2 = NP2;
  se.cpw_t = (BA_TE_THREADS / 64) * kn.te_chunks;
  se.Rt = (nchunks + se.cpw_t - 1) / se.cpw_t;
  se.nlone = (int)fp.lone.size();
  ba_se_split(se, BA_SE_RANGES);
  if (fp.lone.empty()) fp.lone.push_back(0);
  b->se_lds_fixed = se_fixed_lds; b->se_waves = se_nw; b->rm_lds = rm_lds;
  b->n_runs = fp.n_runs; b->rm_points = P_rm;
  tick("chunks");
  return 1;
}

// developer / test entry, host only (no device needed): what the device-side planner gives for a window -- ba_plan_fast on the host, then the
// expansion kernels' bodies run over host arrays.  Same outputs and sizes as cms_ba_debug_plan (run_lane is not produced: the vector
// variant's table); returns CMS_OK with counts[7] = 0 when the window is not one this planner takes.
extern "C" int cms_ba_debug_plan_fast(int K, const uint8_t* fixed, int P, int E, const int* e_pose, const int* e_point, int* pinv_out, int* perm_out,
                                      uint32_t* info_out, int* chunk_pt0_out, int* rm_chunk_out, int* counts, uint32_t* run_mf_out, uint32_t* run_fl_out) {
  if (K < 1 || P < 1 || E < 1 || !fixed || !e_pose || !e_point || !pinv_out || !perm_out || !info_out || !chunk_pt0_out || !rm_chunk_out || !counts)
    return cms_fail(CMS_ERR_ARG, "cms_ba_debug_plan_fast: bad argument");
  cms_ba* b = new cms_ba;
  b->K = K; b->P = P; b->E = E;
  BaFastPlan fp;
  std::vector<int8_t> face(E, 0);
  const bool timing = ba_knobs().create_timing;
  auto t_last = std::chrono::steady_clock::now();
  const int rc = ba_plan_fast(b, fp, K, fixed, P, E, e_pose, e_point, face.data(), [&](const char* what) {
    const auto now = std::chrono::steady_clock::now();
    if (timing) fprintf(stderr, "[cms_ba_debug_plan_fast] %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  });
  for (int i = 0; i < 8; ++i) counts[i] = 0;
  if (rc < 0) { delete b; return cms_fail(CMS_ERR_ARG, "cms_ba_debug_plan_fast: index out of range"); }
  if (rc == 0) { delete b; return CMS_OK; }
  std::vector<int> s_pose(E), s_point(E); std::vector<int8_t> s_face(E);
  std::vector<uint32_t> run_mf((size_t)std::max(fp.n_runs, 1) * 64), run_fl((size_t)std::max(fp.n_runs, 1) * 64 * 12);
  BaExpand x;
  memset(&x, 0, sizeof(x));
  x.K = K; x.P = P; x.E = E; x.np = b->np;
  x.e_pose = e_pose; x.e_point = e_point; x.e_face = face.data(); x.cedge = fp.grouped ? nullptr : fp.cedge.data(); x.cpo = fp.cpo.data();
  x.prank = fp.prank.data(); x.pinv = fp.pinv.data(); x.pt_off = fp.pt_off.data(); x.pcopy = fp.pcopy.data(); x.pose_slot = fp.pose_slot.data();
  x.perm = perm_out; x.iperm = nullptr; x.s_pose = s_pose.data(); x.s_point = s_point.data(); x.s_face = s_face.data(); x.info = info_out;
  x.ce0 = fp.ce0.data(); x.n_rm = fp.n_rm; x.nchunks = fp.nchunks; x.run_sig = fp.run_sig.data(); x.n_runs = fp.n_runs; x.run_mf = run_mf.data(); x.run_fl = run_fl.data();
  x.lo_copy = fp.lo_copy.data(); x.e_lo0 = fp.pt_off[fp.P_rm];
  for (int i = 0; i < E; ++i) ba_expand_edge_at(x, i);
  for (int u = 0; u < 64 * fp.n_runs; ++u) ba_expand_table_at(x, u);
  memcpy(pinv_out, fp.pinv.data(), (size_t)P * sizeof(int));
  memcpy(chunk_pt0_out, fp.chunk_pt0.data(), fp.chunk_pt0.size() * sizeof(int));
  if (fp.n_rm > 0) memcpy(rm_chunk_out, fp.rm_chunk.data(), (size_t)fp.n_rm * sizeof(int4));
  if (run_mf_out && fp.n_runs > 0) memcpy(run_mf_out, run_mf.data(), (size_t)fp.n_runs * 64 * sizeof(uint32_t));
  if (run_fl_out && fp.n_runs > 0) memcpy(run_fl_out, run_fl.data(), (size_t)fp.n_runs * 64 * 12 * sizeof(uint32_t));
  counts[0] = fp.nchunks; counts[1] = fp.n_rm; counts[2] = fp.n_runs; counts[3] = b->np; counts[4] = fp.P_rm; counts[5] = b->se.R_rm; counts[6] = b->se.R; counts[7] = 1;
  delete b;
  return CMS_OK;
}

// developer / test entry, host only: the tables of the one-wavefront run workgroups (cms_ba_schur_runwg.hip) as either planner makes them.
// run_fg: runs x 64 x 24 words, rm_cut: 2 x 1025, counts[4] = runs, run chunks, class-0 run chunks, free key frames; fast != 0: the device-side
// planner's host part + the expansion kernel's body on the host (counts[0] = -1 when it does not take the window)
extern "C" int cms_ba_debug_run_fg(int K, const uint8_t* fixed, int P, int E, const int* e_pose, const int* e_point, int fast, uint32_t* run_fg_out, int* rm_cut_out, int* counts) {
  if (K < 1 || P < 1 || E < 1 || !fixed || !e_pose || !e_point || !run_fg_out || !rm_cut_out || !counts) return cms_fail(CMS_ERR_ARG, "cms_ba_debug_run_fg: bad argument");
  for (int e = 0; e < E; ++e)
    if (e_pose[e] < 0 || e_pose[e] >= K || e_point[e] < 0 || e_point[e] >= P) return cms_fail(CMS_ERR_ARG, "cms_ba_debug_run_fg: index out of range");
  struct Force { Force() { ba_force_rw_tables = true; } ~Force() { ba_force_rw_tables = false; } } force;
  cms_ba* b = new cms_ba;
  b->K = K; b->P = P; b->E = E;
  for (int i = 0; i < 4; ++i) counts[i] = 0;
  if (fast) {
    BaFastPlan fp;
    std::vector<int8_t> face(E, 0);
    const int rc = ba_plan_fast(b, fp, K, fixed, P, E, e_pose, e_point, face.data(), [](const char*) {});
    if (rc <= 0) { delete b; counts[0] = -1; return rc < 0 ? cms_fail(CMS_ERR_ARG, "cms_ba_debug_run_fg: index out of range") : CMS_OK; }
    for (int r = 0; r < fp.n_runs; ++r) {
      BaRunSig rs;
      ba_run_decode(fp.run_sig[r], fp.pose_slot.data(), rs);
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 24; ++i) run_fg_out[((size_t)r * 64 + l) * 24 + i] = ba_run_fg_word(rs, b->np, l, i);
    }
    memcpy(rm_cut_out, fp.rm_cut.data(), fp.rm_cut.size() * sizeof(int));
    counts[0] = fp.n_runs; counts[1] = fp.n_rm; counts[2] = fp.n_rmA; counts[3] = b->np;
  } else {
    BaPlan pl;
    ba_plan(b, pl, K, fixed, P, E, e_pose, e_point, nullptr, nullptr, nullptr, [](const char*) {});
    if (!pl.se_built) { delete b; counts[0] = -1; return CMS_OK; }
    if (b->n_runs > 0) memcpy(run_fg_out, pl.run_fg.data(), (size_t)b->n_runs * 64 * 24 * sizeof(uint32_t));
    if (!pl.rm_cut.empty()) memcpy(rm_cut_out, pl.rm_cut.data(), pl.rm_cut.size() * sizeof(int));
    counts[0] = b->n_runs; counts[1] = b->se.n_rm; counts[2] = pl.n_rmA; counts[3] = b->np;
  }
  delete b;
  return CMS_OK;
}
