// cms_ba_schur_runwg.hip -- the run-major Schur body of cms_ba_schur_runs.hip re-decomposed: ONE WAVEFRONT PER WORKGROUP, no workgroup copy of the
// reduced system, sums flushed straight to the window's ONE global copy (BaSe::gsum, global_atomic_add_f64).  OPT-IN (CMS_BA_RUN_WG=1): parity-green
// and slower than the monolith -- see "What it measured" below; the default stays kb_ba_lin_schur_runs.
//
// Why it was built (round 6; round 5's verdict).  kb_ba_lin_schur_runs is a monolith: one kernel holds the run body, the key frame's 27 own sums AND
// the edge-major chunk loop of the left-over points (256 VGPRs, 48 B of scratch), and every workgroup owns a copy of the whole reduced system in LDS
// (158 KB at 19 free key frames): one workgroup per CU, two wavefronts per SIMD, and 27 % of a wavefront's life is s_waitcnt there
// (profiles/r05_pmc_instruction_mix.json).  A run touches only the kf (kf + 1) / 2 pose pairs of its signature, though, and its sums live in MFMA
// accumulators for as long as the run lasts: the LDS copy only exists so that eight wavefronts and the left-over chunks can share one write-out.  Here
//
//   * a workgroup is one wavefront with 12.6 KB of LDS (its chunk buffer: 64 rows of 18 doubles + 32 point slots; the key frames' rotations):
//     twelve of them fit a CU, so the launch runs at three wavefronts per SIMD (the register budget is set for that: amdgpu_waves_per_eu(3, 3));
//   * a unit of work = a range of a window's run chunks of equal estimated cost (rm_cut: 1024 precomputed cut points per class); when a run
//     ends inside the range (or the range does) the wavefront adds its accumulators to the global copy: ~350 global additions for the tiles
//     (run_fg: per lane and accumulator the offset into BaSe::partial) and 33 per key frame of the signature for the key frames' own blocks;
//   * the key frames' own blocks (27 sums per lane in the monolith = 54 of its registers; with them the run body alone wants ~260 registers, with or
//     without the edge-major loop next to it) are added up ACROSS the lanes of an observation position every chunk, through the chunk buffer, by one
//     lane per (position, sum): four doubles per lane instead of 27, ~170 vector instructions per chunk more;
//   * nothing but runs: the left-over chunks go through kb_ba_lin_schur_edges (cms_ba_schur_edges.hip), launched behind this kernel;
//   * two instantiations: CLS 0 takes the signatures whose stacked matrix fits two tile rows (6 kf + 1 <= 32: three resident upper tiles),
//     CLS 1 the ones with six or seven free key frames (six resident tiles, two wavefronts per SIMD).  The planner orders the runs by class
//     (class 0 first), so that each launch walks a contiguous range of chunks.
//
// What it measured (MI355X, 16 tracked configs[3] windows per launch, profiles/r06_runwg_experiment.txt): the three launches take 73 + 59 + 29 us
// against the monolith's 90; in bench.py's step 18.1-20.0 k frames/s against 20.0-21.5 k.  Cycle stamps of a unit (k = 4, kf = 4 chunks, three
// wavefronts per SIMD): 16.8 k cycles per chunk and wavefront = 5.6 k per chunk and SIMD, against ~6.1 k ISSUE cycles of the chunk's instructions
// (~900 vector instructions x 4 + 36 matrix instructions x 68; FP64 matrix and vector instructions share one pipe on gfx950): at three wavefronts
// per SIMD the kernel is bound by FP64 ISSUE, and the monolith's loop already runs at ~two thirds of that bound with fewer instructions (63 FMAs
// for the own sums instead of ~250 through the buffer).  Occupancy bought +45 % per SIMD (1 -> 3 wavefronts: 105 -> 73 us), not the 1 / 0.59
// the wait-count share suggested, and the extra instructions, the per-unit start (cut points -> descriptors -> operands: three dependent memory
// round trips, 6 k cycles) and the balance of ~5-chunk units (durations 27 .. 69 us around a mean of 48: the chunk-cost model was fitted to the
// monolith) cost more than that.  What would actually move this kernel is fewer FP64 issue cycles per observation, not more wavefronts.
//
// The arithmetic of a chunk -- residual with the reference's float round trip, Huber weight, Jacobians, the point's 3x3 factorisation, W = B L^-T,
// G += Y D^-1 Y^T on v_mfma_f64_16x16x4_f64 -- is the monolith's, statement for statement (ba_schur_runs_mfma_body): the same sums in a different
// grouping, so everything behind it (kb_ba_trial_solve3r, kb_ba_trial_edges) is unchanged.  Diagonal pose pairs receive the UPPER triangle only
// (the solve kernel mirrors it when it reads: ba_trial_solve3_body).
// (block_solver.hpp:367-437: Hschur -= Bi Dinv Bj^T, bschur -= Bi Dinv bl; base_binary_edge.hpp:54-120.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#define BA_RW_CUTS 1024                        /* cut points per class: unit u of U takes the chunks [cut[u 1024 / U], cut[(u + 1) 1024 / U]) */
#define BA_RW_NONE 0xFFFFFFFFu
// LDS of one workgroup (= one wavefront): point slots | rows | rotation and translation of every key frame
static inline size_t ba_rw_lds(int K) { return ((size_t)BA_RM_BUF + (size_t)K * 12) * sizeof(double); }
// class of a signature with kf free key frames: 0 -> two tile rows, 1 -> three
__host__ __device__ constexpr int ba_rw_class(int kf) { return 6 * kf + 1 <= 32 ? 0 : 1; }

// the free key frames of a signature (a 64-bit set of key frames), in ascending key-frame order: position of the observation within its point
// (points' observations are sorted by key frame) and free-pose slot; at most eight are kept (signatures of runs have <= 7).  Packed (8 bits
// each): indexable arrays would live in scratch memory on the device.
struct BaRunSig {
  int kf; unsigned long long fpos8, fslot8;
  __host__ __device__ int fpos(int a) const { return (int)((fpos8 >> (8 * a)) & 0xFFu); }
  __host__ __device__ int fslot(int a) const { return (int)((fslot8 >> (8 * a)) & 0xFFu); }
  __host__ __device__ void push(int pos, int slot) {
    if (kf < 8) { fpos8 |= (unsigned long long)pos << (8 * kf); fslot8 |= (unsigned long long)slot << (8 * kf); }
    ++kf;
  }
};
__host__ __device__ inline void ba_run_decode(uint64_t sig, const int* pose_slot, BaRunSig& rs) {
  rs.kf = 0; rs.fpos8 = 0; rs.fslot8 = 0;
  int pos = 0;
  while (sig) {
    const int k = __builtin_ctzll(sig);
    sig &= sig - 1;
    const int s = pose_slot[k];
    if (s >= 0) rs.push(pos, s);
    ++pos;
  }
}
// run_fg[(run * 64 + lane) * 24 + 4 t + g]: where accumulator g of tile t -- (0,0) (0,1) (1,1) (0,2) (1,2) (2,2) -- of lane l goes: the offset (doubles)
// into BaSe::partial of element (r1, r2) of the pose pair (s1 <= s2), dense pair enumeration, 42 doubles per pair: the 6x6 block row major, then the
// pair's right-hand side (diagonal pairs).  Diagonal pairs get their UPPER triangle only.  Lower triangle of G, padding: BA_RW_NONE.
__host__ __device__ inline uint32_t ba_run_fg_word(const BaRunSig& rs, int np, int l, int idx) {
  const int n6 = 6 * rs.kf;
  if (n6 + 1 > 48) return BA_RW_NONE;
  const int t = idx >> 2, g = idx & 3;
  const int ti = (0x210100 >> (4 * t)) & 15, tj = (0x222110 >> (4 * t)) & 15;
  const int I = 16 * ti + (l >> 4) + 4 * g, N = 16 * tj + (l & 15);
  if (!(I < n6 && N <= n6 && (N == n6 || I <= N))) return BA_RW_NONE;
  const int a1 = I / 6, r1 = I % 6, s1 = rs.fslot(a1);
  if (N == n6) return (uint32_t)((s1 * np - ((s1 * (s1 - 1)) >> 1)) * 42 + 36 + r1);
  const int a2 = N / 6, r2 = N % 6, s2 = rs.fslot(a2);
  return (uint32_t)((s1 * np - ((s1 * (s1 - 1)) >> 1) + (s2 - s1)) * 42 + 6 * r1 + r2);
}
// cut points of the chunks [lo, hi) by estimated cost (cost = running sum over all chunks, cost[c + 1] - cost[c] = chunk c): cut[i] = first chunk c in
// [lo, hi] with cost[c] - cost[lo] >= ceil(i total / 1024); cut[0] = lo, cut[1024] = hi
template <class COST> static inline void ba_rw_make_cuts(const COST& cost, int lo, int hi, int* cut) {
  const unsigned long long c0 = cost[lo], total = (unsigned long long)cost[hi] - c0;
  int c = lo;
  for (int i = 0; i <= BA_RW_CUTS; ++i) {
    const unsigned long long target = (total * (unsigned long long)i + BA_RW_CUTS - 1) / BA_RW_CUTS;
    while (c < hi && (unsigned long long)cost[c] - c0 < target) ++c;
    cut[i] = c;
  }
  cut[0] = lo; cut[BA_RW_CUTS] = hi;
}

// the order of a window's runs: class 0 first, first appearance inside a class (a stable partition; run_of_group follows).  Both planners call this,
// so that their device arrays stay byte-identical
template <class RUN, class KF> static inline void ba_rw_order_runs(std::vector<RUN>& runs, std::vector<int>& run_of_group, KF&& kf_of) {
  const int n = (int)runs.size();
  std::vector<int> newid(n, 0);
  std::vector<RUN> out; out.reserve(n);
  for (int cls = 0; cls < 2; ++cls)
    for (int r = 0; r < n; ++r) if (ba_rw_class(kf_of(runs[r])) == cls) { newid[r] = (int)out.size(); out.push_back(runs[r]); }
  runs.swap(out);
  for (int& g : run_of_group) if (g >= 0) g = newid[g];
}

// where element i (0 .. 20: upper triangle, row by row) of a key frame's own 6x6 block sits inside the 6x6 block of its diagonal pair: 6 r + q
__device__ __forceinline__ int ba_rw_upper_pos(int i) {
  const unsigned long long lo = 0ull | (1ull << 6) | (2ull << 12) | (3ull << 18) | (4ull << 24) | (5ull << 30) | (7ull << 36) | (8ull << 42) | (9ull << 48) | (10ull << 54);
  const unsigned long long hi = 11ull | (14ull << 6) | (15ull << 12) | (16ull << 18) | (17ull << 24) | (21ull << 30) | (22ull << 36) | (23ull << 42) | (28ull << 48) | (29ull << 54);
  return i < 10 ? (int)((lo >> (6 * i)) & 63ull) : i < 20 ? (int)((hi >> (6 * (i - 10))) & 63ull) : 35;
}

#ifdef BA_RW_TS
// developer instrumentation (tools/ab_build.sh rwts -DBA_RW_TS): start / end of every unit on the 100 MHz real-time counter; cms_ba_debug_rw_ts reads it
__device__ long long ba_rw_ts[2 * 2 * 4096];
#endif
template <int CLS>
__device__ __forceinline__ void ba_schur_run_wg_body(int BX, int U, BaDevG d, BaSeG se, double* __restrict__ Hll, double* __restrict__ bl, double lambda,
                                                     const double* __restrict__ poses, const double* __restrict__ pts, int robust, double delta) {
#pragma clang fp contract(fast)
  constexpr int NTILES = CLS == 0 ? 3 : 6;
  extern __shared__ __align__(16) double rw_lds[];
  const int lane = threadIdx.x;                                     // (blockDim.x == 64)
#ifdef BA_RW_TS
  const long long ts0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
#ifdef BA_RM_CLK
  const bool clk_on = BX == BA_RM_CLK && blockIdx.z == 0 && CLS == 0;      // (developer build: the unit whose cycle stamps are kept)
  long long clk_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long clk_last = (long long)__builtin_readcyclecounter();
#endif
  double* slots = rw_lds;                                           // D^-1 | y of up to 32 points (6 doubles each)
  double* buf = slots + BA_RM_PTS * 6;                              // a row of 18 doubles per lane
  double* prt = slots + BA_RM_BUF;                                  // K x 12
  // ---- the unit's range of chunks
  const int c_lo = CLS == 0 ? 0 : se.n_rmA, c_hi = CLS == 0 ? se.n_rmA : se.n_rm;
  if (c_hi <= c_lo) return;
  int cb, ce;
  {
    const BA_AS1 int* cut = se.rm_cut + (size_t)CLS * (BA_RW_CUTS + 1);
    const int i0 = (int)(((long long)BX * BA_RW_CUTS) / U), i1 = (int)(((long long)(BX + 1) * BA_RW_CUTS) / U);
    cb = __builtin_amdgcn_readfirstlane(cut[i0]); ce = __builtin_amdgcn_readfirstlane(cut[i1]);
  }
  if (ce <= cb) return;
  for (int i = lane; i < BA_RM_BUF; i += 64) slots[i] = 0.0;        // whatever the matrix phase reads must be finite
  for (int k = lane; k < d.K; k += 64) {
    double R[9];
    quat_to_R(poses + 7 * k + 3, R);
#pragma unroll
    for (int i = 0; i < 9; ++i) prt[12 * k + i] = R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) prt[12 * k + 9 + i] = poses[7 * k + i];
  }
  __syncthreads();
  const int li = lane & 15, lk = lane >> 4;

  // the key frames' own blocks and gradients (27 sums per key frame of the signature): every chunk the lanes of an observation position add their
  // shares up through the buffer, and ONE lane per (position, sum) keeps the running total -- two or four doubles per lane instead of the 27 per
  // lane of the monolith, which alone were a third of the 168 registers three wavefronts per SIMD leave.  Task t = lane + 64 round of pass h
  // (pass 0: sums 0 .. 13, pass 1: sums 14 .. 26): position t / nv, sum t % nv
  double hacc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  ba_v4d acc[NTILES];                                               // resident upper tiles (0,0) (0,1) (1,1) [(0,2) (1,2) (2,2)]
#pragma unroll
  for (int t = 0; t < NTILES; ++t) acc[t] = (ba_v4d){0.0, 0.0, 0.0, 0.0};
  int cur_run = -1, NT = 1;
  uint32_t ent[3] = {BA_RM_MF_NONE, BA_RM_MF_NONE, BA_RM_MF_NONE};
  uint32_t v_kf = 1;
  auto load_tab = [&](int run) {
    const BA_AS1 uint32_t* t = se.run_mf + (size_t)run * 64;
    ent[0] = t[li]; ent[1] = t[16 + li]; if (CLS == 1) ent[2] = t[32 + li];
    v_kf = t[56];
  };
  auto sdesc = [&](const int4 v) { return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y), __builtin_amdgcn_readfirstlane(v.z), __builtin_amdgcn_readfirstlane(v.w)); };
  const int4 none = make_int4(0, 0, -1, 0);
  int4 d_cur = none, d_nxt = none, v_nn = none;
  d_cur = sdesc(ba_ld4i(se.rm_chunk + cb));
  if (cb + 1 < ce) d_nxt = sdesc(ba_ld4i(se.rm_chunk + (cb + 1)));
  if (cb + 2 < ce) v_nn = ba_ld4i(se.rm_chunk + (cb + 2));
  uint32_t n_info = 0; uint8_t n_lvl = 0; double n_inv = 0.0;
  int n_p = 0, n_e = 0;
  double n_X[3] = {0, 0, 1};
  double2 n_obs = make_double2(0.0, 0.0);
  auto load_chunk = [&](const int4 dc) {
    if (dc.z >= 0) {                                               // (uniform)
      const int ne = dc.y & 255, kk = (dc.y >> 8) & 255;
      const int le = min(lane, ne - 1);
      const int e = dc.x + le;
      n_e = e; n_info = se.e_info[e]; n_lvl = d.level[e]; n_inv = d.e_inv[e];
      n_obs = BA_OBS2(d, e);
      n_p = dc.w + ((le * (int)__builtin_ceilf(65536.0f * __builtin_amdgcn_rcpf((float)kk))) >> 16);
      const double* Xp = pts + 3 * (size_t)n_p;
      n_X[0] = Xp[0]; n_X[1] = Xp[1]; n_X[2] = Xp[2];
    }
  };
  load_chunk(d_cur);

  BA_RM_STAMP(9);                                                  // start of the unit: cut points, LDS, first descriptors and operands requested
  for (int c = cb; c < ce; ++c) {
    BA_RM_STAMP(0);
    const int4 desc = d_cur;
    const int ne = desc.y & 255, k_run = (desc.y >> 8) & 255, m = desc.y >> 16;
    const uint32_t info = n_info;
    const bool live = lane < ne;
    double ow = (live && n_lvl == 0) ? n_inv : 0.0;
    const int pnt = n_p, eid = n_e;
    const double2 obs = n_obs;
    const double X[3] = {n_X[0], n_X[1], n_X[2]};
    const bool new_run = desc.z != cur_run;
    if (new_run) { cur_run = desc.z; load_tab(cur_run); }
    const int invk = (int)__builtin_ceilf(65536.0f * __builtin_amdgcn_rcpf((float)k_run));
    BA_RM_STAMP(1);                                                // operands of the chunk in registers
    // ---------------------------------------------------------------- vector phase
    const int a = info & 31, s_ = (int)((info >> 10) & 63) - 1, face = (info >> 16) & 7, kp = (info >> 19) & 255;
    const bool act = ow != 0.0;
    double Jp[12], Jl[6], o0, o1;
    {
      const double* Rt = prt + 12 * kp;
      double R[9], Xc[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) R[i] = Rt[i];
      ba_se_cam_point(Rt, X, Xc);
      double r[2], rho0;
      edge_error_v(d, face, obs.x, obs.y, Xc, r);
      r[0] = act ? r[0] : 0.0; r[1] = act ? r[1] : 0.0;
      const double om = ow;
      const double w = robust ? huber_w(om * (r[0] * r[0] + r[1] * r[1]), delta, &rho0) : 1.0;
      ow = w * om;
      o0 = -om * r[0] * w; o1 = -om * r[1] * w;
      double lf[3];
      face_local(face, Xc, lf);
      if (!act) { lf[0] = 0.0; lf[1] = 0.0; lf[2] = 1.0; }
      edge_jac_local(d, face, lf, Xc, R, Jp, Jl);
    }
    const double owf = s_ >= 0 ? ow : 0.0;
    BA_RM_STAMP(2);                                                // residual, weight, Jacobians
    {
      constexpr int RR[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5}, QQ[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};
      const double of0 = s_ >= 0 ? o0 : 0.0, of1 = s_ >= 0 ? o1 : 0.0;
      auto share = [&](int idx) -> double {      // (idx is a constant wherever this is called: the loops below are unrolled)
        if (idx < 21) return owf * (Jp[RR[idx < 21 ? idx : 0]] * Jp[QQ[idx < 21 ? idx : 0]] + Jp[6 + RR[idx < 21 ? idx : 0]] * Jp[6 + QQ[idx < 21 ? idx : 0]]);
        if (idx < 27) return Jp[idx - 21 < 6 && idx >= 21 ? idx - 21 : 0] * of0 + Jp[6 + (idx - 21 < 6 && idx >= 21 ? idx - 21 : 0)] * of1;
        return 0.0;
      };
      const uint32_t pstride = (uint32_t)k_run * 14u * 8u;          // from one point's rows to the next one's, bytes
      const char* hb = reinterpret_cast<const char*>(slots);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int nv = h == 0 ? 14 : 13, v0 = h == 0 ? 0 : 14;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        {
          double2* w2 = reinterpret_cast<double2*>(slots + (size_t)lane * 14);
#pragma unroll
          for (int i = 0; i < 7; ++i) w2[i] = make_double2(share(v0 + 2 * i), share(v0 + 2 * i + 1 < v0 + nv ? v0 + 2 * i + 1 : 27));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int T = k_run * nv;
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          if (64 * rd < T) {                                        // (uniform)
            const int t = 64 * rd + lane;
            const int pa = (t * (h == 0 ? 4682 : 5042)) >> 16;      // t / 14, t / 13 for t < 128
            const uint32_t o = (uint32_t)(pa * 14 + (t - nv * pa)) * 8u;
            double tot = 0.0;
            if (t < T)
              for (int j = 0; j < m; ++j) tot += *reinterpret_cast<const double*>(hb + o + (uint32_t)j * pstride);
            hacc[h][rd] += tot;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    BA_RM_STAMP(3);                                                // the key frames' own sums through the buffer
    double hl[9];
    hl[0] = ow * (Jl[0] * Jl[0] + Jl[3] * Jl[3]); hl[1] = ow * (Jl[0] * Jl[1] + Jl[3] * Jl[4]); hl[2] = ow * (Jl[0] * Jl[2] + Jl[3] * Jl[5]);
    hl[3] = ow * (Jl[1] * Jl[1] + Jl[4] * Jl[4]); hl[4] = ow * (Jl[1] * Jl[2] + Jl[4] * Jl[5]); hl[5] = ow * (Jl[2] * Jl[2] + Jl[5] * Jl[5]);
    hl[6] = Jl[0] * o0 + Jl[3] * o1; hl[7] = Jl[1] * o0 + Jl[4] * o1; hl[8] = Jl[2] * o0 + Jl[5] * o1;
    if (live) d.ow[eid] = ow;
    double sum[9];
    if (k_run == 4) {
#pragma unroll
      for (int i = 0; i < 9; ++i) sum[i] = ((ba_quad_bcast<0>(hl[i]) + ba_quad_bcast<1>(hl[i])) + ba_quad_bcast<2>(hl[i])) + ba_quad_bcast<3>(hl[i]);
    } else if (k_run == 2) {
#pragma unroll
      for (int i = 0; i < 9; ++i) sum[i] = ba_quad_perm<0xA0>(hl[i]) + ba_quad_perm<0xF5>(hl[i]);
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      {
        double2* row2 = reinterpret_cast<double2*>(buf + (size_t)lane * 18);
#pragma unroll
        for (int i = 0; i < 4; ++i) row2[i] = make_double2(hl[2 * i], hl[2 * i + 1]);
        row2[4] = make_double2(hl[8], 0.0);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 9; ++i) sum[i] = 0.0;
      for (int j = 0; j < k_run; ++j) {
        const double2* row2 = reinterpret_cast<const double2*>(buf + (size_t)(lane - a + j) * 18);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const double2 u = row2[i]; sum[2 * i] += u.x; sum[2 * i + 1] += u.y; }
        sum[8] += row2[4].x;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    BA_RM_STAMP(4);                                                // exchange: Hll, bl of the point in every lane
    if (live && a == 0) {
      double* H = Hll + 9 * (size_t)pnt; double* bq = bl + 3 * (size_t)pnt;
      H[0] = sum[0]; H[1] = sum[1]; H[2] = sum[2]; H[3] = sum[1]; H[4] = sum[3]; H[5] = sum[4]; H[6] = sum[2]; H[7] = sum[4]; H[8] = sum[5];
      bq[0] = sum[6]; bq[1] = sum[7]; bq[2] = sum[8];
    }
    const int jpt = ((lane - a) * invk) >> 16;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    {
      const double dead = live ? 0.0 : 1.0;
      const double a00 = sum[0] + lambda + dead, a10 = sum[1], a11 = sum[3] + lambda + dead, a20 = sum[2], a21 = sum[4], a22 = sum[5] + lambda + dead;
      const double i0 = 1.0 / a00;
      const double l10 = a10 * i0, l20 = a20 * i0;
      const double d1 = a11 - l10 * a10;
      const double i1 = 1.0 / d1;
      const double l21 = (a21 - l20 * a10) * i1;
      const double d2 = a22 - l20 * a20 - l21 * (l21 * d1);
      const double i2 = 1.0 / d2;
      if (live && a == 0) {
        const double y0 = sum[6], y1 = sum[7] - l10 * y0, y2 = sum[8] - l20 * y0 - l21 * y1;
        double2* pp = reinterpret_cast<double2*>(slots + (size_t)jpt * 6);
        pp[0] = make_double2(i0, i1); pp[1] = make_double2(i2, y0); pp[2] = make_double2(y1, y2);
      }
      double2* row2 = reinterpret_cast<double2*>(buf + (size_t)lane * 18);
#pragma unroll
      for (int r = 0; r < 6; r += 2) {
        double wv[6];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const double q0 = owf * (Jp[r + h] * Jl[0] + Jp[6 + r + h] * Jl[3]);
          const double q1 = owf * (Jp[r + h] * Jl[1] + Jp[6 + r + h] * Jl[4]);
          const double q2 = owf * (Jp[r + h] * Jl[2] + Jp[6 + r + h] * Jl[5]);
          const double w0 = q0, w1 = q1 - w0 * l10, w2 = q2 - w0 * l20 - w1 * l21;
          wv[3 * h] = w0; wv[3 * h + 1] = w1; wv[3 * h + 2] = w2;
        }
        row2[3 * (r >> 1)] = make_double2(wv[0], wv[1]); row2[3 * (r >> 1) + 1] = make_double2(wv[2], wv[3]); row2[3 * (r >> 1) + 2] = make_double2(wv[4], wv[5]);
      }
    }
    BA_RM_STAMP(5);                                                // 3x3 factorisation, W
    const int niter = (m + 3) >> 2;
    if (lane < 4 * niter - m) {
      double2* pp = reinterpret_cast<double2*>(slots + (size_t)(m + lane) * 6);
      pp[0] = make_double2(0.0, 0.0); pp[1] = make_double2(0.0, 0.0); pp[2] = make_double2(0.0, 0.0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // the next chunk's operands (and, when this chunk ends its run, the flush table) are requested here: the matrix phase below gives them time
    d_cur = d_nxt;
    d_nxt = sdesc(v_nn);
    v_nn = none;
    if (c + 3 < ce) v_nn = ba_ld4i(se.rm_chunk + (c + 3));
    load_chunk(d_cur);
    const bool run_ends = d_cur.z != desc.z;                       // (uniform; also at the end of the unit's range: d_cur.z == -1 there)
    if (new_run) { const int kf = __builtin_amdgcn_readfirstlane((int)v_kf); NT = (6 * kf + 1 + 15) >> 4; }
    BA_RM_STAMP(6);                                                // rows published, next chunk's loads issued
    // ---------------------------------------------------------------- matrix phase: G += Y D^-1 Y^T over the chunk's points
    {
      const uint32_t rs8 = (uint32_t)k_run * 144u, row0 = BA_RM_PTS * 48u;
      const uint32_t lim = row0 + (64u * 18u - 1u) * 8u;
      uint32_t ad[3][3], st[3], dd[3];
      int mj[3], mc[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) { const int kap = 4 * u + lk; mj[u] = (kap * 11) >> 5; mc[u] = kap - 3 * mj[u]; }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const bool isw = ent[t] < BA_RM_MF_RHS;
        st[t] = isw ? 4u * rs8 : 192u;
        const uint32_t base = isw ? row0 + ent[t] * 8u : 24u, per = isw ? rs8 : 48u;
#pragma unroll
        for (int u = 0; u < 3; ++u) ad[t][u] = base + (uint32_t)mj[u] * per + (uint32_t)mc[u] * 8u;
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) dd[u] = ((uint32_t)mj[u] * 6u + (uint32_t)mc[u]) * 8u;
      const char* bb = reinterpret_cast<const char*>(slots);
      auto ldsd = [&](uint32_t off) { return *reinterpret_cast<const double*>(bb + off); };
      if (CLS == 0) {
        double nB0[3], nB1[3], nD[3];
        auto fetch = [&](bool last) {
          if (last) {
#pragma unroll
            for (int u = 0; u < 3; ++u) { ad[0][u] = min(ad[0][u], lim); ad[1][u] = min(ad[1][u], lim); }
          }
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            nD[u] = ldsd(dd[u]);
            nB0[u] = ldsd(ad[0][u]);
            nB1[u] = NT > 1 ? ldsd(ad[1][u]) : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 3; ++u) { dd[u] += 192u; ad[0][u] += st[0]; ad[1][u] += st[1]; }
        };
        fetch(niter == 1);
        for (int q = 0; q < niter; ++q) {
          const double Bv0[3] = {nB0[0], nB0[1], nB0[2]}, Bv1[3] = {nB1[0], nB1[1], nB1[2]}, Dv[3] = {nD[0], nD[1], nD[2]};
          if (q + 1 < niter) fetch(q + 2 == niter);
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const double A0 = Bv0[u] * Dv[u];
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, Bv0[u], acc[0], 0, 0, 0);
            if (NT > 1) {
              const double A1 = Bv1[u] * Dv[u];
              acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, Bv1[u], acc[1], 0, 0, 0);
              acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, Bv1[u], acc[2], 0, 0, 0);
            }
          }
        }
      } else {
        for (int q = 0; q < niter; ++q) {
          if (q == niter - 1) {
#pragma unroll
            for (int u = 0; u < 3; ++u) { ad[0][u] = min(ad[0][u], lim); ad[1][u] = min(ad[1][u], lim); ad[2][u] = min(ad[2][u], lim); }
          }
          double c0 = ldsd(ad[0][0]), c1 = ldsd(ad[1][0]), c2 = ldsd(ad[2][0]), cd = ldsd(dd[0]);
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const double b0 = c0, b1 = c1, b2 = c2, dv = cd;
            if (u < 2) { c0 = ldsd(ad[0][u + 1]); c1 = ldsd(ad[1][u + 1]); c2 = ldsd(ad[2][u + 1]); cd = ldsd(dd[u + 1]); }
            const double A0 = b0 * dv, A1 = b1 * dv, A2 = b2 * dv;
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, b1, acc[2], 0, 0, 0);
            acc[NTILES - 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, b2, acc[NTILES - 3], 0, 0, 0);
            acc[NTILES - 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, b2, acc[NTILES - 2], 0, 0, 0);
            acc[NTILES - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(A2, b2, acc[NTILES - 1], 0, 0, 0);
          }
#pragma unroll
          for (int u = 0; u < 3; ++u) { dd[u] += 192u; ad[0][u] += st[0]; ad[1][u] += st[1]; ad[2][u] += st[2]; }
        }
      }
    }
    BA_RM_STAMP(7);                                                // matrix phase
    // ---------------------------------------------------------------- end of the run (or of the unit's range): the sums go to the global copy
    if (run_ends) {
      // the tiles: accumulator g of tile (ti, tj) in lane l is G[16 ti + (l >> 4) + 4 g][16 tj + (l & 15)] -> run_fg (requested here: once per run,
      // a memory round trip the other wavefronts of the SIMD cover; held from the run's first chunk on it cost twelve registers of 168)
      uint32_t fg[4 * NTILES];
      {
        const BA_AS1 uint32_t* f = se.run_fg + ((size_t)desc.z * 64 + lane) * 24;
#pragma unroll
        for (int i = 0; i < 4 * NTILES; ++i) fg[i] = f[i];
      }
#pragma unroll
      for (int t = 0; t < NTILES; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint32_t o = fg[4 * t + g];
          if (o != BA_RW_NONE) ba_gadd(se.partial + o, acc[t][g]);
        }
        acc[t] = (ba_v4d){0.0, 0.0, 0.0, 0.0};
      }
      // the key frames' own blocks and gradients: the lane that kept the total of (position, sum) adds it to the diagonal pair of that key frame
      // (upper triangle | right-hand side) and to bp
      const int np = d.np;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int nv = h == 0 ? 14 : 13, v0 = h == 0 ? 0 : 14;
        const int T = k_run * nv;
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          if (64 * rd < T) {
            const int t = 64 * rd + lane;
            const int pa = min((t * (h == 0 ? 4682 : 5042)) >> 16, k_run - 1);
            const int sa = __shfl(s_, pa);                          // free-pose slot of the observation at position pa (lane pa: point 0 of the chunk)
            const double tot = hacc[h][rd];
            if (t < T && sa >= 0 && tot != 0.0) {
              const int i = v0 + (t - nv * pa);
              const size_t pr = (size_t)ba_se_pair(np, sa, sa) * 42;
              if (i < 21) ba_gadd(se.partial + pr + ba_rw_upper_pos(i), -tot);
              else { ba_gadd(se.partial + pr + 36 + (i - 21), -tot); ba_gadd(se.bp_partial + (size_t)sa * 6 + (i - 21), tot); }
            }
            hacc[h][rd] = 0.0;
          }
        }
      }
      BA_RM_STAMP(8);                                              // flush
    }
  }
#ifdef BA_RM_CLK
  if (clk_on && lane == 0) {
    for (int i = 0; i < 12; ++i) ba_rm_clk[i] = clk_acc[i];
    ba_rm_clk[12] = ce - cb;
  }
#endif
#ifdef BA_RW_TS
  { const int id = (int)blockIdx.z * U + BX; if (lane == 0 && id < 4096) { ba_rw_ts[(CLS * 4096 + id) * 2] = ts0; ba_rw_ts[(CLS * 4096 + id) * 2 + 1] = (long long)__builtin_amdgcn_s_memrealtime(); } }
#endif
}

// class 0: three wavefronts per SIMD (<= 168 registers); class 1: six resident tiles, two wavefronts per SIMD
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
kb_ba_lin_schur_run_wg0(const BaItem* __restrict__ items, BaDyn dyn, int phase, int units) {
  BA_ITEM(phase, units)
  ba_schur_run_wg_body<0>(blockIdx.x, units, it.d, it.se, it.Hll, it.bl, ba_lambda, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta);
}
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
kb_ba_lin_schur_run_wg1(const BaItem* __restrict__ items, BaDyn dyn, int phase, int units) {
  BA_ITEM(phase, units)
  ba_schur_run_wg_body<1>(blockIdx.x, units, it.d, it.se, it.Hll, it.bl, ba_lambda, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta);
}
