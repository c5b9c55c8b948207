// cms_ba_schur_edges.hip -- Schur complement of the local BA, EDGE-major: one lane per observation, the reduced system of a workgroup's
// slice of the window accumulated in LDS with ds_add_f64.
//
// The pair-owner kernel (cms_ba_schur_points.hip) is deterministic but pays for it: one lane per co-visible pose pair walks that
// pair's tuples of a staged batch, and with ~1.6 tuples per pair and batch the waves run at a third of their lanes (profiles/r01:
// 75 us for eight 80k-edge windows, 6 % of the FP64 rate).  Duplicating it in bench.py's step costs 4.6 ms of 16.4: it is the largest
// consumer of the chip in the whole step.  Here the unit of work is the observation:
//
//   * edges are sorted by point (CSR); the host cuts them into CHUNKS of whole points with at most 64 edges; a wavefront takes a chunk,
//     lane = edge;
//   * the lane rebuilds its edge's 6x3 block B = ow Jp^T Jl from the estimate (as the pair-owner kernel's stagers do), factors the
//     point's A = Hll + lambda I = L D L^T (3x3, no square roots) and forms W = B L^-T, so that  B_a A^-1 B_b^T = W_a D^-1 W_b^T;
//     W goes to a wave-private LDS row, W D^-1 stays in registers;
//   * the point's k(k-1)/2 off-diagonal tuples are spread over its k lanes: in step d = 1 .. k/2 lane a does the tuple
//     (a, a + d mod k) -- the partner's W is a 144-byte LDS read from the neighbouring row, no barrier (LDS operations of one
//     wavefront execute in order);
//   * the LDS takes the 64 lanes of a ds_add_f64 in four groups of 16 consecutive lanes: two clocks per group when its addresses fall on
//     16 different f64 banks (address mod 16 doubles), two more for every further lane on the fullest bank, much more for lanes on the
//     SAME address (tools/probe/lds_atomics.hip: 7.8 additions per clock and CU on consecutive addresses, 2.7 for random pose pairs,
//     0.33 on one address).  Three things follow.  The diagonal tuples (a, a) -- 64 lanes on ~19 diagonal blocks cannot avoid each other --
//     go to one of four copies of the diagonal blocks.  A 6x6 block is laid out so that element (r, q) and its transpose are 16 doubles
//     apart (ba_se_off): a wrapped partner, which holds the transposed block, hits the same banks as an unwrapped one.  And the host
//     composes the chunks (cms_api_ba.hip, ba_compose_chunks): which points share a chunk, in which order, and which diagonal copy an
//     edge adds to, so that the lanes of a group repeat banks as little as possible;
//   * (a pose-major kernel for the diagonal tuples -- deterministic, butterfly reduction like ba_lin_poses_body -- was measured too:
//     35 us for eight windows, bound by the three gathered cache lines per edge; dropped)
//   * every product element is added to the workgroup's copy of the reduced system in LDS (36 doubles per off-diagonal pose pair; per copy
//     and key frame 21 + 6 + 6 for the diagonal block, its right-hand side and bp) with ds_add_f64; when the loop is done the copy is
//     written to this range's slice of `partial`, and kb_ba_schur_edges_reduce adds the ranges in fixed order.
//
// Where the time goes (profiles/r02_pmc_instruction_mix.json, eight 80k-edge windows, fused variant): 65 us with all windows active; LDS pipe
// 45 % busy (80 % before the composition, a third of it bank conflicts), vector-ALU issue 22 us; 148 KB of LDS per workgroup leave two
// wavefronts per SIMD, so what is left is latency.  Measured and dropped: the exchanges between lanes through wavefront shuffles / DPP
// shifts instead of LDS rows (two workgroups per CU, but ~150 ds_bpermute per chunk and spills at three wavefronts per SIMD: 75 us);
// ten wavefronts with two diagonal copies (59 us).
//
// The LDS additions of different wavefronts interleave in an order that is not fixed: sums may differ in the last bits from run to
// run (the BA parity bar is 1e-4 relative on updates; the oracle's own summation order is different anyway).  CMS_BA_DETERMINISTIC=1
// selects the pair-owner kernel instead.  (block_solver.hpp:367-437: Hschur -= Bi Dinv Bj^T, bschur -= Bi Dinv bl.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef BA_SE_THREADS
#define BA_SE_THREADS 512
#endif
#define BA_SE_SSTRIDE 37          /* doubles per off-diagonal pose pair in the LDS copy: 36, padded to an odd count (bank aliasing of neighbouring pairs) */
#define BA_SE_RANGES 128          /* workgroups per window at most (a window alone; a group shares the chip: ba_group_ranges) */

struct BaSe {                      // device view of the edge-major work list (cms_api_ba.hip)
  int R, nchunks, cpw;            // ranges (workgroups) of the edge-major body over the chunks [n_rm, nchunks), all chunks, chunks per range
  int n_rm, R_rm;                 // chunks [0, n_rm) hold points of ONE observation signature each (cms_ba_schur_runs.hip): R_rm workgroups take them;
                                  // their slices of `partial` come first, the edge-major ranges' slices follow (R_rm + R slices in all)
  const int4* rm_chunk;           // n_rm: first edge | edges + (edges per point << 8) + (points << 16) | run | first (internal) point
  const uint2* run_lane;          // runs x 64: which tuple of the signature a consumer lane multiplies, and where its sum goes (ba_rm_lane_*)
  const uint32_t* run_mf;         // runs x 64: rows / columns of a signature's stacked matrix for the MFMA variant (cms_ba_schur_runs.hip, BA_RM_MF_*)
  const uint32_t* run_fl;         // runs x 64 x 12: per lane, where its 24 MFMA accumulators are added (two 16-bit LDS offsets per word)
  const uint32_t* rm_cost;        // n_rm + 1: running sum of the run chunks' estimated cost (ba_rm_chunk_cost): the wavefronts' ranges are cut by cost, not by count
  // one-wavefront workgroups (cms_ba_schur_runwg.hip): the runs are ordered by class (signatures with two tile rows first), chunks [0, n_rmA) are class 0
  const uint32_t* run_fg;         // runs x 64 x 24: per lane and MFMA accumulator its offset into `partial` (doubles; 0xFFFFFFFF: nowhere)
  const int* rm_cut;              // 2 x 1025: per class, the chunk at which i / 1024 of the class's estimated cost is reached
  int n_rmA;
  int Rt, cpw_t;                  // the same chunks cut into more, shorter ranges for the edge-major trial kernel (no LDS copy of the system to amortise)
  int npairs2;                    // np (np + 1) / 2: pose pairs s1 <= s2 enumerated densely, row by row
  const int* chunk_e0;            // nchunks + 1: first edge of every chunk (whole points, <= 64 edges)
  const uint32_t* e_info;         // per edge: index within its point (5 bits) | edges of the point << 5 | (free-pose slot + 1) << 10 | face << 16 |
                                  // key frame << 19 (8 bits) | copy of the diagonal blocks its (a, a) tuple goes to << 27 (2 bits)
  double* partial;                // R x npairs2 x 42
  double* bp_partial;             // R x np x 6: the ranges' sums of bp (fused linearisation: the diagonal blocks of `partial` then hold S - Hpp and s - bp)
  const int* lone; int nlone;     // points without any observation (in no chunk): the trial kernel copies their position
  int gsum;                       // 1: the workgroups ADD their LDS copies to slice 0 of partial / bp_partial (global_atomic_add_f64) instead of writing a slice
                                  // each; the solve kernel reads that one slice and puts the zeros back (kb_ba_trial_solve3r)
  int det;                        // 1: a deterministic window in a fixed-order group (kb_ba_first_pass sums the key frames' diagonals in chunk and slice order)
  int strided;                    // 1: the run-major body's wavefronts take the left-over chunks strided (chunk n_rm + wavefront, + wavefronts, ...) instead of cut by cost
  int pad_[2];                    // (sizeof(BaItem) stays a multiple of 16: the items travel as 16-byte words)
};

struct BaSeG {                     // BaSe with global-memory pointer types (see BaDevG, cms_ba_kernels.hip): what the device bodies take
  int R, nchunks, cpw, n_rm, R_rm;
  const BA_AS1 int4* rm_chunk; const BA_AS1 uint2* run_lane; const BA_AS1 uint32_t* run_mf; const BA_AS1 uint32_t* run_fl; const BA_AS1 uint32_t* rm_cost;
  const BA_AS1 uint32_t* run_fg; const BA_AS1 int* rm_cut; int n_rmA;
  int Rt, cpw_t, npairs2;
  const BA_AS1 int* chunk_e0; const BA_AS1 uint32_t* e_info;
  BA_AS1 double* partial; BA_AS1 double* bp_partial;
  const BA_AS1 int* lone; int nlone;
  int gsum, det, strided;
  __device__ __forceinline__ BaSeG() {}
  __device__ __forceinline__ BaSeG(const BaSe& s)
      : R(s.R), nchunks(s.nchunks), cpw(s.cpw), n_rm(s.n_rm), R_rm(s.R_rm), rm_chunk(ba_g(s.rm_chunk)), run_lane(ba_g(s.run_lane)), run_mf(ba_g(s.run_mf)),
        run_fl(ba_g(s.run_fl)), rm_cost(ba_g(s.rm_cost)), run_fg(ba_g(s.run_fg)), rm_cut(ba_g(s.rm_cut)), n_rmA(s.n_rmA), Rt(s.Rt), cpw_t(s.cpw_t), npairs2(s.npairs2), chunk_e0(ba_g(s.chunk_e0)), e_info(ba_g(s.e_info)), partial(ba_g(s.partial)),
        bp_partial(ba_g(s.bp_partial)), lone(ba_g(s.lone)), nlone(s.nlone), gsum(s.gsum), det(s.det), strided(s.strided) {}
};

__device__ __forceinline__ int ba_se_pair(int np, int s1, int s2) { return s1 * np - ((s1 * (s1 - 1)) >> 1) + (s2 - s1); }   // s1 <= s2, dense (with diagonal)
__device__ __forceinline__ int ba_se_opair(int np, int s1, int s2) { return s1 * np - ((s1 * (s1 + 1)) >> 1) + (s2 - s1 - 1); }   // s1 < s2, off-diagonal only
// layout of a 6x6 block of S in LDS: strictly upper elements (r < q) at 0..14, their transposes at 16..30, the diagonal at 15, 31, 32..35
__host__ __device__ constexpr int ba_se_upper(int r, int q) { return r * 5 - (r * (r - 1)) / 2 + (q - r - 1); }      // r < q
__host__ __device__ constexpr int ba_se_off(int r, int q) {
  return r < q ? ba_se_upper(r, q) : r > q ? 16 + ba_se_upper(q, r) : (r == 0 ? 15 : r == 1 ? 31 : 30 + r);
}
#ifndef BA_SE_DCOPIES
#define BA_SE_DCOPIES 4
#endif
#define BA_SE_DSTRIDE 33          /* 21 (upper triangle) + 6 (right-hand side) + 6 (bp, fused linearisation only): an ODD number of doubles, so that consecutive
                                     (copy, key frame) rows walk through all 16 f64 banks of the LDS (28 left them on four) */

// key-frame-frame point from the LDS copy of a pose, in cam_point's operation order and WITHOUT contraction: the residual of the fused
// linearisation must be the one ba_errors_body / the trial kernels compute (the projection is rounded to float: edge_error_v)
__device__ __forceinline__ void ba_se_cam_point(const double* Rt, const double* X, double* Xc) {
#pragma unroll
  for (int i = 0; i < 3; ++i) Xc[i] = Rt[3 * i] * X[0] + Rt[3 * i + 1] * X[1] + Rt[3 * i + 2] * X[2] + Rt[9 + i];
}

// FUSED = true: the kernel also LINEARISES (what kb_ba_lin does in front of the first trial of an iteration): a lane computes its edge's
// residual, robust weight and Jacobians from the current estimate; the lanes of a point exchange their 3x3 / 3x1 contributions through
// their LDS rows, so that every one of them holds Hll and bl of the point (its first lane stores them for the trial kernel); the key
// frame's own block  ow Jp^T Jp  and gradient ride on the diagonal tuple's additions (the diagonal block of `partial` becomes
// S_aa - Hpp_aa, its right-hand side  s_a - bp_a; bp alone goes to six more slots: the gain ratio needs it).  From the second iteration
// of a stage on no other kernel linearises: kb_ba_lin + kb_ba_maxdiag drop out of the round (25 + 7 us of ~165 for eight windows), and a
// rejected trial merely repeats arithmetic this kernel had to do anyway (it rebuilt the Jacobians from the estimate before, too).
// ---- FIXED-ORDER additions (deterministic windows: cms_ba_set_deterministic, the kernels kb_ba_lin_schur_runs_det / kb_ba_lin_schur_edges_det).
// The only thing that makes the sums of this kernel family differ in their last bits from run to run is the ORDER in which the wavefronts of a
// workgroup add to the workgroup's LDS copy (the workgroups' copies go out as slices that the solve kernel adds in slice order, every wavefront's
// chunks are a fixed range, and inside a wavefront the instruction order is the program's).  The deterministic kernels give every set of additions
// a KEY that follows from the window's plan alone -- t = the estimated cost of the wavefront's chunks up to and including the one the additions
// belong to (run-major body and its left-over chunks), or the chunk's index (edge-major body) -- and perform the sets in ascending (t, wavefront)
// order: L[w] (LDS) holds a lower bound of wavefront w's next key, published at the start of every chunk once its previous additions are through
// (lgkmcnt(0)); a wavefront adds when every other wavefront's bound lies above its key.  Nobody waits unless the others are BEHIND it in
// estimated time, so the wavefronts still overlap their vector / matrix phases; the wavefront with the smallest pending key can always go (no
// deadlock: there is no workgroup barrier between the first publication and the last).
#define BA_DET_DONE 0xFFFFFFFFu
#define BA_AS3 __attribute__((address_space(3)))
typedef volatile BA_AS3 uint32_t* ba_det_ptr;                      // (an LDS pointer by type: a volatile generic pointer is accessed with flat instructions)
__device__ __forceinline__ void ba_det_publish(ba_det_ptr L, int wave, uint32_t t) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // this wavefront's LDS additions so far have been performed
  if ((threadIdx.x & 63) == 0) L[wave] = t == BA_DET_DONE ? BA_DET_DONE : ((t << 3) | (uint32_t)wave);
}
__device__ __forceinline__ void ba_det_wait(ba_det_ptr L, int wave, int nw, uint32_t t) {
  const uint32_t key = (t << 3) | (uint32_t)wave;
  const int lane = threadIdx.x & 63;
  // (the wait is bounded by the lifetime of the workgroup's slowest wavefront, a few hundred microseconds.  Should the protocol ever be broken -- a key sequence
  // that does not ascend, a wavefront that leaves without publishing BA_DET_DONE -- the kernel must not spin for ever on somebody's GPU: after ~2^32 shader
  // clocks, seconds, it traps; the launch fails and the call returns an error)
  uint32_t polls = 0;
  long long t0 = 0;
  for (;;) {
    const uint32_t v = lane < nw ? L[lane] : BA_DET_DONE;
    if (__ballot(lane != wave && v <= key) == 0) break;
    __builtin_amdgcn_s_sleep(1);
    if ((++polls & 0xFFFu) == 0u) {
      const long long now = (long long)__builtin_readcyclecounter();
      if (t0 == 0) t0 = now;
      else if (now - t0 > (1ll << 32)) __builtin_trap();
    }
  }
  asm volatile("" ::: "memory");
}

template <bool FUSED> __device__ __forceinline__ void ba_se_writeout(int slice, int np, int NP2, const double* S, const double* Dg, const BaSeG& se);
// The chunks [c_begin, c_end) in steps of c_step, worked on by ONE wavefront: its 64 rows and row slots in LDS, the workgroup's copy (S, Dg) of the
// reduced system and the key frames' rotations / translations (prt) are the caller's -- the edge-major body below (a wavefront takes every nw-th
// chunk of the workgroup's range) and the run-major body (cms_ba_schur_runs.hip: a wavefront's share of the left-over chunks behind its run chunks).
// DET (fixed-order additions, see above): 0 no order; 1 key = chunk index - det_base; 2 key = se.rm_cost[chunk + 1] - det_base; 3 key = det_base + the
// estimated cost of the chunks this call has worked on so far (a strided share of the chunks: the run-major body's deterministic variant)
template <bool FUSED, int DET = 0>
__device__ __forceinline__ void ba_se_wave_chunks(const int c_begin, const int c1, const int c_step, const BaDevG& d, const BaSeG& se, double* __restrict__ Hll,
                                                  double* __restrict__ bl, const double lambda, const double* __restrict__ pts, const int robust, const double delta,
                                                  double* S, double* Dg, double* myrows, int* myslot, const double* prt,
                                                  ba_det_ptr detL = nullptr, const uint32_t det_base = 0) {
#pragma clang fp contract(fast)
  const int lane = threadIdx.x & 63;
  const int det_wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), det_nw = (int)(blockDim.x >> 6);
  const int np = d.np;
  // The loop is software pipelined over a wave's chunks: the per-edge words of chunk c + nw are requested before chunk c is worked on,
  // its per-point operands (position, Hll, bl) right after chunk c's rows are published -- the atomics section hides their latency.
  int n_p = 0, n_e = 0; uint32_t n_info = 0; double n_ow = 0.0;          // FUSED: n_ow carries the edge's information, n_lvl its exclusion flag -- RAW: the
  uint8_t n_lvl = 0;                                                      // selection happens where the chunk is worked on (a select here would wait for both loads at once)
  double n_X[3] = {0, 0, 0}, n_H[6] = {1, 0, 1, 0, 0, 1}, n_b[3] = {0, 0, 0};
  double2 n_obs = make_double2(0.0, 0.0);
  auto load1 = [&](int c) {
    n_info = 0; n_ow = 0.0; n_p = 0; n_e = 0; n_lvl = 0;
    if (c < c1) {
      const int e = se.chunk_e0[c] + lane;
      if (e < se.chunk_e0[c + 1]) {
        n_p = d.e_point[e]; n_info = se.e_info[e]; n_e = e;
        if (FUSED) { n_ow = d.e_inv[e]; n_lvl = d.level[e]; n_obs = BA_OBS2(d, e); }
        else n_ow = d.ow[e];
      }
    }
  };
  auto load2 = [&]() {
    if (n_info != 0) {
      const double* Xp = pts + 3 * (size_t)n_p;
      n_X[0] = Xp[0]; n_X[1] = Xp[1]; n_X[2] = Xp[2];
      if (!FUSED) {
        const double* H = Hll + 9 * (size_t)n_p; const double* bp = bl + 3 * (size_t)n_p;
        n_H[0] = H[0]; n_H[1] = H[3]; n_H[2] = H[4]; n_H[3] = H[6]; n_H[4] = H[7]; n_H[5] = H[8];
        n_b[0] = bp[0]; n_b[1] = bp[1]; n_b[2] = bp[2];
      }
    }
  };
  load1(c_begin);
  load2();
  uint32_t det_acc = det_base;
  for (int c = c_begin; c < c1; c += c_step) {
    uint32_t det_t = 0;
    if (DET) {
      if (DET == 3) det_acc += se.rm_cost[c + 1] - se.rm_cost[c];
      det_t = DET == 1 ? (uint32_t)c - det_base : DET == 2 ? se.rm_cost[c + 1] - det_base : det_acc;
      ba_det_publish(detL, det_wave, det_t);
    }
    const uint32_t info = n_info;
    double ow = (FUSED && n_lvl != 0) ? 0.0 : n_ow;
    const int pnt = n_p, eid = n_e;
    const double2 obs = n_obs;
    const double X[3] = {n_X[0], n_X[1], n_X[2]};
    double Hc[6] = {n_H[0], n_H[1], n_H[2], n_H[3], n_H[4], n_H[5]}, bc[3] = {n_b[0], n_b[1], n_b[2]};
    load1(c + c_step);
    int slot = -1, a = 0, k = 1;
    double W[18], WD[18], z[3];
    double Jp[12], Jl[6], o0 = 0.0, o1 = 0.0;         // FUSED: kept for the key frame's own block on the diagonal tuple
#pragma unroll
    for (int i = 0; i < 18; ++i) { W[i] = 0.0; WD[i] = 0.0; }
    z[0] = z[1] = z[2] = 0.0;
    bool have_jac = false;
    if (info != 0) {
      a = info & 31; k = (info >> 5) & 31;
      const int s = (int)((info >> 10) & 63) - 1, face = (info >> 16) & 7, kp = (info >> 19) & 255;
      if (FUSED ? (ow != 0.0) : (s >= 0 && ow != 0.0)) {
        const double* Rt = prt + 12 * kp;
        double R[9], Xc[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = Rt[i];
        if (FUSED) {
          ba_se_cam_point(Rt, X, Xc);
          double r[2], rho0;
          edge_error_v(d, face, obs.x, obs.y, Xc, r);
          const double om = ow;
          const double w = robust ? huber_w(om * (r[0] * r[0] + r[1] * r[1]), delta, &rho0) : 1.0;
          ow = w * om;
          o0 = -om * r[0] * w; o1 = -om * r[1] * w;
        } else {
#pragma unroll
          for (int i = 0; i < 3; ++i) Xc[i] = R[3 * i] * X[0] + R[3 * i + 1] * X[1] + R[3 * i + 2] * X[2] + Rt[9 + i];
        }
        edge_jac_face(d, face, Xc, R, Jp, Jl);
        have_jac = true;
        if (s >= 0) slot = s;
      }
    }
    if (FUSED) {
      // ---- Hll and bl of the point: every lane publishes its edge's share (upper triangle | gradient) in its row, then adds the rows of its
      // point's edges in edge order -- the order ba_lin_points_body adds them in; all lanes of a point end up with the same bits
      double hl[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) hl[i] = 0.0;
      if (have_jac) {
        hl[0] = ow * (Jl[0] * Jl[0] + Jl[3] * Jl[3]); hl[1] = ow * (Jl[0] * Jl[1] + Jl[3] * Jl[4]); hl[2] = ow * (Jl[0] * Jl[2] + Jl[3] * Jl[5]);
        hl[3] = ow * (Jl[1] * Jl[1] + Jl[4] * Jl[4]); hl[4] = ow * (Jl[1] * Jl[2] + Jl[4] * Jl[5]); hl[5] = ow * (Jl[2] * Jl[2] + Jl[5] * Jl[5]);
        hl[6] = Jl[0] * o0 + Jl[3] * o1; hl[7] = Jl[1] * o0 + Jl[4] * o1; hl[8] = Jl[2] * o0 + Jl[5] * o1;
      }
      if (info != 0) d.ow[eid] = have_jac ? ow : 0.0;          // the trial kernel rebuilds the edge's block from it
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      {
        double2* row2 = reinterpret_cast<double2*>(myrows + (size_t)lane * 18);
#pragma unroll
        for (int i = 0; i < 5; ++i) row2[i] = make_double2(hl[2 * i], hl[2 * i + 1]);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      double sum[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) sum[i] = 0.0;
      int kmax = k;
      for (int o = 32; o > 0; o >>= 1) kmax = max(kmax, __shfl_xor(kmax, o));
      for (int j = 0; j < kmax; ++j) {
        if (info != 0 && j < k) {
          const double2* row2 = reinterpret_cast<const double2*>(myrows + (size_t)(lane - a + j) * 18);
#pragma unroll
          for (int i = 0; i < 5; ++i) { const double2 u = row2[i]; sum[2 * i] += u.x; sum[2 * i + 1] += u.y; }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();                       // the rows are reused for W below
      if (info != 0 && a == 0) {
        double* H = Hll + 9 * (size_t)pnt; double* bq = bl + 3 * (size_t)pnt;
        H[0] = sum[0]; H[1] = sum[1]; H[2] = sum[2]; H[3] = sum[1]; H[4] = sum[3]; H[5] = sum[4]; H[6] = sum[2]; H[7] = sum[4]; H[8] = sum[5];
        bq[0] = sum[6]; bq[1] = sum[7]; bq[2] = sum[8];
      }
      Hc[0] = sum[0]; Hc[1] = sum[1]; Hc[2] = sum[3]; Hc[3] = sum[2]; Hc[4] = sum[4]; Hc[5] = sum[5];      // (0,0) (1,0) (1,1) (2,0) (2,1) (2,2)
      bc[0] = sum[6]; bc[1] = sum[7]; bc[2] = sum[8];
    }
    if (slot >= 0) {
        // A = Hll + lambda I = L D L^T (unit lower L)
        const double a00 = Hc[0] + lambda, a10 = Hc[1], a11 = Hc[2] + lambda, a20 = Hc[3], a21 = Hc[4], a22 = Hc[5] + lambda;
        const double i0 = 1.0 / a00;
        const double l10 = a10 * i0, l20 = a20 * i0;
        const double d1 = a11 - l10 * a10;
        const double i1 = 1.0 / d1;
        const double l21 = (a21 - l20 * a10) * i1;
        const double d2 = a22 - l20 * a20 - l21 * (l21 * d1);
        const double i2 = 1.0 / d2;
        const double y0 = bc[0], y1 = bc[1] - l10 * y0, y2 = bc[2] - l20 * y0 - l21 * y1;      // y = L^-1 bl
        z[0] = i0 * y0; z[1] = i1 * y1; z[2] = i2 * y2;                                          // D^-1 y
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const double q0 = ow * (Jp[r] * Jl[0] + Jp[6 + r] * Jl[3]);                  // row r of B = ow Jp^T Jl
          const double q1 = ow * (Jp[r] * Jl[1] + Jp[6 + r] * Jl[4]);
          const double q2 = ow * (Jp[r] * Jl[2] + Jp[6 + r] * Jl[5]);
          const double w0 = q0, w1 = q1 - w0 * l10, w2 = q2 - w0 * l20 - w1 * l21;     // W L^T = B
          W[3 * r] = w0; W[3 * r + 1] = w1; W[3 * r + 2] = w2;
          WD[3 * r] = w0 * i0; WD[3 * r + 1] = w1 * i1; WD[3 * r + 2] = w2 * i2;
        }
    }
    // ---- publish this lane's W (LDS operations of one wavefront execute in program order: rows written here are what the reads
    // below see, and the reads of the previous chunk are through before these writes)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    {
      double2* row2 = reinterpret_cast<double2*>(myrows + (size_t)lane * 18);
#pragma unroll
      for (int i = 0; i < 9; ++i) row2[i] = make_double2(W[2 * i], W[2 * i + 1]);
      myslot[lane] = slot;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    load2();                     // next chunk's per-point operands travel while this chunk's products are added
#if defined(BA_DET_NO_WAIT_LEFT)      /* developer A/B (tools/ab_build.sh): what the order costs -- results are then NOT repeatable */
    if (DET == 1) ba_det_wait(detL, det_wave, det_nw, det_t);
#else
    if (DET) ba_det_wait(detL, det_wave, det_nw, det_t);      // this chunk's additions: behind those of every smaller key
#endif
    // ---- diagonal tuple (a, a): W D^-1 W^T (upper triangle) and the right-hand side W D^-1 y, into this edge's copy of the diagonal blocks
    if (slot >= 0) {
      double* base = Dg + ((size_t)(info >> 27) * np + slot) * BA_SE_DSTRIDE;
      int cidx = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int q = r; q < 6; ++q) {
          double v = WD[3 * r] * W[3 * q] + WD[3 * r + 1] * W[3 * q + 1] + WD[3 * r + 2] * W[3 * q + 2];
          if (FUSED) v -= ow * (Jp[r] * Jp[q] + Jp[6 + r] * Jp[6 + q]);          // S_aa - Hpp_aa
          unsafeAtomicAdd(base + (cidx++), v);
        }
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double v = W[3 * r] * z[0] + W[3 * r + 1] * z[1] + W[3 * r + 2] * z[2];
        if (FUSED) {
          const double g = Jp[r] * o0 + Jp[6 + r] * o1;                          // bp_a
          v -= g;
          unsafeAtomicAdd(base + 27 + r, g);
        }
        unsafeAtomicAdd(base + 21 + r, v);
      }
    }
    // ---- off-diagonal tuples: step d pairs lane a with (a + d) mod k; for even k the last step is done by the lower half only
    int kh = k >> 1;
    for (int o = 32; o > 0; o >>= 1) kh = max(kh, __shfl_xor(kh, o));
    for (int dd = 1; dd <= kh; ++dd) {
      const bool mine = slot >= 0 && (2 * dd < k || (2 * dd == k && a < dd));
      int b = a + dd;
      const bool wrapped = b >= k;
      if (wrapped) b -= k;
      const int lb = mine ? lane - a + b : lane;
      const int sb = myslot[lb];
      if (mine && sb >= 0) {
        double Wb[18];
        const double2* row2 = reinterpret_cast<const double2*>(myrows + (size_t)lb * 18);
#pragma unroll
        for (int i = 0; i < 9; ++i) { const double2 u = row2[i]; Wb[2 * i] = u.x; Wb[2 * i + 1] = u.y; }
        // the block of pair (s1 < s2) is W_1 D^-1 W_2^T; a wrapped partner has the lower slot (edges of a point ascend by key frame),
        // this lane then holds the TRANSPOSE of the pair's block: same products, element (q, r) instead of (r, q).  The block's layout
        // (ba_se_off) puts an element and its transpose 16 doubles apart -- the same LDS bank: what the host balances per group of 16
        // lanes, the pair's bank class, holds for every lane of the instruction, wrapped or not.
        double* base = S + (size_t)(wrapped ? ba_se_opair(np, sb, slot) : ba_se_opair(np, slot, sb)) * BA_SE_SSTRIDE;
        double* base_up = base + (wrapped ? 16 : 0);      // elements r < q of this lane's product
        double* base_lo = base + (wrapped ? 0 : 16);      // elements r > q
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            const double v = WD[3 * r] * Wb[3 * q] + WD[3 * r + 1] * Wb[3 * q + 1] + WD[3 * r + 2] * Wb[3 * q + 2];
            double* dst = r < q ? base_up + ba_se_upper(r, q) : r > q ? base_lo + ba_se_upper(q, r) : base + ba_se_off(r, r);
            unsafeAtomicAdd(dst, v);      // (the host refuses this kernel for windows in which a point is seen twice by one key frame: sb != slot)
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <bool FUSED, bool DET = false>
__device__ __forceinline__ void ba_schur_edges_body(int BX, BaDevG d, BaSeG se, double* __restrict__ Hll, double* __restrict__ bl, double lambda,
                                                    const double* __restrict__ poses, const double* __restrict__ pts, int robust, double delta) {
  extern __shared__ __align__(16) double se_lds[];
  __shared__ uint32_t det_L_[8];                                  // DET: the wavefronts' key bounds (ba_det_publish / ba_det_wait)
  const ba_det_ptr det_L = (ba_det_ptr)det_L_;
  const int tid = threadIdx.x, wave = tid >> 6, nw = blockDim.x >> 6;
  if (DET && tid < 8) det_L_[tid] = 0u;
  const int np = d.np, NP2 = se.npairs2, NPO = NP2 - np;
  double* S = se_lds;                                              // NPO x 37: off-diagonal pairs s1 < s2
  double* Dg = S + ((NPO * BA_SE_SSTRIDE + 1) & ~1);               // 4 x np x 33: copies of the diagonal blocks (upper triangle | rhs | bp)
  double* rows = Dg + (size_t)BA_SE_DCOPIES * np * BA_SE_DSTRIDE;  // nw x 64 x 18 (16-byte aligned)
  double* prt = rows + (size_t)nw * 64 * 18;                       // K x 12: rotation (row major) | translation of every key frame
  int* rslot = reinterpret_cast<int*>(prt + (size_t)d.K * 12);     // nw x 64: free-pose slot of the edge in a row, -1 = contributes nothing
  for (int i = tid; i < (int)(rows - S); i += blockDim.x) S[i] = 0.0;
  for (int k = tid; k < d.K; k += blockDim.x) {
    double R[9];
    quat_to_R(poses + 7 * k + 3, R);
#pragma unroll
    for (int i = 0; i < 9; ++i) prt[12 * k + i] = R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) prt[12 * k + 9 + i] = poses[7 * k + i];
  }
  __syncthreads();
  const int c0 = se.n_rm + BX * se.cpw, c1 = min(se.nchunks, c0 + se.cpw);      // (the chunks in front of n_rm belong to the run-major body)
  if (DET) {      // additions in chunk order, whatever the number of wavefronts
    ba_se_wave_chunks<FUSED, 1>(c0 + wave, c1, nw, d, se, Hll, bl, lambda, pts, robust, delta, S, Dg, rows + (size_t)wave * 64 * 18, rslot + wave * 64, prt, det_L, (uint32_t)c0);
    ba_det_publish(det_L, __builtin_amdgcn_readfirstlane(wave), BA_DET_DONE);
  } else
  ba_se_wave_chunks<FUSED>(c0 + wave, c1, nw, d, se, Hll, bl, lambda, pts, robust, delta, S, Dg, rows + (size_t)wave * 64 * 18, rslot + wave * 64, prt);
  __syncthreads();
  ba_se_writeout<FUSED>(se.R_rm + BX, np, NP2, S, Dg, se);
}

// ---- what a stage's FIRST iteration needs before its first trial, in one pass over the window's chunks (dyn.fused_lin: every later linearisation
// happens inside the Schur kernel): the residuals and the robust chi2 of the starting estimate (computeActiveErrors + activeRobustChi2) and the
// largest diagonal entry of the Hessian (computeLambdaInit: lambda = 1e-5 x that).  kb_ba_errors, kb_ba_reduce, kb_ba_lin and kb_ba_maxdiag did
// this in four launches, building all of Hpp / Hll / bl only to look at their diagonals (the Schur kernel rebuilds them anyway).  Here a lane
// takes one observation: its squared Jacobian columns go to the key frame's six diagonal sums (LDS, then one global addition per workgroup and
// entry) and, summed over the lanes of its point, to the point's three -- of which only the maximum is kept.
__device__ __forceinline__ void ba_first_pass_body(int BX, int GX, BaDevG d, BaSeG se, const double* __restrict__ poses, const double* __restrict__ pts, int robust,
                                                   double delta, double* __restrict__ partial, double* __restrict__ pose_diag, unsigned long long* __restrict__ pt_max) {
#pragma clang fp contract(fast)
  extern __shared__ __align__(16) double te_lds[];     // K x 12 (rotation | translation) | np x 6 (diagonal sums of the free key frames)
  __shared__ double sh[16];
  __shared__ uint32_t det_L_[8];                        // se.det: the key frames' sums are added in chunk order (ba_det_publish / ba_det_wait) ...
  const ba_det_ptr det_L = (ba_det_ptr)det_L_;
  const bool det = se.det != 0;                         // ... and handed over as this workgroup's slice, which the window's last workgroup adds in slice order
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  if (tid < 8) det_L_[tid] = 0u;
  double* prc = te_lds;
  double* pd = prc + (size_t)d.K * 12;
  for (int k = tid; k < d.K; k += blockDim.x) {
    const double* pose = poses + 7 * k;
    double R[9];
    quat_to_R(pose + 3, R);
#pragma unroll
    for (int i = 0; i < 9; ++i) prc[12 * k + i] = R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) prc[12 * k + 9 + i] = pose[i];
  }
  for (int i = tid; i < 6 * d.np; i += blockDim.x) pd[i] = 0.0;
  __syncthreads();
  double chi = 0, mx = 0;
  const int c0 = BX * se.cpw_t, c1 = min(se.nchunks, c0 + se.cpw_t);
  for (int c = c0 + wave; c < c1; c += nw) {
    if (det) ba_det_publish(det_L, __builtin_amdgcn_readfirstlane(wave), (uint32_t)(c - c0));
    const int e0 = se.chunk_e0[c], e1 = se.chunk_e0[c + 1];
    const int e = e0 + lane;
    const bool have = e < e1;
    int a = 0, k = 1;
    double dl[3] = {0, 0, 0};
    double dp[6] = {0, 0, 0, 0, 0, 0}; int dps = -1;
    if (have) {
      const uint32_t info = se.e_info[e];
      const int p = d.e_point[e];
      const bool act = d.level[e] == 0;
      const double einv = d.e_inv[e];
      const double2 ob = BA_OBS2(d, e);
      a = info & 31; k = (info >> 5) & 31;
      const int s = (int)((info >> 10) & 63) - 1, face = (info >> 16) & 7, kp = (info >> 19) & 255;
      if (act) {
        const double X[3] = {pts[3 * (size_t)p], pts[3 * (size_t)p + 1], pts[3 * (size_t)p + 2]};
        const double* Rt = prc + 12 * kp;
        double R[9], Xc[3], r[2], Jp[12], Jl[6], rho0;
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = Rt[i];
        ba_se_cam_point(Rt, X, Xc);
        edge_error_v(d, face, ob.x, ob.y, Xc, r);
        BA_ERR2_ST(d, e, r[0], r[1]);
        const double c2 = einv * (r[0] * r[0] + r[1] * r[1]);
        double w = 1.0;
        if (robust) w = huber_w(c2, delta, &rho0); else rho0 = c2;
        chi += rho0;
        const double ow = w * einv;
        edge_jac_face(d, face, Xc, R, Jp, Jl);
        if (s >= 0) {
          dps = s;
#pragma unroll
          for (int i = 0; i < 6; ++i) dp[i] = ow * (Jp[i] * Jp[i] + Jp[6 + i] * Jp[6 + i]);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) dl[j] = ow * (Jl[j] * Jl[j] + Jl[3 + j] * Jl[3 + j]);
      }
    }
    if (det) ba_det_wait(det_L, __builtin_amdgcn_readfirstlane(wave), nw, (uint32_t)(c - c0));      // (all lanes are back together here)
    if (dps >= 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) unsafeAtomicAdd(pd + 6 * dps + i, dp[i]);
    }
    // the point's first lane adds its lanes' shares in edge order
    int kmax = k;
    for (int o = 32; o > 0; o >>= 1) kmax = max(kmax, __shfl_xor(kmax, o));
    const bool head = have && a == 0;
    double sl[3] = {dl[0], dl[1], dl[2]};
    for (int dd = 1; dd < kmax; ++dd) {
      const int src = min(lane + dd, 63);
      const double v0 = __shfl(dl[0], src), v1 = __shfl(dl[1], src), v2 = __shfl(dl[2], src);
      if (head && dd < k) { sl[0] += v0; sl[1] += v1; sl[2] += v2; }
    }
    if (head) mx = fmax(mx, fmax(sl[0], fmax(sl[1], sl[2])));
  }
  if (det) ba_det_publish(det_L, __builtin_amdgcn_readfirstlane(wave), BA_DET_DONE);
  const double s1 = block_sum(chi, sh);
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
  __syncthreads();
  if (lane == 0) sh[wave] = mx;
  __syncthreads();                                     // (also: every lane's LDS additions to pd are through)
  // hand-over to the window's last workgroup (kb_ba_first_pass): returning atomics only, every one waited for before the barrier behind which
  // thread 0 takes its ticket
  for (int i = tid; i < 6 * d.np; i += blockDim.x) {           // (up to 62 free key frames: more entries than threads)
    const double v = pd[i];
    // (a non-finite sum stays out, like the fmax of the kernel this one replaced dropped NaNs: lambda's start must not be poisoned by one degenerate edge)
    if (det) {      // this workgroup's slice (parked in `partial` of the Schur kernel, which has not run yet): the last workgroup adds the slices in order
      const unsigned long long o = __hip_atomic_exchange(reinterpret_cast<BA_AS1 unsigned long long*>(se.partial) + ((size_t)BX * 6 * d.np + i),
                                                         (unsigned long long)__double_as_longlong(isfinite(v) ? v : 0.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" :: "v"(o) : "memory");
    } else
    if (v != 0.0 && isfinite(v)) { const double o = atomicAdd(pose_diag + i, v); asm volatile("" :: "v"(o) : "memory"); }
  }
  if (tid == 0) {
    double m = 0;
    for (int i = 0; i < nw; ++i) m = fmax(m, sh[i]);
    m = (isfinite(m) && m > 0.0) ? m : 0.0;                    // the integer maximum below orders bit patterns: a NaN's (above +inf's) must never enter
    const unsigned long long o1 = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(partial + BX), (unsigned long long)__double_as_longlong(s1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long o2 = atomicMax(pt_max, (unsigned long long)__double_as_longlong(m));      // (non-negative doubles order like their bit patterns)
    asm volatile("" :: "v"(o1), "v"(o2) : "memory");
  }
  __syncthreads();
}

// ---- a workgroup's LDS copy of the reduced system -> its slice of `partial`, in the dense pair enumeration the reduction and the solve kernel use
template <bool FUSED>
__device__ __forceinline__ void ba_se_writeout(int slice, int np, int NP2, const double* S, const double* Dg, const BaSeG& se) {
  // One pose pair per wavefront and step, lane = element (0 .. 35 the 6x6 block, 36 .. 41 the right-hand side, 42 .. 47 bp of a diagonal pair):
  // the pair (s1, s2) is wave-uniform and advanced incrementally on the scalar unit, the element's place inside a block is a per-lane constant.
  // (One thread per element of the slice, with a division by 42, a search for s1 and the block-layout arithmetic per element, was ~1300 vector
  // instructions per thread: a sixth of everything the run-major body issues for a workgroup of ten chunks per wavefront.)
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int BX = slice;
  int off_od = -1, off_dg = -1;
  if (lane < 36) {
    const int r = lane / 6, q = lane - 6 * r, lo = min(r, q), hi = max(r, q);
    off_od = ba_se_off(r, q);
    off_dg = lo * 6 - ((lo * (lo - 1)) >> 1) + (hi - lo);
  } else if (lane < 42) off_dg = 21 + (lane - 36);
  else if (FUSED && lane < 48) off_dg = 27 + (lane - 42);
  int s1 = 0, s2 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  while (s2 >= np && s1 < np) { ++s1; s2 = s2 - np + s1; }
  for (int pr = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); pr < NP2; pr += nw) {
    double v = 0.0;
    if (s1 == s2) {
      if (off_dg >= 0) {
#pragma unroll
        for (int cp = 0; cp < BA_SE_DCOPIES; ++cp) v += Dg[((size_t)cp * np + s1) * BA_SE_DSTRIDE + off_dg];
      }
    } else if (off_od >= 0) {
      v = S[(size_t)ba_se_opair(np, s1, s2) * BA_SE_SSTRIDE + off_od];
    }
    if (lane < 42) {
      if (se.gsum) { if (v != 0.0) ba_gadd(&se.partial[(size_t)pr * 42 + lane], v); }
      else se.partial[((size_t)BX * NP2 + pr) * 42 + lane] = v;
    } else if (FUSED && lane < 48 && s1 == s2) {
      if (se.gsum) { if (v != 0.0) ba_gadd(&se.bp_partial[(size_t)s1 * 6 + (lane - 42)], v); }
      else se.bp_partial[((size_t)BX * np + s1) * 6 + (lane - 42)] = v;
    }
    s2 += nw;
    while (s2 >= np && s1 < np) { ++s1; s2 = s2 - np + s1; }
  }
}

// ---- landmark back-substitution + update + residuals at the trial state, EDGE-major (what ba_trial_points_body does with one thread per
// point): the same chunks, lane = edge.  The lanes of a point hand their  ow Jl^T (Jp x_p)  to the point's first lane in edge order
// (same summation order as the per-point loop), that lane solves for the landmark step and updates the point, every lane then evaluates
// its own residual at the trial state.  One thread per point left the chip at 2.7 wavefronts per SIMD walking dependent loads (29 us for
// eight windows against ~10 us of traffic and arithmetic); here every edge is a lane and the key frames' poses sit in LDS.
// partial[BX] = chi2 sum, partial[GX + BX] = gain-denominator sum of this workgroup's points.
#ifndef BA_TE_THREADS
#define BA_TE_THREADS 256
#endif
__device__ __forceinline__ void ba_trial_edges_body(int BX, int GX, BaDevG d, BaSeG se, const double* __restrict__ bl, const double* __restrict__ Hll,
                                                    const double* __restrict__ xp, double lambda, const double* __restrict__ pts, double* __restrict__ pts_new,
                                                    const double* __restrict__ poses_cur, const double* __restrict__ poses_new, int robust, double delta,
                                                    double* __restrict__ partial, bool handover = false) {
  extern __shared__ __align__(16) double te_lds[];     // K x 12 (current poses) | K x 12 (trial poses) | np x 6 (pose update)
  __shared__ double sh[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  double* prc = te_lds;
  double* prn = prc + (size_t)d.K * 12;
  double* xps = prn + (size_t)d.K * 12;
  for (int k = tid; k < 2 * d.K; k += blockDim.x) {
    const int kk = k < d.K ? k : k - d.K;
    const double* pose = (k < d.K ? poses_cur : poses_new) + 7 * kk;
    double* dst = (k < d.K ? prc : prn) + 12 * kk;
    double R[9];
    quat_to_R(pose + 3, R);
#pragma unroll
    for (int i = 0; i < 9; ++i) dst[i] = R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) dst[9 + i] = pose[i];
  }
  for (int i = tid; i < 6 * d.np; i += blockDim.x) xps[i] = xp[i];
  __syncthreads();
  double sc = 0, chi = 0;
  const int c0 = BX * se.cpw_t, c1 = min(se.nchunks, c0 + se.cpw_t);
  // Everything a lane reads from global memory is requested up front, in two waves of loads: the per-edge words (they depend only on the edge
  // index), then what hangs on the point index -- position and, for the point's first lane, Hll and bl.  Left where they are used (weight and
  // observation inside the branches, Hll / bl behind the shuffle loop) they were five dependent round trips per chunk instead of three.
  for (int c = c0 + wave; c < c1; c += nw) {
    const int e0 = se.chunk_e0[c], e1 = se.chunk_e0[c + 1];
    const int e = e0 + lane;
    const bool have = e < e1;
    uint32_t info = 0;
    int p = 0, a = 0, k = 1, kp = 0, face = 0, s = -1;
    bool act = false;
    double X[3] = {0, 0, 0}, cj[3] = {0, 0, 0};
    double ow = 0.0, einv = 0.0;
    double2 ob = make_double2(0.0, 0.0);
    double Hp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bp3[3] = {0, 0, 0};
    if (have) {
      info = se.e_info[e]; p = d.e_point[e];
      act = d.level[e] == 0;
      ow = d.ow[e]; einv = d.e_inv[e];
      ob = BA_OBS2(d, e);
      a = info & 31; k = (info >> 5) & 31; s = (int)((info >> 10) & 63) - 1; face = (info >> 16) & 7; kp = (info >> 19) & 255;
      X[0] = pts[3 * (size_t)p]; X[1] = pts[3 * (size_t)p + 1]; X[2] = pts[3 * (size_t)p + 2];
      if (a == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) Hp[i] = Hll[9 * (size_t)p + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) bp3[i] = bl[3 * (size_t)p + i];
      }
      if (act && s >= 0) {
        const double* Rt = prc + 12 * kp;
        double R[9], Xc[3], Jp[12], Jl[6];
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = Rt[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) Xc[i] = R[3 * i] * X[0] + R[3 * i + 1] * X[1] + R[3 * i + 2] * X[2] + Rt[9 + i];
        edge_jac_face(d, face, Xc, R, Jp, Jl);
        double t0 = 0, t1 = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) { t0 += Jp[i] * xps[6 * s + i]; t1 += Jp[6 + i] * xps[6 * s + i]; }
#pragma unroll
        for (int j = 0; j < 3; ++j) cj[j] = ow * (Jl[j] * t0 + Jl[3 + j] * t1);
      }
    }
    // ---- the point's first lane collects the contributions in edge order
    int kmax = k;
    for (int o = 32; o > 0; o >>= 1) kmax = max(kmax, __shfl_xor(kmax, o));
    const int nact_me = act ? 1 : 0;
    int nact = nact_me;
    double cl[3] = {0, 0, 0};
    const bool head = have && a == 0;
    if (head) { cl[0] = bp3[0] - cj[0]; cl[1] = bp3[1] - cj[1]; cl[2] = bp3[2] - cj[2]; }
    for (int dd = 1; dd < kmax; ++dd) {
      const int src = min(lane + dd, 63);
      const double v0 = __shfl(cj[0], src), v1 = __shfl(cj[1], src), v2 = __shfl(cj[2], src);
      const int na = __shfl(nact_me, src);
      if (head && dd < k) { cl[0] -= v0; cl[1] -= v1; cl[2] -= v2; nact += na; }
    }
    double Xn[3] = {X[0], X[1], X[2]};
    if (head) {
      double D[9], Di[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) D[i] = Hp[i] + ((i & 3) == 0 ? lambda : 0.0);
      inv3(D, Di);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double xl = nact > 0 ? Di[3 * i] * cl[0] + Di[3 * i + 1] * cl[1] + Di[3 * i + 2] * cl[2] : 0.0;
        Xn[i] = X[i] + xl;
        pts_new[3 * (size_t)p + i] = Xn[i];
        sc += xl * (lambda * xl + bp3[i]);
      }
    }
    // ---- every lane gets its point's trial position from the point's first lane, then its own residual at the trial state
    {
      const int src = lane - a;
      Xn[0] = __shfl(Xn[0], src); Xn[1] = __shfl(Xn[1], src); Xn[2] = __shfl(Xn[2], src);
    }
    if (have && act) {
      const double* Rt = prn + 12 * kp;
      double Xc[3], r[2], rho0;
#pragma unroll
      for (int i = 0; i < 3; ++i) Xc[i] = Rt[3 * i] * Xn[0] + Rt[3 * i + 1] * Xn[1] + Rt[3 * i + 2] * Xn[2] + Rt[9 + i];
      edge_error_v(d, face, ob.x, ob.y, Xc, r);
      BA_ERR2_ST(d, e, r[0], r[1]);
      const double c2 = einv * (r[0] * r[0] + r[1] * r[1]);
      if (robust) { huber_w(c2, delta, &rho0); chi += rho0; } else chi += c2;
    }
  }
  for (int i = BX * blockDim.x + tid; i < se.nlone; i += GX * blockDim.x) {      // points nobody observes keep their position (x_l = 0)
    const int p = se.lone[i];
    pts_new[3 * (size_t)p] = pts[3 * (size_t)p]; pts_new[3 * (size_t)p + 1] = pts[3 * (size_t)p + 1]; pts_new[3 * (size_t)p + 2] = pts[3 * (size_t)p + 2];
  }
  const double s1 = block_sum(chi, sh);
  const double s2 = block_sum(sc, sh);
  if (threadIdx.x == 0) {
    if (handover) {
      // (kb_ba_trial_edges, dyn.fold_reduce) the sums are read by another workgroup of this launch: atomic exchanges, and their returned
      // values waited for -- both are performed before this thread takes its ticket
      const unsigned long long o1 = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(partial + BX), (unsigned long long)__double_as_longlong(s1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long o2 = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(partial + GX + BX), (unsigned long long)__double_as_longlong(s2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" :: "v"(o1), "v"(o2) : "memory");
    } else { partial[BX] = s1; partial[GX + BX] = s2; }
  }
}
