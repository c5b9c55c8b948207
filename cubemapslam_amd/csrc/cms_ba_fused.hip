// cms_ba_fused.hip -- the Levenberg trial of the local BA in FOUR launches (windows of up to 27 free key frames):
//
//   k_ba_dinv            per point (Hll + lambda I)^-1 and Dinv*bl                               (cms_ba_kernels.hip)
//   k_ba_schur_chunks    chunk sums of B1 Dinv B2^T over the co-visibility tuples               (cms_ba_kernels.hip)
//   k_ba_trial_solve     assembles the reduced system straight from Hpp / the chunk sums (no Hs matrix in memory),
//                        blocked 6x6 LDL^T with one-step look-ahead, block back substitution, then exp(x)*T for every
//                        free pose and the pose part of the gain denominator
//   k_ba_trial_points    per point: landmark back substitution, X + x_l, and the residuals / robust chi2 of the
//                        point's own edges at the TRIAL state (edges are CSR-sorted by point), block partials
//   k_ba_reduce2         both partial arrays -> chi2(trial), gain denominator
//
// (block_solver.hpp:367-485 Schur + solve + back substitution, sparse_optimizer.cpp:61-114,422-435 update + errors,
//  optimization_algorithm_levenberg.cpp:102-127,182-189.)
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void ba_factor_diag(double* a, int J, double* Ld, double* ybuf, double* idg, int* bad) {
  // The solver is outside the bit-exact part of the path (BA parity bar: 1e-4 relative), so the serial chains use explicit
  // fused multiply-adds and a Newton reciprocal (hardware estimate + two steps, <= 1 ulp) instead of the IEEE division sequence.
  double idl[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const double dc = a[7 * c];
    if (!(isfinite(dc)) || dc == 0.0) *bad = 1;
    double x = __builtin_amdgcn_rcp(dc);
    x = __builtin_fma(__builtin_fma(-dc, x, 1.0), x, x);
    x = __builtin_fma(__builtin_fma(-dc, x, 1.0), x, x);
    idl[c] = x;
    double lc[6];
#pragma unroll
    for (int r = c + 1; r < 6; ++r) lc[r] = a[6 * r + c] * x;
#pragma unroll
    for (int r = c + 1; r < 6; ++r) {
#pragma unroll
      for (int q = c + 1; q <= r; ++q) a[6 * r + q] = __builtin_fma(-lc[r], a[6 * q + c], a[6 * r + q]);   // a[q][c] still unscaled
    }
#pragma unroll
    for (int r = c + 1; r < 6; ++r) a[6 * r + c] = lc[r];
  }
  double z[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) z[r] = ybuf[6 * J + r];     // loads before the stores below (same shared array)
#pragma unroll
  for (int r = 0; r < 6; ++r) {
#pragma unroll
    for (int c = 0; c < r; ++c) z[r] = __builtin_fma(-a[6 * r + c], z[c], z[r]);
  }
#pragma unroll
  for (int r = 1; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < r; ++c) Ld[6 * r + c] = a[6 * r + c];      // strictly lower part is all anybody reads
#pragma unroll
  for (int r = 0; r < 6; ++r) { ybuf[6 * J + r] = z[r]; idg[6 * J + r] = idl[r]; }
}

// T <- exp(x) T for the free poses (types_six_dof_expmap.h:73-76), plain copy for the fixed ones; pose part of the gain denominator and the
// solver status -> scal[5], scal[4] (shared by the two solve kernels)
__device__ __forceinline__ void ba_trial_pose_update(BaDevG d, const double* ybuf, const double* __restrict__ bp, double lambda, const double* __restrict__ poses,
                                                     double* __restrict__ poses_new, double* __restrict__ scal, bool isbad) {
  const int tid = threadIdx.x;
  double sc = 0;
  for (int k = tid; k < d.K; k += blockDim.x) {
    const double* T = poses + 7 * k;
    double* Tn = poses_new + 7 * k;
    const int s = d.pose_slot[k];
    if (s < 0) { for (int i = 0; i < 7; ++i) Tn[i] = T[i]; continue; }
    const double* u = ybuf + 6 * s;
    for (int i = 0; i < 6; ++i) sc += u[i] * (lambda * u[i] + bp[6 * s + i]);
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double Om[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double Om2[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Om2[3 * i + j] = Om[3 * i] * Om[j] + Om[3 * i + 1] * Om[3 + j] + Om[3 * i + 2] * Om[6 + j];
    double R[9], V[9];
    if (theta < 0.00001) {
      for (int i = 0; i < 9; ++i) { R[i] = ((i & 3) == 0 ? 1.0 : 0.0) + Om[i] + Om2[i]; V[i] = R[i]; }
    } else {
      double sn, cs;
      sincos(theta, &sn, &cs);
      const double sa = sn / theta, sb = (1 - cs) / (theta * theta), scc = (theta - sn) / (theta * theta * theta);
      for (int i = 0; i < 9; ++i) {
        const double Id = ((i & 3) == 0 ? 1.0 : 0.0);
        R[i] = Id + sa * Om[i] + sb * Om2[i];
        V[i] = Id + sb * Om[i] + scc * Om2[i];
      }
    }
    double Eq[4], Et[3], RE[9];
    R_to_quat(R, Eq);
    for (int i = 0; i < 3; ++i) Et[i] = V[3 * i] * up[0] + V[3 * i + 1] * up[1] + V[3 * i + 2] * up[2];
    normalize_rot(Eq);
    quat_to_R(Eq, RE);
    for (int i = 0; i < 3; ++i) Tn[i] = Et[i] + RE[3 * i] * T[0] + RE[3 * i + 1] * T[1] + RE[3 * i + 2] * T[2];
    const double* A = Eq; const double* B = T + 3;
    double q[4];
    q[3] = A[3] * B[3] - A[0] * B[0] - A[1] * B[1] - A[2] * B[2];
    q[0] = A[3] * B[0] + A[0] * B[3] + A[1] * B[2] - A[2] * B[1];
    q[1] = A[3] * B[1] + A[1] * B[3] + A[2] * B[0] - A[0] * B[2];
    q[2] = A[3] * B[2] + A[2] * B[3] + A[0] * B[1] - A[1] * B[0];
    normalize_rot(q);
    for (int i = 0; i < 4; ++i) Tn[3 + i] = q[i];
  }
  // pose part of the gain denominator + solver status (status travels as a double next to the other scalars)
  __shared__ double shs[16];
  for (int o = 32; o > 0; o >>= 1) sc += __shfl_xor(sc, o);
  if ((tid & 63) == 0) shs[tid >> 6] = sc;
  __syncthreads();
  if (tid == 0) {
    double s = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += shs[i];
    scal[5] = s;
    int st = isbad ? 0 : 1;
    double stv = 0;
    memcpy(&stv, &st, sizeof(int));
    scal[4] = stv;
  }
}

__device__ __forceinline__ void ba_trial_solve_body(int BX, int GX, BaDevG d, const double* __restrict__ Hpp, const double* __restrict__ bp, double lambda,
                 const int* __restrict__ pair_of_block, const int* __restrict__ pair_chunk_off,
                 const double* __restrict__ chunk_sum, const double* __restrict__ poses, double* __restrict__ poses_new,
                 double* __restrict__ xp_out, double* __restrict__ scal, long long* __restrict__ clk = nullptr) {
  extern __shared__ __align__(16) double sm[];
#define BA_CLK(i) do { if (clk && threadIdx.x == 0) clk[i] = (long long)wall_clock64(); } while (0)
  BA_CLK(0);
  const int nb = d.np, n = 6 * nb, nblk = nb * (nb + 1) / 2;
  // LDS (doubles): panels of L | diagonal factors | W double buffer | y | 1/D            = 36 nblk + 84 nb
  // A panel (column J, rows I = J+1 .. nb-1, m = nb-J-1) is stored "pair-interleaved transposed": element q of row i sits at
  // ((q >> 1) * m + i) * 2 + (q & 1), so that the lanes of a wave (consecutive i) move consecutive 16-byte words -- the
  // row-major block layout cost 3-5 way bank conflicts on every one of the 72 stores of the panel step.
  double* Lp = sm;
  double* Ldg = Lp + 36 * (size_t)(nb * (nb - 1) / 2);
  double* Wbuf = Ldg + 36 * (size_t)nb;
  double* ybuf = Wbuf + 72 * (size_t)nb;
  double* idg = ybuf + n;
  __shared__ int bad;
  const int tid = threadIdx.x;
  // thread <-> block (I, K), I >= K, COLUMN-major: the blocks of one column are consecutive lanes, so the panel of a column
  // lives in one or two waves and the trailing blocks K > J are a dense tail of the thread range
  int I = 0, K = 0;
  const bool have = tid < nblk;
  if (have) {
    int off = 0;
    while (off + (nb - K) <= tid) { off += nb - K; ++K; }
    I = K + (tid - off);
  }
  if (tid == 0) bad = 0;
  // ---- assemble this thread's block of  blockdiag(Hpp + lambda I) - sum_chunks(B1 Dinv B2^T)  and the reduced rhs.
  // A pair has at most ceil(tuples / BA_TUP_CHUNK) chunk sums (<= 5 for a diagonal pair of an 80k-edge window); two chunks
  // (84 loads) are in flight per round trip.
  double a[36];
  if (have) {
#pragma unroll
    for (int q = 0; q < 36; ++q) a[q] = 0.0;
    if (I == K) {
#pragma unroll
      for (int q = 0; q < 36; ++q) a[q] = Hpp[36 * I + q];
#pragma unroll
      for (int r = 0; r < 6; ++r) a[7 * r] += lambda;
    }
    const int pr = pair_of_block[I * (I + 1) / 2 + K];       // the host table is row-major over the lower triangle
    double yb[6] = {0, 0, 0, 0, 0, 0};
    if (pr >= 0) {
      const int c0 = pair_chunk_off[pr], c1 = pair_chunk_off[pr + 1];
      for (int c = c0; c < c1; c += 2) {
        const double* cs0 = chunk_sum + (size_t)c * 42;
        const bool two = c + 1 < c1;
        const double* cs1 = two ? cs0 + 42 : cs0;
        double u[42], v[42];
#pragma unroll
        for (int q = 0; q < 42; ++q) { u[q] = cs0[q]; v[q] = cs1[q]; }
        // chunk sums are stored for the pair (s1 = K) <= (s2 = I), i.e. for block (K, I): transpose into (I, K)
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int q = 0; q < 6; ++q) a[6 * r + q] -= two ? (u[6 * q + r] + v[6 * q + r]) : u[6 * q + r];
        if (I == K) {
#pragma unroll
          for (int r = 0; r < 6; ++r) yb[r] += two ? (u[36 + r] + v[36 + r]) : u[36 + r];
        }
      }
    }
    if (I == K) {
#pragma unroll
      for (int r = 0; r < 6; ++r) ybuf[6 * I + r] = bp[6 * I + r] - yb[r];
    }
  }
  __syncthreads();
  BA_CLK(1);
  if (have && I == 0 && K == 0) ba_factor_diag(a, 0, Ldg, ybuf, idg, &bad);
  __syncthreads();
  for (int J = 0; J < nb && !bad; ++J) {
    const int m = nb - J - 1;                                            // rows below the diagonal of column J
    double2* W2 = reinterpret_cast<double2*>(Wbuf + (J & 1) * 36 * (size_t)nb);
    double2* L2 = reinterpret_cast<double2*>(Lp + 36 * (size_t)(J * nb - J * (J + 1) / 2));
    if (have && K == J && I > J) {                   // panel: W_IJ = A_IJ L_JJ^-T, L_IJ = W_IJ D_J^-1, y_I -= L_IJ z_J
      const double* Ld = Ldg + 36 * (size_t)J;
      // every LDS operand is read into registers first and all results are stored at the end: stores into the shared array
      // between the loads would force the compiler to re-read (it cannot prove the destinations do not alias idg / ybuf)
      double ld[15], idj[6], zj[6], yi[6];
#pragma unroll
      for (int c = 1; c < 6; ++c)
#pragma unroll
        for (int q = 0; q < c; ++q) ld[c * (c - 1) / 2 + q] = Ld[6 * c + q];
#pragma unroll
      for (int c = 0; c < 6; ++c) { idj[c] = idg[6 * J + c]; zj[c] = ybuf[6 * J + c]; yi[c] = ybuf[6 * I + c]; }
      double w[36], lo[36];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double v = a[6 * r + c];
#pragma unroll
          for (int q = 0; q < c; ++q) v = __builtin_fma(-w[6 * r + q], ld[c * (c - 1) / 2 + q], v);
          w[6 * r + c] = v;
        }
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const double l = w[6 * r + c] * idj[c];
          lo[6 * r + c] = l;
          yi[r] = __builtin_fma(-l, zj[c], yi[r]);
        }
      const int i = I - J - 1;
#pragma unroll
      for (int qp = 0; qp < 18; ++qp) {
        W2[qp * m + i] = make_double2(w[2 * qp], w[2 * qp + 1]);
        L2[qp * m + i] = make_double2(lo[2 * qp], lo[2 * qp + 1]);
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) ybuf[6 * I + r] = yi[r];
    }
    __syncthreads();
    if (J == 0) BA_CLK(5);
    if (have && K > J) {                             // trailing update, then look-ahead factorisation of the next diagonal
      const int iw = I - J - 1, ik = K - J - 1;
      double wv[36], lk[36];
#pragma unroll
      for (int qp = 0; qp < 18; ++qp) {
        const double2 x = W2[qp * m + iw], y = L2[qp * m + ik];
        wv[2 * qp] = x.x; wv[2 * qp + 1] = x.y; lk[2 * qp] = y.x; lk[2 * qp + 1] = y.y;
      }
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          double v = a[6 * r + q];
#pragma unroll
          for (int c = 0; c < 6; ++c) v = __builtin_fma(-wv[6 * r + c], lk[6 * q + c], v);
          a[6 * r + q] = v;
        }
      if (I == J + 1 && K == J + 1) {
        if (clk && J == 0) clk[6] = (long long)wall_clock64();
        ba_factor_diag(a, J + 1, Ldg + 36 * (size_t)(J + 1), ybuf, idg, &bad);
        if (clk && J == 0) clk[7] = (long long)wall_clock64();
      }
    }
    __syncthreads();
  }
  const bool isbad = bad != 0;
  BA_CLK(2);
  if (!isbad && tid < 64) {
    for (int i = tid; i < n; i += 64) ybuf[i] *= idg[i];
    __builtin_amdgcn_wave_barrier();
    for (int J = nb - 1; J >= 0; --J) {
      const double* Ld = Ldg + 36 * (size_t)J;
      double xj[6];
#pragma unroll
      for (int r = 5; r >= 0; --r) {
        xj[r] = ybuf[6 * J + r];
#pragma unroll
        for (int q = r + 1; q < 6; ++q) xj[r] -= Ld[6 * q + r] * xj[q];
      }
      __builtin_amdgcn_wave_barrier();
      if (tid == 0) {
#pragma unroll
        for (int r = 0; r < 6; ++r) ybuf[6 * J + r] = xj[r];
      }
      for (int o = tid; o < 6 * J; o += 64) {
        const int Kq = o / 6, c = o - 6 * Kq;
        // L_{J,Kq}: row i = J-Kq-1 of the panel of column Kq
        const int mq = nb - Kq - 1, iq = J - Kq - 1;
        const double* Lq = Lp + 36 * (size_t)(Kq * nb - Kq * (Kq + 1) / 2);
        double v = ybuf[o];
#pragma unroll
        for (int r = 0; r < 6; ++r) { const int q = 6 * r + c; v -= Lq[((q >> 1) * mq + iq) * 2 + (q & 1)] * xj[r]; }
        ybuf[o] = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  if (isbad) for (int i = tid; i < n; i += blockDim.x) ybuf[i] = 0.0;
  __syncthreads();
  BA_CLK(3);
  for (int i = tid; i < n; i += blockDim.x) xp_out[i] = ybuf[i];
  ba_trial_pose_update(d, ybuf, bp, lambda, poses, poses_new, scal, isbad);
  BA_CLK(4);
#undef BA_CLK
}

// ---- the same trial solve with every 6x6 block of the reduced system split over THREE lanes (two rows each).
// One lane per block (above) leaves <= 171 lanes busy with 216 dependent FMAs per panel step: 19 steps of ~2.2 us = 43 of the kernel's 57 us,
// pure latency with the other window groups' kernels waiting behind it.  Here a lane owns rows 2 rp, 2 rp + 1 of block (I, K): the trailing
// update is 72 FMAs per lane, the panel solve 30; the strictly serial part (the 6x6 LDL^T of the next diagonal block) still runs in one
// lane, as a look-ahead behind that lane's own trailing update.  Panels are stored row major, 38 doubles per block (304 B: the three lanes
// of a block read the same L block -- a broadcast -- and neighbouring blocks fall on different banks).
// Round 5 (cycle stamps of thread 0, tools/prof_s3_clk.py, 19 free key frames: 103.7 k cycles = 43 us before, 95.8 k after):
//   * 21 blocks per wavefront (lane 63 idle): a diagonal block's three lanes share a wavefront, so its gather + factorisation follow its own
//     trailing update without a workgroup barrier -- two barriers per panel step instead of three; the wavefront runs at s_setprio 3 meanwhile.
//     (The hoped-for overlap of the factorisation with the OTHER wavefronts' trailing updates is small: a wavefront's own trailing update takes
//     ~1 200 cycles -- 24 16-byte LDS reads with bank conflicts -- whether or not the others run, and the factorisation's 1 260 come behind it.)
//   * the back substitution keeps the right-hand side in registers of wavefront 0 (v_readlane + uniform 6x6 solves, operands requested ahead):
//     17.1 k -> 14.4 k cycles;
//   * R_to_quat without dynamic indexing (it had put the rotation and the quaternion of the pose update into scratch memory).
// Per panel step now: panels ~1 200, trailing update + next diagonal block ~2 400 cycles.
#define BA_S3_STRIDE 38
#define BA_S3_BPW 21          /* blocks (of three lanes) per wavefront */
#define BA_S3_NY 3            /* registers of the back substitution's right-hand side, sixty entries each: 6 np <= 180 (np <= 25, see solve_blk3) */
// threads of the solve kernel for np free key frames: whole wavefronts of BA_S3_BPW blocks
static inline int ba_s3_threads(int np) { return (np * (np + 1) / 2 + BA_S3_BPW - 1) / BA_S3_BPW * 64; }
#ifdef BA_S3_CLK
// developer instrumentation (tools/ab_build.sh s3clk -DBA_S3_CLK): where the solve kernel's time goes, phase by phase (thread 0 of window 0);
// read with cms_ba_debug_s3_clocks
__device__ long long ba_s3_clk[16];
#define BA_S3_STAMP(i) do { if (s3_on) { const long long now_ = (long long)__builtin_readcyclecounter(); s3_acc[i] += now_ - s3_last; s3_last = now_; } } while (0)
#else
#define BA_S3_STAMP(i) do { } while (0)
#endif
__device__ __forceinline__ void ba_trial_solve3_body(BaDevG d, const double* __restrict__ Hpp, const double* __restrict__ bp, double lambda,
                 const int* __restrict__ pair_of_block, const int* __restrict__ pair_chunk_off,
                 const double* __restrict__ chunk_sum, const double* __restrict__ poses, double* __restrict__ poses_new,
                 double* __restrict__ xp_out, double* __restrict__ scal, bool lumped = false, double* partial = nullptr,
                 double* bp_partial = nullptr, int n_slices = 0, int NP2 = 0, bool consume = false, bool presum = false) {
  // lumped: the diagonal blocks of chunk_sum hold S - Hpp and s - bp already (fused linearisation, cms_ba_schur_edges.hip); bp is only
  // read for the gain ratio
  // partial != nullptr (implies lumped): the Schur kernel's range slices [n_slices][NP2][42] are summed here, in slice order, instead of by
  // kb_ba_schur_edges_reduce -- one launch less per round; bp (sum of the slices of bp_partial) is formed in LDS and stored for later readers
  // consume (n_slices == 1): the Schur kernel's workgroups ADDED their copies to the one slice (BaSe::gsum); every element is read by exactly
  // one thread here, which puts the zero back for the next round's additions
  // presum (slices, not consume; the launch carries NP2 x 42 more doubles of LDS): ALL threads first add the slices element by element, in slice order, into
  // LDS -- coalesced loads, many in flight -- and the assembly reads the sums there.  The assembly's own loop walks the slices behind two loads in flight per
  // thread: 66 instead of 42 us for the 16 slices of a deterministic window (profiles/r06_det_experiment.txt)
  extern __shared__ __align__(16) double sm[];
  const int nb = d.np, n = 6 * nb, nblk = nb * (nb + 1) / 2;
  // LDS (doubles): L panels [nb (nb - 1) / 2][38] | diagonal factors [nb][36] | W double buffer [2][nb][38] | diag staging [36] | y [n] | 1/D [n] | bp [n]
  double* Lp = sm;
  double* Ldg = Lp + (size_t)BA_S3_STRIDE * (nb * (nb - 1) / 2);
  double* Wbuf = Ldg + 36 * (size_t)nb;
  double* dstage = Wbuf + 2 * (size_t)BA_S3_STRIDE * nb;
  double* ybuf = dstage + 36;
  double* idg = ybuf + n;
  double* bps = idg + n;
  double* ssum = bps + n;                          // presum: the slices' sums, NP2 x 42
  const bool presummed = partial && presum && !consume && n_slices > 1;
  if (presummed) {
    const int tot = NP2 * 42;
    for (int i = threadIdx.x; i < tot; i += blockDim.x) {
      double v = 0.0;
#pragma unroll 8
      for (int r = 0; r < n_slices; ++r) v += partial[(size_t)r * tot + i];
      ssum[i] = v;
    }
    __syncthreads();
  }
#ifdef BA_S3_CLK
  const bool s3_on = threadIdx.x == 0 && blockIdx.z == 0;
  long long s3_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long s3_last = (long long)__builtin_readcyclecounter();
#endif
  if (partial) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const int s1 = i / 6, c = i - 6 * s1;
      double v = 0.0;
      for (int r = 0; r < n_slices; ++r) v += bp_partial[((size_t)r * nb + s1) * 6 + c];
      if (consume) bp_partial[i] = 0.0;
      bps[i] = v;
      const_cast<double*>(bp)[i] = v;
    }
  }
  __shared__ int bad;
  // lane <-> block: 21 blocks of three lanes per wavefront, lane 63 idle -- the three lanes of a block (of a DIAGONAL block in particular) never
  // sit in two wavefronts, so a diagonal block is gathered and factored inside its wavefront, without a workgroup barrier (below)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bw = (lane * 43) >> 7, rp = lane - 3 * bw;                  // lane / 3, lane % 3 (lane < 64)
  const int blk = BA_S3_BPW * wave + bw;
  int I = 0, K = 0;
  const bool have = lane < 3 * BA_S3_BPW && blk < nblk;
  if (have) {
    int off = 0;
    while (off + (nb - K) <= blk) { off += nb - K; ++K; }
    I = K + (blk - off);
  }
  if (tid == 0) bad = 0;
  const int r0 = 2 * rp;
  double a[12];
  if (have) {
#pragma unroll
    for (int q = 0; q < 12; ++q) a[q] = 0.0;
    if (I == K) {
      if (!lumped) {
#pragma unroll
        for (int q = 0; q < 12; ++q) a[q] = Hpp[36 * I + 6 * r0 + q];
      }
      a[r0] += lambda; a[6 + r0 + 1] += lambda;
    }
    const int pr = pair_of_block[I * (I + 1) / 2 + K];
    double yb[2] = {0, 0};
    if (presummed) {
      const double* cs = ssum + (size_t)pr * 42;      // (LDS)
#pragma unroll
      for (int q = 0; q < 6; ++q) { const double2 v = *reinterpret_cast<const double2*>(cs + 6 * q + r0); a[q] -= v.x; a[6 + q] -= v.y; }
      if (I == K) { const double2 v = *reinterpret_cast<const double2*>(cs + 36 + r0); yb[0] += v.x; yb[1] += v.y; }
    } else if (partial) {
      // dense pair enumeration (se_pob): slice r holds this pair's 42 sums at (r NP2 + pr) 42; two slices' loads are in flight together
      // (the kernel is compiled for up to 1024 threads, 128 registers: four slices spilled).  Rows r0, r0 + 1 of the block are columns r0, r0 + 1
      // of the stored (transposed) block: one 16-byte access per stored row
#pragma unroll 2
      for (int r = 0; r < n_slices; ++r) {
        double* cs = partial + ((size_t)r * NP2 + pr) * 42;
        if (I == K && consume) {
          // a diagonal pair of the ONE global copy: only its upper triangle is complete (the one-wavefront run workgroups add nothing else,
          // cms_ba_schur_runwg.hip; the edge-major write-out adds both): element (r, q) is stored[min][max]
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            a[q] -= cs[6 * min(q, r0) + max(q, r0)];
            a[6 + q] -= cs[6 * min(q, r0 + 1) + max(q, r0 + 1)];
          }
        } else {
#pragma unroll
          for (int q = 0; q < 6; ++q) { const double2 v = *reinterpret_cast<const double2*>(cs + 6 * q + r0); a[q] -= v.x; a[6 + q] -= v.y; }
        }
        if (I == K) { const double2 v = *reinterpret_cast<const double2*>(cs + 36 + r0); yb[0] += v.x; yb[1] += v.y; }
        if (consume) {
#pragma unroll
          for (int q = 0; q < 6; ++q) *reinterpret_cast<double2*>(cs + 6 * q + r0) = make_double2(0.0, 0.0);
          if (I == K) *reinterpret_cast<double2*>(cs + 36 + r0) = make_double2(0.0, 0.0);
        }
      }
    } else if (pr >= 0) {
      for (int c = pair_chunk_off[pr]; c < pair_chunk_off[pr + 1]; ++c) {
        const double* cs = chunk_sum + (size_t)c * 42;
        // chunk sums are stored for the pair (s1 = K) <= (s2 = I), i.e. for block (K, I): transpose into (I, K)
#pragma unroll
        for (int q = 0; q < 6; ++q) { a[q] -= cs[6 * q + r0]; a[6 + q] -= cs[6 * q + r0 + 1]; }
        if (I == K) { yb[0] += cs[36 + r0]; yb[1] += cs[36 + r0 + 1]; }
      }
    }
    if (I == K) {
      ybuf[6 * I + r0] = (lumped ? 0.0 : bp[6 * I + r0]) - yb[0];
      ybuf[6 * I + r0 + 1] = (lumped ? 0.0 : bp[6 * I + r0 + 1]) - yb[1];
    }
  }
  // ordering of LDS accesses between the lanes of ONE wavefront (its LDS operations complete in order; the fences pin the compiler's order)
#define BA_S3_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                               __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
  // a diagonal block is gathered in LDS by its three lanes and factored by the first of them (ba_factor_diag works on a full 6x6): all inside
  // the block's wavefront.  Block (0, 0): wavefront 0, lanes 0 .. 2 -- it also needs every lane's ybuf entries of block 0, which are its own
  if (wave == 0) {
    if (have && I == 0 && K == 0) {
#pragma unroll
      for (int q = 0; q < 12; ++q) dstage[6 * r0 + q] = a[q];
    }
    BA_S3_WAVE_SYNC();
    if (tid == 0) {
      double f[36];
#pragma unroll
      for (int q = 0; q < 36; ++q) f[q] = dstage[q];
      ba_factor_diag(f, 0, Ldg, ybuf, idg, &bad);
    }
  }
  __syncthreads();
  BA_S3_STAMP(0);                                  // assembly (the Schur kernel's sums read, zeros put back) + first diagonal block
  for (int J = 0; J < nb && !bad; ++J) {
    double* Wp = Wbuf + (size_t)(J & 1) * BA_S3_STRIDE * nb;
    double* Lpan = Lp + (size_t)BA_S3_STRIDE * (J * nb - J * (J + 1) / 2);
    if (have && K == J && I > J) {                   // panel rows: W = A L_JJ^-T, L = W D_J^-1, y_I -= L z_J (this lane's two rows)
      const double* Ld = Ldg + 36 * (size_t)J;
      double ld[15], idj[6], zj[6];
#pragma unroll
      for (int c = 1; c < 6; ++c)
#pragma unroll
        for (int q = 0; q < c; ++q) ld[c * (c - 1) / 2 + q] = Ld[6 * c + q];
#pragma unroll
      for (int c = 0; c < 6; ++c) { idj[c] = idg[6 * J + c]; zj[c] = ybuf[6 * J + c]; }
      double yi[2] = {ybuf[6 * I + r0], ybuf[6 * I + r0 + 1]};
      double w[12], lo[12];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double v = a[6 * r + c];
#pragma unroll
          for (int q = 0; q < c; ++q) v = __builtin_fma(-w[6 * r + q], ld[c * (c - 1) / 2 + q], v);
          w[6 * r + c] = v;
        }
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const double l = w[6 * r + c] * idj[c];
          lo[6 * r + c] = l;
          yi[r] = __builtin_fma(-l, zj[c], yi[r]);
        }
      const int i = I - J - 1;
      double2* W2 = reinterpret_cast<double2*>(Wp + (size_t)i * BA_S3_STRIDE + 6 * r0);
      double2* L2 = reinterpret_cast<double2*>(Lpan + (size_t)i * BA_S3_STRIDE + 6 * r0);
#pragma unroll
      for (int q = 0; q < 6; ++q) { W2[q] = make_double2(w[2 * q], w[2 * q + 1]); L2[q] = make_double2(lo[2 * q], lo[2 * q + 1]); }
      ybuf[6 * I + r0] = yi[0]; ybuf[6 * I + r0 + 1] = yi[1];
    }
    __syncthreads();
    BA_S3_STAMP(2);                                // panels
    const bool next_diag = have && I == J + 1 && K == J + 1;
    // the wavefront that holds block (J + 1, J + 1) has the step's serial chain behind its trailing update: it goes first on its SIMD
    const int blkd = (J + 1) * nb - (J + 1) * J / 2, wd = blkd / BA_S3_BPW;
    const bool crit = J + 1 < nb && wave == wd;
    if (crit) __builtin_amdgcn_s_setprio(3);
    if (have && K > J) {                             // trailing update of this lane's two rows: A_IK -= W_IJ L_KJ^T
      const int iw = I - J - 1, ik = K - J - 1;
      double wv[12], lk[36];
      const double2* W2 = reinterpret_cast<const double2*>(Wp + (size_t)iw * BA_S3_STRIDE + 6 * r0);
      const double2* L2 = reinterpret_cast<const double2*>(Lpan + (size_t)ik * BA_S3_STRIDE);
#pragma unroll
      for (int q = 0; q < 6; ++q) { const double2 x = W2[q]; wv[2 * q] = x.x; wv[2 * q + 1] = x.y; }
#pragma unroll
      for (int q = 0; q < 18; ++q) { const double2 y = L2[q]; lk[2 * q] = y.x; lk[2 * q + 1] = y.y; }
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          double v = a[6 * r + q];
#pragma unroll
          for (int c = 0; c < 6; ++c) v = __builtin_fma(-wv[6 * r + c], lk[6 * q + c], v);
          a[6 * r + q] = v;
        }
      if (next_diag) {
#pragma unroll
        for (int q = 0; q < 12; ++q) dstage[6 * r0 + q] = a[q];
      }
    }
    // look-ahead factorisation of block (J + 1, J + 1) by the first of its lanes, inside its wavefront: no barrier in front of it, and the other
    // wavefronts' trailing updates (the phase is bound by LDS bandwidth) run beside its serial chain
    if (crit) {
      BA_S3_WAVE_SYNC();
      if (lane == 3 * (blkd - BA_S3_BPW * wd)) {
        double f[36];
#pragma unroll
        for (int q = 0; q < 36; ++q) f[q] = dstage[q];
        ba_factor_diag(f, J + 1, Ldg + 36 * (size_t)(J + 1), ybuf, idg, &bad);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    BA_S3_STAMP(3);                                // trailing updates + next diagonal block
  }
  const bool isbad = bad != 0;
  // back substitution L^T x = D^-1 y by wavefront 0, the right-hand side in REGISTERS: entry o = 60 g + lane of register g (ten block rows per
  // register, lanes 60 .. 63 idle: a block row never straddles two registers).  Per block row J the six unknowns come out of their lanes with
  // v_readlane, every lane runs the 6x6 triangular solve on them (uniform values; the farthest unknown first: five dependent steps, not
  // fifteen), lane 0 stores the solution, and every lane subtracts L_{J,Kq}^T x_J from its entries above.  The L operands do not depend on x: a
  // step's are requested at its top (the diagonal factor one step ahead), from addresses that are valid for every lane -- no predicated loads.
  // (Through ybuf in LDS it was 900 cycles per block row: three dependent round trips.)
  if (!isbad && wave == 0) {
    const int ln = min(lane, 59);
    double y[BA_S3_NY];
    const double* Lo[BA_S3_NY];
#pragma unroll
    for (int g = 0; g < BA_S3_NY; ++g) {
      const int o = 60 * g + ln;
      y[g] = o < n ? ybuf[o] * idg[o] : 0.0;
      const int Kq = o / 6, c = o - 6 * Kq;
      // L_{J,Kq}[r][c]: block J - Kq - 1 of the panel of column Kq, row major -> Lo + J stride + 6 r
      Lo[g] = Lp + (ptrdiff_t)BA_S3_STRIDE * (Kq * nb - Kq * (Kq + 1) / 2 - Kq - 1) + c;
    }
    const double* Lsafe = Lp - BA_S3_STRIDE;         // + J stride: block (J, 0), there for every J >= 1
    double nld[15];
    auto fetch = [&](int J) {                        // the diagonal factor of block row J
      const double* Ld = Ldg + 36 * (size_t)J;
#pragma unroll
      for (int q = 1; q < 6; ++q)
#pragma unroll
        for (int r = 0; r < q; ++r) nld[q * (q - 1) / 2 + r] = Ld[6 * q + r];
    };
    fetch(nb - 1);
#pragma unroll
    for (int g = BA_S3_NY - 1; g >= 0; --g) {
      for (int J = min(nb, 10 * (g + 1)) - 1; J >= 10 * g; --J) {
        double ld[15], u[BA_S3_NY][6];
#pragma unroll
        for (int i = 0; i < 15; ++i) ld[i] = nld[i];
        bool upd[BA_S3_NY];
#pragma unroll
        for (int h = 0; h <= g; ++h) {
          upd[h] = lane < 60 && 60 * h + lane < 6 * J;
          const double* src = (upd[h] ? Lo[h] : Lsafe) + (ptrdiff_t)BA_S3_STRIDE * max(J, 1);
#pragma unroll
          for (int r = 0; r < 6; ++r) u[h][r] = src[6 * r];
        }
        if (J > 0) fetch(J - 1);
        double xj[6];
        const int sl = 6 * (J - 10 * g);             // (uniform)
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const int lo_ = __builtin_amdgcn_readlane(__double2loint(y[g]), sl + r), hi_ = __builtin_amdgcn_readlane(__double2hiint(y[g]), sl + r);
          xj[r] = __hiloint2double(hi_, lo_);
        }
#pragma unroll
        for (int r = 4; r >= 0; --r)
#pragma unroll
          for (int q = 5; q > r; --q) xj[r] = __builtin_fma(-ld[q * (q - 1) / 2 + r], xj[q], xj[r]);
        if (lane == 0) {
#pragma unroll
          for (int r = 0; r < 6; ++r) ybuf[6 * J + r] = xj[r];
        }
#pragma unroll
        for (int h = 0; h <= g; ++h) {
          double v = y[h];
#pragma unroll
          for (int r = 0; r < 6; ++r) v = __builtin_fma(-u[h][r], xj[r], v);
          y[h] = upd[h] ? v : y[h];
        }
      }
    }
  }
  __syncthreads();
  BA_S3_STAMP(5);                                  // back substitution
  if (isbad) for (int i = tid; i < n; i += blockDim.x) ybuf[i] = 0.0;
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) xp_out[i] = ybuf[i];
  ba_trial_pose_update(d, ybuf, partial ? bps : bp, lambda, poses, poses_new, scal, isbad);      // (bps was complete before the first barrier above)
  BA_S3_STAMP(6);                                  // pose update
#ifdef BA_S3_CLK
  if (s3_on) { for (int i = 0; i < 10; ++i) ba_s3_clk[i] = s3_acc[i]; ba_s3_clk[10] = nb; }
#endif
#undef BA_S3_WAVE_SYNC
}

// per point: x_l = Dinv (b_l - sum B^T x_p), X_new = X + x_l, gain-denominator partial; then residuals + robust chi2 of the
// point's edges at the trial state.  partial[blk] = chi2 sum, partial[nblk + blk] = denominator sum.
__device__ __forceinline__ void ba_trial_points_body(int BX, int GX, BaDevG d, const double* __restrict__ bl, const double* __restrict__ Hpl, const double* __restrict__ Dinv,
                  const double* __restrict__ xp, double lambda, const double* __restrict__ pts, double* __restrict__ pts_new,
                  const double* __restrict__ poses_new, int robust, double delta, double* __restrict__ partial,
                  const double* __restrict__ Hll = nullptr, const double* __restrict__ poses_cur = nullptr) {
  // poses_cur != nullptr: B^T xp of an edge is evaluated from its Jacobians at the linearisation point and the stored weight
  // (Jl^T ow (Jp xp)) instead of reading the 6x3 block
  // Hll != nullptr: (Hll + lambda I)^-1 is evaluated here (same arithmetic as k_ba_dinv) and `Dinv` is not read -- the batched
  // driver then needs no separate inversion kernel
  __shared__ double sh[16];
  const int p = BX * blockDim.x + threadIdx.x;
  double sc = 0, chi = 0;
  if (p < d.P) {
    const int e0 = d.pt_off[p], e1 = d.pt_off[p + 1];
    double cl[3] = {bl[3 * (size_t)p], bl[3 * (size_t)p + 1], bl[3 * (size_t)p + 2]};
    int nact = 0;
    for (int a = e0; a < e1; ++a) {
      if (d.level[a] != 0) continue;
      ++nact;
      const int s = d.pose_slot[d.e_pose[a]];
      if (s < 0) continue;
      if (poses_cur) {
        const double* pose = poses_cur + 7 * d.e_pose[a];
        double R[9], Xc[3], Jp[12], Jl[6];
        quat_to_R(pose + 3, R);
        cam_point(pose, R, pts + 3 * (size_t)p, Xc);
        edge_jac(d, a, Xc, R, Jp, Jl);
        double t0 = 0, t1 = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) { t0 += Jp[i] * xp[6 * s + i]; t1 += Jp[6 + i] * xp[6 * s + i]; }
        const double ow = d.ow[a];
#pragma unroll
        for (int j = 0; j < 3; ++j) cl[j] -= ow * (Jl[j] * t0 + Jl[3 + j] * t1);
      } else {
        const double2* q = reinterpret_cast<const double2*>(Hpl + 18 * (size_t)a);   // nine 16-byte loads (see ba_lin_points_body)
        double B[18];
#pragma unroll
        for (int i = 0; i < 9; ++i) { const double2 u = q[i]; B[2 * i] = u.x; B[2 * i + 1] = u.y; }
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int i = 0; i < 6; ++i) cl[j] -= B[3 * i + j] * xp[6 * s + i];
      }
    }
    double Dl[9];
    if (Hll) {
      double D[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) D[i] = Hll[9 * (size_t)p + i] + ((i & 3) == 0 ? lambda : 0.0);
      inv3(D, Dl);
    }
    const double* Di = Hll ? Dl : Dinv + 9 * (size_t)p;
    double X[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double xl = nact > 0 ? Di[3 * i] * cl[0] + Di[3 * i + 1] * cl[1] + Di[3 * i + 2] * cl[2] : 0.0;
      X[i] = pts[3 * (size_t)p + i] + xl;
      pts_new[3 * (size_t)p + i] = X[i];
      sc += xl * (lambda * xl + bl[3 * (size_t)p + i]);
    }
    for (int e = e0; e < e1; ++e) {
      if (d.level[e] != 0) continue;
      const double* pose = poses_new + 7 * d.e_pose[e];
      double R[9], Xc[3], r[2], rho0;
      quat_to_R(pose + 3, R);
      cam_point(pose, R, X, Xc);
      edge_error(d, e, Xc, r);
      BA_ERR2_ST(d, e, r[0], r[1]);
      const double c2 = d.e_inv[e] * (r[0] * r[0] + r[1] * r[1]);
      if (robust) { huber_w(c2, delta, &rho0); chi += rho0; } else chi += c2;
    }
  }
  const double s1 = block_sum(chi, sh);
  const double s2 = block_sum(sc, sh);
  if (threadIdx.x == 0) { partial[BX] = s1; partial[GX + BX] = s2; }
}

// scal[1] = sum(partial[0..n)), scal[2] = sum(partial[n..2n)) + scal[5]
__device__ __forceinline__ void ba_reduce2_body(int BX, int GX, const double* __restrict__ partial, int n, double* __restrict__ scal,
                                                double* __restrict__ hscal = nullptr) {
  __shared__ double sh[16];
  double v1 = 0, v2 = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { v1 += partial[i]; v2 += partial[n + i]; }
  const double s1 = block_sum(v1, sh);
  const double s2 = block_sum(v2, sh);
  if (threadIdx.x == 0) {
    const double den = s2 + scal[5];
    scal[1] = s1; scal[2] = den;
    if (hscal) { hscal[1] = s1; hscal[2] = den; hscal[4] = scal[4]; }   // pinned host mirror (batched driver)
  }
}
