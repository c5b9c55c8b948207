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

__device__ __forceinline__ void ba_factor_diag(double* a, int J, int tid_blk, double* Lblk, double* ybuf, double* idg, int* bad) {
  double idl[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const double dc = a[7 * c];
    if (!(isfinite(dc)) || dc == 0.0) *bad = 1;
    idl[c] = 1.0 / dc;
    double lc[6];
#pragma unroll
    for (int r = c + 1; r < 6; ++r) lc[r] = a[6 * r + c] * idl[c];
#pragma unroll
    for (int r = c + 1; r < 6; ++r) {
#pragma unroll
      for (int q = c + 1; q <= r; ++q) a[6 * r + q] -= lc[r] * lc[q] * dc;
      a[6 * r + c] = lc[r];
    }
  }
  double* Ld = Lblk + 36 * (size_t)tid_blk;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) Ld[6 * r + c] = c < r ? a[6 * r + c] : (c == r ? 1.0 : 0.0);
  double z[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    z[r] = ybuf[6 * J + r];
#pragma unroll
    for (int c = 0; c < r; ++c) z[r] -= a[6 * r + c] * z[c];
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) { ybuf[6 * J + r] = z[r]; idg[6 * J + r] = idl[r]; }
}

__device__ __forceinline__ void ba_trial_solve_body(int BX, int GX, BaDev d, const double* __restrict__ Hpp, const double* __restrict__ bp, double lambda,
                 const int* __restrict__ pair_of_block, const int* __restrict__ pair_chunk_off,
                 const double* __restrict__ chunk_sum, const double* __restrict__ poses, double* __restrict__ poses_new,
                 double* __restrict__ xp_out, double* __restrict__ scal) {
  extern __shared__ __align__(16) double sm[];
  const int nb = d.np, n = 6 * nb, nblk = nb * (nb + 1) / 2;
  double* Lblk = sm;
  double* Wbuf = Lblk + 36 * (size_t)nblk;
  double* ybuf = Wbuf + 72 * (size_t)nb;
  double* idg = ybuf + n;
  __shared__ int bad;
  const int tid = threadIdx.x;
  int I = 0, K = 0;
  const bool have = tid < nblk;
  if (have) {
    I = (int)((sqrt(8.0 * tid + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= tid) ++I;
    while (I * (I + 1) / 2 > tid) --I;
    K = tid - I * (I + 1) / 2;
  }
  if (tid == 0) bad = 0;
  // ---- assemble this thread's block of  blockdiag(Hpp + lambda I) - sum_chunks(B1 Dinv B2^T)  and the reduced rhs
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
    const int pr = pair_of_block[tid];
    double yb[6] = {0, 0, 0, 0, 0, 0};
    if (pr >= 0) {
      for (int c = pair_chunk_off[pr]; c < pair_chunk_off[pr + 1]; ++c) {
        const double* cs = chunk_sum + (size_t)c * 42;
        // chunk sums are stored for the pair (s1 = K) <= (s2 = I), i.e. for block (K, I): transpose into (I, K)
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int q = 0; q < 6; ++q) a[6 * r + q] -= cs[6 * q + r];
        if (I == K) {
#pragma unroll
          for (int r = 0; r < 6; ++r) yb[r] += cs[36 + r];
        }
      }
    }
    if (I == K) {
#pragma unroll
      for (int r = 0; r < 6; ++r) ybuf[6 * I + r] = bp[6 * I + r] - yb[r];
    }
  }
  __syncthreads();
  if (have && I == 0 && K == 0) ba_factor_diag(a, 0, tid, Lblk, ybuf, idg, &bad);
  __syncthreads();
  for (int J = 0; J < nb && !bad; ++J) {
    double* W = Wbuf + (J & 1) * 36 * (size_t)nb;
    if (have && K == J && I > J) {                   // panel: W_IJ = A_IJ L_JJ^-T, L_IJ = W_IJ D_J^-1, y_I -= L_IJ z_J
      const double* Ld = Lblk + 36 * (size_t)(J * (J + 1) / 2 + J);
      double w[36];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double v = a[6 * r + c];
#pragma unroll
          for (int q = 0; q < c; ++q) v -= w[6 * r + q] * Ld[6 * c + q];
          w[6 * r + c] = v;
        }
      double* Wd = W + 36 * (size_t)I;
      double* Lo = Lblk + 36 * (size_t)tid;
      double yi[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) yi[r] = ybuf[6 * I + r];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const double l = w[6 * r + c] * idg[6 * J + c];
          Wd[6 * r + c] = w[6 * r + c];
          Lo[6 * r + c] = l;
          yi[r] -= l * ybuf[6 * J + c];
        }
#pragma unroll
      for (int r = 0; r < 6; ++r) ybuf[6 * I + r] = yi[r];
    }
    __syncthreads();
    if (have && K > J) {                             // trailing update, then look-ahead factorisation of the next diagonal
      const double* Wi = W + 36 * (size_t)I;
      const double* Lk = Lblk + 36 * (size_t)(K * (K + 1) / 2 + J);
      double lk[36];
#pragma unroll
      for (int q = 0; q < 36; ++q) lk[q] = Lk[q];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double wr[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) wr[c] = Wi[6 * r + c];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          double v = a[6 * r + q];
#pragma unroll
          for (int c = 0; c < 6; ++c) v -= wr[c] * lk[6 * q + c];
          a[6 * r + q] = v;
        }
      }
      if (I == J + 1 && K == J + 1) ba_factor_diag(a, J + 1, tid, Lblk, ybuf, idg, &bad);
    }
    __syncthreads();
  }
  const bool isbad = bad != 0;
  if (!isbad && tid < 64) {
    for (int i = tid; i < n; i += 64) ybuf[i] *= idg[i];
    __builtin_amdgcn_wave_barrier();
    for (int J = nb - 1; J >= 0; --J) {
      const double* Ld = Lblk + 36 * (size_t)(J * (J + 1) / 2 + J);
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
        const double* Ljk = Lblk + 36 * (size_t)(J * (J + 1) / 2 + Kq);
        double v = ybuf[o];
#pragma unroll
        for (int r = 0; r < 6; ++r) v -= Ljk[6 * r + c] * xj[r];
        ybuf[o] = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  if (isbad) for (int i = tid; i < n; i += blockDim.x) ybuf[i] = 0.0;
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) xp_out[i] = ybuf[i];
  // ---- T <- exp(x) T for the free poses (types_six_dof_expmap.h:73-76), plain copy for the fixed ones
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
      const double sa = sin(theta) / theta, sb = (1 - cos(theta)) / (theta * theta), scc = (theta - sin(theta)) / (theta * theta * theta);
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
  __shared__ double shs[8];
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

// per point: x_l = Dinv (b_l - sum B^T x_p), X_new = X + x_l, gain-denominator partial; then residuals + robust chi2 of the
// point's edges at the trial state.  partial[blk] = chi2 sum, partial[nblk + blk] = denominator sum.
__device__ __forceinline__ void ba_trial_points_body(int BX, int GX, BaDev d, const double* __restrict__ bl, const double* __restrict__ Hpl, const double* __restrict__ Dinv,
                  const double* __restrict__ xp, double lambda, const double* __restrict__ pts, double* __restrict__ pts_new,
                  const double* __restrict__ poses_new, int robust, double delta, double* __restrict__ partial) {
  __shared__ double sh[4];
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
      const double* B = Hpl + 18 * (size_t)a;
      for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 6; ++i) cl[j] -= B[3 * i + j] * xp[6 * s + i];
    }
    const double* Di = Dinv + 9 * (size_t)p;
    double X[3];
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
      d.err[2 * e] = r[0]; d.err[2 * e + 1] = r[1];
      const double c2 = d.e_inv[e] * (r[0] * r[0] + r[1] * r[1]);
      if (robust) { huber_w(c2, delta, &rho0); chi += rho0; } else chi += c2;
    }
  }
  const double s1 = block_sum(chi, sh);
  const double s2 = block_sum(sc, sh);
  if (threadIdx.x == 0) { partial[BX] = s1; partial[GX + BX] = s2; }
}

// scal[1] = sum(partial[0..n)), scal[2] = sum(partial[n..2n)) + scal[5]
__device__ __forceinline__ void ba_reduce2_body(int BX, int GX, const double* __restrict__ partial, int n, double* __restrict__ scal) {
  __shared__ double sh[4];
  double v1 = 0, v2 = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { v1 += partial[i]; v2 += partial[n + i]; }
  const double s1 = block_sum(v1, sh);
  const double s2 = block_sum(v2, sh);
  if (threadIdx.x == 0) { scal[1] = s1; scal[2] = s2 + scal[5]; }
}
