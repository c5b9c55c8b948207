// cms_pose_opt.hip -- Optimizer::PoseOptimization (src/Optimizer.cpp:48-190) as ONE kernel launch for a batch of frames.
//
// The reference runs, per tracked frame and 1-3 times per frame (Tracking.cpp:585,647,688), a pose-only Levenberg-Marquardt over
// N unary multi-pinhole edges (EdgeSE3ProjectXYZMultiPinholeOnlyPose, g2o_cubemap_vertices_edges.cpp:61-134; quadratic form
// base_unary_edge.hpp:43-72): 4 rounds x optimize(10) from the same initial pose, every edge re-classified with chi2 > 5.991
// after each round, Huber dropped for the last round, BlockSolver_6_3 without Schur + dense LDL^T on the 6x6 system.
//
// That is a latency problem (a few hundred edges, ~40 dependent linearise / trial passes), so the whole procedure --
// including the accept / reject logic of optimization_algorithm_levenberg.cpp:61-164 -- runs inside one workgroup per frame
// with no host round trip: thread 0 owns the LM state and the 6x6 solve, all 256 threads share the edge passes, a
// reduce-scatter butterfly folds the 28 partial sums (21 H + 6 b + chi2) per pass.  Frames of a batch (camera streams, or the
// 2-3 calls of one frame's tracking step replayed) are independent workgroups of the same launch.
#include <hip/hip_runtime.h>
#include <stdint.h>

struct PoseDev {
  int nf;
  const int* off;                 // nf + 1: edge range of every frame
  const double* Xw;               // E x 3 world points (never moved)
  const double* obs;              // E x 2 measurement inside its face
  const double* inv;              // E     invSigma2 of the key point's octave
  const int8_t* face;             // E
  uint8_t* outlier;               // E out: pFrame->mvbOutlier (== g2o level of the edge)
  double* err;                    // E x 2 scratch: the persistent _error of every edge
  double* poses;                  // nf x 7 in / out
  int* result;                    // nf x 8: [0] inliers returned, [1] nBad, [2] rounds, [3] reserved, [4..8) iterations per round
  double fx, fy, cx, cy;
};

// T <- exp(u) * T  (types_six_dof_expmap.h:73-76, se3quat.h:217-257); same arithmetic as the pose update of k_ba_trial_solve
__device__ __forceinline__ void pose_exp_mul(const double* u, const double* T, double* Tn) {
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
    sincos(theta, &sn, &cs);            // one argument reduction for both (this runs in one lane, in the serial part of every trial)
    const double it = 1.0 / theta, it2 = it * it;
    const double sa = sn * it, sb = (1 - cs) * it2, scc = (theta - sn) * (it2 * it);
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

// dense LDL^T of the 6x6 system (LinearSolverDense); false on a zero / non-finite pivot (-> the trial is rejected)
__device__ __forceinline__ bool pose_solve6(const double* H, const double* b, double lambda, double* x) {
  double A[36], D[6], RD[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) A[i] = H[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) A[7 * i] += lambda;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[7 * j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= A[6 * j + k] * A[6 * j + k] * D[k];
    if (!isfinite(d) || d == 0.0) ok = false;
    D[j] = d;
    const double rd = 1.0 / d;             // one division per pivot; the column is scaled by the reciprocal (<= 1 ulp from the quotient)
    RD[j] = rd;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double s = A[6 * i + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= A[6 * i + k] * A[6 * j + k] * D[k];
      A[6 * i + j] = s * rd;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = b[i];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int k = 0; k < i; ++k) x[i] -= A[6 * i + k] * x[k];
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] *= RD[i];
#pragma unroll
  for (int i = 5; i >= 0; --i)
#pragma unroll
    for (int k = i + 1; k < 6; ++k) x[i] -= A[6 * k + i] * x[k];
  return ok;
}

// sum of acc[0..28) over the workgroup -> out[0..28) in LDS (valid for every thread after the call)
__device__ __forceinline__ void pose_block_sum28(const double* acc, double (*sh)[28], double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double v14[14], v7[7], v4[4], v2[2], v1[1], v0[1];
  rs_step<28, 14>(acc, v14, (lane & 32) != 0, 32);
  rs_step<14, 7>(v14, v7, (lane & 16) != 0, 16);
  rs_step<7, 4>(v7, v4, (lane & 8) != 0, 8);
  rs_step<4, 2>(v4, v2, (lane & 4) != 0, 4);
  rs_step<2, 1>(v2, v1, (lane & 2) != 0, 2);
  rs_step<1, 1>(v1, v0, (lane & 1) != 0, 1);
  int idx = 0, s = 28;
  { const bool h = lane & 32; idx += h ? 14 : 0; s = h ? max(0, s - 14) : min(14, s); }
  { const bool h = lane & 16; idx += h ? 7 : 0; s = h ? max(0, s - 7) : min(7, s); }
  { const bool h = lane & 8; idx += h ? 4 : 0; s = h ? max(0, s - 4) : min(4, s); }
  { const bool h = lane & 4; idx += h ? 2 : 0; s = h ? max(0, s - 2) : min(2, s); }
  { const bool h = lane & 2; idx += h ? 1 : 0; s = h ? max(0, s - 1) : min(1, s); }
  { const bool h = lane & 1; idx += h ? 1 : 0; s = h ? max(0, s - 1) : min(1, s); }
  __syncthreads();                       // previous readers of sh / out are done
  if (s > 0) sh[wave][idx] = v0[0];
  __syncthreads();
  if (threadIdx.x < 28) out[threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
  __syncthreads();
}

// (generic variant: edges stay in global memory; used for frames with more than 256 x PO_MAXJ edges)
extern "C" __global__ void __launch_bounds__(256) k_pose_optimize_g(PoseDev P) {
  __shared__ double sh28[4][28];
  __shared__ double sum[28];            // [0..21) upper triangle of H row by row, [21..27) b, [27] chi2
  __shared__ double s_pose[7], s_trial[7];
  __shared__ int s_ctl[2];              // [0] another trial, [1] another iteration
  const int f = blockIdx.x, tid = threadIdx.x;
  const int e0 = P.off[f], e1 = P.off[f + 1], n = e1 - e0;
  int* res = P.result + 8 * f;
  for (int e = e0 + tid; e < e1; e += 256) P.outlier[e] = 0;   // pFrame->mvbOutlier[i] = false (Optimizer.cpp:91)
  if (n < 3) {                          // Optimizer.cpp:131-132: pose untouched, 0 returned
    if (tid < 8) res[tid] = 0;
    return;
  }
  BaDev d;                              // the edge arithmetic of the local BA (cms_ba_kernels.hip) on this frame's arrays
  d.e_face = P.face; d.e_obs = P.obs; d.e_inv = P.inv; d.fx = P.fx; d.fy = P.fy; d.cx = P.cx; d.cy = P.cy;
  const double delta = sqrt(5.991);
  double pose0[7];
  {
    const double* p = P.poses + 7 * f;
    for (int i = 0; i < 7; ++i) pose0[i] = p[i];
    normalize_rot(pose0 + 3);           // SE3Quat constructor (se3quat.h:58-64)
  }
  int nBad = 0, rounds = 0;
  int its[4] = {0, 0, 0, 0};
  for (int round = 0; round < 4; ++round) {
    const int robust = round < 3;
    if (tid < 7) s_pose[tid] = pose0[tid];                      // setEstimate(pFrame->mTcw) before every round
    int mine = 0;
    for (int e = e0 + tid; e < e1; e += 256) mine += P.outlier[e] == 0;
    const int nact = __syncthreads_count(mine > 0);             // also publishes s_pose
    int done = 0;
    if (nact > 0) {                                             // no level-0 edge -> no active vertex -> optimize() returns at once
      double lambda = -1, ni = 2, currentChi = 0, iniChi = 0, rho = 0;
      int nBadIt = 0;
      for (int it = 0; it < 10; ++it) {
        // ---- computeActiveErrors + activeRobustChi2 + buildSystem at s_pose
        double acc[28];
#pragma unroll
        for (int i = 0; i < 28; ++i) acc[i] = 0;
        {
          double R[9];
          quat_to_R(s_pose + 3, R);
          for (int e = e0 + tid; e < e1; e += 256) {
            if (P.outlier[e]) continue;
            double Xc[3], r[2], Jp[12], Jl[6];
            cam_point(s_pose, R, P.Xw + 3 * (size_t)e, Xc);
            edge_error(d, e, Xc, r);
            P.err[2 * (size_t)e] = r[0]; P.err[2 * (size_t)e + 1] = r[1];
            const double om = P.inv[e], c2 = om * (r[0] * r[0] + r[1] * r[1]);
            double w = 1.0, rho0 = c2;
            if (robust) w = huber_w(c2, delta, &rho0);
            edge_jac(d, e, Xc, R, Jp, Jl);
            const double ow = w * om;
            int k = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
              for (int j = i; j < 6; ++j) acc[k++] += ow * (Jp[i] * Jp[j] + Jp[6 + i] * Jp[6 + j]);
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[21 + i] -= ow * (Jp[i] * r[0] + Jp[6 + i] * r[1]);
            acc[27] += rho0;
          }
        }
        pose_block_sum28(acc, sh28, sum);
        double H[36], b[6];
        if (tid == 0) {
          int k = 0;
          for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j) { H[6 * i + j] = sum[k]; H[6 * j + i] = sum[k]; ++k; }
          for (int i = 0; i < 6; ++i) b[i] = sum[21 + i];
          currentChi = sum[27]; iniChi = currentChi;
          if (it == 0) {
            double md = 0;
            for (int j = 0; j < 6; ++j) md = fmax(fabs(H[7 * j]), md);
            lambda = 1e-5 * md; ni = 2; nBadIt = 0;
          }
          rho = 0;
        }
        int qmax = 0;
        for (;;) {   // do { } while (rho < 0 && qmax < 10)
          double scale = 0;
          bool ok2 = true;
          if (tid == 0) {
            double x[6];
            ok2 = pose_solve6(H, b, lambda, x);
            if (!ok2) for (int i = 0; i < 6; ++i) x[i] = 0;
            double Tn[7];
            pose_exp_mul(x, s_pose, Tn);
            for (int i = 0; i < 7; ++i) s_trial[i] = Tn[i];
            for (int j = 0; j < 6; ++j) scale += x[j] * (lambda * x[j] + b[j]);
          }
          __syncthreads();
          // ---- computeActiveErrors at the trial pose: the stored errors ARE the trial's from here on (kept on rejection, g2o)
          double chi = 0;
          {
            double R[9];
            quat_to_R(s_trial + 3, R);
            for (int e = e0 + tid; e < e1; e += 256) {
              if (P.outlier[e]) continue;
              double Xc[3], r[2];
              cam_point(s_trial, R, P.Xw + 3 * (size_t)e, Xc);
              edge_error(d, e, Xc, r);
              P.err[2 * (size_t)e] = r[0]; P.err[2 * (size_t)e + 1] = r[1];
              const double c2 = P.inv[e] * (r[0] * r[0] + r[1] * r[1]);
              double rho0 = c2;
              if (robust) huber_w(c2, delta, &rho0);
              chi += rho0;
            }
          }
          const double tempChi0 = block_sum(chi, &sh28[0][0]);
          if (tid == 0) {
            double tempChi = ok2 ? tempChi0 : 1.7976931348623157e308;
            rho = (currentChi - tempChi) / (scale + 1e-3);
            if (rho > 0 && isfinite(tempChi)) {
              const double t21 = 2 * rho - 1; double alpha = 1. - t21 * t21 * t21;     // pow(2 rho - 1, 3): the cube, two roundings instead of a ~200-instruction pow in the serial part
              alpha = fmin(alpha, 2. / 3.);
              lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi;
              for (int i = 0; i < 7; ++i) s_pose[i] = s_trial[i];
            } else {
              lambda *= ni; ni *= 2;
            }
            ++qmax;
            s_ctl[0] = (rho < 0 && qmax < 10) ? 1 : 0;
          }
          __syncthreads();
          if (!s_ctl[0]) break;
        }
        ++done;
        if (tid == 0) {
          bool stop = (qmax == 10 || rho == 0);
          if (!stop) {
            if ((iniChi - currentChi) * 1e3 < iniChi) ++nBadIt; else nBadIt = 0;
            if (nBadIt >= 3) stop = true;
          }
          s_ctl[1] = stop ? 0 : 1;
        }
        __syncthreads();
        if (!s_ctl[1]) break;
      }
    }
    its[round] = done;
    // ---- re-classification (Optimizer.cpp:142-173): former outliers get a fresh error at the round's final pose
    int bad = 0;
    {
      double R[9];
      quat_to_R(s_pose + 3, R);
      for (int e = e0 + tid; e < e1; e += 256) {
        double r0, r1;
        if (P.outlier[e]) {
          double Xc[3], r[2];
          cam_point(s_pose, R, P.Xw + 3 * (size_t)e, Xc);
          edge_error(d, e, Xc, r);
          P.err[2 * (size_t)e] = r[0]; P.err[2 * (size_t)e + 1] = r[1];
          r0 = r[0]; r1 = r[1];
        } else { r0 = P.err[2 * (size_t)e]; r1 = P.err[2 * (size_t)e + 1]; }
        const float chi2 = (float)(P.inv[e] * (r0 * r0 + r1 * r1));     // const float chi2 = e->chi2()
        const int o = chi2 > 5.991f ? 1 : 0;
        P.outlier[e] = (uint8_t)o;
        bad += o;
      }
    }
    // nBad of this round (integer sum over the workgroup)
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    __shared__ int s_bad[4];
    __syncthreads();
    if ((tid & 63) == 0) s_bad[tid >> 6] = bad;
    __syncthreads();
    nBad = s_bad[0] + s_bad[1] + s_bad[2] + s_bad[3];
    rounds = round + 1;
    if (n < 10) break;                                           // optimizer.edges().size() < 10
  }
  if (tid < 7) P.poses[7 * f + tid] = s_pose[tid];
  if (tid == 0) {
    res[0] = n - nBad; res[1] = nBad; res[2] = rounds; res[3] = 0;
    for (int i = 0; i < 4; ++i) res[4 + i] = its[i];
  }
}

// The same procedure with a frame's edges held in REGISTERS: thread t owns edges t, t + 256, ... (PO_MAXJ of them, i.e. frames of up to
// 256 x PO_MAXJ edges); world point, measurement, information, face, the persistent error and the outlier flag never go back to memory
// during the ~80 dependent passes of the four rounds -- the global-memory latency of every pass was most of the kernel's time.
// Arithmetic and order of operations are those of k_pose_optimize_g above.
#define PO_MAXJ 4
extern "C" __global__ void __launch_bounds__(256) k_pose_optimize(PoseDev P) {
  __shared__ double sh28[4][28];
  __shared__ double sum[28];            // [0..21) upper triangle of H row by row, [21..27) b, [27] chi2
  __shared__ double s_pose[7], s_trial[7];
  __shared__ int s_ctl[2];              // [0] another trial, [1] another iteration
  const int f = blockIdx.x, tid = threadIdx.x;
  const int e0 = P.off[f], e1 = P.off[f + 1], n = e1 - e0;
  int* res = P.result + 8 * f;
  for (int e = e0 + tid; e < e1; e += 256) P.outlier[e] = 0;   // pFrame->mvbOutlier[i] = false (Optimizer.cpp:91)
  bool has[PO_MAXJ]; int cf[PO_MAXJ], cout_[PO_MAXJ];
  double cX[PO_MAXJ][3], co[PO_MAXJ][2], ci[PO_MAXJ], cer[PO_MAXJ][2];
#pragma unroll
  for (int j = 0; j < PO_MAXJ; ++j) {
    const int e = e0 + tid + 256 * j;
    has[j] = e < e1; cf[j] = 0; cout_[j] = 0; ci[j] = 0; cer[j][0] = 0; cer[j][1] = 0;
    cX[j][0] = cX[j][1] = cX[j][2] = 0; co[j][0] = co[j][1] = 0;
    if (has[j]) {
      cX[j][0] = P.Xw[3 * (size_t)e]; cX[j][1] = P.Xw[3 * (size_t)e + 1]; cX[j][2] = P.Xw[3 * (size_t)e + 2];
      co[j][0] = P.obs[2 * (size_t)e]; co[j][1] = P.obs[2 * (size_t)e + 1]; ci[j] = P.inv[e]; cf[j] = P.face[e];
    }
  }
  if (n < 3) {                          // Optimizer.cpp:131-132: pose untouched, 0 returned
    if (tid < 8) res[tid] = 0;
    return;
  }
  BaDev d;                              // the edge arithmetic of the local BA (cms_ba_kernels.hip) on this frame's arrays
  d.e_face = P.face; d.e_obs = P.obs; d.e_inv = P.inv; d.fx = P.fx; d.fy = P.fy; d.cx = P.cx; d.cy = P.cy;
  const double delta = sqrt(5.991);
  double pose0[7];
  {
    const double* p = P.poses + 7 * f;
    for (int i = 0; i < 7; ++i) pose0[i] = p[i];
    normalize_rot(pose0 + 3);           // SE3Quat constructor (se3quat.h:58-64)
  }
  int nBad = 0, rounds = 0;
  int its[4] = {0, 0, 0, 0};
  for (int round = 0; round < 4; ++round) {
    const int robust = round < 3;
    if (tid < 7) s_pose[tid] = pose0[tid];                      // setEstimate(pFrame->mTcw) before every round
    int mine = 0;
#pragma unroll
    for (int j = 0; j < PO_MAXJ; ++j) mine += (has[j] && cout_[j] == 0) ? 1 : 0;
    const int nact = __syncthreads_count(mine > 0);             // also publishes s_pose
    int done = 0;
    if (nact > 0) {                                             // no level-0 edge -> no active vertex -> optimize() returns at once
      double lambda = -1, ni = 2, currentChi = 0, iniChi = 0, rho = 0;
      int nBadIt = 0;
      for (int it = 0; it < 10; ++it) {
        // ---- computeActiveErrors + activeRobustChi2 + buildSystem at s_pose
        double acc[28];
#pragma unroll
        for (int i = 0; i < 28; ++i) acc[i] = 0;
        {
          double R[9];
          quat_to_R(s_pose + 3, R);
#pragma unroll
          for (int j = 0; j < PO_MAXJ; ++j) {
            if (!has[j] || cout_[j]) continue;
            double Xc[3], r[2], Jp[12], Jl[6];
            cam_point(s_pose, R, cX[j], Xc);
            edge_error_v(d, cf[j], co[j][0], co[j][1], Xc, r);
            cer[j][0] = r[0]; cer[j][1] = r[1];
            const double om = ci[j], c2 = om * (r[0] * r[0] + r[1] * r[1]);
            double w = 1.0, rho0 = c2;
            if (robust) w = huber_w(c2, delta, &rho0);
            edge_jac_face(d, cf[j], Xc, R, Jp, Jl);
            const double ow = w * om;
            int k = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
              for (int jj = i; jj < 6; ++jj) acc[k++] += ow * (Jp[i] * Jp[jj] + Jp[6 + i] * Jp[6 + jj]);
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[21 + i] -= ow * (Jp[i] * r[0] + Jp[6 + i] * r[1]);
            acc[27] += rho0;
          }
        }
        pose_block_sum28(acc, sh28, sum);
        double H[36], b[6];
        if (tid == 0) {
          int k = 0;
          for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j) { H[6 * i + j] = sum[k]; H[6 * j + i] = sum[k]; ++k; }
          for (int i = 0; i < 6; ++i) b[i] = sum[21 + i];
          currentChi = sum[27]; iniChi = currentChi;
          if (it == 0) {
            double md = 0;
            for (int j = 0; j < 6; ++j) md = fmax(fabs(H[7 * j]), md);
            lambda = 1e-5 * md; ni = 2; nBadIt = 0;
          }
          rho = 0;
        }
        int qmax = 0;
        for (;;) {   // do { } while (rho < 0 && qmax < 10)
          double scale = 0;
          bool ok2 = true;
          if (tid == 0) {
            double x[6];
            ok2 = pose_solve6(H, b, lambda, x);
            if (!ok2) for (int i = 0; i < 6; ++i) x[i] = 0;
            double Tn[7];
            pose_exp_mul(x, s_pose, Tn);
            for (int i = 0; i < 7; ++i) s_trial[i] = Tn[i];
            for (int j = 0; j < 6; ++j) scale += x[j] * (lambda * x[j] + b[j]);
          }
          __syncthreads();
          // ---- computeActiveErrors at the trial pose: the stored errors ARE the trial's from here on (kept on rejection, g2o)
          double chi = 0;
          {
            double R[9];
            quat_to_R(s_trial + 3, R);
#pragma unroll
            for (int j = 0; j < PO_MAXJ; ++j) {
              if (!has[j] || cout_[j]) continue;
              double Xc[3], r[2];
              cam_point(s_trial, R, cX[j], Xc);
              edge_error_v(d, cf[j], co[j][0], co[j][1], Xc, r);
              cer[j][0] = r[0]; cer[j][1] = r[1];
              const double c2 = ci[j] * (r[0] * r[0] + r[1] * r[1]);
              double rho0 = c2;
              if (robust) huber_w(c2, delta, &rho0);
              chi += rho0;
            }
          }
          const double tempChi0 = block_sum(chi, &sh28[0][0]);
          if (tid == 0) {
            double tempChi = ok2 ? tempChi0 : 1.7976931348623157e308;
            rho = (currentChi - tempChi) / (scale + 1e-3);
            if (rho > 0 && isfinite(tempChi)) {
              const double t21 = 2 * rho - 1; double alpha = 1. - t21 * t21 * t21;     // pow(2 rho - 1, 3): the cube, two roundings instead of a ~200-instruction pow in the serial part
              alpha = fmin(alpha, 2. / 3.);
              lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi;
              for (int i = 0; i < 7; ++i) s_pose[i] = s_trial[i];
            } else {
              lambda *= ni; ni *= 2;
            }
            ++qmax;
            s_ctl[0] = (rho < 0 && qmax < 10) ? 1 : 0;
          }
          __syncthreads();
          if (!s_ctl[0]) break;
        }
        ++done;
        if (tid == 0) {
          bool stop = (qmax == 10 || rho == 0);
          if (!stop) {
            if ((iniChi - currentChi) * 1e3 < iniChi) ++nBadIt; else nBadIt = 0;
            if (nBadIt >= 3) stop = true;
          }
          s_ctl[1] = stop ? 0 : 1;
        }
        __syncthreads();
        if (!s_ctl[1]) break;
      }
    }
    its[round] = done;
    // ---- re-classification (Optimizer.cpp:142-173): former outliers get a fresh error at the round's final pose
    int bad = 0;
    {
      double R[9];
      quat_to_R(s_pose + 3, R);
#pragma unroll
      for (int j = 0; j < PO_MAXJ; ++j) {
        if (!has[j]) continue;
        if (cout_[j]) {
          double Xc[3], r[2];
          cam_point(s_pose, R, cX[j], Xc);
          edge_error_v(d, cf[j], co[j][0], co[j][1], Xc, r);
          cer[j][0] = r[0]; cer[j][1] = r[1];
        }
        const double r0 = cer[j][0], r1 = cer[j][1];
        const float chi2 = (float)(ci[j] * (r0 * r0 + r1 * r1));        // const float chi2 = e->chi2()
        const int o = chi2 > 5.991f ? 1 : 0;
        cout_[j] = o;
        bad += o;
      }
    }
    // nBad of this round (integer sum over the workgroup)
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    __shared__ int s_bad[4];
    __syncthreads();
    if ((tid & 63) == 0) s_bad[tid >> 6] = bad;
    __syncthreads();
    nBad = s_bad[0] + s_bad[1] + s_bad[2] + s_bad[3];
    rounds = round + 1;
    if (n < 10) break;                                           // optimizer.edges().size() < 10
  }
#pragma unroll
  for (int j = 0; j < PO_MAXJ; ++j) if (has[j]) P.outlier[e0 + tid + 256 * j] = (uint8_t)cout_[j];
  if (tid < 7) P.poses[7 * f + tid] = s_pose[tid];
  if (tid == 0) {
    res[0] = n - nBad; res[1] = nBad; res[2] = rounds; res[3] = 0;
    for (int i = 0; i < 4; ++i) res[4 + i] = its[i];
  }
}
