// cms_ba_wrappers.hip -- __global__ entry points of the local-BA kernels.
//
// Every kernel body lives in cms_ba_kernels.hip / cms_ba_fused.hip as a __device__ function taking its block index (BX)
// and grid size (GX).  Two thin entry points exist per body:
//   k_ba_*   one window per launch (explicit arguments)
//   kb_ba_*  MANY windows per launch: blockIdx.z selects a BaItem (one per window) holding that window's pointers,
//            lambda and phase; a window whose phase differs from the launch's phase, or whose own grid is smaller than
//            the launch grid, exits at once.  n independent Levenberg-Marquardt problems then cost the launches (and
//            host synchronisations) of one, and their small kernels fill the chip together.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BA_MAX_GROUP 64
// Levenberg-Marquardt state of one window when the accept / reject logic runs on the device (batched driver): what
// OptimizationAlgorithmLevenberg::solve keeps between trials (optimization_algorithm_levenberg.cpp:61-164).  The last kernel of a phase
// updates it (ba_lm_after_iter / ba_lm_after_trial) and mirrors it into pinned host memory; every kernel of the following launches
// reads `next` to know whether it has work: the host can enqueue several iterations back to back without waiting.
struct BaLmDev {
  double lambda, ni, currentChi, iniChi, rho, chi_ini, chi_fin, lam_fin;
  int it, iterations, qmax, nBad, done, next, cur, rounds;   // next: 0 = start an iteration, 1 = one more trial, 2 = finished
  int n_out[2], pad[2];                                      // outliers counted by kb_ba_classify: [0] mid-way (set_level), [1] final
};
// static description of one window (device resident, uploaded once per optimisation stage) ...
struct BaItem {
  BaDev d;
  double* poses[2]; double* pts[2];          // estimate double buffer; BaDyn::cur says which one is current
  double* Hll; double* bl; double* Hpl; double* Hpp; double* bp; double* pose_partial; double* Dinv; double* db;
  double* chunk_sum; double* x; double* partial; double* scal;
  double* hscal;                             // pinned host mirror of scal[0..6), written by the last kernel of a phase
  const int2* chunk_range; const int2* tup; const int* pair_of_block; const int* pair_chunk_off;
  uint8_t* flags;
  int nblk_e, nblk_p, nchunks, block_free;   // block_free: the 6x3 pose-point blocks are rebuilt from the estimate, never stored (per-point and edge-major Schur paths)
  BaSp sp;                                   // per-point Schur work lists (sp.R == 0: window uses the tuple-chunk kernel)
  BaSe se;                                   // edge-major Schur work list (se.R > 0: the group runs kb_ba_schur_edges instead of kb_ba_schur_points)
  BaLmDev* lm; BaLmDev* hlm;                 // device-side LM state and its pinned host mirror (dyn.dev_lm)
};
// the same with global-memory pointer types (see BaDevG): what a kb_ba_* kernel builds from its window's BaItem before it calls a body.
// BaGP<T> holds the pointer in address space 1; used directly ([] / ->) it is a global access, handed to a body's `T*` parameter it converts
// with ONE address-space cast, which the compiler's address-space inference follows back after inlining (a cast pair would fold away)
template <class T> struct BaGP {
  BA_AS1 T* p;
  __device__ __forceinline__ BaGP() {}
  __device__ __forceinline__ BaGP(T* q) : p((BA_AS1 T*)q) {}
  __device__ __forceinline__ operator T*() const { return (T*)p; }
  __device__ __forceinline__ BA_AS1 T& operator[](size_t i) const { return p[i]; }
  __device__ __forceinline__ BA_AS1 T* operator->() const { return p; }
};
struct BaGP2 {      // the two estimate buffers: picked with a select (an indexed array member would put the whole view into scratch memory)
  BaGP<double> a, b;
  __device__ __forceinline__ BaGP<double> operator[](int i) const { BaGP<double> r; r.p = i ? b.p : a.p; return r; }
};
struct BaItemG {
  BaDevG d;
  BaGP2 poses, pts;
  BaGP<double> Hll, bl, Hpl, Hpp, bp, pose_partial, Dinv, db, chunk_sum, x, partial, scal, hscal;
  BaGP<const int2> chunk_range, tup;
  BaGP<const int> pair_of_block, pair_chunk_off;
  BaGP<uint8_t> flags;
  int nblk_e, nblk_p, nchunks, block_free;
  BaSpG sp; BaSeG se;
  BaGP<BaLmDev> lm, hlm;
  __device__ __forceinline__ BaItemG(const BaItem& t)
      : d(t.d), Hll(t.Hll), bl(t.bl), Hpl(t.Hpl), Hpp(t.Hpp), bp(t.bp), pose_partial(t.pose_partial), Dinv(t.Dinv), db(t.db), chunk_sum(t.chunk_sum), x(t.x),
        partial(t.partial), scal(t.scal), hscal(t.hscal), chunk_range(t.chunk_range), tup(t.tup), pair_of_block(t.pair_of_block),
        pair_chunk_off(t.pair_chunk_off), flags(t.flags), nblk_e(t.nblk_e), nblk_p(t.nblk_p), nchunks(t.nchunks), block_free(t.block_free), sp(t.sp), se(t.se),
        lm(t.lm), hlm(t.hlm) {
    poses.a = BaGP<double>(t.poses[0]); poses.b = BaGP<double>(t.poses[1]); pts.a = BaGP<double>(t.pts[0]); pts.b = BaGP<double>(t.pts[1]);
  }
};
// ... and what changes from launch to launch, passed BY VALUE as a kernel argument: no host->device copy per Levenberg step
struct BaDyn {
  double lambda[BA_MAX_GROUP];
  uint8_t phase[BA_MAX_GROUP], cur[BA_MAX_GROUP], first_iter[BA_MAX_GROUP];
  int robust, set_level;
  double delta, chi2_th;
  int dev_lm, fold_finish;                           // 1: phase / cur / lambda / first_iter come from BaItem::lm instead of the arrays above
  int fused_lin, fold_reduce;                        // fused_lin 1: kb_ba_schur_edges linearises itself (cms_ba_schur_edges.hip): no ITER phase after a stage's first iteration
  int solve_presum, pad_;                            // kb_ba_trial_solve3r: the slices summed into LDS by all threads first (the launch carries the LDS for it)
};
enum { BA_PHASE_IDLE = 0, BA_PHASE_ITER = 1, BA_PHASE_TRIAL = 2, BA_PHASE_CLASSIFY = 3 };

// ------------------------------------------------------------------------------------------------ one window per launch
extern "C" __global__ void __launch_bounds__(256)
k_ba_errors(BaDev d, const double* poses, const double* pts, int robust, double delta, double* partial) {
  ba_errors_body(blockIdx.x, gridDim.x, d, poses, pts, robust, delta, partial);
}
extern "C" __global__ void __launch_bounds__(256)
k_ba_reduce(const double* partial, int n, double* out, int add) { ba_reduce_body(blockIdx.x, gridDim.x, partial, n, out, add); }
extern "C" __global__ void __launch_bounds__(128)
k_ba_lin_points(BaDev d, const double* poses, const double* pts, int robust, double delta, double* Hll, double* bl, double* Hpl) {
  ba_lin_points_body(blockIdx.x, gridDim.x, d, poses, pts, robust, delta, Hll, bl, Hpl);
}
extern "C" __global__ void __launch_bounds__(256)
k_ba_lin_poses(BaDev d, const double* poses, const double* pts, int robust, double delta, double* pose_partial) {
  ba_lin_poses_body(blockIdx.x, gridDim.x, d, poses, pts, robust, delta, pose_partial);
}
extern "C" __global__ void __launch_bounds__(64)
k_ba_pose_finish(int np, const double* pose_partial, double* Hpp, double* bp) { ba_pose_finish_body(blockIdx.x, gridDim.x, np, pose_partial, Hpp, bp); }
extern "C" __global__ void __launch_bounds__(256)
k_ba_maxdiag(int np, int P, const double* Hpp, const double* Hll, double* out) { ba_maxdiag_body(blockIdx.x, gridDim.x, np, P, Hpp, Hll, out); }
extern "C" __global__ void __launch_bounds__(256)
k_ba_dinv(int P, const double* Hll, const double* bl, double lambda, double* Dinv, double* db) {
  ba_dinv_body(blockIdx.x, gridDim.x, P, Hll, bl, lambda, Dinv, db);
}
extern "C" __global__ void __launch_bounds__(256)
k_ba_schur_chunks(BaDev d, const int2* chunk_range, const int2* tup, const double* Hpl, const double* Dinv, const double* db, double* chunk_sum) {
  ba_schur_chunks_body(blockIdx.x, gridDim.x, d, chunk_range, tup, Hpl, Dinv, db, chunk_sum);
}
extern "C" __global__ void __launch_bounds__(BA_SP_MAX_THREADS + BA_SP_STAGERS) __attribute__((amdgpu_waves_per_eu(1, 3)))
k_ba_schur_points(BaDev d, BaSp sp, const double* Hpl, const double* Dinv, const double* db) { ba_schur_points_body(blockIdx.x, d, sp, Hpl, Dinv, db); }
extern "C" __global__ void __launch_bounds__(256)
k_ba_schur_reduce(BaSp sp, double* pair_sum) { ba_schur_reduce_body(blockIdx.x, sp, pair_sum); }
extern "C" __global__ void __launch_bounds__(256)
k_ba_classify(BaDev d, const double* poses, const double* pts, double chi2_th, int set_level, uint8_t* flags) {
  ba_classify_body(blockIdx.x, gridDim.x, d, poses, pts, chi2_th, set_level, flags);
}
extern "C" __global__ void __launch_bounds__(384)
k_ba_trial_solve(BaDev d, const double* Hpp, const double* bp, double lambda, const int* pair_of_block, const int* pair_chunk_off,
                 const double* chunk_sum, const double* poses, double* poses_new, double* xp_out, double* scal) {
  ba_trial_solve_body(blockIdx.x, gridDim.x, d, Hpp, bp, lambda, pair_of_block, pair_chunk_off, chunk_sum, poses, poses_new, xp_out, scal,
                      reinterpret_cast<long long*>(scal + 8));
}
extern "C" __global__ void __launch_bounds__(128)
k_ba_trial_points(BaDev d, const double* bl, const double* Hpl, const double* Dinv, const double* xp, double lambda, const double* pts,
                  double* pts_new, const double* poses_new, int robust, double delta, double* partial) {
  ba_trial_points_body(blockIdx.x, gridDim.x, d, bl, Hpl, Dinv, xp, lambda, pts, pts_new, poses_new, robust, delta, partial);
}
extern "C" __global__ void __launch_bounds__(256)
k_ba_reduce2(const double* partial, int n, double* scal) { ba_reduce2_body(blockIdx.x, gridDim.x, partial, n, scal); }

// ------------------------------------------------------------------------------------------------ many windows per launch
#define BA_ITEM(PHASE, NBLK)                                                                              \
  const int z = blockIdx.z;                                                                               \
  const BaItemG it(items[z]);                                                                             \
  int ba_ph = dyn.phase[z], cur = dyn.cur[z];                                                             \
  double ba_lambda = dyn.lambda[z];                                                                       \
  bool ba_first = dyn.first_iter[z] != 0;                                                                 \
  if (dyn.dev_lm && (PHASE) != BA_PHASE_CLASSIFY) {                                                       \
    const int nx = it.lm->next;                                                                           \
    ba_ph = nx == 0 ? BA_PHASE_ITER : nx == 1 ? BA_PHASE_TRIAL : BA_PHASE_IDLE;                           \
    cur = it.lm->cur; ba_lambda = it.lm->lambda; ba_first = it.lm->it == 0;                               \
  }                                                                                                       \
  if (ba_ph != (PHASE) || (int)blockIdx.x >= (NBLK)) return;                                              \
  const int nxt = cur ^ 1;                                                                                \
  (void)nxt; (void)ba_lambda; (void)ba_first;

// end of the ITER phase (computeActiveErrors + buildSystem done): what the host driver's ba_finish_iter_start does
__device__ __forceinline__ void ba_lm_after_iter(BaLmDev* L, BaLmDev* H, double chi, double maxdiag) {
  L->currentChi = chi; L->iniChi = chi;
  if (L->it == 0) { L->chi_ini = chi; L->lambda = 1e-5 * maxdiag; L->ni = 2; L->nBad = 0; }
  L->rho = 0; L->qmax = 0; L->next = 1;
  *H = *L;
}
// end of a TRIAL: accept / reject, lambda update, the 10-trial rule and the termination tests (ba_finish_trial)
__device__ __forceinline__ void ba_lm_after_trial(BaLmDev* L, BaLmDev* H, double tempChi, double den, int ok2, bool fused_lin = false) {
  if (!ok2) tempChi = 1.7976931348623157e308;
  double rho = (L->currentChi - tempChi) / (den + 1e-3);
  L->rho = rho;
  if (rho > 0 && isfinite(tempChi)) {
    double alpha = 1. - pow(2 * rho - 1, 3.0);
    alpha = fmin(alpha, 2. / 3.);
    L->lambda *= fmax(1. / 3., alpha);
    L->ni = 2; L->currentChi = tempChi;
    L->cur ^= 1;
  } else {
    L->lambda *= L->ni; L->ni *= 2;
  }
  ++L->qmax;
  if (rho < 0 && L->qmax < 10) { L->next = 1; *H = *L; return; }
  ++L->done;
  L->chi_fin = L->currentChi; L->lam_fin = L->lambda;
  bool terminate = (L->qmax == 10 || rho == 0);
  if (!terminate) {
    if ((L->iniChi - L->currentChi) * 1e3 < L->iniChi) ++L->nBad; else L->nBad = 0;
    if (L->nBad >= 3) terminate = true;
  }
  ++L->it;
  L->next = (terminate || L->it >= L->iterations) ? 2 : 0;
  if (fused_lin && L->next == 0) {      // the next iteration's linearisation happens inside its first trial: what ba_lm_after_iter would do
    L->iniChi = L->currentChi; L->rho = 0; L->qmax = 0; L->next = 1;
  }
  *H = *L;
}

// Device-side Levenberg state: from the second iteration of a stage on, the estimate an iteration starts from is the trial the previous
// iteration accepted -- its residuals are already in d.err and its chi2 is currentChi (kb_ba_trial_points / kb_ba_reduce2 produced
// both), so computeActiveErrors + activeRobustChi2 would only recompute them: these two kernels have work on a stage's first iteration only.
extern "C" __global__ void __launch_bounds__(256) kb_ba_errors(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.nblk_e)
  if (dyn.dev_lm && !ba_first) return;
  ba_errors_body(blockIdx.x, it.nblk_e, it.d, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta, it.partial);
}
// chi2 of the current estimate -> scal[0]
extern "C" __global__ void __launch_bounds__(256) kb_ba_reduce(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, 1)
  if (dyn.dev_lm && !ba_first) return;
  ba_reduce_body(0, 1, it.partial, it.nblk_e, it.scal, 0);
}
extern "C" __global__ void __launch_bounds__(128) kb_ba_lin_points(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.nblk_p)
  ba_lin_points_body(blockIdx.x, it.nblk_p, it.d, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta, it.Hll, it.bl,
                     it.block_free ? nullptr : it.Hpl);   // per-point / edge-major Schur paths: blocks are rebuilt on the fly, only the weights are stored
}
extern "C" __global__ void __launch_bounds__(256) kb_ba_lin_poses(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.d.K)        // blockIdx.y = slice of the pose's edge list
  ba_lin_poses_body(blockIdx.x, it.d.K, it.d, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta, it.pose_partial);
}
// Both linearisation passes in one launch: the per-point and the per-pose accumulation read the same estimate and residuals and write
// disjoint outputs, and neither fills the chip on its own (one after the other they cost 19 + 28 us per iteration of a four-window
// group).  Workgroups [0, npb) take 256 points each, the rest take one (pose, slice) pair each.
extern "C" __global__ void __launch_bounds__(256) kb_ba_lin(const BaItem* __restrict__ items, BaDyn dyn, int phase, int npb) {
  BA_ITEM(phase, 1 << 30)
  if ((int)blockIdx.x < npb) {
    if ((int)blockIdx.x * 256 >= it.d.P) return;
    ba_lin_points_body(blockIdx.x, npb, it.d, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta, it.Hll, it.bl, it.block_free ? nullptr : it.Hpl);
  } else {
    const int j = (int)blockIdx.x - npb, k = j / BA_POSE_CHUNKS, ch = j - k * BA_POSE_CHUNKS;
    if (k >= it.d.K) return;
    ba_lin_poses_body(k, it.d.K, it.d, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta, it.pose_partial, ch);
  }
}
extern "C" __global__ void __launch_bounds__(64) kb_ba_pose_finish(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.d.np)
  ba_pose_finish_body(blockIdx.x, it.d.np, it.d.np, it.pose_partial, it.Hpp, it.bp);
}
// last kernel of the ITER phase, one workgroup per window: computeLambdaInit's max diagonal on the first iteration, then the
// phase's scalars are published to the pinned host mirror (the host only has to wait for the stream, no D2H copy)
extern "C" __global__ void __launch_bounds__(1024) kb_ba_maxdiag(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, 1)
  __shared__ double shm[16];
  if (dyn.fold_finish) {      // the slice sums of kb_ba_pose_finish, done here: one launch less per iteration (540 sums of 8 for K = 20)
    for (int idx = threadIdx.x; idx < 27 * it.d.np; idx += blockDim.x) {
      const int slot = idx / 27, t = idx - 27 * slot;
      double s = 0;
      for (int ch = 0; ch < BA_POSE_CHUNKS; ++ch) s += it.pose_partial[((size_t)slot * BA_POSE_CHUNKS + ch) * 27 + t];
      if (t < 21) {
        int i = 0, rem = t;
        while (rem >= 6 - i) { rem -= 6 - i; ++i; }
        const int j = i + rem;
        it.Hpp[36 * slot + 6 * i + j] = s; it.Hpp[36 * slot + 6 * j + i] = s;
      } else {
        it.bp[6 * slot + (t - 21)] = s;
      }
    }
    __syncthreads();
  }
  double mx_all = it.scal[3];
  if (ba_first) {
    double m = 0;
    for (int i = threadIdx.x; i < 6 * it.d.np; i += blockDim.x) m = fmax(m, fabs(it.Hpp[36 * (i / 6) + 7 * (i % 6)]));
    for (int i = threadIdx.x; i < 3 * it.d.P; i += blockDim.x) m = fmax(m, fabs(it.Hll[9 * (size_t)(i / 3) + 4 * (i % 3)]));
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      double mx = 0;
      for (int i = 0; i < (int)(blockDim.x >> 6); ++i) mx = fmax(mx, shm[i]);
      it.scal[3] = mx; it.hscal[3] = mx; mx_all = mx;
    }
  }
  if (threadIdx.x == 0) {
    it.hscal[0] = it.scal[0];
    if (dyn.dev_lm) ba_lm_after_iter(it.lm, it.hlm, ba_first ? it.scal[0] : it.lm->currentChi, mx_all);
  }
}
extern "C" __global__ void __launch_bounds__(256) kb_ba_dinv(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, (it.d.P + 255) / 256)
  ba_dinv_body(blockIdx.x, 0, it.d.P, it.Hll, it.bl, ba_lambda, it.Dinv, it.db);
}
extern "C" __global__ void __launch_bounds__(256) kb_ba_schur_chunks(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.nchunks)
  ba_schur_chunks_body(blockIdx.x, it.nchunks, it.d, it.chunk_range, it.tup, it.Hpl, it.Dinv, it.db, it.chunk_sum);
}
extern "C" __global__ void __launch_bounds__(BA_SP_MAX_THREADS + BA_SP_STAGERS) __attribute__((amdgpu_waves_per_eu(1, 3))) kb_ba_schur_points(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.sp.R)
  ba_schur_points_body(blockIdx.x, it.d, it.sp, it.Hpl, it.Dinv, it.db, it.Hll, it.bl, ba_lambda, it.poses[cur], it.pts[cur]);   // inverts Hll + lambda I itself
}
extern "C" __global__ void __launch_bounds__(BA_SE_THREADS) kb_ba_schur_edges(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.se.R)
  ba_schur_edges_body<false>(blockIdx.x, it.d, it.se, it.Hll, it.bl, ba_lambda, it.poses[cur], it.pts[cur], 0, 0.0);
}
// ... and with the linearisation inside (dyn.fused_lin)
extern "C" __global__ void __launch_bounds__(BA_SE_THREADS) kb_ba_lin_schur_edges(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.se.R)
  ba_schur_edges_body<true>(blockIdx.x, it.d, it.se, it.Hll, it.bl, ba_lambda, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta);
}
// ... and with the points of frequent observation signatures taken run-major (cms_ba_schur_runs.hip): the first R_rm workgroups of a window
// work through its run chunks, the rest through the left-over chunks edge-major -- one launch, one LDS layout, one kind of slice
extern "C" __global__ void __launch_bounds__(BA_SE_THREADS) kb_ba_lin_schur_runs(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.se.R_rm + it.se.R)
  if ((int)blockIdx.x < it.se.R_rm) ba_schur_runs_mfma_body(blockIdx.x, it.d, it.se, it.Hll, it.bl, ba_lambda, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta);
  else ba_schur_edges_body<true>((int)blockIdx.x - it.se.R_rm, it.d, it.se, it.Hll, it.bl, ba_lambda, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta);
}
// ... and the two with their LDS additions in a fixed order (deterministic windows, cms_ba_set_deterministic: bit-identical sums from run to run; the
// workgroups store slices, se.gsum = 0, which the solve kernel adds in slice order)
extern "C" __global__ void __launch_bounds__(BA_SE_THREADS) kb_ba_lin_schur_runs_det(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.se.R_rm + it.se.R)
  if ((int)blockIdx.x < it.se.R_rm) ba_schur_runs_mfma_body<true>(blockIdx.x, it.d, it.se, it.Hll, it.bl, ba_lambda, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta);
  else ba_schur_edges_body<true, true>((int)blockIdx.x - it.se.R_rm, it.d, it.se, it.Hll, it.bl, ba_lambda, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta);
}
extern "C" __global__ void __launch_bounds__(BA_SE_THREADS) kb_ba_lin_schur_edges_det(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.se.R)
  ba_schur_edges_body<true, true>(blockIdx.x, it.d, it.se, it.Hll, it.bl, ba_lambda, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta);
}
// ... the same with the runs' products on the vector ALU (producer / consumer wavefront pairs; CMS_BA_RM_VALU=1, A/B)
extern "C" __global__ void __launch_bounds__(BA_SE_THREADS) kb_ba_lin_schur_runs_valu(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.se.R_rm + it.se.R)
  if ((int)blockIdx.x < it.se.R_rm) ba_schur_runs_body(blockIdx.x, it.d, it.se, it.Hll, it.bl, ba_lambda, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta);
  else ba_schur_edges_body<true>((int)blockIdx.x - it.se.R_rm, it.d, it.se, it.Hll, it.bl, ba_lambda, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta);
}
extern "C" __global__ void __launch_bounds__(256) kb_ba_schur_edges_reduce(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.se.nchunks > 0 ? it.se.npairs2 : 0)
  BaSpG v;                                         // the range sum only looks at these three fields
  v.R = it.se.R_rm + it.se.R; v.npairs = it.se.npairs2; v.partial = it.se.partial;
  ba_schur_reduce_body(blockIdx.x, v, it.chunk_sum);
  if (dyn.fused_lin && (int)blockIdx.x < it.d.np) {      // bp of key frame blockIdx.x: the ranges' sums, added in fixed order
    __shared__ double bps[6 * BA_SE_RANGES];
    const int R = min(it.se.R_rm + it.se.R, BA_SE_RANGES);
    for (int t = threadIdx.x; t < 6 * R; t += blockDim.x) bps[t] = it.se.bp_partial[((size_t)(t / 6) * it.d.np + blockIdx.x) * 6 + (t % 6)];
    __syncthreads();
    if (threadIdx.x < 6) {
      double sum = 0.0;
      for (int r = 0; r < R; ++r) sum += bps[6 * r + threadIdx.x];
      it.bp[6 * blockIdx.x + threadIdx.x] = sum;
    }
  }
}
extern "C" __global__ void __launch_bounds__(256) kb_ba_schur_reduce(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.sp.R > 0 ? it.sp.npairs : 0)
  ba_schur_reduce_body(blockIdx.x, it.sp, it.chunk_sum);
}
extern "C" __global__ void __launch_bounds__(384) kb_ba_trial_solve(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, 1)
  ba_trial_solve_body(0, 1, it.d, it.Hpp, it.bp, ba_lambda, it.pair_of_block, it.pair_chunk_off, it.chunk_sum, it.poses[cur], it.poses[nxt],
                      it.x, it.scal);
}
extern "C" __global__ void __launch_bounds__(1024) kb_ba_trial_solve3(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, 1)
  ba_trial_solve3_body(it.d, it.Hpp, it.bp, ba_lambda, it.pair_of_block, it.pair_chunk_off, it.chunk_sum, it.poses[cur], it.poses[nxt], it.x, it.scal,
                       dyn.fused_lin != 0);
}
// ... and with the sum over the Schur kernel's range slices done by the assembly itself (no kb_ba_schur_edges_reduce launch in the round)
extern "C" __global__ void __launch_bounds__(1024) kb_ba_trial_solve3r(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, 1)
  ba_trial_solve3_body(it.d, it.Hpp, it.bp, ba_lambda, it.pair_of_block, it.pair_chunk_off, it.chunk_sum, it.poses[cur], it.poses[nxt], it.x, it.scal,
                       true, (double*)it.se.partial, (double*)it.se.bp_partial, it.se.gsum ? 1 : it.se.R_rm + it.se.R, it.se.npairs2, it.se.gsum != 0, false);
}
// ... for groups whose windows keep slices (deterministic windows, CMS_BA_NO_GLOBAL_SUM): all threads add the slices into LDS first (a kernel of its own: the
// extra phase cost the default kernel 96 B of scratch when it was a run-time switch)
extern "C" __global__ void __launch_bounds__(1024) kb_ba_trial_solve3rp(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, 1)
  ba_trial_solve3_body(it.d, it.Hpp, it.bp, ba_lambda, it.pair_of_block, it.pair_chunk_off, it.chunk_sum, it.poses[cur], it.poses[nxt], it.x, it.scal,
                       true, (double*)it.se.partial, (double*)it.se.bp_partial, it.se.R_rm + it.se.R, it.se.npairs2, false, true);
}
extern "C" __global__ void __launch_bounds__(128) kb_ba_trial_points(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.nblk_p)
  ba_trial_points_body(blockIdx.x, it.nblk_p, it.d, it.bl, it.Hpl, it.Dinv, it.x, ba_lambda, it.pts[cur], it.pts[nxt], it.poses[nxt], dyn.robust,
                       dyn.delta, it.partial, it.block_free ? it.Hll : nullptr, it.block_free ? it.poses[cur] : nullptr);
}
// last step of the TRIAL phase: chi2(trial), gain denominator, accept / reject, then publish (see kb_ba_maxdiag)
__device__ __forceinline__ void ba_round_count(const BaItem* __restrict__ items) {
  // round counter of the group (one thread per round calls this), mirrored into pinned host memory AFTER window 0's state: the host driver
  // paces its launches on it without ever synchronising the stream (see ba_optimize_stage_batched_dev); bumped first, the host saw the round
  // end before the windows' "finished" flags and queued idle rounds (5 per window group and call instead of 2).  The mirror is fine-grained
  // coherent host memory (uncached on the device, hipHostMallocCoherent): stores of one thread arrive in order, no fence -- a system-scope
  // release would write back this XCD's whole L2, the 19 us the pacing scheme exists to avoid.  The other windows' blocks are not ordered
  // against this counter; the host only uses the mirrored state to decide whether to queue another round (an early counter costs at most
  // one idle round) and reads the final state after synchronising the stream.
  BaLmDev* L0 = items[0].lm;
  const int r = L0->rounds + 1;
  L0->rounds = r;
  *reinterpret_cast<volatile int*>(&items[0].hlm->rounds) = r;
}
__device__ __forceinline__ void ba_trial_publish(const BaItemG& it, const BaDyn& dyn) {
  if (dyn.dev_lm && threadIdx.x == 0) {
    const int ok2 = (int)__double_as_longlong(it.scal[4]);      // (the solver's status word sits in the low half of scal[4])
    ba_lm_after_trial(it.lm, it.hlm, it.scal[1], it.scal[2], ok2, dyn.fused_lin != 0);
  }
}
// dyn.fold_reduce: the trial kernel is the round's LAST launch -- the workgroup of a window that finishes last sums the window's partial sums
// (what kb_ba_reduce2 did in a launch of its own: 8 us alone, 24 us inside the step, plus the gap in front of it) and decides the trial.
// A workgroup hands its two sums over with returning atomic exchanges (read-modify-writes are coherent across the XCDs' L2 caches, plain
// stores only become so at the end of the kernel), waits for the returned values -- the exchanges are then performed -- and takes a ticket;
// the holder of the last ticket reads the partial sums back the same way (an atomic addition of zero) and adds them in kb_ba_reduce2's
// order: the same bits as the separate launch, whichever workgroup comes last.  No fence: nothing but these atomics is handed over.
#define BA_TICKET_SLOT 7       /* scal[7] (the group keeps eight scalars per window): the window's ticket counter (an int; zeroed by kb_ba_lm_load, put back to zero by the last workgroup) */
extern "C" __global__ void __launch_bounds__(BA_TE_THREADS) kb_ba_trial_edges(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  // the group's round counter follows window 0's state when that window takes part in the round (its last workgroup, below); a window 0 that
  // is through counts the round right away (the host then queues the next round a little early: at most one idle round, see ba_round_count)
  if (dyn.fold_reduce && dyn.dev_lm && blockIdx.z == 0 && blockIdx.x == 0 && threadIdx.x == 0 && (items[0].lm->next != 1 || items[0].se.Rt < 1)) ba_round_count(items);
  BA_ITEM(phase, it.se.Rt)
  ba_trial_edges_body(blockIdx.x, it.se.Rt, it.d, it.se, it.bl, it.Hll, it.x, ba_lambda, it.pts[cur], it.pts[nxt], it.poses[cur], it.poses[nxt], dyn.robust,
                      dyn.delta, it.partial, dyn.fold_reduce != 0);
  if (!dyn.fold_reduce) return;
  __shared__ int last_sh;
  __shared__ double sh2[16];
  const int n = it.se.Rt;
  int* ticket = reinterpret_cast<int*>((double*)it.scal + BA_TICKET_SLOT);
  if (threadIdx.x == 0) last_sh = atomicAdd(ticket, 1) == n - 1;      // (the body's thread 0 waited for its two exchanges before it returned)
  __syncthreads();
  if (!last_sh) return;
  double* partial = it.partial;
  double v1 = 0, v2 = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { v1 += atomicAdd(partial + i, 0.0); v2 += atomicAdd(partial + n + i, 0.0); }
  const double s1 = block_sum(v1, sh2);
  const double s2 = block_sum(v2, sh2);
  if (threadIdx.x == 0) {
    const double den = s2 + it.scal[5];
    it.scal[1] = s1; it.scal[2] = den;
    it.hscal[1] = s1; it.hscal[2] = den; it.hscal[4] = it.scal[4];      // pinned host mirror (batched driver)
    atomicExch(ticket, 0);
  }
  ba_trial_publish(it, dyn);
  if (dyn.dev_lm && blockIdx.z == 0 && threadIdx.x == 0) ba_round_count(items);
}
// ITER phase of a fused group in ONE launch (ba_first_pass_body): the last workgroup of a window to finish adds the workgroups' chi2 sums in a
// fixed order, takes the maximum of the diagonal sums and does what kb_ba_maxdiag did at the end of the phase (lambda, Levenberg state, mirror)
#define BA_PTMAX_SLOT 6        /* scal[6]: largest diagonal entry over the window's points, as the bits of a non-negative double (zeroed by kb_ba_lm_load) */
extern "C" __global__ void __launch_bounds__(BA_TE_THREADS) kb_ba_first_pass(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.se.Rt)
  unsigned long long* pt_max = reinterpret_cast<unsigned long long*>((double*)it.scal + BA_PTMAX_SLOT);
  ba_first_pass_body(blockIdx.x, it.se.Rt, it.d, it.se, it.poses[cur], it.pts[cur], dyn.robust, dyn.delta, it.partial, it.Hpp, pt_max);
  __shared__ int last_sh;
  __shared__ double sh2[16];
  const int n = it.se.Rt;
  int* ticket = reinterpret_cast<int*>((double*)it.scal + BA_TICKET_SLOT);
  if (threadIdx.x == 0) last_sh = atomicAdd(ticket, 1) == n - 1;
  __syncthreads();
  if (!last_sh) return;
  double* partial = it.partial;
  double v = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) v += atomicAdd(partial + i, 0.0);
  const double chi = block_sum(v, sh2);
  double m = 0;
  double* pose_diag = it.Hpp;
  if (it.se.det) {      // a deterministic window: the workgroups' slices of the diagonal sums, added in slice order
    for (int i = threadIdx.x; i < 6 * it.d.np; i += blockDim.x) {
      double sum = 0.0;
      for (int r = 0; r < n; ++r)
        sum += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<BA_AS1 unsigned long long*>(it.se.partial) + ((size_t)r * 6 * it.d.np + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      m = fmax(m, fabs(sum));
    }
  } else
  for (int i = threadIdx.x; i < 6 * it.d.np; i += blockDim.x)       // (read and put back to zero for the next stage)
    m = fmax(m, fabs(__longlong_as_double((long long)__hip_atomic_exchange(reinterpret_cast<unsigned long long*>(pose_diag + i), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))));
  if (threadIdx.x == 0) m = fmax(m, __longlong_as_double((long long)atomicExch(pt_max, 0ull)));
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh2[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    double mx = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) mx = fmax(mx, sh2[i]);
    it.scal[0] = chi; it.hscal[0] = chi; it.scal[3] = mx; it.hscal[3] = mx;
    atomicExch(ticket, 0);
    if (dyn.dev_lm) ba_lm_after_iter(it.lm, it.hlm, chi, mx);
  }
}
// last kernel of the TRIAL phase when the trial kernel does not fold it
__device__ __forceinline__ void kb_ba_reduce2_window(const BaItem* __restrict__ items, const BaDyn& dyn, int phase) {
  BA_ITEM(phase, 1)
  ba_reduce2_body(0, 1, it.partial, it.se.Rt > 0 ? it.se.Rt : it.nblk_p, it.scal, it.hscal);   // partial sums of kb_ba_trial_edges / kb_ba_trial_points
  ba_trial_publish(it, dyn);
}
extern "C" __global__ void __launch_bounds__(256) kb_ba_reduce2(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  kb_ba_reduce2_window(items, dyn, phase);
  if (dyn.dev_lm && blockIdx.z == 0 && threadIdx.x == 0) ba_round_count(items);
}
extern "C" __global__ void __launch_bounds__(256) kb_ba_classify(const BaItem* __restrict__ items, BaDyn dyn, int phase) {
  BA_ITEM(phase, it.nblk_e)
  ba_classify_body(blockIdx.x, it.nblk_e, it.d, it.poses[cur], it.pts[cur], dyn.chi2_th, dyn.set_level, it.flags);
  // the driver only needs the NUMBER of outliers of a window (cms_ba_stats); the flags stay on the device for cms_ba_read
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = __syncthreads_count(e < it.d.E && it.flags[e] != 0);
  if (threadIdx.x == 0 && n > 0) atomicAdd((int*)&it.lm->n_out[dyn.set_level ? 0 : 1], n);
}
// start of a stage: the windows' Levenberg state comes from the pinned host block the driver just filled (no H2D copy in the stream),
// the outlier counters are cleared by the first stage of a call
extern "C" __global__ void __launch_bounds__(64) kb_ba_lm_load(const BaItem* __restrict__ items, int n, int clear_counts) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n) return;
  BaLmDev L = *items[w].hlm;
  const BaLmDev old = *items[w].lm;
  L.n_out[0] = clear_counts ? 0 : old.n_out[0];      // the first stage of a call clears both counters (the final classification
  L.n_out[1] = clear_counts ? 0 : old.n_out[1];      // also runs when a stop request skips the second stage)
  *items[w].lm = L;
  reinterpret_cast<int*>(items[w].scal + 7)[0] = 0;        // ticket counter of the trial kernel's folded reduction (BA_TICKET_SLOT)
  items[w].scal[6] = 0.0;                                  // kb_ba_first_pass: the points' largest diagonal entry (BA_PTMAX_SLOT) ...
  for (int i = 0; i < 6 * items[w].d.np; ++i) items[w].Hpp[i] = 0.0;      // ... and the key frames' diagonal sums (the head of Hpp, unused by fused groups)
}
// after a classification: the windows' outlier counters -> pinned host block
extern "C" __global__ void __launch_bounds__(64) kb_ba_counts_publish(const BaItem* __restrict__ items, int n) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n) return;
  items[w].hlm->n_out[0] = items[w].lm->n_out[0];
  items[w].hlm->n_out[1] = items[w].lm->n_out[1];
}
