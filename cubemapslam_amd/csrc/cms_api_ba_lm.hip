// cms_api_ba_lm.hip -- Levenberg-Marquardt driver of the device-resident local BA, included by cms_api_ba.hip.
//
// Control flow restated from OptimizationAlgorithmLevenberg::solve (optimization_algorithm_levenberg.cpp:61-164),
// SparseOptimizer::optimize (sparse_optimizer.cpp:354-419) and the two-stage schedule of
// Optimizer::LocalBundleAdjustment (Optimizer.cpp:359-412), as a resumable per-window state machine: one host thread
// advances any number of windows in lock-step (every round it enqueues the next step of every unfinished window on
// that window's own HIP stream, then collects the few scalars each step returns).  Windows of different streams thus
// overlap on the device without competing host threads; a single window is just the n = 1 case.

struct BaLm {
  int iterations = 0, robust = 0;
  double delta = 0;
  int it = 0, qmax = 0, nBad = 0, done = 0, next = 0;   // next: 0 = start an iteration, 1 = one more trial, 2 = finished
  double lambda = -1, ni = 2, currentChi = 0, iniChi = 0, rho = 0;
  double chi_ini = 0, chi_fin = 0, lam_fin = 0;
};

static bool ba_stopped(const volatile uint8_t* stop) { return stop && *stop; }

// computeActiveErrors + activeRobustChi2 + buildSystem (+ computeLambdaInit on the first iteration); scalars -> h_pin
static int ba_enqueue_iter_start(cms_ba* b, const BaLm& st) {
  hipStream_t s = b->stream;
  const int cur = b->cur;
  ba_errors(b, cur, st.robust, st.delta, 0);
  hipLaunchKernelGGL(k_ba_lin_points, dim3(b->nblk_p), dim3(128), 0, s, b->d, (const double*)b->d_poses[cur],
                     (const double*)b->d_pts[cur], st.robust, st.delta, b->d_Hll, b->d_bl, b->d_Hpl);
  hipLaunchKernelGGL(k_ba_lin_poses, dim3(b->K, BA_POSE_CHUNKS), dim3(256), 0, s, b->d, (const double*)b->d_poses[cur],
                     (const double*)b->d_pts[cur], st.robust, st.delta, b->d_pose_partial);
  if (b->np > 0)
    hipLaunchKernelGGL(k_ba_pose_finish, dim3(b->np), dim3(64), 0, s, b->np, (const double*)b->d_pose_partial, b->d_Hpp, b->d_bp);
  if (st.it == 0) {
    HIPCHK(hipMemsetAsync(b->d_scal + 3, 0, sizeof(double), s));
    hipLaunchKernelGGL(k_ba_maxdiag, dim3(64), dim3(256), 0, s, b->np, b->P, (const double*)b->d_Hpp, (const double*)b->d_Hll, b->d_scal + 3);
  }
  HIPCHK(hipMemcpyAsync(b->h_pin, b->d_scal, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
  return CMS_OK;
}
static void ba_finish_iter_start(cms_ba* b, BaLm& st) {
  st.currentChi = b->h_pin[0];
  st.iniChi = st.currentChi;
  if (st.it == 0) { st.chi_ini = st.iniChi; st.lambda = 1e-5 * b->h_pin[3]; st.ni = 2; st.nBad = 0; }
  st.rho = 0; st.qmax = 0;
  st.next = 1;
}

// one Levenberg trial: setLambda + Schur solve + update + computeActiveErrors at the trial state; scalars -> h_pin
static int ba_enqueue_trial(cms_ba* b, const BaLm& st) {
  hipStream_t s = b->stream;
  const int cur = b->cur, nxt = cur ^ 1, n = 6 * b->np;
  const double lambda = st.lambda;
  if (b->solve_blk) {
    // fused trial (cms_ba_fused.hip): 5 launches, no reduced matrix in memory
    hipLaunchKernelGGL(k_ba_dinv, dim3((b->P + 255) / 256), dim3(256), 0, s, b->P, (const double*)b->d_Hll, (const double*)b->d_bl, lambda,
                       b->d_Dinv, b->d_db);
    // A window on its own is a latency problem: the tuple-chunk kernel (hundreds of short workgroups, 13 us) beats the
    // per-point kernel (64 workgroups walking 6 batches each, 40 us).  The per-point kernel pays off when several windows
    // share the launches and the chunk kernel becomes bandwidth bound -- the batched driver below uses it.
    const bool sp = b->sp.R > 0 && ba_knobs().schur_points;
    if (sp) {
      hipLaunchKernelGGL(k_ba_schur_points, dim3(b->sp.R), dim3(b->sp_threads), b->sp_lds, s, b->d, b->sp, (const double*)b->d_Hpl,
                         (const double*)b->d_Dinv, (const double*)b->d_db);
      hipLaunchKernelGGL(k_ba_schur_reduce, dim3(b->npairs), dim3(256), 0, s, b->sp, b->d_sp_sum);
    } else if (b->npairs > 0) {
      hipLaunchKernelGGL(k_ba_schur_chunks, dim3(b->nchunks), dim3(256), 0, s, b->d, (const int2*)b->d_chunk_range, (const int2*)b->d_tup,
                         (const double*)b->d_Hpl, (const double*)b->d_Dinv, (const double*)b->d_db, b->d_chunk_sum);
    }
    hipLaunchKernelGGL(k_ba_trial_solve, dim3(1), dim3(384), b->blk_lds, s, b->d, (const double*)b->d_Hpp, (const double*)b->d_bp, lambda,
                       (const int*)b->d_pair_of_block, (const int*)(sp ? b->d_sp_chunk_off : b->d_pair_chunk_off),
                       (const double*)(sp ? b->d_sp_sum : b->d_chunk_sum), (const double*)b->d_poses[cur], b->d_poses[nxt], b->d_x, b->d_scal);
    hipLaunchKernelGGL(k_ba_trial_points, dim3(b->nblk_p), dim3(128), 0, s, b->d, (const double*)b->d_bl, (const double*)b->d_Hpl,
                       (const double*)b->d_Dinv, (const double*)b->d_x, lambda, (const double*)b->d_pts[cur], b->d_pts[nxt],
                       (const double*)b->d_poses[nxt], st.robust, st.delta, b->d_partial);
    hipLaunchKernelGGL(k_ba_reduce2, dim3(1), dim3(256), 0, s, (const double*)b->d_partial, b->nblk_p, b->d_scal);
  } else {
    // larger windows: reduced matrix in global memory, scalar LDL^T
    if (n > 0)
      hipLaunchKernelGGL(k_ba_schur_init, dim3(std::min((n * n + 255) / 256, 256)), dim3(256), 0, s, b->np, (const double*)b->d_Hpp,
                         (const double*)b->d_bp, lambda, b->d_Hs, b->d_bs);
    hipLaunchKernelGGL(k_ba_dinv, dim3((b->P + 255) / 256), dim3(256), 0, s, b->P, (const double*)b->d_Hll, (const double*)b->d_bl, lambda,
                       b->d_Dinv, b->d_db);
    if (b->npairs > 0) {
      hipLaunchKernelGGL(k_ba_schur_chunks, dim3(b->nchunks), dim3(256), 0, s, b->d, (const int2*)b->d_chunk_range, (const int2*)b->d_tup,
                         (const double*)b->d_Hpl, (const double*)b->d_Dinv, (const double*)b->d_db, b->d_chunk_sum);
      hipLaunchKernelGGL(k_ba_schur_finish, dim3(b->npairs), dim3(64), 0, s, b->np, (const int*)b->d_pair_s1, (const int*)b->d_pair_s2,
                         (const int*)b->d_pair_chunk_off, (const double*)b->d_chunk_sum, b->d_Hs, b->d_bs);
    }
    if (b->solve_in_lds)
      hipLaunchKernelGGL(k_ba_solve_r192, dim3(1), dim3(512), b->solve_lds, s, n, (const double*)b->d_Hs, (const double*)b->d_bs, b->d_x, b->d_status);
    else
      hipLaunchKernelGGL(k_ba_solve, dim3(1), dim3(256), 0, s, n, b->d_Hs, b->d_bs, b->d_x, b->d_Dg, b->d_status);
    hipLaunchKernelGGL(k_ba_backsub, dim3(b->nblk_p), dim3(128), 0, s, b->d, (const double*)b->d_bl, (const double*)b->d_Hpl,
                       (const double*)b->d_Dinv, (const double*)b->d_x, lambda, (const double*)b->d_pts[cur], b->d_pts[nxt], b->d_partial);
    hipLaunchKernelGGL(k_ba_update_poses, dim3(1), dim3(64), 0, s, b->d, (const double*)b->d_x, (const double*)b->d_bp, lambda,
                       (const double*)b->d_poses[cur], b->d_poses[nxt], b->d_scal + 2);
    hipLaunchKernelGGL(k_ba_reduce, dim3(1), dim3(256), 0, s, (const double*)b->d_partial, b->nblk_p, b->d_scal + 2, 1);
    ba_errors(b, nxt, st.robust, st.delta, 1);
  }
  HIPCHK(hipMemcpyAsync(b->h_pin, b->d_scal, 5 * sizeof(double), hipMemcpyDeviceToHost, s));
  return CMS_OK;
}
static void ba_finish_trial(cms_ba* b, BaLm& st, const volatile uint8_t* stop) {
  const double* t = b->h_pin;
  int ok2 = 0;
  memcpy(&ok2, &t[4], sizeof(int));
  double tempChi = t[1];
  if (!ok2) tempChi = DBL_MAX;
  st.rho = (st.currentChi - tempChi);
  const double scale = t[2] + 1e-3;
  st.rho /= scale;
  if (st.rho > 0 && std::isfinite(tempChi)) {
    double alpha = 1. - std::pow((2 * st.rho - 1), 3);
    alpha = std::min(alpha, 2. / 3.);
    st.lambda *= std::max(1. / 3., alpha);
    st.ni = 2; st.currentChi = tempChi;
    b->cur ^= 1;   // discardTop(): the trial state becomes the estimate
  } else {
    st.lambda *= st.ni; st.ni *= 2;   // pop(): old estimate kept; stored edge errors stay those of the rejected trial (as in g2o)
  }
  ++st.qmax;
  if (st.rho < 0 && st.qmax < 10 && !ba_stopped(stop)) { st.next = 1; return; }   // do { } while (rho<0 && qmax<max && !terminate())
  ++st.done;
  st.chi_fin = st.currentChi; st.lam_fin = st.lambda;
  bool terminate = (st.qmax == 10 || st.rho == 0);
  if (!terminate) {
    if ((st.iniChi - st.currentChi) * 1e3 < st.iniChi) ++st.nBad; else st.nBad = 0;
    if (st.nBad >= 3) terminate = true;
  }
  ++st.it;
  st.next = (terminate || st.it >= st.iterations || ba_stopped(stop)) ? 2 : 0;
}

// SparseOptimizer::optimize(iterations) for every window, in lock-step
static int ba_optimize_stage_many(cms_ba** bas, int n, std::vector<BaLm>& st, const volatile uint8_t* stop) {
  std::vector<int> stepped(n);
  for (;;) {
    int nact = 0;
    for (int w = 0; w < n; ++w) {
      stepped[w] = -1;
      if (st[w].next == 2) continue;
      if (st[w].next == 0 && (st[w].it >= st[w].iterations || ba_stopped(stop))) { st[w].next = 2; continue; }
      HIPCHK(hipSetDevice(bas[w]->device));
      const int rc = st[w].next == 0 ? ba_enqueue_iter_start(bas[w], st[w]) : ba_enqueue_trial(bas[w], st[w]);
      if (rc) return rc;
      stepped[w] = st[w].next;
      ++nact;
    }
    if (nact == 0) break;
    for (int w = 0; w < n; ++w) {
      if (stepped[w] < 0) continue;
      HIPCHK(hipStreamSynchronize(bas[w]->stream));
      if (stepped[w] == 0) ba_finish_iter_start(bas[w], st[w]);
      else ba_finish_trial(bas[w], st[w], stop);
    }
  }
  HIPCHK(hipGetLastError());
  return CMS_OK;
}

// ---- batched variant: the windows share every launch (kb_ba_* kernels, blockIdx.z = window) and every synchronisation.
// Used when all windows fit the fused trial path and live on one device; results are identical to the per-window path.
static bool ba_can_batch(cms_ba** bas, int n) {
  const bool single_old = ba_knobs().single_host_lm;      // A/B: a single window through the host-driven per-window path
  if (n < (single_old ? 2 : 1) || n > BA_MAX_GROUP) return false;      // (a single window too: device-side Levenberg loop, edge-major kernels)
  for (int w = 0; w < n; ++w) if (!bas[w]->solve_blk || bas[w]->device != bas[0]->device) return false;
  return true;
}
static int ba_group_reserve(cms_ba* owner, int n) {
  if (owner->grp_cap >= n) return CMS_OK;
  // (device blocks of an outgrown group stay in the window's slabs until it is destroyed; the pinned ones go back to the pool)
  if (owner->grp_items_host) ba_pin_give(owner->device, owner->grp_items_host, owner->grp_pin_bytes[0]);
  if (owner->grp_scal_host) ba_pin_give(owner->device, owner->grp_scal_host, owner->grp_pin_bytes[1]);
  if (owner->grp_lm_host) ba_pin_give(owner->device, owner->grp_lm_host, owner->grp_pin_bytes[2]);
  owner->grp_items_host = nullptr; owner->grp_scal_host = nullptr; owner->grp_lm_host = nullptr;
  owner->grp_cap = 0;
  const int cap = std::max(n, 8);
  { char* q = nullptr; int rc = ba_alloc(owner, &q, (size_t)cap * sizeof(BaItem)); if (rc) return rc; owner->grp_items_dev = q; }
  { int rc = ba_alloc(owner, &owner->grp_scal_dev, (size_t)cap * 8); if (rc) return rc; }
  { char* q = nullptr; int rc = ba_alloc(owner, &q, (size_t)cap * sizeof(BaLmDev)); if (rc) return rc; owner->grp_lm_dev = q; }
  // pinned, device-visible and explicitly coherent (fine-grained, uncached on the device side): kernels publish the windows' state and the
  // round counter into these while they run, so visibility must not depend on HIP_HOST_COHERENT's default
  HIPCHK(ba_pin_take(owner->device, (size_t)cap * sizeof(BaItem), &owner->grp_items_host, &owner->grp_pin_bytes[0]));
  HIPCHK(ba_pin_take(owner->device, (size_t)cap * 8 * sizeof(double), (void**)&owner->grp_scal_host, &owner->grp_pin_bytes[1]));
  HIPCHK(ba_pin_take(owner->device, (size_t)cap * sizeof(BaLmDev), &owner->grp_lm_host, &owner->grp_pin_bytes[2]));
  n = cap;
  owner->grp_cap = n;
  return CMS_OK;
}
// static part of the windows' descriptions -> device, once per stage
static bool ba_all_sp(cms_ba** bas, int n) {
  for (int w = 0; w < n; ++w) if (bas[w]->sp.R <= 0) return false;
  return true;
}
// the edge-major Schur kernel (LDS accumulation, cms_ba_schur_edges.hip) runs when every window of the group has its work list and the
// per-point path is available too (the two share the block-free linearisation); CMS_BA_DETERMINISTIC=1 keeps the pair-owner kernel
static bool ba_use_se(cms_ba** bas, int n) {
  if (ba_knobs().host_lm) return false;   // (the host-driven A/B driver only knows the pair-owner kernel)
  for (int w = 0; w < n; ++w) if (bas[w]->se.nchunks <= 0 || bas[w]->det_points) return false;      // (det_points: deterministic windows on the pair-owner kernel)
  return true;
}
// ... and the edge-major trial kernel when, in addition, every point of every window has an observation
static bool ba_use_te(cms_ba** bas, int n) {
  if (ba_knobs().trial_points || !ba_use_se(bas, n)) return false;
  for (int w = 0; w < n; ++w) if (bas[w]->se.Rt <= 0) return false;
  return true;
}
// Workgroups (ranges of chunks) per window for the edge-major Schur kernel of a GROUP: one workgroup fills a CU (148 KB of LDS), so the launch
// should have at most as many workgroups as the chip has CUs -- 11 windows x 32 ranges = 352 workgroups run as one full round plus a round
// at 37 %.  Windows keep the 32 ranges their buffers are sized for as the upper limit.
static int ba_device_cus(int dev) {
  static std::mutex mu;
  static int cus_of[64] = {0};                  // compute units per device (a process may drive several)
  int cus = 256;
  std::lock_guard<std::mutex> lk(mu);
  if (dev >= 0 && dev < 64) {
    if (cus_of[dev] == 0) { hipDeviceProp_t pr; cus_of[dev] = hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    cus = cus_of[dev];
  }
  return cus;
}
static int ba_group_ranges(cms_ba** bas, int n) {
  const int cus = ba_device_cus(bas[0]->device);
  int R = BA_SE_RANGES;
  // CMS_BA_RESERVE_CUS=k: the group's Schur launch leaves k compute units free.  A Schur workgroup fills its CU (158 KB of LDS, 8 x 256 registers):
  // while a launch of 256 of them runs nothing else does -- the other window group's solve and trial kernels (16 workgroups; latency bound) and the frame
  // path wait for it to drain.  With a few CUs left over they run NEXT to it instead of behind it
  static const int reserve = [] { const char* v = getenv("CMS_BA_RESERVE_CUS"); return v ? std::max(0, atoi(v)) : 0; }();
  if (!ba_knobs().fixed_ranges && n > 0) R = std::max(4, std::min(BA_SE_RANGES, std::max(n, cus - reserve) / n));
  // deterministic windows: how a window's chunks are cut into workgroups decides which sums meet in which order, so the cut must not depend on
  // the company the window is optimised in -- always the count fixed at its creation (default 16: 16 windows x 16 = the chip once; the solve kernel adds the slices)
  if (n > 0 && bas[0]->deterministic && !bas[0]->det_points) return std::max(2, bas[0]->det_ranges);      // (fixed when the window was created; a group's windows share it: `kind`)
  // CMS_BA_RANGES_PER_WINDOW=k: k workgroups per window whatever the chip has (A/B: shorter workgroups let the other group's small kernels in sooner)
  static const int per_window = [] { const char* v = getenv("CMS_BA_RANGES_PER_WINDOW"); return v ? atoi(v) : 0; }();
  if (per_window > 0) R = std::max(2, std::min(BA_SE_RANGES, per_window));
  return R;
}
// Which kernels a group's rounds are made of: decided ONCE per group from the windows' lists and the knobs (ba_upload_items and the stage
// driver both use it).  fused: the Schur kernel linearises itself (edge-major kernels + three-lane solve); rm: windows with signature runs
// send them through the run-major body (needs the fused path and full 512-thread workgroups)
struct BaGroupMode { bool use_se, use_te, use_s3, fused, rm, gsum, run_wg, det; int se_waves; };
static BaGroupMode ba_group_mode(cms_ba** bas, int n) {
  BaGroupMode m;
  m.use_se = ba_use_se(bas, n); m.use_te = ba_use_te(bas, n);
  m.use_s3 = !ba_knobs().solve1;
  m.se_waves = BA_SE_THREADS / 64;
  size_t rm_lds = 0;
  for (int w = 0; w < n; ++w) {
    m.use_s3 = m.use_s3 && bas[w]->solve_blk3;
    if (bas[w]->se_waves > 0) m.se_waves = std::min(m.se_waves, bas[w]->se_waves);
    rm_lds = std::max(rm_lds, bas[w]->rm_lds);
  }
  m.fused = m.use_se && m.use_te && m.use_s3 && !ba_knobs().no_fused;
  m.rm = m.fused && rm_lds > 0 && m.se_waves == BA_SE_THREADS / 64 && !ba_knobs().runs_as_edges;
  if (ba_knobs().se_waves_cap > 0 && !ba_knobs().rm_valu) m.se_waves = std::max(1, std::min(m.se_waves, ba_knobs().se_waves_cap));      // (the MFMA body takes any count)
  // one global copy of the reduced system per window, added to by all its workgroups (not with the A/B knobs that want the slices or launch a
  // kernel of the round twice)
  // deterministic windows (fused chain, fixed order): cms_ba_optimize_many forms groups of one kind, so either all windows are or none is
  m.det = m.fused && n > 0 && bas[0]->deterministic && !bas[0]->det_points;
  for (int w = 0; w < n; ++w) if ((bas[w]->deterministic && !bas[w]->det_points) != m.det) m.det = false;
  m.gsum = m.fused && ba_knobs().global_sum && !ba_knobs().separate_reduce && ba_knobs().dup == 0 && !m.det;      // (global FP64 atomics have no order)
  // the runs through one-wavefront workgroups (cms_ba_schur_runwg.hip): they add to the global copy, so they need it; the MFMA body only
  m.run_wg = m.rm && m.gsum && ba_knobs().run_wg && !ba_knobs().rm_valu && !m.det;
  return m;
}
static int ba_upload_items(cms_ba** bas, int n) {
  cms_ba* g = bas[0];
  BaItem* items = reinterpret_cast<BaItem*>(g->grp_items_host);
  const bool all_sp = ba_all_sp(bas, n);
  const BaGroupMode gm = ba_group_mode(bas, n);
  const bool use_se = gm.use_se;
  for (int w = 0; w < n; ++w) {
    cms_ba* b = bas[w];
    BaItem& it = items[w];
    it.d = b->d;
    it.poses[0] = b->d_poses[0]; it.poses[1] = b->d_poses[1]; it.pts[0] = b->d_pts[0]; it.pts[1] = b->d_pts[1];
    it.Hll = b->d_Hll; it.bl = b->d_bl; it.Hpl = b->d_Hpl; it.Hpp = b->d_Hpp; it.bp = b->d_bp; it.pose_partial = b->d_pose_partial;
    it.Dinv = b->d_Dinv; it.db = b->d_db; it.chunk_sum = b->d_chunk_sum; it.x = b->d_x; it.partial = b->d_partial;
    it.scal = g->grp_scal_dev + 8 * w; it.hscal = g->grp_scal_host + 8 * w;
    it.chunk_range = b->d_chunk_range; it.tup = b->d_tup; it.pair_of_block = b->d_pair_of_block; it.pair_chunk_off = b->d_pair_chunk_off;
    it.flags = b->d_flags;
    it.nblk_e = b->nblk_e; it.nblk_p = b->nblk_p; it.nchunks = b->nchunks; it.block_free = (all_sp || use_se) ? 1 : 0;
    it.sp = b->sp;
    it.lm = reinterpret_cast<BaLmDev*>(g->grp_lm_dev) + w; it.hlm = reinterpret_cast<BaLmDev*>(g->grp_lm_host) + w;
    if (!all_sp) it.sp.R = 0;                       // one launch sequence for the whole group: per-point Schur only if every window has it
    if (all_sp) { it.chunk_sum = b->d_sp_sum; it.pair_chunk_off = b->d_sp_chunk_off; }
    it.se = b->se;
    if (!use_se) { it.se.R = 0; it.se.Rt = 0; }
    if (!use_se) { it.se.nchunks = 0; it.se.n_rm = 0; it.se.R_rm = 0; }
    if (use_se && !gm.use_te) it.se.Rt = 0;
    if (use_se) { it.chunk_sum = b->d_se_sum; it.pair_chunk_off = b->d_se_chunk_off; it.pair_of_block = b->d_se_pob; }
    if (use_se) {
      // workgroups of this window in the group's Schur launch: one workgroup fills a CU, so the group shares the chip's CUs; a window's share
      // is split between its run chunks and its left-over chunks (without the run-major body the edge-major one takes every chunk)
      if (!gm.rm) it.se.n_rm = 0;
      ba_se_split(it.se, ba_group_ranges(bas, n), gm.run_wg);
    }
    it.se.gsum = (use_se && gm.gsum) ? 1 : 0;
    it.se.det = gm.det ? 1 : 0;
    // the run-major body's wavefronts take the left-over chunks strided (one or two each, behind their runs) instead of cut by cost (the whole range of the
    // window's last workgroups: 145 chunks' ds_add_f64 on two or three CUs' LDS pipes).  Found while ordering the deterministic kernel's additions:
    // 94 -> 84-87 us per 16-window launch alone, +1.9 % frames/s in bench.py's step (three A/B pairs, profiles/r06_det_experiment.txt).  CMS_BA_LEFT_BY_COST=1: round 4's cut
    { static const bool left_by_cost = getenv("CMS_BA_LEFT_BY_COST") != nullptr; it.se.strided = left_by_cost ? 0 : 1; }
    b->grp_se = it.se;
    if (use_se && b->d_se_partial) {
      // the global copy of the reduced system (slice 0) must be zero when the first round of a gsum group adds to it; the consuming solve kernel
      // leaves it zero, a slice-storing group (CMS_BA_NO_GLOBAL_SUM, mixed groups) does not
      if (it.se.gsum && !b->gsum_clean) {
        HIPCHK(hipMemsetAsync(b->d_se_partial, 0, (size_t)b->se.npairs2 * 42 * sizeof(double), g->stream));
        HIPCHK(hipMemsetAsync(b->d_se_bp_partial, 0, (size_t)b->np * 6 * sizeof(double), g->stream));
      }
      // (dirty from here on: only a call that ends with every round's solve kernel through leaves the copy zero again -- ba_optimize_group
      // marks it clean at its successful end; after an error or an aborted call the next gsum group clears it)
      b->gsum_clean = false;
    }
  }
  // (fetched from the pinned block by a kernel instead of a copy-engine transfer: one queue entry on the group's own stream; measured neutral
  // inside the bench's step -- CMS_BA_ITEMS_COPY_ENGINE=1 is the transfer of round 3)
  static_assert(sizeof(BaItem) % 16 == 0, "BaItem is copied in 16-byte words");
  static const bool items_by_copy_engine = getenv("CMS_BA_ITEMS_COPY_ENGINE") != nullptr;      // developer A/B
  if (items_by_copy_engine) HIPCHK(hipMemcpyAsync(g->grp_items_dev, items, (size_t)n * sizeof(BaItem), hipMemcpyHostToDevice, g->stream));
  else {      // (wavefront-sized workgroups: a 1024-thread workgroup waited up to a millisecond for sixteen free wavefront slots on ONE CU next to the frame path)
    const int n16 = (int)((size_t)n * sizeof(BaItem) / 16);
    hipLaunchKernelGGL(k_copy16, dim3((n16 + 63) / 64), dim3(64), 0, g->stream, (uint4*)g->grp_items_dev, (const uint4*)items, n16);
  }
  HIPCHK(hipGetLastError());
  return CMS_OK;
}
static int ba_optimize_stage_batched(cms_ba** bas, int n, std::vector<BaLm>& st, const volatile uint8_t* stop) {
  cms_ba* g = bas[0];
  hipStream_t s = g->stream;
  const BaItem* ditems = reinterpret_cast<const BaItem*>(g->grp_items_dev);
  int max_e = 0, max_p = 0, max_K = 0, max_np = 0, max_chunks = 0, max_P = 0, max_R = 0, max_spt = 64, max_pairs = 0;
  size_t lds = 0, sp_lds = 0;
  const bool all_sp = ba_all_sp(bas, n);
  for (int w = 0; w < n; ++w) {
    max_R = std::max(max_R, bas[w]->sp.R); max_spt = std::max(max_spt, bas[w]->sp_threads); max_pairs = std::max(max_pairs, bas[w]->npairs);
    sp_lds = std::max(sp_lds, bas[w]->sp_lds);
    max_e = std::max(max_e, bas[w]->nblk_e); max_p = std::max(max_p, bas[w]->nblk_p); max_K = std::max(max_K, bas[w]->K);
    max_np = std::max(max_np, bas[w]->np); max_chunks = std::max(max_chunks, bas[w]->nchunks); max_P = std::max(max_P, bas[w]->P);
    lds = std::max(lds, bas[w]->blk_lds);
  }
  int rc = CMS_OK;
  {   // the classification after this stage counts outliers in the device-side state block: clear its slot
    BaLmDev* hlm0 = reinterpret_cast<BaLmDev*>(g->grp_lm_host);
    for (int w = 0; w < n; ++w) memset(&hlm0[w], 0, sizeof(BaLmDev));
    hipLaunchKernelGGL(kb_ba_lm_load, dim3((n + 63) / 64), dim3(64), 0, s, ditems, n, st[0].robust ? 1 : 0);
  }
  BaDyn dyn{};
  memset(&dyn, 0, sizeof(dyn));
  dyn.robust = st[0].robust; dyn.delta = st[0].delta; dyn.chi2_th = 5.991; dyn.set_level = 0;
  for (;;) {
    int n_iter = 0, n_trial = 0;
    for (int w = 0; w < n; ++w) {
      dyn.phase[w] = BA_PHASE_IDLE;
      if (st[w].next == 0 && (st[w].it >= st[w].iterations || ba_stopped(stop))) st[w].next = 2;
      if (st[w].next == 0) { dyn.phase[w] = BA_PHASE_ITER; ++n_iter; }
      else if (st[w].next == 1) { dyn.phase[w] = BA_PHASE_TRIAL; ++n_trial; }
      dyn.cur[w] = (uint8_t)bas[w]->cur; dyn.first_iter[w] = st[w].it == 0; dyn.lambda[w] = st[w].lambda;
    }
    if (n_iter + n_trial == 0) break;
    if (n_iter) {
      hipLaunchKernelGGL(kb_ba_errors, dim3(max_e, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_ITER);
      hipLaunchKernelGGL(kb_ba_reduce, dim3(1, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_ITER);
      hipLaunchKernelGGL(kb_ba_lin_points, dim3(max_p, 1, n), dim3(128), 0, s, ditems, dyn, (int)BA_PHASE_ITER);
      hipLaunchKernelGGL(kb_ba_lin_poses, dim3(max_K, BA_POSE_CHUNKS, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_ITER);
      hipLaunchKernelGGL(kb_ba_pose_finish, dim3(std::max(max_np, 1), 1, n), dim3(64), 0, s, ditems, dyn, (int)BA_PHASE_ITER);
      hipLaunchKernelGGL(kb_ba_maxdiag, dim3(1, 1, n), dim3(1024), 0, s, ditems, dyn, (int)BA_PHASE_ITER);
    }
    if (n_trial) {
      if (!all_sp)    // the point-major Schur kernel and the trial-points kernel invert Hll + lambda I themselves
        hipLaunchKernelGGL(kb_ba_dinv, dim3((max_P + 255) / 256, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_TRIAL);
      if (all_sp) {
        hipLaunchKernelGGL(kb_ba_schur_points, dim3(max_R, 1, n), dim3(max_spt), sp_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        hipLaunchKernelGGL(kb_ba_schur_reduce, dim3(max_pairs, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_TRIAL);
      } else if (max_chunks > 0) {
        hipLaunchKernelGGL(kb_ba_schur_chunks, dim3(max_chunks, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_TRIAL);
      }
      hipLaunchKernelGGL(kb_ba_trial_solve, dim3(1, 1, n), dim3(384), lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
      hipLaunchKernelGGL(kb_ba_trial_points, dim3(max_p, 1, n), dim3(128), 0, s, ditems, dyn, (int)BA_PHASE_TRIAL);
      hipLaunchKernelGGL(kb_ba_reduce2, dim3(1, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_TRIAL);
    }
    HIPCHK(hipStreamSynchronize(s));          // the last kernel of each phase wrote the scalars into the pinned mirror
    for (int w = 0; w < n; ++w) {
      if (dyn.phase[w] == BA_PHASE_IDLE) continue;
      memcpy(bas[w]->h_pin, g->grp_scal_host + 8 * w, 8 * sizeof(double));
      if (dyn.phase[w] == BA_PHASE_ITER) ba_finish_iter_start(bas[w], st[w]);
      else ba_finish_trial(bas[w], st[w], stop);
    }
  }
  HIPCHK(hipGetLastError());
  return CMS_OK;
}
// ---- the same stage with the Levenberg logic on the device: every kernel reads the window's BaLmDev to know whether it has work, the
// last kernel of a phase updates it, and the host enqueues whole rounds (one Levenberg trial of every window) back to back without ever
// synchronising the stream inside a stage.  Without a stop flag two rounds are kept in the queue; with one, a round is only added once
// the previous one has finished and the flag has been looked at -- the abort point is then the trial boundary g2o's terminate() polls
// (optimization_algorithm_levenberg.cpp:127, sparse_optimizer.cpp:376).  Env CMS_BA_HOST_LM=1 selects the host-driven variant above (A/B).
static int ba_optimize_stage_batched_dev(cms_ba** bas, int n, std::vector<BaLm>& st, const volatile uint8_t* stop) {
  cms_ba* g = bas[0];
  hipStream_t s = g->stream;
  const BaItem* ditems = reinterpret_cast<const BaItem*>(g->grp_items_dev);
  int max_e = 0, max_p = 0, max_K = 0, max_np = 0, max_chunks = 0, max_P = 0, max_R = 0, max_spt = 64, max_pairs = 0, max_it = 0;
  size_t lds = 0, sp_lds = 0;
  const bool all_sp = ba_all_sp(bas, n);
  for (int w = 0; w < n; ++w) {
    max_R = std::max(max_R, bas[w]->sp.R); max_spt = std::max(max_spt, bas[w]->sp_threads); max_pairs = std::max(max_pairs, bas[w]->npairs);
    sp_lds = std::max(sp_lds, bas[w]->sp_lds);
    max_e = std::max(max_e, bas[w]->nblk_e); max_p = std::max(max_p, bas[w]->nblk_p); max_K = std::max(max_K, bas[w]->K);
    max_np = std::max(max_np, bas[w]->np); max_chunks = std::max(max_chunks, bas[w]->nchunks); max_P = std::max(max_P, bas[w]->P);
    lds = std::max(lds, bas[w]->blk_lds);
    max_it = std::max(max_it, st[w].iterations);
  }
  BaLmDev* hlm = reinterpret_cast<BaLmDev*>(g->grp_lm_host);
  for (int w = 0; w < n; ++w) {
    BaLmDev& L = hlm[w];
    memset(&L, 0, sizeof(L));
    L.lambda = st[w].lambda; L.ni = st[w].ni; L.it = st[w].it; L.iterations = st[w].iterations; L.nBad = st[w].nBad; L.done = st[w].done;
    L.cur = bas[w]->cur;
    L.next = (st[w].it >= st[w].iterations || ba_stopped(stop)) ? 2 : 0;
  }
  // the device copy is filled by a one-wave kernel reading the pinned block (a copy command in the stream costs ~20 us of queue time)
  hipLaunchKernelGGL(kb_ba_lm_load, dim3((n + 63) / 64), dim3(64), 0, s, ditems, n, st[0].robust ? 1 : 0);
  BaDyn dyn{};
  memset(&dyn, 0, sizeof(dyn));
  dyn.robust = st[0].robust; dyn.delta = st[0].delta; dyn.chi2_th = 5.991; dyn.set_level = 0; dyn.dev_lm = 1; dyn.fold_finish = 1;
  bool first_round = false;
  for (int w = 0; w < n; ++w) first_round = first_round || st[w].it == 0;
  const BaGroupMode gm = ba_group_mode(bas, n);
  const bool use_se = gm.use_se, use_te = gm.use_te;
  bool iter_phase = true;           // fused linearisation: only the first round of a stage has an ITER phase
  // trial solve: three lanes per 6x6 block where every window allows it (CMS_BA_SOLVE1=1: one lane per block, the single-window kernel's scheme)
  const bool use_s3 = gm.use_s3;
  int s3_threads = 64; size_t lds3 = 0;
  for (int w = 0; w < n; ++w) { s3_threads = std::max(s3_threads, ba_s3_threads(bas[w]->np)); lds3 = std::max(lds3, bas[w]->blk3_lds); }
  {
    // CMS_BA_SOLVE_LDS_KB=k: the solve kernel asks for k KB of LDS whatever it needs -- with ~150 it has its compute unit to itself as far as LDS-using kernels
    // go (developer A/B: how much of its 2-3x longer duration inside bench.py's step is other kernels' wavefronts on the same CU?)
    static const int solve_kb = [] { const char* v = getenv("CMS_BA_SOLVE_LDS_KB"); return v ? atoi(v) : 0; }();
    if (solve_kb > 0) lds3 = std::max(lds3, std::min((size_t)solve_kb * 1024, (size_t)BA_LDS_CEILING));
  }
  int max_seR = 0, max_np2 = 0, max_Rt = 0; size_t se_lds = 0, te_lds = 0, rm_lds = 0;
  const int se_waves = gm.se_waves;
  bool any_runs = false, any_rw0 = false, any_rw1 = false;
  for (int w = 0; w < n; ++w) {
    const BaSe& gs = bas[w]->grp_se;             // this group's split of the window (ba_upload_items)
    if (gm.run_wg) { any_rw0 = any_rw0 || gs.n_rmA > 0; any_rw1 = any_rw1 || gs.n_rm > gs.n_rmA; }
    max_seR = std::max(max_seR, gs.R_rm + gs.R); max_np2 = std::max(max_np2, gs.npairs2); se_lds = std::max(se_lds, bas[w]->se_lds_fixed);
    any_runs = any_runs || gs.R_rm > 0;
    if (gs.R_rm > 0) rm_lds = std::max(rm_lds, bas[w]->rm_lds);
    max_Rt = std::max(max_Rt, gs.Rt); te_lds = std::max(te_lds, ((size_t)24 * bas[w]->K + 6 * (size_t)std::max(bas[w]->np, 1)) * sizeof(double));
  }
  // One "round" = the launches of one Levenberg trial (plus the linearisation in front of it for the windows that start an iteration).
  // The host never synchronises the stream inside a stage: a synchronisation -- or an event -- after a round makes the chip publish
  // its caches before the next kernel starts, 19 us of idle time per round, 0.4 ms per window group.  Instead the last kernel of a
  // round bumps a counter in pinned host memory (next to the windows' Levenberg state, which the kernels mirror there too) and the
  // host keeps at most two rounds in the queue: the one that is running and the one behind it (one, if the caller passed a stop flag:
  // the flag is then honoured at the very next trial boundary).  It looks at the mirrored state and at the caller's stop flag before
  // every round it adds; the price is at most two rounds of idle launches after the last window finished.
  se_lds += (size_t)se_waves * ((size_t)64 * 18 * sizeof(double) + 64 * sizeof(int));      // the group runs with the wavefront count its largest window allows
  se_lds = std::max(se_lds, rm_lds);                                                        // (the run-major body's chunk buffers, when a window has runs)
  const int se_threads = 64 * se_waves;
  // one-wavefront workgroups: units per window so that the launch fills the chip's wavefront slots about once
  const size_t rw_lds = ba_rw_lds(std::max(max_K, 1));
  const int rw_units0 = std::max(8, std::min(BA_RW_CUTS, ba_device_cus(bas[0]->device) * ba_knobs().rw_waves_cu0 / std::max(n, 1)));
  const int rw_units1 = std::max(8, std::min(BA_RW_CUTS, ba_device_cus(bas[0]->device) * ba_knobs().rw_waves_cu1 / std::max(n, 1)));
  // linearisation inside the Schur kernel (cms_ba_schur_edges.hip, FUSED): needs the edge-major kernels and the three-lane solve
  const bool fused = gm.fused;
  // the range slices summed by the solve kernel's assembly (kb_ba_trial_solve3r) instead of by a launch of their own
  const bool solve_reduces = fused && (gm.gsum || (!ba_knobs().separate_reduce && max_seR <= ba_knobs().solve_reduce_max));
  dyn.fused_lin = fused ? 1 : 0;
  {
    // slices (deterministic windows, CMS_BA_NO_GLOBAL_SUM): the solve kernel adds them into LDS with all its threads first, where the LDS allows it (<= 21 free key frames)
    static const bool no_presum = getenv("CMS_BA_NO_SOLVE_PRESUM") != nullptr;      // developer A/B
    const size_t need = lds3 + (size_t)max_np2 * 42 * sizeof(double);
    if (use_s3 && solve_reduces && !gm.gsum && !no_presum && need <= BA_LDS_CEILING) { dyn.solve_presum = 1; lds3 = need; }
  }
  // (deterministic windows, se.det: kb_ba_first_pass adds the key frames' diagonal sums -- lambda starts from their maximum -- in chunk and slice order)
  const bool first_pass = fused && use_te && first_round && !ba_knobs().separate_first_pass;      // (a stage's windows all start at it == 0)
  // the trial kernel sums its own partial sums and decides the trial (kb_ba_trial_edges) -- not with CMS_BA_DUP: the deciding kernel is not
  // idempotent (its last workgroup advances the Levenberg state and counts the round), a duplicated launch would evaluate a different state
  dyn.fold_reduce = (use_te && !ba_knobs().separate_reduce2 && ba_knobs().dup == 0) ? 1 : 0;
  int k = 0;
  const int pk = g->prof_kernel;
  const int dup = ba_knobs().dup;   // developer knob: launch kernel <id> of every round twice (1 lin, 3 the Schur kernel, 4 schur_reduce, 5 trial_solve,
                                    // 6 the trial kernel: idempotent once the global sum and the folded reduce are off, which dup != 0 does): how much does the step pay for it?
  auto bracket = [&](int id, int which) {     // HIP events around the one kernel the caller asked to have timed (cms_ba_profile_kernel)
    if (pk != id) return;
    const size_t i = 2 * (size_t)k + which;
    while (g->prof_ev.size() <= i) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return; g->prof_ev.push_back(e); }
    hipEventRecord(g->prof_ev[i], s);
  };
  auto enqueue_round = [&]() {
      if (first_round && first_pass) {      // errors + chi2 + lambda's maximum diagonal of the stage's starting estimate in one launch (kb_ba_first_pass)
        bracket(1, 0);
        hipLaunchKernelGGL(kb_ba_first_pass, dim3(max_Rt, 1, n), dim3(BA_TE_THREADS), te_lds, s, ditems, dyn, (int)BA_PHASE_ITER);
        bracket(1, 1);
        iter_phase = false;
      } else if (first_round) {     // residuals of the stage's starting estimate: later iterations carry the accepted trial's over (every window starts at it == 0)
        hipLaunchKernelGGL(kb_ba_errors, dim3(max_e, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_ITER);
        hipLaunchKernelGGL(kb_ba_reduce, dim3(1, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_ITER);
      }
      first_round = false;
      if (iter_phase) {
        const int npb = (max_P + 255) / 256;
        bracket(1, 0);
        hipLaunchKernelGGL(kb_ba_lin, dim3(npb + max_K * BA_POSE_CHUNKS, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_ITER, npb);
        if (dup == 1) hipLaunchKernelGGL(kb_ba_lin, dim3(npb + max_K * BA_POSE_CHUNKS, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_ITER, npb);
        bracket(1, 1);
        bracket(2, 0);
        hipLaunchKernelGGL(kb_ba_maxdiag, dim3(1, 1, n), dim3(1024), 0, s, ditems, dyn, (int)BA_PHASE_ITER);      // + the pose slice sums (fold_finish)
        bracket(2, 1);
      }
      if (fused) iter_phase = false;
      if (!all_sp && !use_se)
        hipLaunchKernelGGL(kb_ba_dinv, dim3((max_P + 255) / 256, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_TRIAL);
      if (use_se) {
        bracket(3, 0);
        if (fused && any_runs && ba_knobs().rm_valu) {
          hipLaunchKernelGGL(kb_ba_lin_schur_runs_valu, dim3(max_seR, 1, n), dim3(se_threads), se_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
          if (dup == 3) hipLaunchKernelGGL(kb_ba_lin_schur_runs_valu, dim3(max_seR, 1, n), dim3(se_threads), se_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        } else if (fused && gm.run_wg && (any_rw0 || any_rw1)) {
          // the runs: one wavefront per workgroup, as many units per window as fill the chip once (class 1 first: its units are the longer ones);
          // then the left-over chunks, edge-major
          if (any_rw1) hipLaunchKernelGGL(kb_ba_lin_schur_run_wg1, dim3(rw_units1, 1, n), dim3(64), rw_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL, rw_units1);
          if (any_rw0) hipLaunchKernelGGL(kb_ba_lin_schur_run_wg0, dim3(rw_units0, 1, n), dim3(64), rw_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL, rw_units0);
          if (max_seR > 0) hipLaunchKernelGGL(kb_ba_lin_schur_edges, dim3(max_seR, 1, n), dim3(se_threads), se_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        } else if (fused && gm.det) {      // additions in a fixed order, slices instead of the global copy
          if (any_runs) hipLaunchKernelGGL(kb_ba_lin_schur_runs_det, dim3(max_seR, 1, n), dim3(se_threads), se_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
          else hipLaunchKernelGGL(kb_ba_lin_schur_edges_det, dim3(max_seR, 1, n), dim3(se_threads), se_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        } else if (fused && any_runs) {
          hipLaunchKernelGGL(kb_ba_lin_schur_runs, dim3(max_seR, 1, n), dim3(se_threads), se_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
          if (dup == 3) hipLaunchKernelGGL(kb_ba_lin_schur_runs, dim3(max_seR, 1, n), dim3(se_threads), se_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        } else if (fused) {
          hipLaunchKernelGGL(kb_ba_lin_schur_edges, dim3(max_seR, 1, n), dim3(se_threads), se_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
          if (dup == 3) hipLaunchKernelGGL(kb_ba_lin_schur_edges, dim3(max_seR, 1, n), dim3(se_threads), se_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        } else {
          hipLaunchKernelGGL(kb_ba_schur_edges, dim3(max_seR, 1, n), dim3(se_threads), se_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
          if (dup == 3) hipLaunchKernelGGL(kb_ba_schur_edges, dim3(max_seR, 1, n), dim3(se_threads), se_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        }
        bracket(3, 1);
        bracket(4, 0);
        if (!solve_reduces) hipLaunchKernelGGL(kb_ba_schur_edges_reduce, dim3(max_np2, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        bracket(4, 1);
      } else if (all_sp) {
        bracket(3, 0);
        hipLaunchKernelGGL(kb_ba_schur_points, dim3(max_R, 1, n), dim3(max_spt), sp_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        if (dup == 3) hipLaunchKernelGGL(kb_ba_schur_points, dim3(max_R, 1, n), dim3(max_spt), sp_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        bracket(3, 1);
        bracket(4, 0);
        hipLaunchKernelGGL(kb_ba_schur_reduce, dim3(max_pairs, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        if (dup == 4) hipLaunchKernelGGL(kb_ba_schur_reduce, dim3(max_pairs, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        bracket(4, 1);
      } else if (max_chunks > 0) {
        hipLaunchKernelGGL(kb_ba_schur_chunks, dim3(max_chunks, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_TRIAL);
      }
      bracket(5, 0);
      if (use_s3 && solve_reduces && dyn.solve_presum) {
        hipLaunchKernelGGL(kb_ba_trial_solve3rp, dim3(1, 1, n), dim3(s3_threads), lds3, s, ditems, dyn, (int)BA_PHASE_TRIAL);
      } else if (use_s3 && solve_reduces) {
        hipLaunchKernelGGL(kb_ba_trial_solve3r, dim3(1, 1, n), dim3(s3_threads), lds3, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        if (dup == 5) hipLaunchKernelGGL(kb_ba_trial_solve3r, dim3(1, 1, n), dim3(s3_threads), lds3, s, ditems, dyn, (int)BA_PHASE_TRIAL);
      } else if (use_s3) {
        hipLaunchKernelGGL(kb_ba_trial_solve3, dim3(1, 1, n), dim3(s3_threads), lds3, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        if (dup == 5) hipLaunchKernelGGL(kb_ba_trial_solve3, dim3(1, 1, n), dim3(s3_threads), lds3, s, ditems, dyn, (int)BA_PHASE_TRIAL);
      } else {
        hipLaunchKernelGGL(kb_ba_trial_solve, dim3(1, 1, n), dim3(384), lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        if (dup == 5) hipLaunchKernelGGL(kb_ba_trial_solve, dim3(1, 1, n), dim3(384), lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
      }
      bracket(5, 1);
      bracket(6, 0);
      if (use_te) {
        hipLaunchKernelGGL(kb_ba_trial_edges, dim3(max_Rt, 1, n), dim3(BA_TE_THREADS), te_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        if (dup == 6) hipLaunchKernelGGL(kb_ba_trial_edges, dim3(max_Rt, 1, n), dim3(BA_TE_THREADS), te_lds, s, ditems, dyn, (int)BA_PHASE_TRIAL);
      } else {
        hipLaunchKernelGGL(kb_ba_trial_points, dim3(max_p, 1, n), dim3(128), 0, s, ditems, dyn, (int)BA_PHASE_TRIAL);
        if (dup == 6) hipLaunchKernelGGL(kb_ba_trial_points, dim3(max_p, 1, n), dim3(128), 0, s, ditems, dyn, (int)BA_PHASE_TRIAL);
      }
      bracket(6, 1);
      bracket(7, 0);
      if (!dyn.fold_reduce) hipLaunchKernelGGL(kb_ba_reduce2, dim3(1, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_TRIAL);      // (otherwise the trial kernel's last workgroup per window)
      bracket(7, 1);
      ++k;
  };
  const volatile BaLmDev* vh = reinterpret_cast<const volatile BaLmDev*>(hlm);
  hipError_t werr = hipSuccess;
  for (long spins = 0;;) {
    bool any = false;
    for (int w = 0; w < n; ++w) any = any || vh[w].next != 2;
    if (!any || (k > 0 && ba_stopped(stop))) break;
    if (k - vh[0].rounds < (stop ? 1 : 2)) { enqueue_round(); spins = 0; continue; }
    if (spins < 256) std::this_thread::yield();                         // a round takes 100-250 us: spin briefly, then back off so that
    else std::this_thread::sleep_for(std::chrono::microseconds(20));    // several groups' host threads do not burn a core each
    if ((++spins & 0xFFF) == 0 && hipStreamQuery(s) != hipErrorNotReady) {      // the stream drained (or failed) without the counter moving
      werr = hipStreamSynchronize(s);
      if (werr != hipSuccess || k - vh[0].rounds >= 2) { if (werr == hipSuccess) werr = hipErrorUnknown; break; }
    }
  }
  // (the group driver's waits stay the runtime's spinning ones: they are short -- the mirror already said the windows are through -- and sit on the
  // chain's critical path; queries between sleeps cost 0.3-0.5 ms per call here, measured)
  const hipError_t serr = hipStreamSynchronize(s);                  // final state of every window: mirrored into hlm by the kernels
  HIPCHK(werr);
  HIPCHK(serr);
  HIPCHK(hipGetLastError());
  if (pk > 0)
    for (int r = 0; r < k && 2 * (size_t)r + 1 < g->prof_ev.size(); ++r) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, g->prof_ev[2 * r], g->prof_ev[2 * r + 1]) == hipSuccess) { g->prof_ms += ms; ++g->prof_launches; }
    }
  for (int w = 0; w < n; ++w) {
    const BaLmDev& L = hlm[w];
    BaLm& t = st[w];
    t.lambda = L.lambda; t.ni = L.ni; t.currentChi = L.currentChi; t.iniChi = L.iniChi; t.rho = L.rho; t.it = L.it; t.qmax = L.qmax;
    t.nBad = L.nBad; t.done = L.done; t.next = 2;
    if (L.done > 0 || L.it > 0) { t.chi_ini = L.chi_ini; t.chi_fin = L.chi_fin; t.lam_fin = L.lam_fin; }
    if (L.next == 1 && L.qmax > 0 && ba_stopped(stop)) {   // stopped inside an iteration's trial loop (after a rejected trial): g2o leaves the
      ++t.done; t.chi_ini = L.chi_ini; t.chi_fin = L.currentChi; t.lam_fin = L.lambda;   // do-while, counts the iteration and keeps the old estimate
    }
    bas[w]->cur = L.cur;
  }
  return CMS_OK;
}

static int ba_classify_batched(cms_ba** bas, int n, int set_level, std::vector<int>& counts) {
  cms_ba* g = bas[0];
  hipStream_t s = g->stream;
  BaDyn dyn{};
  memset(&dyn, 0, sizeof(dyn));
  dyn.chi2_th = 5.991; dyn.set_level = set_level;
  int max_e = 0;
  for (int w = 0; w < n; ++w) {
    dyn.phase[w] = BA_PHASE_CLASSIFY; dyn.cur[w] = (uint8_t)bas[w]->cur;
    max_e = std::max(max_e, bas[w]->nblk_e);
  }
  const BaItem* ditems = reinterpret_cast<const BaItem*>(g->grp_items_dev);
  hipLaunchKernelGGL(kb_ba_classify, dim3(max_e, 1, n), dim3(256), 0, s, ditems, dyn, (int)BA_PHASE_CLASSIFY);
  hipLaunchKernelGGL(kb_ba_counts_publish, dim3((n + 63) / 64), dim3(64), 0, s, ditems, n);      // counts -> pinned block; flags stay on the device
  HIPCHK(hipStreamSynchronize(s));
  const BaLmDev* hlm = reinterpret_cast<const BaLmDev*>(g->grp_lm_host);
  counts.resize(n);
  for (int w = 0; w < n; ++w) counts[w] = hlm[w].n_out[set_level ? 0 : 1];
  return CMS_OK;
}

static int ba_optimize_group(cms_ba** bas, int n, int its_robust, int its_final, const volatile uint8_t* stop, cms_ba_stats* stats);
// Windows are independent problems; the grouped driver wants a group of one kind (one device; all with the edge-major work list, or none:
// a window that has it carries no other list) of at most BA_MAX_GROUP windows.  A call that mixes kinds is run as several groups, one
// after the other -- same results, the caller's order of `stats` kept.
extern "C" int cms_ba_optimize_many(cms_ba** bas, int n, int its_robust, int its_final, const volatile uint8_t* stop, cms_ba_stats* stats) {
  if (!bas || n < 1) return cms_fail(CMS_ERR_ARG, "cms_ba_optimize_many: bad argument");
  for (int w = 0; w < n; ++w) if (!bas[w]) return cms_fail(CMS_ERR_ARG, "null ba");
  // (device-planned windows among themselves: their groups always run the fused kernels, which is all they carry lists for)
  // (a deterministic window's sums must not depend on its company: windows that would change the group's kernels -- fewer wavefronts per workgroup,
  // no signature runs -- form groups of their own)
  auto kind = [&](int w) {
    const cms_ba* b = bas[w];
    int k = b->device * 8 + (b->fast_plan ? 4 : 0) + (b->se.nchunks > 0 ? 2 : 0) + (b->deterministic ? 1 : 0);
    if (b->deterministic) k += 1024 * (1 + (b->det_points ? 1 : 0) + 2 * (b->rm_lds > 0 ? 1 : 0) + 4 * std::max(0, std::min(15, b->se_waves)) + 64 * b->det_ranges);
    return k;
  };
  bool one = n <= BA_MAX_GROUP;
  for (int w = 1; w < n && one; ++w) one = kind(w) == kind(0);
  if (one) return ba_optimize_group(bas, n, its_robust, its_final, stop, stats);
  std::vector<char> done(n, 0);
  int rc_all = CMS_OK;
  for (int w0 = 0; w0 < n; ++w0) {
    if (done[w0]) continue;
    std::vector<cms_ba*> grp; std::vector<int> idx;
    for (int w = w0; w < n && (int)grp.size() < BA_MAX_GROUP; ++w)
      if (!done[w] && kind(w) == kind(w0)) { grp.push_back(bas[w]); idx.push_back(w); done[w] = 1; }
    std::vector<cms_ba_stats> gs(grp.size());
    const int rc = ba_optimize_group(grp.data(), (int)grp.size(), its_robust, its_final, stop, gs.data());
    if (stats) for (size_t i = 0; i < idx.size(); ++i) stats[idx[i]] = gs[i];
    if (rc < 0) return rc;                 // an error ends the call; "stopped" (1) lets the remaining groups see the flag themselves
    if (rc != CMS_OK) rc_all = rc;
  }
  return rc_all;
}
static int ba_optimize_group(cms_ba** bas, int n, int its_robust, int its_final, const volatile uint8_t* stop, cms_ba_stats* stats) {
  const bool batched = ba_can_batch(bas, n);
  static const bool call_timing = getenv("CMS_BA_CALL_TIMING") != nullptr;      // developer knob: where a call's host time goes (stderr, ms since entry)
  const auto t_in = std::chrono::steady_clock::now();
  std::string t_log;
  auto tick = [&](const char* what) {
    if (!call_timing) return;
    char buf[64];
    snprintf(buf, sizeof(buf), " %s %.3f", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count());
    t_log += buf;
  };
  // a window handed to its stream by cms_ba_set_stream: the stream waits for the window's set-up (the host does not) -- whichever driver runs the
  // group.  (Until round 6 only the batched branch did this: windows that cannot batch -- more free key frames than the blocked solve takes --
  // started their kernels on the new stream with nothing ordering them behind the upload and the set-up kernel on the old one.)
  for (int w = 0; w < n; ++w) { HIPCHK(hipSetDevice(bas[w]->device)); HIPCHK(ba_order_behind_setup(bas[w])); }
  if (!batched)
    for (int w = 0; w < n; ++w)      // the one-window drivers enqueue on the window's own stream and synchronise it: nothing else may still be pending there
      if (bas[w]->async_pending) { HIPCHK(ba_wait_stream(bas[w]->stream)); bas[w]->async_pending = false; }
  if (batched) {
    HIPCHK(hipSetDevice(bas[0]->device));
    for (int w = 0; w < n; ++w) {     // pending uploads / resets on the windows' streams
      // a window on a stream of its own: the group's stream (bas[0]'s) must not start before that window's pending work is through -- the host waits.
      // A window that already sits on the group's stream is ordered by the stream itself.
      if (bas[w]->async_pending && bas[w]->stream != bas[0]->stream) { HIPCHK(ba_wait_stream(bas[w]->stream)); bas[w]->async_pending = false; }
    }
    if (bas[0]->async_pending && bas[0]->own_stream) { HIPCHK(ba_wait_stream(bas[0]->stream)); bas[0]->async_pending = false; }
    tick("uploads-waited");
    int rcg = ba_group_reserve(bas[0], n);
    if (rcg) return rcg;
    rcg = ba_upload_items(bas, n);          // static descriptions of the windows: once per call
    if (rcg) return rcg;
    // from here on kernels of the group may be in flight on the shared stream: a window destroyed after an early error return must wait for
    // them (cms_ba_destroy synchronises a shared stream only for windows with async_pending); cleared at the successful end of the call
    for (int w = 0; w < n; ++w) { bas[w]->async_pending = true; bas[w]->grp_stream = bas[0]->stream; }
    tick("items");
  }
  std::vector<cms_ba_stats> local(n);
  for (int w = 0; w < n; ++w) memset(&local[w], 0, sizeof(cms_ba_stats));
  if (ba_stopped(stop)) {   // Optimizer.cpp:359-361 (no round ran: the global copies cleared by ba_upload_items stay zero once the stream is through)
    if (batched) for (int w = 0; w < n; ++w) bas[w]->gsum_clean = bas[w]->grp_se.gsum != 0;
    if (stats) memcpy(stats, local.data(), n * sizeof(cms_ba_stats));
    return 1;
  }
  const double delta = std::sqrt(5.991);
  std::vector<BaLm> st(n);
  for (int w = 0; w < n; ++w) { st[w] = BaLm(); st[w].iterations = its_robust; st[w].robust = 1; st[w].delta = delta; }
  const bool host_lm = ba_knobs().host_lm;
  for (int w = 0; w < n; ++w)      // (cannot happen while the knobs are read once: the lists were chosen by the same switches)
    if (bas[w]->se_only && (!batched || host_lm || !ba_use_se(bas, n)))
      return cms_fail(CMS_ERR_UNSUPPORTED, "cms_ba_optimize: the window carries only the edge-major work list but another kernel path was selected");
  int rc = batched ? (host_lm ? ba_optimize_stage_batched(bas, n, st, stop) : ba_optimize_stage_batched_dev(bas, n, st, stop)) : ba_optimize_stage_many(bas, n, st, stop);
  if (rc) return rc;
  tick("stage1");
  for (int w = 0; w < n; ++w) {
    local[w].iterations_done[0] = st[w].done; local[w].chi2_initial[0] = st[w].chi_ini; local[w].chi2_final[0] = st[w].chi_fin;
    local[w].lambda_final[0] = st[w].lam_fin;
  }
  std::vector<std::vector<uint8_t>> flags(n);
  std::vector<int> counts(n, 0);
  auto classify = [&](int set_level) -> int {
    if (batched) return ba_classify_batched(bas, n, set_level, counts);
    for (int w = 0; w < n; ++w) {
      cms_ba* b = bas[w];
      HIPCHK(hipSetDevice(b->device));
      flags[w].resize(b->E);
      hipLaunchKernelGGL(k_ba_classify, dim3(b->nblk_e), dim3(256), 0, b->stream, b->d, (const double*)b->d_poses[b->cur],
                         (const double*)b->d_pts[b->cur], 5.991, set_level, b->d_flags);
      HIPCHK(hipMemcpyAsync(flags[w].data(), b->d_flags, b->E, hipMemcpyDeviceToHost, b->stream));
    }
    for (int w = 0; w < n; ++w) HIPCHK(hipStreamSynchronize(bas[w]->stream));
    for (int w = 0; w < n; ++w) { counts[w] = 0; for (int e = 0; e < bas[w]->E; ++e) counts[w] += flags[w][e]; }
    return CMS_OK;
  };
  if (!ba_stopped(stop)) {   // Optimizer.cpp:366-397: exclude outliers, drop the kernel, optimize(10)
    rc = classify(1);
    if (rc) return rc;
    for (int w = 0; w < n; ++w) local[w].n_outliers_mid = counts[w];
    tick("classified");
    for (int w = 0; w < n; ++w) { st[w] = BaLm(); st[w].iterations = its_final; st[w].robust = 0; st[w].delta = delta; }
    rc = batched ? (host_lm ? ba_optimize_stage_batched(bas, n, st, stop) : ba_optimize_stage_batched_dev(bas, n, st, stop)) : ba_optimize_stage_many(bas, n, st, stop);
    if (rc) return rc;
    for (int w = 0; w < n; ++w) {
      local[w].iterations_done[1] = st[w].done; local[w].chi2_initial[1] = st[w].chi_ini; local[w].chi2_final[1] = st[w].chi_fin;
      local[w].lambda_final[1] = st[w].lam_fin;
    }
    tick("stage2");
  }
  rc = classify(0);          // Optimizer.cpp:399-412
  if (rc) return rc;
  tick("classified");
  if (call_timing) fprintf(stderr, "[cms_ba_optimize_many] %d windows:%s\n", n, t_log.c_str());
  for (int w = 0; w < n; ++w) local[w].n_outliers_final = counts[w];
  if (stats) memcpy(stats, local.data(), n * sizeof(cms_ba_stats));
  if (batched)      // classify synchronised the group's stream: nothing of the windows is in flight, and every round's solve kernel left the global copy zero
    for (int w = 0; w < n; ++w) { bas[w]->async_pending = false; bas[w]->grp_stream = nullptr; bas[w]->gsum_clean = bas[w]->grp_se.gsum != 0; }
  return CMS_OK;
}

extern "C" int cms_ba_optimize(cms_ba* b, int its_robust, int its_final, const volatile uint8_t* stop, cms_ba_stats* st) {
  if (!b) return cms_fail(CMS_ERR_ARG, "null ba");
  return cms_ba_optimize_many(&b, 1, its_robust, its_final, stop, st);
}
