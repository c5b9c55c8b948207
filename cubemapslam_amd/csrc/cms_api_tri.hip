// cms_api_tri.hip -- host side of LocalMapping::CreateNewMapPoints and of the search half of ORBMatcher::Fuse, included by
// cms_lib.hip after cms_api_track.hip.
#include <cmath>
#include <cstring>
#include <vector>

namespace {
// cv::gemm semantics used by ComputeE12 (LocalMapping.cpp:469-482) and the epipole (ORBMatcher.cpp:976-982); cv::gemm semantics as listed in DESIGN.md section 2
inline float tri_h_small(const float* a, const float* b, int bs) {
  float t = a[0] * b[0];
  t = t + a[1] * b[bs];
  t = t + a[2] * b[2 * bs];
  return t;
}
void tri_h_e12(const float* R1w, const float* t1w, const float* R2w, const float* t2w, float* E12) {
  float R12[9], M[9], t12[3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0;                                                  // R1w * R2w.t(): transposed operand -> generic path, double accumulation
      for (int k = 0; k < 3; ++k) s += (double)R1w[3 * r + k] * (double)R2w[3 * c + k];
      R12[3 * r + c] = (float)(s * 1.0);
      M[3 * r + c] = (float)(s * -1.0);
    }
  for (int r = 0; r < 3; ++r) t12[r] = (float)((double)tri_h_small(M + 3 * r, t2w, 1) * 1.0 + (double)t1w[r] * 1.0);
  const float tx[9] = {0, -t12[2], t12[1], t12[2], 0, -t12[0], -t12[1], t12[0], 0};   // SkewSymmetricMatrix (LocalMapping.cpp:621-626)
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) E12[3 * r + c] = (float)((double)tri_h_small(tx + 3 * r, R12 + c, 3) * 1.0);
}
}  // namespace

extern "C" int cms_create_new_map_points(cms_ctx* c, int njobs, const cms_keyframe* cur, const int* neigh_off, const cms_keyframe* neigh,
                                         int check_orientation, int cap_per_job, int* n_new, int* out_neigh, int* out_idx1, int* out_idx2,
                                         float* out_x3d) {
  if (!c || njobs < 0 || cap_per_job < 0 || (njobs > 0 && (!cur || !neigh_off || !n_new)) ||
      (njobs > 0 && cap_per_job > 0 && (!out_neigh || !out_idx1 || !out_idx2 || !out_x3d)))
    return cms_fail(CMS_ERR_ARG, "cms_create_new_map_points: bad argument");
  if (njobs == 0) return CMS_OK;
  if (c->g.nlevels > 16 || c->g.nlevels < 2) return cms_fail(CMS_ERR_UNSUPPORTED, "cms_create_new_map_points: 2..16 pyramid levels");
  const int nneigh = neigh_off[njobs];
  if (nneigh < 0 || (nneigh > 0 && !neigh)) return cms_fail(CMS_ERR_ARG, "cms_create_new_map_points: bad neighbour list");
  // ---- flatten: key frames = the njobs current ones, then all neighbours
  const int nkf = njobs + nneigh;
  std::vector<CmsTriKF> kfs((size_t)nkf);
  std::vector<CmsTriPair> pairs((size_t)nneigh);
  std::vector<CmsTriJob> jobs((size_t)njobs);
  size_t nf = 0, nn = 0, nno = 0, nnf = 0;
  auto kf_at = [&](int i) -> const cms_keyframe& { return i < njobs ? cur[i] : neigh[i - njobs]; };
  for (int i = 0; i < nkf; ++i) {
    const cms_keyframe& k = kf_at(i);
    if (k.n < 0 || k.n > CMS_TRI_MAXF || k.nnodes < 0 || (k.n > 0 && (!k.kps || !k.desc || !k.rays || !k.mp)) ||
        (k.nnodes > 0 && (!k.node_id || !k.node_off || !k.node_feat)))
      return cms_fail(CMS_ERR_ARG, "cms_create_new_map_points: bad key frame (at most 4096 features)");
    CmsTriKF& d = kfs[(size_t)i];
    d.f0 = (int)nf; d.n = k.n; d.node0 = (int)nn; d.nnodes = k.nnodes; d.noff0 = (int)nno; d.nfeat0 = (int)nnf;
    std::memcpy(d.Rcw, k.Rcw, sizeof(d.Rcw)); std::memcpy(d.tcw, k.tcw, sizeof(d.tcw)); std::memcpy(d.Ow, k.Ow, sizeof(d.Ow));
    nf += (size_t)k.n; nn += (size_t)k.nnodes; nno += (size_t)k.nnodes + 1; nnf += k.nnodes > 0 ? (size_t)k.node_off[k.nnodes] : 0;
  }
  std::vector<CmsKeyPoint> kp(nf + 1);
  std::vector<uint8_t> desc(32 * nf + 32);
  std::vector<float> rays(3 * nf + 3);
  std::vector<int> mp(nf + 1), feat_node(nf + 1, -1), node_id(nn + 1), node_off(nno + 1), node_feat(nnf + 1);
  for (int i = 0; i < nkf; ++i) {
    const cms_keyframe& k = kf_at(i);
    const CmsTriKF& d = kfs[(size_t)i];
    if (k.n > 0) {
      std::memcpy(&kp[(size_t)d.f0], k.kps, (size_t)k.n * sizeof(CmsKeyPoint));
      std::memcpy(&desc[32 * (size_t)d.f0], k.desc, 32 * (size_t)k.n);
      std::memcpy(&rays[3 * (size_t)d.f0], k.rays, 12 * (size_t)k.n);
      std::memcpy(&mp[(size_t)d.f0], k.mp, 4 * (size_t)k.n);
    }
    if (k.nnodes > 0) {
      std::memcpy(&node_id[(size_t)d.node0], k.node_id, 4 * (size_t)k.nnodes);
      std::memcpy(&node_off[(size_t)d.noff0], k.node_off, 4 * ((size_t)k.nnodes + 1));
      const int tot = k.node_off[k.nnodes];
      std::memcpy(&node_feat[(size_t)d.nfeat0], k.node_feat, 4 * (size_t)tot);
      for (int e = 0; e < k.nnodes; ++e) {
        if (e > 0 && k.node_id[e] <= k.node_id[e - 1]) return cms_fail(CMS_ERR_ARG, "cms_create_new_map_points: FeatureVector node ids must ascend");
        for (int q = k.node_off[e]; q < k.node_off[e + 1]; ++q) {
          const int f = k.node_feat[q];
          if (f < 0 || f >= k.n) return cms_fail(CMS_ERR_ARG, "cms_create_new_map_points: FeatureVector index out of range");
          feat_node[(size_t)d.f0 + f] = e;
        }
      }
    } else {
      node_off[(size_t)d.noff0] = 0;
    }
  }
  // ---- per pair: baseline test, essential matrix, epipole (host, the reference's float arithmetic)
  const int F = c->g.F;
  for (int j = 0; j < njobs; ++j) {
    jobs[(size_t)j].kf1 = j; jobs[(size_t)j].pair0 = neigh_off[j]; jobs[(size_t)j].npairs = neigh_off[j + 1] - neigh_off[j];
    if (jobs[(size_t)j].npairs < 0) return cms_fail(CMS_ERR_ARG, "cms_create_new_map_points: neigh_off must ascend");
    const cms_keyframe& k1 = cur[j];
    for (int p = neigh_off[j]; p < neigh_off[j + 1]; ++p) {
      const cms_keyframe& k2 = neigh[p];
      CmsTriPair& pr = pairs[(size_t)p];
      pr.kf2 = njobs + p;
      double s = 0;
      for (int k = 0; k < 3; ++k) { const float v = k2.Ow[k] - k1.Ow[k]; s += (double)v * (double)v; }
      const float baseline = (float)std::sqrt(s);
      const float ratioBaselineDepth = baseline / k2.median_depth;
      pr.skip = ratioBaselineDepth < 0.01;                             // LocalMapping.cpp:243-247
      tri_h_e12(k1.Rcw, k1.tcw, k2.Rcw, k2.tcw, pr.E12);
      float C2[3];
      for (int r = 0; r < 3; ++r) C2[r] = (float)((double)tri_h_small(k2.Rcw + 3 * r, k1.Ow, 1) * 1.0 + (double)k2.tcw[r] * 1.0);
      track_rays_to_cubemap(F, C2[0], C2[1], C2[2], pr.ex, pr.ey);
    }
  }
  // ---- device arena
  HIPCHK(hipSetDevice(c->device));
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes + 16); return at; };
  const size_t o_kf = take(kfs.size() * sizeof(CmsTriKF)), o_pair = take(pairs.size() * sizeof(CmsTriPair)), o_job = take(jobs.size() * sizeof(CmsTriJob)),
               o_kp = take(kp.size() * sizeof(CmsKeyPoint)), o_desc = take(desc.size()), o_rays = take(rays.size() * 4), o_mp = take(mp.size() * 4),
               o_fn = take(feat_node.size() * 4), o_nid = take(node_id.size() * 4), o_noff = take(node_off.size() * 4), o_nfeat = take(node_feat.size() * 4),
               o_nnew = take((size_t)njobs * 4), o_on = take((size_t)njobs * cap_per_job * 4), o_o1 = take((size_t)njobs * cap_per_job * 4),
               o_o2 = take((size_t)njobs * cap_per_job * 4), o_ox = take((size_t)njobs * cap_per_job * 12);
  int rc = cms_scratch(c, o);
  if (rc) return rc;
  uint8_t* p = (uint8_t*)c->d_match;
  hipStream_t s = c->stream;
  auto up = [&](size_t at, const void* src, size_t bytes) { return bytes ? hipMemcpyAsync(p + at, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess; };
  HIPCHK(up(o_kf, kfs.data(), kfs.size() * sizeof(CmsTriKF)));
  HIPCHK(up(o_pair, pairs.data(), pairs.size() * sizeof(CmsTriPair)));
  HIPCHK(up(o_job, jobs.data(), jobs.size() * sizeof(CmsTriJob)));
  HIPCHK(up(o_kp, kp.data(), kp.size() * sizeof(CmsKeyPoint)));
  HIPCHK(up(o_desc, desc.data(), desc.size()));
  HIPCHK(up(o_rays, rays.data(), rays.size() * 4));
  HIPCHK(up(o_mp, mp.data(), mp.size() * 4));
  HIPCHK(up(o_fn, feat_node.data(), feat_node.size() * 4));
  HIPCHK(up(o_nid, node_id.data(), node_id.size() * 4));
  HIPCHK(up(o_noff, node_off.data(), node_off.size() * 4));
  HIPCHK(up(o_nfeat, node_feat.data(), node_feat.size() * 4));
  CmsTriArgs a;
  a.kf = (const CmsTriKF*)(p + o_kf); a.pair = (const CmsTriPair*)(p + o_pair); a.job = (const CmsTriJob*)(p + o_job);
  a.kp = (const CmsKeyPoint*)(p + o_kp); a.desc = (const uint4*)(p + o_desc); a.rays = (const float*)(p + o_rays); a.mp = (const int*)(p + o_mp);
  a.feat_node = (const int*)(p + o_fn); a.node_id = (const int*)(p + o_nid); a.node_off = (const int*)(p + o_noff); a.node_feat = (const int*)(p + o_nfeat);
  a.F = F;
  {                                                                    // CamModelGeneral::SetCosFovTh (CamModelGeneral.h:224-229), float
    const float fov = (float)c->cam.fov_deg;
    const float pif = 3.1415926535897932384626f;
    a.cos_fov = std::cos(fov / 2 * (pif / 180));
  }
  a.ratio_factor = 1.5f * c->scale[1];                                // 1.5f * mpCurrentKeyFrame->mfScaleFactor
  a.check_orientation = check_orientation;
  for (int l = 0; l < 16; ++l) { a.sf[l] = l < c->g.nlevels ? c->scale[l] : 1.0f; a.sigma2[l] = l < c->g.nlevels ? c->sigma2[l] : 1.0f; }
  a.cap = cap_per_job;
  a.n_new = (int*)(p + o_nnew); a.out_neigh = (int*)(p + o_on); a.out_idx1 = (int*)(p + o_o1); a.out_idx2 = (int*)(p + o_o2); a.out_x3d = (float*)(p + o_ox);
  hipLaunchKernelGGL(k_create_new_map_points, dim3(njobs), dim3(512), 0, s, a);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(n_new, p + o_nnew, (size_t)njobs * 4, hipMemcpyDeviceToHost, s));
  if (cap_per_job > 0) {
    const size_t nb = (size_t)njobs * cap_per_job;
    HIPCHK(hipMemcpyAsync(out_neigh, p + o_on, nb * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_idx1, p + o_o1, nb * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_idx2, p + o_o2, nb * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_x3d, p + o_ox, nb * 12, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  for (int j = 0; j < njobs; ++j)
    if (n_new[j] > cap_per_job) return cms_fail(CMS_ERR_OVERFLOW, "cms_create_new_map_points: more new points than cap_per_job (n_new holds the counts)");
  return CMS_OK;
}

// search half of ORBMatcher::Fuse(pKF, vpMapPoints, th) for key frame slot b (cms_area_set_keypoints / _descriptors + cms_area_grid first)
extern "C" int cms_fuse_search(cms_ctx* c, int b, const float* pose15, int nmp, const uint8_t* skip, const float* pos, const float* normal,
                               const float* min_dist, const float* max_dist, const uint8_t* mp_desc, float th, int* best_idx, int* best_dist) {
  if (!c || !pose15 || nmp < 0 || (nmp > 0 && (!pos || !normal || !min_dist || !max_dist || !mp_desc || !best_idx || !best_dist)))
    return cms_fail(CMS_ERR_ARG, "cms_fuse_search: bad argument");
  if (b < 0 || b >= c->area_frames) return cms_fail(CMS_ERR_ARG, "cms_fuse_search: no grid for this key frame (cms_area_grid first)");
  if (nmp == 0) return CMS_OK;
  if (c->g.nlevels > 16) return cms_fail(CMS_ERR_UNSUPPORTED, "cms_fuse_search: more than 16 pyramid levels");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t n4 = (size_t)nmp * 4;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
  const size_t o_pose = take(64), o_skip = take(nmp), o_pos = take(3 * n4), o_nrm = take(3 * n4), o_min = take(n4), o_max = take(n4),
               o_desc = take((size_t)nmp * 32), o_qx = take(n4), o_qy = take(n4), o_qr = take(n4), o_qmin = take(n4), o_qmax = take(n4), o_lvl = take(n4),
               o_cnt = take(n4), o_off = take(n4 + 4), o_tot = take(16), o_qf = take(n4), o_bi = take(n4), o_bd = take(n4);
  const size_t fixed = o;
  int cap = 64 * nmp + 1024;
  for (int attempt = 0; attempt < 2; ++attempt) {
    int rc = cms_scratch(c, fixed + al((size_t)cap * 4));
    if (rc) return rc;
    uint8_t* p = (uint8_t*)c->d_match;
    const size_t o_idx = fixed;
    HIPCHK(hipMemcpyAsync(p + o_pose, pose15, 60, hipMemcpyHostToDevice, s));
    if (skip) HIPCHK(hipMemcpyAsync(p + o_skip, skip, nmp, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(p + o_pos, pos, 3 * n4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(p + o_nrm, normal, 3 * n4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(p + o_min, min_dist, n4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(p + o_max, max_dist, n4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(p + o_desc, mp_desc, (size_t)nmp * 32, hipMemcpyHostToDevice, s));
    const std::vector<int> qf((size_t)nmp, b);
    HIPCHK(hipMemcpyAsync(p + o_qf, qf.data(), n4, hipMemcpyHostToDevice, s));
    CmsFuseArgs fa;
    fa.pose15 = (const float*)(p + o_pose); fa.mp_frame = nullptr; fa.n = nmp; fa.skip = skip ? p + o_skip : nullptr;
    fa.P = (const float*)(p + o_pos); fa.normal = (const float*)(p + o_nrm); fa.min_dist = (const float*)(p + o_min); fa.max_dist = (const float*)(p + o_max);
    fa.th = th; fa.log_scale = std::log(c->scale[1]); fa.nlevels = c->g.nlevels; fa.F = c->g.F;
    for (int l = 0; l < 16; ++l) fa.sf[l] = l < c->g.nlevels ? c->scale[l] : 0.0f;
    fa.qx = (float*)(p + o_qx); fa.qy = (float*)(p + o_qy); fa.qr = (float*)(p + o_qr); fa.qmin = (int*)(p + o_qmin); fa.qmax = (int*)(p + o_qmax);
    fa.level = (int*)(p + o_lvl);
    hipLaunchKernelGGL(k_fuse_project, dim3((nmp + 255) / 256), dim3(256), 0, s, fa);
    HIPCHK(hipGetLastError());
    rc = cms_features_in_area_batch_device(c, nmp, p + o_qf, p + o_qx, p + o_qy, p + o_qr, p + o_qmin, p + o_qmax, p + o_cnt, p + o_off, p + o_idx, cap,
                                           p + o_tot);
    if (rc) return rc;
    int tot = 0;
    HIPCHK(hipMemcpyAsync(&tot, p + o_tot, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (tot > cap) { cap = tot + 64; continue; }
    CmsFuseScanArgs sa;
    sa.n = nmp; sa.qx = (const float*)(p + o_qx); sa.qy = (const float*)(p + o_qy); sa.level = (const int*)(p + o_lvl); sa.mp_desc = (const uint4*)(p + o_desc);
    sa.cand_off = (const int*)(p + o_off); sa.cand_idx = (const int*)(p + o_idx); sa.kp = (const CmsKeyPoint*)c->d_kps; sa.t_desc = (const uint4*)c->d_desc;
    for (int l = 0; l < 16; ++l) sa.inv_sigma2[l] = l < c->g.nlevels ? c->inv_sigma2[l] : 0.0f;
    sa.best_idx = (int*)(p + o_bi); sa.best_dist = (int*)(p + o_bd);
    hipLaunchKernelGGL(k_fuse_scan, dim3((nmp * 8 + 255) / 256), dim3(256), 0, s, sa);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(best_idx, p + o_bi, n4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(best_dist, p + o_bd, n4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const int base = b * c->g.kp_cap;
    for (int i = 0; i < nmp; ++i) if (best_idx[i] >= 0) best_idx[i] -= base;       // batch row -> key point index of slot b
    return CMS_OK;
  }
  return cms_fail(CMS_ERR_OVERFLOW, "cms_fuse_search: candidate lists kept growing");
}
