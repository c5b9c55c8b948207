// cms_api_tri.hip -- host side of LocalMapping::CreateNewMapPoints and of the search half of ORBMatcher::Fuse, included by
// cms_lib.hip after cms_api_track.hip.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
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

namespace {
struct TriDev {                                           // device views of the key-frame data (flattened per call, or a resident store)
  const CmsTriKF* kf; const CmsKeyPoint* kp; const uint4* desc; const float* rays; const int* mp; const int* feat_node;
  const int* node_id; const int* node_off; const int* node_feat;
};
inline size_t tri_al(size_t v) { return (v + 255) & ~(size_t)255; }
size_t tri_work_bytes(int njobs, int nneigh, int max_n1, int cap) {
  const size_t nb = (size_t)njobs * (size_t)cap;
  return tri_al((size_t)nneigh * sizeof(CmsTriPair) + 16) + tri_al((size_t)njobs * sizeof(CmsTriJob) + 16) + tri_al((size_t)nneigh * 4 + 16) +
         tri_al((size_t)njobs * 4 + 16) + 3 * tri_al(nb * 4 + 16) + tri_al(nb * 12 + 16) + tri_al((size_t)nneigh * (size_t)max_n1 * sizeof(CmsTriCand) + 16);
}
int tri_check_keyframe(const cms_keyframe& k, const char* who) {
  if (k.n < 0 || k.n > CMS_TRI_MAXF || k.nnodes < 0 || (k.n > 0 && (!k.kps || !k.desc || !k.rays || !k.mp)) ||
      (k.nnodes > 0 && (!k.node_id || !k.node_off || !k.node_feat)))
    return cms_fail(CMS_ERR_ARG, who);
  for (int e = 0; e < k.nnodes; ++e) {
    if (e > 0 && k.node_id[e] <= k.node_id[e - 1]) return cms_fail(CMS_ERR_ARG, "FeatureVector node ids must ascend");
    if (k.node_off[e + 1] < k.node_off[e]) return cms_fail(CMS_ERR_ARG, "FeatureVector offsets must ascend");
    for (int q = k.node_off[e]; q < k.node_off[e + 1]; ++q)
      if (k.node_feat[q] < 0 || k.node_feat[q] >= k.n) return cms_fail(CMS_ERR_ARG, "FeatureVector index out of range");
  }
  return CMS_OK;
}
void tri_feat_node(const cms_keyframe& k, int* feat_node /* k.n entries */) {
  for (int i = 0; i < k.n; ++i) feat_node[i] = -1;
  for (int e = 0; e < k.nnodes; ++e)
    for (int q = k.node_off[e]; q < k.node_off[e + 1]; ++q) feat_node[k.node_feat[q]] = e;
}

// pairs (baseline test, essential matrix, epipole: host, the reference's float arithmetic), launch, fetch.  hkf / hmedian: host copies of
// the key frames' CmsTriKF records and median depths, indexed like the device array; cur_idx / neigh_idx index them.
int tri_run(cms_ctx* c, const TriDev& dev, const CmsTriKF* hkf, const float* hmedian, int njobs, const int* cur_idx, const int* neigh_off,
            const int* neigh_idx, int check_orientation, int cap, uint8_t* work, int* n_new, int* out_neigh, int* out_idx1, int* out_idx2, float* out_x3d) {
  const int nneigh = neigh_off[njobs];
  static const bool timing = getenv("CMS_TRI_TIMING") != nullptr;      // developer knob: where a call's host time goes (stderr)
  const auto t_0 = std::chrono::steady_clock::now();
  std::vector<CmsTriPair> pairs((size_t)nneigh + 1);
  std::vector<CmsTriJob> jobs((size_t)njobs);
  std::vector<int> pair_job((size_t)nneigh + 1, 0);
  const int F = c->g.F;
  int max_n1 = 1, max_pairs = 0;
  for (int j = 0; j < njobs; ++j) {
    jobs[(size_t)j].kf1 = cur_idx[j]; jobs[(size_t)j].pair0 = neigh_off[j]; jobs[(size_t)j].npairs = neigh_off[j + 1] - neigh_off[j];
    if (jobs[(size_t)j].npairs < 0) return cms_fail(CMS_ERR_ARG, "cms_create_new_map_points: neigh_off must ascend");
    const CmsTriKF& k1 = hkf[cur_idx[j]];
    max_n1 = std::max(max_n1, k1.n); max_pairs = std::max(max_pairs, jobs[(size_t)j].npairs);
    for (int q = neigh_off[j]; q < neigh_off[j + 1]; ++q) {
      const CmsTriKF& k2 = hkf[neigh_idx[q]];
      CmsTriPair& pr = pairs[(size_t)q];
      pair_job[(size_t)q] = j;
      pr.kf2 = neigh_idx[q];
      double s = 0;
      for (int k = 0; k < 3; ++k) { const float v = k2.Ow[k] - k1.Ow[k]; s += (double)v * (double)v; }
      const float baseline = (float)std::sqrt(s);
      const float ratioBaselineDepth = baseline / hmedian[neigh_idx[q]];
      pr.skip = ratioBaselineDepth < 0.01;                             // LocalMapping.cpp:243-247
      tri_h_e12(k1.Rcw, k1.tcw, k2.Rcw, k2.tcw, pr.E12);
      float C2[3];
      for (int r = 0; r < 3; ++r) C2[r] = (float)((double)tri_h_small(k2.Rcw + 3 * r, k1.Ow, 1) * 1.0 + (double)k2.tcw[r] * 1.0);
      track_rays_to_cubemap(F, C2[0], C2[1], C2[2], pr.ex, pr.ey);
    }
  }
  const size_t nb = (size_t)njobs * (size_t)cap;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += tri_al(bytes + 16); return at; };
  const size_t o_pair = take((size_t)nneigh * sizeof(CmsTriPair)), o_job = take((size_t)njobs * sizeof(CmsTriJob)), o_pjob = take((size_t)nneigh * 4),
               o_nnew = take((size_t)njobs * 4), o_on = take(nb * 4), o_o1 = take(nb * 4), o_o2 = take(nb * 4), o_ox = take(nb * 12),
               o_cand = take((size_t)nneigh * (size_t)max_n1 * sizeof(CmsTriCand));
  (void)o_cand;
  hipStream_t s = c->stream;
  uint8_t* p = work;
  // inputs (pairs | jobs | pair -> job) and outputs (counts | neighbour | idx1 | idx2 | x3d) are contiguous in the work block and mirrored in the
  // context's pinned staging block.  Neither direction uses a copy engine: the inputs (~20 KB) are fetched from the pinned block by a one-workgroup
  // kernel, and the kernels store the new points straight into it (a few thousand posted 4-byte writes) -- two queue entries less per call.
  // Measured inside the bench's step the call takes the same 0.9-1.1 ms either way (CMS_TRI_COPY_ENGINE=1: the copies of round 3; 0.3 ms alone):
  // what made it 3 ms there was not the transfers but the Python wrapper's per-call arrays (see api.KeyframeStore.create_new_map_points).
  static const bool copy_engine = getenv("CMS_TRI_COPY_ENGINE") != nullptr;
  const size_t in_bytes = o_nnew, out_bytes = o_cand - o_nnew;
  int rcs = cms_hstage(c, in_bytes + out_bytes);
  if (rcs) return rcs;
  uint8_t* hin = c->h_stage;
  uint8_t* hout = c->h_stage + in_bytes;
  if (nneigh > 0) {
    memcpy(hin + o_pair, pairs.data(), (size_t)nneigh * sizeof(CmsTriPair));
    memcpy(hin + o_pjob, pair_job.data(), (size_t)nneigh * 4);
  }
  memcpy(hin + o_job, jobs.data(), (size_t)njobs * sizeof(CmsTriJob));
  if (copy_engine) HIPCHK(hipMemcpyAsync(p, hin, in_bytes, hipMemcpyHostToDevice, s));
  else hipLaunchKernelGGL(k_copy16, dim3((unsigned)((in_bytes / 16 + 63) / 64)), dim3(64), 0, s, (uint4*)p, (const uint4*)hin, (int)(in_bytes / 16));      // (offsets are multiples of 256; wavefront-sized workgroups find a slot at once)
  uint8_t* po = copy_engine ? p + o_nnew : hout;          // where the kernels put their results
  CmsTriArgs a;
  a.kf = dev.kf; a.pair = (const CmsTriPair*)(p + o_pair); a.job = (const CmsTriJob*)(p + o_job);
  a.kp = dev.kp; a.desc = dev.desc; a.rays = dev.rays; a.mp = dev.mp; a.feat_node = dev.feat_node;
  a.node_id = dev.node_id; a.node_off = dev.node_off; a.node_feat = dev.node_feat;
  a.F = F;
  {                                                                    // CamModelGeneral::SetCosFovTh (CamModelGeneral.h:224-229), float
    const float fov = (float)c->cam.fov_deg;
    const float pif = 3.1415926535897932384626f;
    a.cos_fov = std::cos(fov / 2 * (pif / 180));
  }
  a.ratio_factor = 1.5f * c->scale[1];                                // 1.5f * mpCurrentKeyFrame->mfScaleFactor
  a.check_orientation = check_orientation;
  for (int l = 0; l < 16; ++l) { a.sf[l] = l < c->g.nlevels ? c->scale[l] : 1.0f; a.sigma2[l] = l < c->g.nlevels ? c->sigma2[l] : 1.0f; }
  a.cap = cap;
  a.n_new = (int*)po; a.out_neigh = (int*)(po + (o_on - o_nnew)); a.out_idx1 = (int*)(po + (o_o1 - o_nnew)); a.out_idx2 = (int*)(po + (o_o2 - o_nnew));
  a.out_x3d = (float*)(po + (o_ox - o_nnew));
  // (CMS_TRI_SEQUENTIAL is read per call on purpose: test_create_new_map_points toggles it inside one process to hold the two kernels to identical records)
  if (check_orientation || max_pairs > 64 || nneigh == 0 || getenv("CMS_TRI_SEQUENTIAL")) {
    hipLaunchKernelGGL(k_create_new_map_points, dim3(njobs), dim3(512), 0, s, a);      // neighbour after neighbour (the rotation histogram of a
  } else {                                                                            // neighbour depends on which features are still free)
    hipLaunchKernelGGL(k_tri_candidates, dim3((max_n1 + 255) / 256, nneigh), dim3(256), 0, s, a, (const int*)(p + o_pjob), (CmsTriCand*)(p + o_cand), max_n1);
    hipLaunchKernelGGL(k_tri_resolve, dim3(njobs), dim3(1024), 0, s, a, (const CmsTriCand*)(p + o_cand), max_n1);
  }
  HIPCHK(hipGetLastError());
  if (copy_engine) HIPCHK(hipMemcpyAsync(hout, p + o_nnew, cap > 0 ? out_bytes : tri_al((size_t)njobs * 4 + 16), hipMemcpyDeviceToHost, s));
  const auto t_1 = std::chrono::steady_clock::now();
  HIPCHK(hipStreamSynchronize(s));
  if (timing) {
    const auto t_2 = std::chrono::steady_clock::now();
    fprintf(stderr, "[tri_run] jobs %d pairs %d: host + enqueue %.3f ms, wait %.3f ms\n", njobs, nneigh, std::chrono::duration<double, std::milli>(t_1 - t_0).count(),
            std::chrono::duration<double, std::milli>(t_2 - t_1).count());
  }
  memcpy(n_new, hout, (size_t)njobs * 4);
  for (int j = 0; j < njobs && cap > 0; ++j) {                        // (entries behind a job's count were never written)
    const size_t at = (size_t)j * (size_t)cap, cnt = (size_t)std::max(0, std::min(n_new[j], cap));
    memcpy(out_neigh + at, hout + (o_on - o_nnew) + 4 * at, cnt * 4);
    memcpy(out_idx1 + at, hout + (o_o1 - o_nnew) + 4 * at, cnt * 4);
    memcpy(out_idx2 + at, hout + (o_o2 - o_nnew) + 4 * at, cnt * 4);
    memcpy(out_x3d + 3 * at, hout + (o_ox - o_nnew) + 12 * at, cnt * 12);
  }
  for (int j = 0; j < njobs; ++j)
    if (n_new[j] > cap) return cms_fail(CMS_ERR_OVERFLOW, "cms_create_new_map_points: more new points than cap_per_job (n_new holds the counts)");
  return CMS_OK;
}
}  // namespace

// ---- key frames of a call, flattened (features, FeatureVector, poses concatenated) and uploaded into the context's scratch arena; the work
// area of the call follows at o_work.  kf_at(i): the i-th key frame of the call; the first n_current of them are "current" key frames
// (max_n1 = their largest feature count); work_bytes(max_n1): what the caller needs behind the key frames.
template <class KfAt, class WorkBytes>
static int tri_flatten_upload(cms_ctx* c, int nkf, KfAt&& kf_at, int n_current, const char* who, std::vector<CmsTriKF>& kfs, std::vector<float>& median,
                              TriDev& dev, size_t& o_work, int& max_n1, WorkBytes&& work_bytes) {
  kfs.assign((size_t)nkf, CmsTriKF());
  median.assign((size_t)nkf, 0.f);
  size_t nf = 0, nn = 0, nno = 0, nnf = 0;
  max_n1 = 1;
  for (int i = 0; i < nkf; ++i) {
    const cms_keyframe& k = kf_at(i);
    const int rc = tri_check_keyframe(k, who);
    if (rc) return rc;
    CmsTriKF& d = kfs[(size_t)i];
    d.f0 = (int)nf; d.n = k.n; d.node0 = (int)nn; d.nnodes = k.nnodes; d.noff0 = (int)nno; d.nfeat0 = (int)nnf;
    std::memcpy(d.Rcw, k.Rcw, sizeof(d.Rcw)); std::memcpy(d.tcw, k.tcw, sizeof(d.tcw)); std::memcpy(d.Ow, k.Ow, sizeof(d.Ow));
    median[(size_t)i] = k.median_depth;
    nf += (size_t)k.n; nn += (size_t)k.nnodes; nno += (size_t)k.nnodes + 1; nnf += k.nnodes > 0 ? (size_t)k.node_off[k.nnodes] : 0;
    if (i < n_current) max_n1 = std::max(max_n1, k.n);
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
      tri_feat_node(k, &feat_node[(size_t)d.f0]);
    }
    if (k.nnodes > 0) {
      std::memcpy(&node_id[(size_t)d.node0], k.node_id, 4 * (size_t)k.nnodes);
      std::memcpy(&node_off[(size_t)d.noff0], k.node_off, 4 * ((size_t)k.nnodes + 1));
      std::memcpy(&node_feat[(size_t)d.nfeat0], k.node_feat, 4 * (size_t)k.node_off[k.nnodes]);
    } else {
      node_off[(size_t)d.noff0] = 0;
    }
  }
  HIPCHK(hipSetDevice(c->device));
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += tri_al(bytes + 16); return at; };
  const size_t o_kf = take(kfs.size() * sizeof(CmsTriKF)), o_kp = take(kp.size() * sizeof(CmsKeyPoint)), o_desc = take(desc.size()), o_rays = take(rays.size() * 4),
               o_mp = take(mp.size() * 4), o_fn = take(feat_node.size() * 4), o_nid = take(node_id.size() * 4), o_noff = take(node_off.size() * 4),
               o_nfeat = take(node_feat.size() * 4);
  o_work = o;
  int rc = cms_scratch(c, o_work + work_bytes(max_n1));
  if (rc) return rc;
  uint8_t* p = (uint8_t*)c->d_match;
  hipStream_t s = c->stream;
  auto up = [&](size_t at, const void* src, size_t bytes) { return bytes ? hipMemcpyAsync(p + at, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess; };
  HIPCHK(up(o_kf, kfs.data(), kfs.size() * sizeof(CmsTriKF)));
  HIPCHK(up(o_kp, kp.data(), kp.size() * sizeof(CmsKeyPoint)));
  HIPCHK(up(o_desc, desc.data(), desc.size()));
  HIPCHK(up(o_rays, rays.data(), rays.size() * 4));
  HIPCHK(up(o_mp, mp.data(), mp.size() * 4));
  HIPCHK(up(o_fn, feat_node.data(), feat_node.size() * 4));
  HIPCHK(up(o_nid, node_id.data(), node_id.size() * 4));
  HIPCHK(up(o_noff, node_off.data(), node_off.size() * 4));
  HIPCHK(up(o_nfeat, node_feat.data(), node_feat.size() * 4));
  HIPCHK(hipStreamSynchronize(s));      // the sources are this function's vectors (the resident store, cms_kfstore_*, is the path without this copy)
  dev.kf = (const CmsTriKF*)(p + o_kf); dev.kp = (const CmsKeyPoint*)(p + o_kp); dev.desc = (const uint4*)(p + o_desc); dev.rays = (const float*)(p + o_rays);
  dev.mp = (const int*)(p + o_mp); dev.feat_node = (const int*)(p + o_fn); dev.node_id = (const int*)(p + o_nid); dev.node_off = (const int*)(p + o_noff);
  dev.node_feat = (const int*)(p + o_nfeat);
  return CMS_OK;
}

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
  std::vector<CmsTriKF> kfs;
  std::vector<float> median;
  auto kf_at = [&](int i) -> const cms_keyframe& { return i < njobs ? cur[i] : neigh[i - njobs]; };
  TriDev dev;
  size_t o_work = 0;
  int max_n1 = 1;
  int rc = tri_flatten_upload(c, nkf, kf_at, njobs, "cms_create_new_map_points: bad key frame (at most 4096 features)", kfs, median, dev, o_work, max_n1,
                              [&](int mn1) { return tri_work_bytes(njobs, nneigh, mn1, cap_per_job); });
  if (rc) return rc;
  uint8_t* p = (uint8_t*)c->d_match;
  std::vector<int> cur_idx((size_t)njobs), neigh_idx((size_t)nneigh + 1);
  for (int j = 0; j < njobs; ++j) cur_idx[(size_t)j] = j;
  for (int q = 0; q < nneigh; ++q) neigh_idx[(size_t)q] = njobs + q;
  return tri_run(c, dev, kfs.data(), median.data(), njobs, cur_idx.data(), neigh_off, neigh_idx.data(), check_orientation, cap_per_job, p + o_work, n_new,
                 out_neigh, out_idx1, out_idx2, out_x3d);
}

// ORBMatcher::SearchForTriangulation(pKF1, pKF2, E12, vMatchedPairs) (include/ORBMatcher.h:61, src/ORBMatcher.cpp:971-1125) on its own: what
// LocalMapping::CreateNewMapPoints calls per neighbour (LocalMapping.cpp:254) before it triangulates.  E12 = NULL: ComputeE12 of the two poses.
extern "C" int cms_search_for_triangulation(cms_ctx* c, const cms_keyframe* kf1, const cms_keyframe* kf2, const float* E12, int check_orientation,
                                            int* matches12, int* n_matches) {
  if (!c || !kf1 || !kf2 || !matches12 || !n_matches) return cms_fail(CMS_ERR_ARG, "cms_search_for_triangulation: bad argument");
  if (c->g.nlevels > 16 || c->g.nlevels < 2) return cms_fail(CMS_ERR_UNSUPPORTED, "cms_search_for_triangulation: 2..16 pyramid levels");
  *n_matches = 0;
  if (kf1->n == 0) return CMS_OK;
  std::vector<CmsTriKF> kfs;
  std::vector<float> median;
  auto kf_at = [&](int i) -> const cms_keyframe& { return i == 0 ? *kf1 : *kf2; };
  TriDev dev;
  size_t o_work = 0;
  int max_n1 = 1;
  const size_t o_pair = 0, o_job = tri_al(sizeof(CmsTriPair) + 16), o_m = o_job + tri_al(sizeof(CmsTriJob) + 16);
  int rc = tri_flatten_upload(c, 2, kf_at, 1, "cms_search_for_triangulation: bad key frame (at most 4096 features)", kfs, median, dev, o_work, max_n1,
                              [&](int mn1) { return o_m + tri_al((size_t)mn1 * 4 + 16) + 256; });
  if (rc) return rc;
  uint8_t* p = (uint8_t*)c->d_match + o_work;
  const size_t o_n = o_m + tri_al((size_t)max_n1 * 4 + 16);
  CmsTriPair pr;
  pr.kf2 = 1; pr.skip = 0;
  if (E12) std::memcpy(pr.E12, E12, sizeof(pr.E12));
  else tri_h_e12(kfs[0].Rcw, kfs[0].tcw, kfs[1].Rcw, kfs[1].tcw, pr.E12);
  float C2[3];                                                       // the epipole in the second image (ORBMatcher.cpp:976-982)
  for (int r = 0; r < 3; ++r) C2[r] = (float)((double)tri_h_small(kfs[1].Rcw + 3 * r, kfs[0].Ow, 1) * 1.0 + (double)kfs[1].tcw[r] * 1.0);
  track_rays_to_cubemap(c->g.F, C2[0], C2[1], C2[2], pr.ex, pr.ey);
  CmsTriJob jb; jb.kf1 = 0; jb.pair0 = 0; jb.npairs = 1;
  hipStream_t s = c->stream;
  HIPCHK(hipMemcpyAsync(p + o_pair, &pr, sizeof(pr), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(p + o_job, &jb, sizeof(jb), hipMemcpyHostToDevice, s));
  CmsTriArgs a;
  std::memset(&a, 0, sizeof(a));
  a.kf = dev.kf; a.pair = (const CmsTriPair*)(p + o_pair); a.job = (const CmsTriJob*)(p + o_job);
  a.kp = dev.kp; a.desc = dev.desc; a.rays = dev.rays; a.mp = dev.mp; a.feat_node = dev.feat_node;
  a.node_id = dev.node_id; a.node_off = dev.node_off; a.node_feat = dev.node_feat;
  a.F = c->g.F;
  a.check_orientation = check_orientation;
  for (int l = 0; l < 16; ++l) { a.sf[l] = l < c->g.nlevels ? c->scale[l] : 1.0f; a.sigma2[l] = l < c->g.nlevels ? c->sigma2[l] : 1.0f; }
  hipLaunchKernelGGL(k_tri_search, dim3(1), dim3(1024), 0, s, a, (int*)(p + o_m), (int*)(p + o_n));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(matches12, p + o_m, (size_t)kf1->n * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(n_matches, p + o_n, 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return CMS_OK;
}

// ---- resident key frames: the map's key frames live on the device in fixed-size slots (features, FeatureVector, pose); a call names
// slots, so nothing but ~100 bytes per (current, neighbour) pair travels to the device per CreateNewMapPoints.
struct cms_kfstore {
  cms_ctx* c = nullptr;
  int maxkf = 0, maxf = 0, maxn = 0;
  CmsTriKF* d_kf = nullptr; CmsKeyPoint* d_kp = nullptr; uint8_t* d_desc = nullptr; float* d_rays = nullptr; int* d_mp = nullptr; int* d_fn = nullptr;
  int* d_nid = nullptr; int* d_noff = nullptr; int* d_nfeat = nullptr;
  uint8_t* d_work = nullptr; size_t work_bytes = 0;
  // frame grid of every slot (KeyFrame::AssignFeaturesToGrid), built by cms_kfstore_put: the Fuse search runs on resident key frames too
  uint16_t* d_sorted = nullptr; int* d_cell_start = nullptr; int* d_nvalid = nullptr; int* d_kp_cnt = nullptr;
  std::vector<CmsTriKF> h_kf; std::vector<float> h_median; std::vector<uint8_t> used;
  // cms_kfstore_put_from_frame / cms_kfstore_update_poses: what still comes from the host (FeatureVector, map-point slots, poses) travels through a
  // pinned block per slot that the copying kernel reads itself -- no copy-engine transfer, no synchronisation; an event per slot guards its reuse
  // (one event per CALL, shared by the slots the call filled: a slot's block is rewritten only after the call that last read it is through)
  struct PutCall { hipEvent_t ev = nullptr; ~PutCall() { if (ev) (void)hipEventDestroy(ev); } };
  int* h_ff = nullptr; size_t ff_stride = 0; std::vector<std::shared_ptr<PutCall>> ff_call;
  void* h_items = nullptr; int items_gen = 0; std::shared_ptr<PutCall> items_call[2];      // the batch's descriptors: two pinned arrays, used alternately
  unsigned upd_call = 0; hipEvent_t upd_ev[4] = {nullptr, nullptr, nullptr, nullptr}; bool upd_ev_set[4] = {false, false, false, false};
  // slots that work enqueued on the store's stream reads or writes (pose updates, CreateNewMapPoints, the Fuse searches) since the slot was last
  // filled: cms_kfstore_put_from_frames copies on the FRAME context's stream, so before it overwrites such a slot that stream waits for the store's
  std::vector<uint8_t> busy; hipEvent_t order_ev = nullptr;
  // copies of cms_kfstore_put_from_frames that the store's stream has not been told to wait for yet: the put runs on the FRAME thread, and a call that
  // touches the store's stream from there (hipStreamWaitEvent) queues inside the runtime behind the mapping thread's synchronous calls on that stream
  // (measured: 2.8 ms per put call of 16 key frames, the length of a CreateNewMapPoints / Fuse call).  The wait is inserted by the store's next own
  // operation instead (kfstore_order_behind_puts)
  std::mutex put_mu; std::vector<std::shared_ptr<PutCall>> pending_puts;
  float* h_upd = nullptr; std::vector<uint8_t> upd_par;      // two 16-float blocks per SLOT, used alternately: a block is rewritten only by the SECOND later update
                                                             // of the same slot, long after the kernel of the first has read it (no event, no wait)
};

static hipError_t kfstore_order_behind_puts(cms_kfstore* st);
extern "C" void cms_kfstore_destroy(cms_kfstore* st) {
  if (!st) return;
  if (st->c) (void)hipSetDevice(st->c->device);
  void* bufs[] = {st->d_kf, st->d_kp, st->d_desc, st->d_rays, st->d_mp, st->d_fn, st->d_nid, st->d_noff, st->d_nfeat, st->d_work,
                  st->d_sorted, st->d_cell_start, st->d_nvalid, st->d_kp_cnt};
  for (void* b : bufs) if (b) (void)hipFree(b);
  if (st->h_ff) (void)hipHostFree(st->h_ff);
  if (st->h_items) (void)hipHostFree(st->h_items);
  if (st->h_upd) (void)hipHostFree(st->h_upd);
  for (hipEvent_t e : st->upd_ev) if (e) (void)hipEventDestroy(e);
  if (st->order_ev) (void)hipEventDestroy(st->order_ev);
  delete st;
}

extern "C" int cms_kfstore_create(cms_kfstore** out, cms_ctx* c, int max_keyframes, int max_features, int max_nodes) {
  if (!out || !c || max_keyframes < 1 || max_features < 1 || max_features > CMS_AREA_MAXKP || max_nodes < 1)
    return cms_fail(CMS_ERR_ARG, "cms_kfstore_create: bad argument (at most 16383 features per key frame)");
  HIPCHK(hipSetDevice(c->device));
  { const int rca = cms_area_grid_attr(c->device); if (rca) return rca; }      // cms_kfstore_put launches k_area_grid with up to 128 KB of LDS
  cms_kfstore* st = new cms_kfstore();
  st->c = c; st->maxkf = max_keyframes; st->maxf = max_features; st->maxn = max_nodes;
  const size_t K = (size_t)max_keyframes, Fq = (size_t)max_features, Nq = (size_t)max_nodes;
#define KF_ALLOC(ptr, bytes) do { if (hipMalloc((void**)&(ptr), (bytes)) != hipSuccess) { cms_kfstore_destroy(st); return cms_fail(CMS_ERR_HIP, "cms_kfstore_create: out of device memory"); } } while (0)
  KF_ALLOC(st->d_kf, K * sizeof(CmsTriKF));
  KF_ALLOC(st->d_kp, K * Fq * sizeof(CmsKeyPoint));
  KF_ALLOC(st->d_desc, K * Fq * 32);
  KF_ALLOC(st->d_rays, K * Fq * 12);
  KF_ALLOC(st->d_mp, K * Fq * 4);
  KF_ALLOC(st->d_fn, K * Fq * 4);
  KF_ALLOC(st->d_nid, K * Nq * 4);
  KF_ALLOC(st->d_noff, K * (Nq + 1) * 4);
  KF_ALLOC(st->d_nfeat, K * Fq * 4);
  KF_ALLOC(st->d_sorted, K * Fq * sizeof(uint16_t));
  KF_ALLOC(st->d_cell_start, K * (CMS_AREA_CELLS + 1) * sizeof(int));
  KF_ALLOC(st->d_nvalid, K * sizeof(int));
  KF_ALLOC(st->d_kp_cnt, K * sizeof(int));
#undef KF_ALLOC
  st->h_kf.assign(K, CmsTriKF{}); st->h_median.assign(K, 1.0f); st->used.assign(K, 0); st->busy.assign(K, 0);
  *out = st;
  return CMS_OK;
}

// upload / replace the key frame in `slot`
extern "C" int cms_kfstore_put(cms_kfstore* st, int slot, const cms_keyframe* kf) {
  if (!st || !kf || slot < 0 || slot >= st->maxkf) return cms_fail(CMS_ERR_ARG, "cms_kfstore_put: bad argument");
  int rc = tri_check_keyframe(*kf, "cms_kfstore_put: bad key frame");
  if (rc) return rc;
  if (kf->n > st->maxf || kf->nnodes > st->maxn) return cms_fail(CMS_ERR_ARG, "cms_kfstore_put: key frame larger than the store's slots");
  cms_ctx* c = st->c;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(kfstore_order_behind_puts(st));
  hipStream_t s = c->stream;
  const size_t f0 = (size_t)slot * st->maxf, n0 = (size_t)slot * st->maxn, o0 = (size_t)slot * (st->maxn + 1);
  CmsTriKF d;
  d.f0 = (int)f0; d.n = kf->n; d.node0 = (int)n0; d.nnodes = kf->nnodes; d.noff0 = (int)o0; d.nfeat0 = (int)f0;
  std::memcpy(d.Rcw, kf->Rcw, sizeof(d.Rcw)); std::memcpy(d.tcw, kf->tcw, sizeof(d.tcw)); std::memcpy(d.Ow, kf->Ow, sizeof(d.Ow));
  std::vector<int> fn((size_t)kf->n + 1, -1);
  tri_feat_node(*kf, fn.data());
  const int zero = 0;
  if (kf->n > 0) {
    HIPCHK(hipMemcpyAsync(st->d_kp + f0, kf->kps, (size_t)kf->n * sizeof(CmsKeyPoint), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(st->d_desc + 32 * f0, kf->desc, 32 * (size_t)kf->n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(st->d_rays + 3 * f0, kf->rays, 12 * (size_t)kf->n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(st->d_mp + f0, kf->mp, 4 * (size_t)kf->n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(st->d_fn + f0, fn.data(), 4 * (size_t)kf->n, hipMemcpyHostToDevice, s));
  }
  if (kf->nnodes > 0) {
    HIPCHK(hipMemcpyAsync(st->d_nid + n0, kf->node_id, 4 * (size_t)kf->nnodes, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(st->d_noff + o0, kf->node_off, 4 * ((size_t)kf->nnodes + 1), hipMemcpyHostToDevice, s));
    if (kf->node_off[kf->nnodes] > 0) HIPCHK(hipMemcpyAsync(st->d_nfeat + f0, kf->node_feat, 4 * (size_t)kf->node_off[kf->nnodes], hipMemcpyHostToDevice, s));
  } else {
    HIPCHK(hipMemcpyAsync(st->d_noff + o0, &zero, 4, hipMemcpyHostToDevice, s));
  }
  HIPCHK(hipMemcpyAsync(st->d_kf + slot, &d, sizeof(d), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(st->d_kp_cnt + slot, &kf->n, sizeof(int), hipMemcpyHostToDevice, s));
  {
    const float inv = (float)(3 * CMS_AREA_G) / (float)c->g.W;          // mfGridElementLengthInv (Frame.cpp:149)
    hipLaunchKernelGGL(k_area_grid, dim3(1), dim3(1024), (size_t)(st->maxf + 1) * 8, s, (const CmsKeyPoint*)(st->d_kp + f0), (const int*)(st->d_kp_cnt + slot), st->maxf, c->g.F, inv,
                       st->d_sorted + f0, st->d_cell_start + (size_t)slot * (CMS_AREA_CELLS + 1), st->d_nvalid + slot);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(s));
  st->h_kf[(size_t)slot] = d; st->h_median[(size_t)slot] = kf->median_depth; st->used[(size_t)slot] = 1;
  return CMS_OK;
}

// ---- LocalMapping::ProcessNewKeyFrame's device half (LocalMapping.cpp:52-117: the frame Tracking just made a key frame enters the map): the
// key points, descriptors, key rays and the frame grid of frame b are ALREADY on the device, in `src`'s frame context -- extraction (k_cull,
// k_describe) and cms_area_grid left them there.  cms_kfstore_put would send them through the host and back (nine pageable copies, a 53-us
// single-workgroup k_area_grid and a stream synchronisation per key frame); here one kernel copies them device to device into the slot, the
// grid included, and reads what only the host has (FeatureVector, map-point slots) from a pinned block.  Asynchronous: the copy is enqueued on
// `src`'s stream (behind the work that made frame b, in front of the next batch), the store's stream waits for it on the device.
struct CmsKfFromFrame {
  const CmsKeyPoint* kp; const uint8_t* desc; const float* rays; const uint16_t* sorted; const int* cell_start; const int* nvalid;      // source: frame b
  CmsKeyPoint* o_kp; uint8_t* o_desc; float* o_rays; uint16_t* o_sorted; int* o_cell_start; int* o_nvalid; int* o_kp_cnt;              // destination: the slot
  int* o_mp; int* o_fn; int* o_nid; int* o_noff; int* o_nfeat; CmsTriKF* o_kf;
  const int* h_mp; const int* h_fn; const int* h_nid; const int* h_noff; const int* h_nfeat;                                          // pinned host block (h_mp may be NULL: no map points)
  CmsTriKF kf; int n, nnodes, nfeat;
};
extern "C" __global__ void __launch_bounds__(256) k_kf_put_from_frame(const CmsKfFromFrame* __restrict__ items) {      // blockIdx.y = key frame of the batch
  const CmsKfFromFrame a = items[blockIdx.y];
  const int gs = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = a.n;
  {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(a.kp); uint32_t* d = reinterpret_cast<uint32_t*>(a.o_kp);      // 24-byte records
    for (int i = t0; i < 6 * n; i += gs) d[i] = s[i];
  }
  {
    const uint4* s = reinterpret_cast<const uint4*>(a.desc); uint4* d = reinterpret_cast<uint4*>(a.o_desc);
    for (int i = t0; i < 2 * n; i += gs) d[i] = s[i];
  }
  for (int i = t0; i < 3 * n; i += gs) a.o_rays[i] = a.rays[i];
  for (int i = t0; i < n; i += gs) { a.o_sorted[i] = a.sorted[i]; a.o_mp[i] = a.h_mp ? a.h_mp[i] : -1; a.o_fn[i] = a.h_fn[i]; }
  for (int i = t0; i <= CMS_AREA_CELLS; i += gs) a.o_cell_start[i] = a.cell_start[i];
  for (int i = t0; i < a.nnodes; i += gs) a.o_nid[i] = a.h_nid[i];
  for (int i = t0; i <= a.nnodes; i += gs) a.o_noff[i] = a.nnodes > 0 ? a.h_noff[i] : 0;
  for (int i = t0; i < a.nfeat; i += gs) a.o_nfeat[i] = a.h_nfeat[i];
  if (t0 == 0) { *a.o_nvalid = *a.nvalid; *a.o_kp_cnt = n; *a.o_kf = a.kf; }
}
// first thing every operation on the store's stream does: the stream waits (on the device) for the frame-to-store copies enqueued since the last one
static hipError_t kfstore_order_behind_puts(cms_kfstore* st) {
  std::vector<std::shared_ptr<cms_kfstore::PutCall>> w;
  { std::lock_guard<std::mutex> lk(st->put_mu); w.swap(st->pending_puts); }
  for (auto& c : w) { const hipError_t e = hipStreamWaitEvent(st->c->stream, c->ev, 0); if (e != hipSuccess) return e; }
  return hipSuccess;
}
static int kfstore_ff_reserve(cms_kfstore* st) {
  if (st->h_ff) return CMS_OK;
  st->ff_stride = ((size_t)3 * st->maxf + 2 * (size_t)st->maxn + 1 + 63) & ~(size_t)63;      // ints per slot: mp | feat_node | node_feat | node_id | node_off
  HIPCHK(hipHostMalloc((void**)&st->h_ff, (size_t)st->maxkf * st->ff_stride * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
  HIPCHK(hipHostMalloc((void**)&st->h_items, 2 * (size_t)st->maxkf * sizeof(CmsKfFromFrame), hipHostMallocMapped | hipHostMallocCoherent));
  st->ff_call.assign((size_t)st->maxkf, nullptr);
  return CMS_OK;
}
// validation of one key frame of a cms_kfstore_put_from_frames call: touches nothing
static int kf_put_check(cms_kfstore* st, int slot, cms_ctx* src, int b, int n, const float* Rcw, const float* tcw, const float* Ow,
                        const int* mp, int nnodes, const int* node_id, const int* node_off, const int* node_feat) {
  if (slot < 0 || slot >= st->maxkf || b < 0 || b >= src->max_batch || n < 0 || nnodes < 0 || !Rcw || !tcw || !Ow ||
      (nnodes > 0 && (!node_id || !node_off || !node_feat)))
    return cms_fail(CMS_ERR_ARG, "cms_kfstore_put_from_frame: bad argument");
  cms_ctx* c = st->c;
  if (b >= src->area_frames) return cms_fail(CMS_ERR_ARG, "cms_kfstore_put_from_frame: no frame grid for this frame (cms_area_grid on the batch first)");
  if (n > st->maxf || n > src->g.kp_cap || nnodes > st->maxn) return cms_fail(CMS_ERR_ARG, "cms_kfstore_put_from_frame: key frame larger than the store's slots");
  {
    cms_keyframe chk;                                              // (the FeatureVector checks of cms_kfstore_put)
    std::memset(&chk, 0, sizeof(chk));
    chk.n = n; chk.nnodes = nnodes; chk.node_id = node_id; chk.node_off = node_off; chk.node_feat = node_feat;
    chk.kps = reinterpret_cast<const cms_keypoint*>(st); chk.desc = reinterpret_cast<const uint8_t*>(st); chk.rays = reinterpret_cast<const float*>(st); chk.mp = reinterpret_cast<const int*>(st);
    const int rc = tri_check_keyframe(chk, "cms_kfstore_put_from_frame: bad key frame");
    if (rc) return rc;
  }
  const int nfeat = nnodes > 0 ? node_off[nnodes] : 0;
  if (nfeat > st->maxf) return cms_fail(CMS_ERR_ARG, "cms_kfstore_put_from_frame: FeatureVector lists more features than the slot holds");
  (void)c; (void)mp;
  return CMS_OK;
}
// ... then the slot's pinned block and the descriptor (no launch; the store's host-side record of the slot is committed by the caller once the
// kernel is in the stream)
static int kf_put_fill(cms_kfstore* st, int slot, cms_ctx* src, int b, int n, const float* Rcw, const float* tcw, const float* Ow,
                       const int* mp, int nnodes, const int* node_id, const int* node_off, const int* node_feat, CmsKfFromFrame& a) {
  const int nfeat = nnodes > 0 ? node_off[nnodes] : 0;
  if (st->ff_call[(size_t)slot]) { HIPCHK(hipEventSynchronize(st->ff_call[(size_t)slot]->ev)); st->ff_call[(size_t)slot].reset(); }      // (the call that last read this block: long through)
  int* h = st->h_ff + (size_t)slot * st->ff_stride;
  int* h_mp = h; int* h_fn = h + st->maxf; int* h_nfeat = h + 2 * (size_t)st->maxf; int* h_nid = h + 3 * (size_t)st->maxf; int* h_noff = h_nid + st->maxn;
  if (mp && n > 0) std::memcpy(h_mp, mp, 4 * (size_t)n);
  {
    // feature -> node: scattered writes, made in ordinary memory and copied in one piece (the block is uncached, device-coherent host memory: 1 650
    // scattered 4-byte stores into it cost ~170 us per key frame, 5.6 ms of the frame thread per bench step)
    static thread_local std::vector<int> fn;
    fn.assign((size_t)std::max(n, 1), -1);
    for (int e = 0; e < nnodes; ++e)
      for (int q = node_off[e]; q < node_off[e + 1]; ++q) fn[(size_t)node_feat[q]] = e;
    if (n > 0) std::memcpy(h_fn, fn.data(), 4 * (size_t)n);
  }
  if (nnodes > 0) { std::memcpy(h_nid, node_id, 4 * (size_t)nnodes); std::memcpy(h_noff, node_off, 4 * ((size_t)nnodes + 1)); std::memcpy(h_nfeat, node_feat, 4 * (size_t)nfeat); }
  const size_t f0 = (size_t)slot * st->maxf, n0 = (size_t)slot * st->maxn, o0 = (size_t)slot * (st->maxn + 1);
  std::memset(&a, 0, sizeof(a));
  CmsTriKF& d = a.kf;
  d.f0 = (int)f0; d.n = n; d.node0 = (int)n0; d.nnodes = nnodes; d.noff0 = (int)o0; d.nfeat0 = (int)f0;
  std::memcpy(d.Rcw, Rcw, sizeof(d.Rcw)); std::memcpy(d.tcw, tcw, sizeof(d.tcw)); std::memcpy(d.Ow, Ow, sizeof(d.Ow));
  const size_t sb = (size_t)b * src->g.kp_cap;
  a.kp = (const CmsKeyPoint*)src->d_kps + sb; a.desc = src->d_desc + 32 * sb; a.rays = src->d_rays + 3 * sb; a.sorted = src->d_area_sorted + sb;
  a.cell_start = src->d_area_cell_start + (size_t)b * (CMS_AREA_CELLS + 1); a.nvalid = src->d_area_nvalid + b;
  a.o_kp = st->d_kp + f0; a.o_desc = st->d_desc + 32 * f0; a.o_rays = st->d_rays + 3 * f0; a.o_sorted = st->d_sorted + f0;
  a.o_cell_start = st->d_cell_start + (size_t)slot * (CMS_AREA_CELLS + 1); a.o_nvalid = st->d_nvalid + slot; a.o_kp_cnt = st->d_kp_cnt + slot;
  a.o_mp = st->d_mp + f0; a.o_fn = st->d_fn + f0; a.o_nid = st->d_nid + n0; a.o_noff = st->d_noff + o0; a.o_nfeat = st->d_nfeat + f0; a.o_kf = st->d_kf + slot;
  a.h_mp = mp ? h_mp : nullptr; a.h_fn = h_fn; a.h_nid = h_nid; a.h_noff = h_noff; a.h_nfeat = h_nfeat;
  a.n = n; a.nnodes = nnodes; a.nfeat = nfeat;
  return CMS_OK;
}
// several key frames of one batch in one call (a process that tracks many camera streams per GPU inserts one key frame per stream and step): ONE
// kernel, one event, one stream wait for all of them
extern "C" int cms_kfstore_put_from_frames(cms_kfstore* st, cms_ctx* src, int n_items, const cms_kf_from_frame* items) {
  if (!st || !src || n_items < 0 || n_items > st->maxkf || (n_items > 0 && !items)) return cms_fail(CMS_ERR_ARG, "cms_kfstore_put_from_frames: bad argument");
  if (n_items == 0) return CMS_OK;
  cms_ctx* c = st->c;
  if (src->device != c->device || src->g.F != c->g.F || src->g.W != c->g.W) return cms_fail(CMS_ERR_ARG, "cms_kfstore_put_from_frame: the frame context and the store must share device and cubemap geometry");
  HIPCHK(hipSetDevice(c->device));
  { const int rc = kfstore_ff_reserve(st); if (rc) return rc; }
  // every item is checked before anything is touched: a bad item (or a slot named twice: the second would overwrite the first one's pinned block
  // with no event between them) leaves the store as it was
  {
    std::vector<uint8_t> seen((size_t)st->maxkf, 0);
    for (int i = 0; i < n_items; ++i) {
      const cms_kf_from_frame& q = items[i];
      const int rc = kf_put_check(st, q.slot, src, q.b, q.n, q.Rcw, q.tcw, q.Ow, q.mp, q.nnodes, q.node_id, q.node_off, q.node_feat);
      if (rc) return rc;
      if (seen[(size_t)q.slot]) return cms_fail(CMS_ERR_ARG, "cms_kfstore_put_from_frames: a slot is named twice");
      seen[(size_t)q.slot] = 1;
    }
  }
  const int gen = (st->items_gen ^= 1);
  if (st->items_call[gen]) { HIPCHK(hipEventSynchronize(st->items_call[gen]->ev)); st->items_call[gen].reset(); }      // (two calls ago)
  CmsKfFromFrame* h_items = reinterpret_cast<CmsKfFromFrame*>(st->h_items) + (size_t)gen * st->maxkf;
  int max_n = 0;
  for (int i = 0; i < n_items; ++i) {
    const cms_kf_from_frame& q = items[i];
    const int rc = kf_put_fill(st, q.slot, src, q.b, q.n, q.Rcw, q.tcw, q.Ow, q.mp, q.nnodes, q.node_id, q.node_off, q.node_feat, h_items[i]);
    if (rc) return rc;      // (only a failing event wait: nothing of the store's record has changed yet)
    max_n = std::max(max_n, q.n);
  }
  // The copy runs on the FRAME context's stream: right behind the work that produced the frames, and in front of whatever the caller enqueues there
  // next (the next batch overwrites the frame buffers) -- the KeyFrame constructor's copy happens on the Tracking thread in the reference too
  // (Tracking.cpp:1015-1017, KeyFrame.cpp:29-55).  The store's stream then waits for it on the device.
  hipStream_t s = src->stream;
  // ... but work already queued on the STORE's stream may still read or write a slot this call refills (an asynchronous pose update landing after
  // the new record, a search still reading the old key points): then the frame stream waits for the store's stream first.  Slots that nothing on
  // the store's stream referred to since they were last filled -- the usual case: a new key frame goes to a fresh slot -- cost no wait.
  {
    bool wait = false;
    for (int i = 0; i < n_items; ++i) wait = wait || (st->busy[(size_t)items[i].slot] != 0);
    if (wait && c->stream != s) {
      if (!st->order_ev) HIPCHK(hipEventCreateWithFlags(&st->order_ev, hipEventDisableTiming));
      HIPCHK(hipEventRecord(st->order_ev, c->stream));
      HIPCHK(hipStreamWaitEvent(s, st->order_ev, 0));
      std::fill(st->busy.begin(), st->busy.end(), 0);      // (everything queued on the store's stream so far is in front of the copy now)
    }
  }
  hipLaunchKernelGGL(k_kf_put_from_frame, dim3(std::max(1, std::min(32, (6 * max_n + 255) / 256)), n_items), dim3(256), 0, s, (const CmsKfFromFrame*)h_items);
  HIPCHK(hipGetLastError());
  auto call = std::make_shared<cms_kfstore::PutCall>();
  HIPCHK(hipEventCreateWithFlags(&call->ev, hipEventDisableTiming));
  HIPCHK(hipEventRecord(call->ev, s));
  if (c->stream != s) { std::lock_guard<std::mutex> lk(st->put_mu); st->pending_puts.push_back(call); }      // (the store's stream waits for it at its next own operation)
  for (int i = 0; i < n_items; ++i) {      // the kernel is in the stream: commit the store's record of the slots
    const cms_kf_from_frame& q = items[i];
    st->ff_call[(size_t)q.slot] = call;
    st->h_kf[(size_t)q.slot] = h_items[i].kf; st->h_median[(size_t)q.slot] = q.median_depth; st->used[(size_t)q.slot] = 1;
  }
  st->items_call[gen] = call;
  return CMS_OK;
}
extern "C" int cms_kfstore_put_from_frame(cms_kfstore* st, int slot, cms_ctx* src, int b, int n, const float* Rcw, const float* tcw, const float* Ow,
                                          float median_depth, const int* mp, int nnodes, const int* node_id, const int* node_off, const int* node_feat) {
  cms_kf_from_frame q;
  q.slot = slot; q.b = b; q.n = n; q.Rcw = Rcw; q.tcw = tcw; q.Ow = Ow; q.median_depth = median_depth; q.mp = mp; q.nnodes = nnodes;
  q.node_id = node_id; q.node_off = node_off; q.node_feat = node_feat;
  return cms_kfstore_put_from_frames(st, src, 1, &q);
}

// the poses of n resident key frames after a local BA (Optimizer.cpp:419-431 writes them back; LocalMapping's next CreateNewMapPoints reads them):
// one kernel, the values read from a pinned block, no synchronisation -- cms_kfstore_update per key frame is a copy and a stream wait each
// (the call's slot list travels in a pinned ring of four lists: a list is rewritten four calls later)
extern "C" __global__ void __launch_bounds__(64) k_kf_update_poses(CmsTriKF* kf, const float* upd, const int* __restrict__ list, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int slot = list[i] & 0x3FFFFFFF, par = (list[i] >> 30) & 1;
  const float* u = upd + 16 * (2 * (size_t)slot + par);
  CmsTriKF& d = kf[slot];
  for (int j = 0; j < 9; ++j) d.Rcw[j] = u[j];
  for (int j = 0; j < 3; ++j) { d.tcw[j] = u[9 + j]; d.Ow[j] = u[12 + j]; }
}
extern "C" int cms_kfstore_update_poses(cms_kfstore* st, int n, const int* slots, const float* Rcw, const float* tcw, const float* Ow) {
  if (!st || n < 0 || n > st->maxkf || (n > 0 && (!slots || !Rcw || !tcw || !Ow))) return cms_fail(CMS_ERR_ARG, "cms_kfstore_update_poses: bad argument");
  for (int i = 0; i < n; ++i) if (slots[i] < 0 || slots[i] >= st->maxkf || !st->used[(size_t)slots[i]]) return cms_fail(CMS_ERR_ARG, "cms_kfstore_update_poses: bad slot");
  if (n == 0) return CMS_OK;
  cms_ctx* c = st->c;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(kfstore_order_behind_puts(st));
  if (!st->h_upd) {      // two 16-float blocks per slot, then four slot lists of maxkf ints
    HIPCHK(hipHostMalloc((void**)&st->h_upd, (size_t)st->maxkf * (32 * sizeof(float) + 4 * sizeof(int)), hipHostMallocMapped | hipHostMallocCoherent));
    st->upd_par.assign((size_t)st->maxkf, 0);
  }
  const unsigned ring = st->upd_call++ & 3;
  if (!st->upd_ev[ring]) HIPCHK(hipEventCreateWithFlags(&st->upd_ev[ring], hipEventDisableTiming));
  // the call before the last one must be through (it usually is, long since): then every pose block written up to it has been read -- this call
  // writes each slot's OTHER block than the slot's last update did -- and so has the slot list this call is about to reuse
  if (st->upd_ev_set[(ring + 2) & 3]) HIPCHK(hipEventSynchronize(st->upd_ev[(ring + 2) & 3]));
  int* list = reinterpret_cast<int*>(st->h_upd + (size_t)st->maxkf * 32) + (size_t)ring * st->maxkf;
  for (int i = 0; i < n; ++i) {
    const int slot = slots[i];
    const int par = (st->upd_par[(size_t)slot] ^= 1);
    float* u = st->h_upd + 16 * (2 * (size_t)slot + par);
    std::memcpy(u, Rcw + 9 * (size_t)i, 36); std::memcpy(u + 9, tcw + 3 * (size_t)i, 12); std::memcpy(u + 12, Ow + 3 * (size_t)i, 12);
    CmsTriKF& d = st->h_kf[(size_t)slot];
    std::memcpy(d.Rcw, Rcw + 9 * (size_t)i, 36); std::memcpy(d.tcw, tcw + 3 * (size_t)i, 12); std::memcpy(d.Ow, Ow + 3 * (size_t)i, 12);
    list[i] = slot | (par << 30);
    st->busy[(size_t)slot] = 1;      // (asynchronous: a later cms_kfstore_put_from_frames into this slot orders itself behind this kernel)
  }
  hipLaunchKernelGGL(k_kf_update_poses, dim3((n + 63) / 64), dim3(64), 0, c->stream, st->d_kf, (const float*)st->h_upd, (const int*)list, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(st->upd_ev[ring], c->stream));
  st->upd_ev_set[ring] = true;
  return CMS_OK;
}

// developer / test entry: what a slot holds on the device (any pointer may be NULL).  kps / desc / rays / mp / feat_node / sorted: n entries (n = header[1]);
// node_id nnodes, node_off nnodes + 1, node_feat node_off[nnodes]; cell_start 12501 ints; header: the 21 32-bit words of the slot's record (f0, n, node0,
// nnodes, noff0, nfeat0, Rcw[9], tcw[3], Ow[3]); misc[2] = valid grid entries, stored key-point count
extern "C" int cms_kfstore_debug_fetch(cms_kfstore* st, int slot, cms_keypoint* kps, uint8_t* desc, float* rays, int* mp, int* feat_node, uint16_t* sorted,
                                       int* node_id, int* node_off, int* node_feat, int* cell_start, uint32_t* header, int* misc) {
  if (!st || slot < 0 || slot >= st->maxkf || !st->used[(size_t)slot]) return cms_fail(CMS_ERR_ARG, "cms_kfstore_debug_fetch: bad slot");
  cms_ctx* c = st->c;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(kfstore_order_behind_puts(st));
  HIPCHK(hipStreamSynchronize(c->stream));
  CmsTriKF d;
  HIPCHK(hipMemcpy(&d, st->d_kf + slot, sizeof(d), hipMemcpyDeviceToHost));
  static_assert(sizeof(CmsTriKF) == 21 * 4, "CmsTriKF is 21 words");
  if (header) std::memcpy(header, &d, sizeof(d));
  const size_t f0 = (size_t)slot * st->maxf, n0 = (size_t)slot * st->maxn, o0 = (size_t)slot * (st->maxn + 1);
  const size_t n = (size_t)std::max(0, std::min(d.n, st->maxf)), nn = (size_t)std::max(0, std::min(d.nnodes, st->maxn));
  auto dl = [&](void* dst, const void* src, size_t bytes) { return (!dst || bytes == 0) ? hipSuccess : hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost); };
  HIPCHK(dl(kps, st->d_kp + f0, n * sizeof(CmsKeyPoint))); HIPCHK(dl(desc, st->d_desc + 32 * f0, 32 * n)); HIPCHK(dl(rays, st->d_rays + 3 * f0, 12 * n));
  HIPCHK(dl(mp, st->d_mp + f0, 4 * n)); HIPCHK(dl(feat_node, st->d_fn + f0, 4 * n)); HIPCHK(dl(sorted, st->d_sorted + f0, 2 * n));
  HIPCHK(dl(node_id, st->d_nid + n0, 4 * nn)); HIPCHK(dl(node_off, st->d_noff + o0, 4 * (nn + 1)));
  int nfeat = 0;
  if (nn > 0) HIPCHK(hipMemcpy(&nfeat, st->d_noff + o0 + nn, 4, hipMemcpyDeviceToHost));
  HIPCHK(dl(node_feat, st->d_nfeat + f0, 4 * (size_t)std::max(0, std::min(nfeat, st->maxf))));
  HIPCHK(dl(cell_start, st->d_cell_start + (size_t)slot * (CMS_AREA_CELLS + 1), 4 * ((size_t)CMS_AREA_CELLS + 1)));
  if (misc) { HIPCHK(hipMemcpy(misc, st->d_nvalid + slot, 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(misc + 1, st->d_kp_cnt + slot, 4, hipMemcpyDeviceToHost)); }
  return CMS_OK;
}

// what changes on a key frame between CreateNewMapPoints calls: pose (local BA), median depth, map-point slots (any may be NULL: unchanged)
extern "C" int cms_kfstore_update(cms_kfstore* st, int slot, const float* Rcw, const float* tcw, const float* Ow, const float* median_depth, const int* mp) {
  if (!st || slot < 0 || slot >= st->maxkf || !st->used[(size_t)slot]) return cms_fail(CMS_ERR_ARG, "cms_kfstore_update: bad slot");
  cms_ctx* c = st->c;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(kfstore_order_behind_puts(st));
  CmsTriKF& d = st->h_kf[(size_t)slot];
  if (Rcw) std::memcpy(d.Rcw, Rcw, sizeof(d.Rcw));
  if (tcw) std::memcpy(d.tcw, tcw, sizeof(d.tcw));
  if (Ow) std::memcpy(d.Ow, Ow, sizeof(d.Ow));
  if (median_depth) st->h_median[(size_t)slot] = *median_depth;
  if (Rcw || tcw || Ow) HIPCHK(hipMemcpyAsync(st->d_kf + slot, &d, sizeof(d), hipMemcpyHostToDevice, c->stream));
  if (mp && d.n > 0) HIPCHK(hipMemcpyAsync(st->d_mp + (size_t)d.f0, mp, 4 * (size_t)d.n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return CMS_OK;
}

extern "C" int cms_kfstore_create_new_map_points(cms_kfstore* st, int njobs, const int* cur_slot, const int* neigh_off, const int* neigh_slot,
                                                 int check_orientation, int cap_per_job, int* n_new, int* out_neigh, int* out_idx1, int* out_idx2,
                                                 float* out_x3d) {
  if (!st || njobs < 0 || cap_per_job < 0 || (njobs > 0 && (!cur_slot || !neigh_off || !n_new)) ||
      (njobs > 0 && cap_per_job > 0 && (!out_neigh || !out_idx1 || !out_idx2 || !out_x3d)))
    return cms_fail(CMS_ERR_ARG, "cms_kfstore_create_new_map_points: bad argument");
  if (njobs == 0) return CMS_OK;
  cms_ctx* c = st->c;
  if (c->g.nlevels > 16 || c->g.nlevels < 2) return cms_fail(CMS_ERR_UNSUPPORTED, "cms_kfstore_create_new_map_points: 2..16 pyramid levels");
  const int nneigh = neigh_off[njobs];
  if (nneigh < 0 || (nneigh > 0 && !neigh_slot)) return cms_fail(CMS_ERR_ARG, "cms_kfstore_create_new_map_points: bad neighbour list");
  for (int j = 0; j < njobs; ++j)
    if (cur_slot[j] < 0 || cur_slot[j] >= st->maxkf || !st->used[(size_t)cur_slot[j]]) return cms_fail(CMS_ERR_ARG, "cms_kfstore_create_new_map_points: empty slot");
  for (int q = 0; q < nneigh; ++q)
    if (neigh_slot[q] < 0 || neigh_slot[q] >= st->maxkf || !st->used[(size_t)neigh_slot[q]]) return cms_fail(CMS_ERR_ARG, "cms_kfstore_create_new_map_points: empty slot");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(kfstore_order_behind_puts(st));
  const size_t need = tri_work_bytes(njobs, nneigh, st->maxf, cap_per_job);
  if (need > st->work_bytes) {
    if (st->d_work) HIPCHK(hipFree(st->d_work));
    st->d_work = nullptr; st->work_bytes = 0;
    HIPCHK(hipMalloc((void**)&st->d_work, need + need / 4));
    st->work_bytes = need + need / 4;
  }
  TriDev dev;
  dev.kf = st->d_kf; dev.kp = st->d_kp; dev.desc = (const uint4*)st->d_desc; dev.rays = st->d_rays; dev.mp = st->d_mp; dev.feat_node = st->d_fn;
  dev.node_id = st->d_nid; dev.node_off = st->d_noff; dev.node_feat = st->d_nfeat;
  return tri_run(c, dev, st->h_kf.data(), st->h_median.data(), njobs, cur_slot, neigh_off, neigh_slot, check_orientation, cap_per_job, st->d_work, n_new,
                 out_neigh, out_idx1, out_idx2, out_x3d);
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
    fa.src = nullptr;
    fa.bounds_scaled = c->dist_bounds_scaled;
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
    sa.cap = 0; sa.src = nullptr; sa.row_slot = nullptr; sa.maxf = 0;
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

// MapPoint::ComputeDistinctiveDescriptors for a batch of map points (host buffers)
extern "C" int cms_distinctive_descriptors(cms_ctx* c, int npts, const int* obs_off, const uint8_t* desc, int* best_idx) {
  if (!c || npts < 0 || (npts > 0 && (!obs_off || !best_idx))) return cms_fail(CMS_ERR_ARG, "cms_distinctive_descriptors: bad argument");
  if (npts == 0) return CMS_OK;
  const int nobs = obs_off[npts];
  if (nobs < 0 || (nobs > 0 && !desc)) return cms_fail(CMS_ERR_ARG, "cms_distinctive_descriptors: bad observation list");
  for (int p = 0; p < npts; ++p) {
    if (obs_off[p + 1] < obs_off[p]) return cms_fail(CMS_ERR_ARG, "cms_distinctive_descriptors: obs_off must ascend");
    if (obs_off[p + 1] - obs_off[p] > 65535) return cms_fail(CMS_ERR_UNSUPPORTED, "cms_distinctive_descriptors: more than 65535 observations of one point");
  }
  HIPCHK(hipSetDevice(c->device));
  const size_t o_off = 0, o_desc = tri_al((size_t)(npts + 1) * 4), o_out = o_desc + tri_al((size_t)nobs * 32 + 32);
  int rc = cms_scratch(c, o_out + tri_al((size_t)npts * 4));
  if (rc) return rc;
  uint8_t* p = (uint8_t*)c->d_match;
  hipStream_t s = c->stream;
  HIPCHK(hipMemcpyAsync(p + o_off, obs_off, (size_t)(npts + 1) * 4, hipMemcpyHostToDevice, s));
  if (nobs > 0) HIPCHK(hipMemcpyAsync(p + o_desc, desc, (size_t)nobs * 32, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_distinctive, dim3((npts + 3) / 4), dim3(256), 0, s, npts, (const int*)(p + o_off), (const uint4*)(p + o_desc), (int*)(p + o_out));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(best_idx, p + o_out, (size_t)npts * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return CMS_OK;
}

// MapPoint::UpdateNormalAndDepth for a batch of map points (host buffers); points without observations keep their outputs untouched
extern "C" int cms_update_normal_and_depth(cms_ctx* c, int npts, const int* obs_off, const float* pos, const float* obs_Ow, const float* ref_Ow,
                                           const int* ref_level, float* normal, float* min_dist, float* max_dist) {
  if (!c || npts < 0 || (npts > 0 && (!obs_off || !pos || !ref_Ow || !ref_level || !normal || !min_dist || !max_dist)))
    return cms_fail(CMS_ERR_ARG, "cms_update_normal_and_depth: bad argument");
  if (npts == 0) return CMS_OK;
  const int nobs = obs_off[npts];
  if (nobs < 0 || (nobs > 0 && !obs_Ow)) return cms_fail(CMS_ERR_ARG, "cms_update_normal_and_depth: bad observation list");
  for (int p = 0; p < npts; ++p) {
    if (obs_off[p + 1] < obs_off[p]) return cms_fail(CMS_ERR_ARG, "cms_update_normal_and_depth: obs_off must ascend");
    if (ref_level[p] < 0 || ref_level[p] >= c->g.nlevels) return cms_fail(CMS_ERR_ARG, "cms_update_normal_and_depth: reference level out of range");
  }
  HIPCHK(hipSetDevice(c->device));
  const size_t n4 = (size_t)npts * 4;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += tri_al(bytes + 16); return at; };
  const size_t o_off = take(n4 + 4), o_pos = take(3 * n4), o_ow = take((size_t)nobs * 12), o_ref = take(3 * n4), o_lvl = take(n4), o_sf = take(64),
               o_nrm = take(3 * n4), o_min = take(n4), o_max = take(n4);
  int rc = cms_scratch(c, o);
  if (rc) return rc;
  uint8_t* p = (uint8_t*)c->d_match;
  hipStream_t s = c->stream;
  float sf[16];
  for (int l = 0; l < 16; ++l) sf[l] = l < c->g.nlevels ? c->scale[l] : 1.0f;
  HIPCHK(hipMemcpyAsync(p + o_off, obs_off, n4 + 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(p + o_pos, pos, 3 * n4, hipMemcpyHostToDevice, s));
  if (nobs > 0) HIPCHK(hipMemcpyAsync(p + o_ow, obs_Ow, (size_t)nobs * 12, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(p + o_ref, ref_Ow, 3 * n4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(p + o_lvl, ref_level, n4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(p + o_sf, sf, sizeof(sf), hipMemcpyHostToDevice, s));
  // outputs start from the caller's values so that points without observations come back unchanged
  HIPCHK(hipMemcpyAsync(p + o_nrm, normal, 3 * n4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(p + o_min, min_dist, n4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(p + o_max, max_dist, n4, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_update_normal_depth, dim3((npts + 255) / 256), dim3(256), 0, s, npts, (const int*)(p + o_off), (const float*)(p + o_pos),
                     (const float*)(p + o_ow), (const float*)(p + o_ref), (const int*)(p + o_lvl), (const float*)(p + o_sf), c->g.nlevels,
                     (float*)(p + o_nrm), (float*)(p + o_min), (float*)(p + o_max));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(normal, p + o_nrm, 3 * n4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(min_dist, p + o_min, n4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(max_dist, p + o_max, n4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return CMS_OK;
}


// SearchInNeighbors' Fuse calls on resident key frames, all in one launch sequence: job j searches the map points [mp_off[j], mp_off[j+1])
// (host arrays, concatenated) in the key frame of slot job_slot[j].  best_idx[i] = key point of that key frame or -1.
// per map point of a batched Fuse call: its job (binary search in the jobs' offsets) and the job's slot -- on the device: with a quarter of a million
// map points per call (SearchInNeighbors of 16 key frames: each one's ~750 map points into 20 neighbours) the host loops that filled these two
// arrays, their upload and the host loop over the results were a third of the call
extern "C" __global__ void __launch_bounds__(256)
k_fuse_expand_jobs(int nmp, int njobs, const int* __restrict__ mp_off, const int* __restrict__ job_slot, int* __restrict__ mp_job, int* __restrict__ mp_slot,
                   const int* __restrict__ job_set0, int* __restrict__ mp_src) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nmp) return;
  int lo = 0, hi = njobs;                                          // last job whose first map point is <= i (empty jobs skipped by the search)
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (mp_off[mid] <= i) lo = mid; else hi = mid; }
  mp_job[i] = lo; mp_slot[i] = job_slot[lo];
  if (mp_src) mp_src[i] = job_set0[lo] + (i - mp_off[lo]);         // jobs over shared sets of map points: entry -> the set's map point
}
// job_set0 == NULL: every job brings its own map points (entry i of the concatenated arrays IS map point i; npts = entries).  Otherwise the jobs
// refer to SETS of map points uploaded once (job j's entries are the points job_set0[j] .. of the npts-long arrays): SearchInNeighbors sends one key
// frame's map points to each of its ~20 neighbours, and 20 copies of the same positions / normals / descriptors were 15 of the 16 MB a call uploaded.
static int kfstore_fuse_core(cms_kfstore* st, int njobs, const int* job_slot, const int* mp_off, const int* job_set0, int npts, const uint8_t* skip, const float* pos,
                             const float* normal, const float* min_dist, const float* max_dist, const uint8_t* mp_desc, float th, int* best_idx, int* best_dist) {
  if (!st || njobs < 0 || (njobs > 0 && (!job_slot || !mp_off))) return cms_fail(CMS_ERR_ARG, "cms_kfstore_fuse_search: bad argument");
  if (njobs == 0) return CMS_OK;
  const int nmp = mp_off[njobs];
  if (nmp < 0 || mp_off[0] != 0 || (nmp > 0 && (!pos || !normal || !min_dist || !max_dist || !mp_desc || !best_idx || !best_dist)))
    return cms_fail(CMS_ERR_ARG, "cms_kfstore_fuse_search: bad map-point arrays");
  if (nmp == 0) return CMS_OK;
  cms_ctx* c = st->c;
  if (c->g.nlevels > 16) return cms_fail(CMS_ERR_UNSUPPORTED, "cms_kfstore_fuse_search: more than 16 pyramid levels");
  // per JOB on the host: the slot's pose; per MAP POINT everything happens on the device
  static thread_local std::vector<float> pose;
  pose.resize((size_t)njobs * 15);
  for (int j = 0; j < njobs; ++j) {
    const int sl = job_slot[j];
    if (sl < 0 || sl >= st->maxkf || !st->used[(size_t)sl] || mp_off[j + 1] < mp_off[j]) return cms_fail(CMS_ERR_ARG, "cms_kfstore_fuse_search: bad job");
    const CmsTriKF& k = st->h_kf[(size_t)sl];
    memcpy(&pose[15 * (size_t)j], k.Rcw, 36); memcpy(&pose[15 * (size_t)j + 9], k.tcw, 12); memcpy(&pose[15 * (size_t)j + 12], k.Ow, 12);
  }
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(kfstore_order_behind_puts(st));
  hipStream_t s = c->stream;
  if (!job_set0) npts = nmp;
  const size_t n4 = (size_t)nmp * 4, j4 = (size_t)njobs * 4, p4 = (size_t)npts * 4;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
  const size_t o_pose = take(pose.size() * 4), o_joff = take(j4 + 4), o_jslot = take(j4), o_jset = take(j4), o_src = take(n4), o_job = take(n4), o_slot = take(n4), o_skip = take(nmp),
               o_pos = take(3 * p4), o_nrm = take(3 * p4), o_min = take(p4), o_max = take(p4), o_desc = take((size_t)npts * 32), o_qx = take(n4), o_qy = take(n4), o_qr = take(n4),
               o_qmin = take(n4), o_qmax = take(n4), o_lvl = take(n4), o_cnt = take(n4), o_off = take(n4 + 4), o_tot = take(16), o_bi = take(n4), o_bd = take(n4);
  const size_t fixed = o;
  bool direct = false;
  {
    hipPointerAttribute_t a1, a2;
    if (hipPointerGetAttributes(&a1, best_idx) == hipSuccess && hipPointerGetAttributes(&a2, best_dist) == hipSuccess)
      direct = a1.type == hipMemoryTypeHost && a2.type == hipMemoryTypeHost;
    else (void)hipGetLastError();                                  // (a pageable pointer: not an error, just the copies)
  }
  int cap = 64 * nmp + 1024;
  for (int attempt = 0; attempt < 2; ++attempt) {
    int rc = cms_scratch(c, fixed + al((size_t)cap * 4));
    if (rc) return rc;
    uint8_t* p = (uint8_t*)c->d_match;
    const size_t o_idx = fixed;
    HIPCHK(hipMemcpyAsync(p + o_pose, pose.data(), pose.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(p + o_joff, mp_off, j4 + 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(p + o_jslot, job_slot, j4, hipMemcpyHostToDevice, s));
    if (skip) HIPCHK(hipMemcpyAsync(p + o_skip, skip, nmp, hipMemcpyHostToDevice, s));
    if (job_set0) HIPCHK(hipMemcpyAsync(p + o_jset, job_set0, j4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(p + o_pos, pos, 3 * p4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(p + o_nrm, normal, 3 * p4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(p + o_min, min_dist, p4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(p + o_max, max_dist, p4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(p + o_desc, mp_desc, (size_t)npts * 32, hipMemcpyHostToDevice, s));
    const int* src_dev = job_set0 ? (const int*)(p + o_src) : nullptr;
    hipLaunchKernelGGL(k_fuse_expand_jobs, dim3((nmp + 255) / 256), dim3(256), 0, s, nmp, njobs, (const int*)(p + o_joff), (const int*)(p + o_jslot), (int*)(p + o_job), (int*)(p + o_slot),
                       job_set0 ? (const int*)(p + o_jset) : nullptr, job_set0 ? (int*)(p + o_src) : nullptr);
    CmsFuseArgs fa;
    fa.src = src_dev;
    fa.bounds_scaled = c->dist_bounds_scaled;
    fa.pose15 = (const float*)(p + o_pose); fa.mp_frame = (const int*)(p + o_job); fa.n = nmp; fa.skip = skip ? p + o_skip : nullptr;
    fa.P = (const float*)(p + o_pos); fa.normal = (const float*)(p + o_nrm); fa.min_dist = (const float*)(p + o_min); fa.max_dist = (const float*)(p + o_max);
    fa.th = th; fa.log_scale = std::log(c->scale[1]); fa.nlevels = c->g.nlevels; fa.F = c->g.F;
    for (int l = 0; l < 16; ++l) fa.sf[l] = l < c->g.nlevels ? c->scale[l] : 0.0f;
    fa.qx = (float*)(p + o_qx); fa.qy = (float*)(p + o_qy); fa.qr = (float*)(p + o_qr); fa.qmin = (int*)(p + o_qmin); fa.qmax = (int*)(p + o_qmax);
    fa.level = (int*)(p + o_lvl);
    hipLaunchKernelGGL(k_fuse_project, dim3((nmp + 255) / 256), dim3(256), 0, s, fa);
    // window query against the grids of the store's slots
    CmsAreaArgs a;
    a.tmp = nullptr;
    a.kp = (const CmsKeyPoint*)st->d_kp; a.sorted_idx = st->d_sorted; a.cell_start = st->d_cell_start;
    a.qx = fa.qx; a.qy = fa.qy; a.qr = fa.qr; a.qmin = fa.qmin; a.qmax = fa.qmax;
    a.q_frame = (const int*)(p + o_slot); a.kp_cap = st->maxf;
    a.nq = nmp; a.F = c->g.F; a.inv = (float)(3 * CMS_AREA_G) / (float)c->g.W;
    a.cnt = (int*)(p + o_cnt); a.off = (const int*)(p + o_off); a.idx = (int*)(p + o_idx); a.cap = cap; a.idx_base = 0;
    {
      const int nblk = (nmp + 1023) / 1024, qgrid = (nmp * CMS_AREA_QL + 255) / 256;
      rc = cms_area_bsum_reserve(c, nblk);
      if (rc) return rc;
      hipLaunchKernelGGL(k_area_query, dim3(qgrid), dim3(256), 0, s, a, 0);
      hipLaunchKernelGGL(k_area_blocksum, dim3(nblk), dim3(1024), 0, s, (const int*)(p + o_cnt), nmp, c->d_area_bsum);
      hipLaunchKernelGGL(k_area_scan, dim3(nblk), dim3(1024), 0, s, (const int*)(p + o_cnt), nmp, (const int*)c->d_area_bsum, (int*)(p + o_off), (int*)(p + o_tot));
      hipLaunchKernelGGL(k_area_query, dim3(qgrid), dim3(256), 0, s, a, 1);      // (writes at most `cap` candidates: a list that does not fit is noticed below)
    }
    // The scan is enqueued right behind the windows: the total is looked at together with the results (ONE synchronisation per call; the fill pass
    // never writes beyond `cap`, and a call whose lists did not fit is simply repeated with room for them)
    CmsFuseScanArgs sa;
    sa.cap = cap; sa.src = src_dev; sa.row_slot = (const int*)(p + o_slot); sa.maxf = st->maxf;
    sa.n = nmp; sa.qx = fa.qx; sa.qy = fa.qy; sa.level = fa.level; sa.mp_desc = (const uint4*)(p + o_desc);
    sa.cand_off = (const int*)(p + o_off); sa.cand_idx = (const int*)(p + o_idx); sa.kp = (const CmsKeyPoint*)st->d_kp; sa.t_desc = (const uint4*)st->d_desc;
    for (int l = 0; l < 16; ++l) sa.inv_sigma2[l] = l < c->g.nlevels ? c->inv_sigma2[l] : 0.0f;
    // result arrays in pinned (device-visible) host memory: the kernels store there themselves -- copies queued behind kernels of the same stream are
    // blit kernels of the runtime, two more dependent launches that wait for a slot on a busy chip
    sa.best_idx = direct ? best_idx : (int*)(p + o_bi); sa.best_dist = direct ? best_dist : (int*)(p + o_bd);
    hipLaunchKernelGGL(k_fuse_scan, dim3((nmp * 8 + 255) / 256), dim3(256), 0, s, sa);      // (stores the key-point index inside the job's key frame: row - slot x maxf)
    HIPCHK(hipGetLastError());
    int tot = 0;
    HIPCHK(hipMemcpyAsync(&tot, p + o_tot, sizeof(int), hipMemcpyDeviceToHost, s));
    if (!direct) {
      HIPCHK(hipMemcpyAsync(best_idx, p + o_bi, n4, hipMemcpyDeviceToHost, s));
      HIPCHK(hipMemcpyAsync(best_dist, p + o_bd, n4, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    if (tot > cap) { cap = tot + 64; continue; }
    return CMS_OK;
  }
  return cms_fail(CMS_ERR_OVERFLOW, "cms_kfstore_fuse_search: candidate lists kept growing");
}
extern "C" int cms_kfstore_fuse_search(cms_kfstore* st, int njobs, const int* job_slot, const int* mp_off, const uint8_t* skip, const float* pos,
                                       const float* normal, const float* min_dist, const float* max_dist, const uint8_t* mp_desc, float th,
                                       int* best_idx, int* best_dist) {
  return kfstore_fuse_core(st, njobs, job_slot, mp_off, nullptr, 0, skip, pos, normal, min_dist, max_dist, mp_desc, th, best_idx, best_dist);
}
// SearchInNeighbors' shape (LocalMapping.cpp:388-466): nsets sets of map points (set s = points set_off[s] .. set_off[s + 1] of pos / normal / min_dist /
// max_dist / mp_desc), njobs jobs (key frame slot job_slot[j], set job_set[j]); skip / best_idx / best_dist are per ENTRY: job after job, a job's entries
// in the order of its set (skip may be NULL)
extern "C" int cms_kfstore_fuse_search_sets(cms_kfstore* st, int nsets, const int* set_off, const float* pos, const float* normal, const float* min_dist,
                                            const float* max_dist, const uint8_t* mp_desc, int njobs, const int* job_slot, const int* job_set, const uint8_t* skip,
                                            float th, int* best_idx, int* best_dist) {
  if (!st || nsets < 0 || njobs < 0 || (nsets > 0 && !set_off) || (njobs > 0 && (!job_slot || !job_set))) return cms_fail(CMS_ERR_ARG, "cms_kfstore_fuse_search_sets: bad argument");
  if (njobs == 0) return CMS_OK;
  if (nsets == 0 || set_off[0] != 0) return cms_fail(CMS_ERR_ARG, "cms_kfstore_fuse_search_sets: bad sets");
  for (int s = 0; s < nsets; ++s) if (set_off[s + 1] < set_off[s]) return cms_fail(CMS_ERR_ARG, "cms_kfstore_fuse_search_sets: set offsets must ascend");
  static thread_local std::vector<int> off, set0;
  off.resize((size_t)njobs + 1); set0.resize((size_t)njobs);
  off[0] = 0;
  for (int j = 0; j < njobs; ++j) {
    if (job_set[j] < 0 || job_set[j] >= nsets) return cms_fail(CMS_ERR_ARG, "cms_kfstore_fuse_search_sets: bad set index");
    set0[(size_t)j] = set_off[job_set[j]];
    const long long nx = (long long)off[(size_t)j] + (set_off[job_set[j] + 1] - set_off[job_set[j]]);
    if (nx > 0x7FFFFFFF / 80) return cms_fail(CMS_ERR_ARG, "cms_kfstore_fuse_search_sets: too many entries for one call");
    off[(size_t)j + 1] = (int)nx;
  }
  return kfstore_fuse_core(st, njobs, job_slot, off.data(), set0.data(), set_off[nsets], skip, pos, normal, min_dist, max_dist, mp_desc, th, best_idx, best_dist);
}
