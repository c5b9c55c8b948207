#include <mutex>
// cms_api_track.hip -- host side of the "track local map" step (Frame::isInFrustum + ORBMatcher::SearchByProjection over the local
// map points, Tracking::SearchLocalPoints), included by cms_lib.hip after cms_api_area.hip.
#include <cmath>
#include <vector>

extern "C" int cms_area_set_descriptors(cms_ctx* c, int b, int n, const uint8_t* desc) {
  if (!c || b < 0 || b >= c->max_batch || n < 0 || n > c->g.kp_cap || (n > 0 && !desc)) return cms_fail(CMS_ERR_ARG, "cms_area_set_descriptors: bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (n > 0) HIPCHK(hipMemcpyAsync(c->d_desc + (size_t)b * c->g.kp_cap * 32, desc, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return CMS_OK;
}

// Which numbers a caller hands over as min_dist / max_dist of a map point: 0 (default) MapPoint::mfMinDistance / mfMaxDistance themselves
// (private members: a binding needs two accessors), 1 the public MapPoint::GetMinDistanceInvariance() / GetMaxDistanceInvariance()
// (MapPoint.cpp:375-385: 0.8f / 1.2f already applied) -- the reference's headers stay byte-identical.  Applies to cms_search_local_points,
// cms_is_in_frustum_device, cms_fuse_search and cms_kfstore_fuse_search of this context.
extern "C" int cms_set_distance_bounds_mode(cms_ctx* c, int scaled) {
  if (!c || scaled < 0 || scaled > 1) return cms_fail(CMS_ERR_ARG, "cms_set_distance_bounds_mode: mode must be 0 or 1");
  c->dist_bounds_scaled = scaled;
  return CMS_OK;
}

extern "C" int cms_is_in_frustum_device(cms_ctx* c, int nmp, const void* d_mp_frame, const void* d_pose15, const void* d_pos, const void* d_normal,
                                        const void* d_min_dist, const void* d_max_dist, float viewing_cos_limit, float th, void* d_in_view,
                                        void* d_proj_x, void* d_proj_y, void* d_level, void* d_view_cos, void* d_qr, void* d_qmin, void* d_qmax) {
  if (!c || nmp < 0 || (nmp > 0 && (!d_pose15 || !d_pos || !d_normal || !d_min_dist || !d_max_dist || !d_in_view || !d_proj_x || !d_proj_y ||
                                    !d_level || !d_view_cos)) || (d_qr && (!d_qmin || !d_qmax)))
    return cms_fail(CMS_ERR_ARG, "cms_is_in_frustum_device: bad argument");
  if (nmp == 0) return CMS_OK;
  if (c->g.nlevels > 16) return cms_fail(CMS_ERR_UNSUPPORTED, "cms_is_in_frustum_device: more than 16 pyramid levels");
  HIPCHK(hipSetDevice(c->device));
  CmsFrustumArgs a;
  a.pose15 = (const float*)d_pose15; a.mp_frame = (const int*)d_mp_frame; a.n = nmp;
  a.P = (const float*)d_pos; a.normal = (const float*)d_normal; a.min_dist = (const float*)d_min_dist; a.max_dist = (const float*)d_max_dist;
  a.viewing_cos_limit = viewing_cos_limit; a.th = th; a.bounds_scaled = c->dist_bounds_scaled;
  a.log_scale = std::log(c->g.nlevels > 1 ? c->scale[1] : 1.2f);          // mfLogScaleFactor = log(mfScaleFactor), float (Frame.cpp:113)
  a.nlevels = c->g.nlevels; a.F = c->g.F;
  for (int l = 0; l < 16; ++l) a.sf[l] = l < c->g.nlevels ? c->scale[l] : 0.0f;
  a.in_view = (uint8_t*)d_in_view; a.proj_x = (float*)d_proj_x; a.proj_y = (float*)d_proj_y; a.level = (int*)d_level; a.view_cos = (float*)d_view_cos;
  a.qr = (float*)d_qr; a.qmin = (int*)d_qmin; a.qmax = (int*)d_qmax;
  hipLaunchKernelGGL(k_in_frustum, dim3((nmp + 255) / 256), dim3(256), 0, c->stream, a);
  HIPCHK(hipGetLastError());
  return CMS_OK;
}

extern "C" int cms_search_local_points_device(cms_ctx* c, int B, const void* d_mp_off, const void* d_mp_desc, const void* d_cand_off,
                                              const void* d_cand_idx, void* d_pair_dist, float nnratio, int th_high, void* d_kp_mp,
                                              void* d_mp_match, void* d_rounds) {
  if (!c || B < 1 || B > c->area_frames || !d_mp_off || !d_mp_desc || !d_cand_off || !d_cand_idx || !d_pair_dist || !d_kp_mp || !d_mp_match)
    return cms_fail(CMS_ERR_ARG, "cms_search_local_points_device: bad argument (cms_area_grid first)");
  if (c->g.kp_cap > CMS_TRACK_KPMAX) return cms_fail(CMS_ERR_UNSUPPORTED, "cms_search_local_points_device: more than 4096 key points per frame");
  HIPCHK(hipSetDevice(c->device));
  CmsSearchLocalArgs a;
  a.mp_off = (const int*)d_mp_off; a.mp_desc = (const uint4*)d_mp_desc; a.cand_off = (const int*)d_cand_off; a.cand_idx = (const int*)d_cand_idx;
  a.t_desc = (const uint4*)c->d_desc; a.kp = (const CmsKeyPoint*)c->d_kps; a.kp_cap = c->g.kp_cap;
  a.pair_dist = (uint16_t*)d_pair_dist; a.kp_mp = (int*)d_kp_mp; a.mp_match = (int*)d_mp_match; a.rounds = (int*)d_rounds;
  a.nnratio = nnratio; a.th_high = th_high; a.frame0 = 0; a.total = nullptr; a.cap = 0;
  hipLaunchKernelGGL(k_search_local, dim3(B), dim3(1024), 0, c->stream, a);
  HIPCHK(hipGetLastError());
  return CMS_OK;
}

// One frame, host buffers: isInFrustum for the n map points, window query against frame b's grid, greedy search.  kp_mp: one int per
// key point of frame b, in/out (>= 0 on entry = key point already holds a map point with observations; matches are written as the
// index of the map point in this list).
extern "C" int cms_search_local_points(cms_ctx* c, int b, const float* pose15, int nmp, const float* pos, const float* normal,
                                       const float* min_dist, const float* max_dist, const uint8_t* mp_desc, float viewing_cos_limit, float th,
                                       float nnratio, int th_high, int nkp, int* kp_mp, uint8_t* in_view, float* proj_x, float* proj_y,
                                       int* level, float* view_cos, int* mp_match, int* n_matches, int* rounds) {
  if (!c || !pose15 || nmp < 0 || nkp < 0 || nkp > (c ? c->g.kp_cap : 0) || (nmp > 0 && (!pos || !normal || !min_dist || !max_dist || !mp_desc || !mp_match)) ||
      (nkp > 0 && !kp_mp))
    return cms_fail(CMS_ERR_ARG, "cms_search_local_points: bad argument");
  if (b < 0 || b >= c->area_frames) return cms_fail(CMS_ERR_ARG, "cms_search_local_points: no grid for this frame (cms_area_grid first)");
  if (c->g.kp_cap > CMS_TRACK_KPMAX) return cms_fail(CMS_ERR_UNSUPPORTED, "cms_search_local_points: more than 4096 key points per frame");
  if (n_matches) *n_matches = 0;
  if (rounds) *rounds = 0;
  if (nmp == 0) return CMS_OK;
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t n4 = (size_t)nmp * 4, kp4 = (size_t)c->g.kp_cap * 4;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  // one block in (pose .. mp_off), one block out (kp_mp .. rounds): one pinned copy each way, no synchronisation in between
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
  const size_t o_pose = take(64), o_pos = take(3 * n4), o_nrm = take(3 * n4), o_min = take(n4), o_max = take(n4), o_desc = take((size_t)nmp * 32),
               o_qf = take(n4), o_mpoff = take(16), o_kpmp = take(kp4);
  const size_t in_bytes = o;
  const size_t o_vis = take(nmp), o_px = take(n4), o_py = take(n4), o_lvl = take(n4), o_vc = take(n4), o_match = take(n4), o_tot = take(16), o_rounds = take(16);
  const size_t out_begin = o_kpmp, out_bytes = o - o_kpmp;
  const size_t o_qr = take(n4), o_qmin = take(n4), o_qmax = take(n4), o_cnt = take(n4), o_off = take(n4 + 4);
  const size_t fixed = o;
  int cap = 64 * nmp + 1024;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const size_t o_idx = fixed, o_pd = fixed + al((size_t)cap * 4);
    int rc = cms_scratch(c, o_pd + al((size_t)cap * 2));
    if (rc) return rc;
    rc = cms_hstage(c, std::max(in_bytes, out_begin + out_bytes));      // the read-back lands at h + out_begin, not at h
    if (rc) return rc;
    uint8_t* p = (uint8_t*)c->d_match;
    uint8_t* h = c->h_stage;
    memcpy(h + o_pose, pose15, 60);
    memcpy(h + o_pos, pos, 3 * n4); memcpy(h + o_nrm, normal, 3 * n4); memcpy(h + o_min, min_dist, n4); memcpy(h + o_max, max_dist, n4);
    memcpy(h + o_desc, mp_desc, (size_t)nmp * 32);
    { int* qf = reinterpret_cast<int*>(h + o_qf); for (int i = 0; i < nmp; ++i) qf[i] = b; }
    { int* mo = reinterpret_cast<int*>(h + o_mpoff); mo[0] = 0; mo[1] = nmp; }
    { int* km = reinterpret_cast<int*>(h + o_kpmp); for (int k = 0; k < c->g.kp_cap; ++k) km[k] = k < nkp ? kp_mp[k] : -1; }
    HIPCHK(hipMemcpyAsync(p, h, in_bytes, hipMemcpyHostToDevice, s));
    rc = cms_is_in_frustum_device(c, nmp, nullptr, p + o_pose, p + o_pos, p + o_nrm, p + o_min, p + o_max, viewing_cos_limit, th, p + o_vis,
                                  p + o_px, p + o_py, p + o_lvl, p + o_vc, p + o_qr, p + o_qmin, p + o_qmax);
    if (rc) return rc;
    rc = cms_features_in_area_batch_device(c, nmp, p + o_qf, p + o_px, p + o_py, p + o_qr, p + o_qmin, p + o_qmax, p + o_cnt, p + o_off, p + o_idx,
                                           cap, p + o_tot);
    if (rc) return rc;
    CmsSearchLocalArgs a;
    a.mp_off = (const int*)(p + o_mpoff); a.mp_desc = (const uint4*)(p + o_desc); a.cand_off = (const int*)(p + o_off); a.cand_idx = (const int*)(p + o_idx);
    a.t_desc = (const uint4*)c->d_desc; a.kp = (const CmsKeyPoint*)c->d_kps; a.kp_cap = c->g.kp_cap;
    a.pair_dist = (uint16_t*)(p + o_pd);
    a.kp_mp = (int*)(p + o_kpmp) - (size_t)b * c->g.kp_cap;          // indexed by batch row: only frame b's rows are ever touched
    a.mp_match = (int*)(p + o_match); a.rounds = (int*)(p + o_rounds);
    a.nnratio = nnratio; a.th_high = th_high; a.frame0 = b; a.total = (const int*)(p + o_tot); a.cap = cap;
    // (window lists longer than `cap` are cut by the query kernel; the search kernel then sees *total > cap and does nothing)
    hipLaunchKernelGGL(k_search_local, dim3(1), dim3(1024), 0, s, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h + out_begin, p + out_begin, out_bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const int tot = *reinterpret_cast<const int*>(h + o_tot);
    if (tot > cap) { cap = tot + 64; continue; }               // denser than 64 candidates per window: once more with the exact size
    const int* match = reinterpret_cast<const int*>(h + o_match);
    if (nkp > 0) memcpy(kp_mp, h + o_kpmp, (size_t)nkp * 4);
    if (in_view) memcpy(in_view, h + o_vis, nmp);
    if (proj_x) memcpy(proj_x, h + o_px, n4);
    if (proj_y) memcpy(proj_y, h + o_py, n4);
    if (level) memcpy(level, h + o_lvl, n4);
    if (view_cos) memcpy(view_cos, h + o_vc, n4);
    int nm = 0;
    for (int i = 0; i < nmp; ++i) {
      const int m = match[i];
      mp_match[i] = m >= 0 ? m - b * c->g.kp_cap : -1;         // batch row -> key point index of frame b
      nm += m >= 0;
    }
    if (n_matches) *n_matches = nm;
    if (rounds) *rounds = *reinterpret_cast<const int*>(h + o_rounds);
    return CMS_OK;
  }
  return cms_fail(CMS_ERR_OVERFLOW, "cms_search_local_points: candidate lists kept growing");
}

// ---- ORBMatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono): device-pointer pieces (asynchronous on the ctx stream) ...
extern "C" int cms_project_last_frame_device(cms_ctx* c, int n, const void* d_qframe, const void* d_pose12, const void* d_valid, const void* d_Xw,
                                             const void* d_oct, float th, void* d_qx, void* d_qy, void* d_qr, void* d_qmin, void* d_qmax) {
  if (!c || n < 0 || (n > 0 && (!d_pose12 || !d_valid || !d_Xw || !d_oct || !d_qx || !d_qy || !d_qr || !d_qmin || !d_qmax)))
    return cms_fail(CMS_ERR_ARG, "cms_project_last_frame_device: bad argument");
  if (n == 0) return CMS_OK;
  HIPCHK(hipSetDevice(c->device));
  CmsProjectLastArgs a;
  a.pose12 = (const float*)d_pose12; a.q_frame = (const int*)d_qframe; a.n = n; a.valid = (const uint8_t*)d_valid; a.Xw = (const float*)d_Xw;
  a.oct = (const int*)d_oct; a.th = th; a.F = c->g.F;
  {
    const float fov = (float)c->cam.fov_deg;
    const float pif = 3.1415926535897932384626f;
    a.cos_fov = std::cos(fov / 2 * (pif / 180));                      // CamModelGeneral::SetCosFovTh
  }
  for (int l = 0; l < 16; ++l) a.sf[l] = l < c->g.nlevels ? c->scale[l] : 0.0f;
  a.qx = (float*)d_qx; a.qy = (float*)d_qy; a.qr = (float*)d_qr; a.qmin = (int*)d_qmin; a.qmax = (int*)d_qmax;
  hipLaunchKernelGGL(k_project_last, dim3((n + 255) / 256), dim3(256), 0, c->stream, a);
  HIPCHK(hipGetLastError());
  return CMS_OK;
}
extern "C" int cms_rotation_filter_device(cms_ctx* c, int B, const void* d_mp_off, const void* d_last_angle, void* d_kp_mp, void* d_mp_match,
                                          void* d_n_matches, int check_orientation) {
  if (!c || B < 1 || !d_mp_off || !d_last_angle || !d_kp_mp || !d_mp_match) return cms_fail(CMS_ERR_ARG, "cms_rotation_filter_device: bad argument");
  HIPCHK(hipSetDevice(c->device));
  CmsRotFilterArgs a;
  a.mp_off = (const int*)d_mp_off; a.last_angle = (const float*)d_last_angle; a.kp = (const CmsKeyPoint*)c->d_kps; a.kp_mp = (int*)d_kp_mp;
  a.mp_match = (int*)d_mp_match; a.n_matches = (int*)d_n_matches; a.check_orientation = check_orientation; a.total = nullptr; a.cap = 0;
  hipLaunchKernelGGL(k_rot_filter, dim3(B), dim3(1024), 0, c->stream, a);
  HIPCHK(hipGetLastError());
  return CMS_OK;
}

// ... and the one-frame entry with host buffers: frame b's key points / descriptors are the current frame (cms_area_grid first).
// pose12 = Rcw | tcw of CurrentFrame.mTcw.  Per key point i of the last frame: valid[i] (holds a map point, not an outlier), Xw, octave,
// angle, mp_desc (MapPoint::GetDescriptor).  kp_mp in/out like cms_search_local_points; match[i] = key point of the current frame or -1.
extern "C" int cms_search_by_projection(cms_ctx* c, int b, const float* pose12, int nlast, const uint8_t* valid, const float* Xw, const int* octave,
                                        const float* angle, const uint8_t* mp_desc, float th, int check_orientation, int th_high, int nkp, int* kp_mp,
                                        int* match, int* n_matches) {
  if (!c || !pose12 || nlast < 0 || nkp < 0 || nkp > (c ? c->g.kp_cap : 0) || (nlast > 0 && (!valid || !Xw || !octave || !angle || !mp_desc || !match)) ||
      (nkp > 0 && !kp_mp))
    return cms_fail(CMS_ERR_ARG, "cms_search_by_projection: bad argument");
  if (b < 0 || b >= c->area_frames) return cms_fail(CMS_ERR_ARG, "cms_search_by_projection: no grid for this frame (cms_area_grid first)");
  if (c->g.kp_cap > CMS_TRACK_KPMAX) return cms_fail(CMS_ERR_UNSUPPORTED, "cms_search_by_projection: more than 4096 key points per frame");
  if (n_matches) *n_matches = 0;
  if (nlast == 0) return CMS_OK;
  for (int i = 0; i < nlast; ++i)
    if (valid[i] && (octave[i] < 0 || octave[i] >= c->g.nlevels)) return cms_fail(CMS_ERR_ARG, "cms_search_by_projection: octave out of range");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t n4 = (size_t)nlast * 4, kp4 = (size_t)c->g.kp_cap * 4;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
  const size_t o_pose = take(64), o_valid = take(nlast), o_xw = take(3 * n4), o_oct = take(n4), o_ang = take(n4), o_desc = take((size_t)nlast * 32),
               o_qf = take(n4), o_mpoff = take(16), o_kpmp = take(kp4);
  const size_t in_bytes = o;
  const size_t o_match = take(n4), o_nm = take(16), o_tot = take(16);
  const size_t out_begin = o_kpmp, out_bytes = o - o_kpmp;
  const size_t o_qx = take(n4), o_qy = take(n4), o_qr = take(n4), o_qmin = take(n4), o_qmax = take(n4), o_cnt = take(n4), o_off = take(n4 + 4);
  const size_t fixed = o;
  int cap = 64 * nlast + 1024;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const size_t o_idx = fixed, o_pd = fixed + al((size_t)cap * 4);
    int rc = cms_scratch(c, o_pd + al((size_t)cap * 2));
    if (rc) return rc;
    rc = cms_hstage(c, std::max(in_bytes, out_begin + out_bytes));      // the read-back lands at h + out_begin, not at h
    if (rc) return rc;
    uint8_t* p = (uint8_t*)c->d_match;
    uint8_t* h = c->h_stage;
    memcpy(h + o_pose, pose12, 48);
    memcpy(h + o_valid, valid, nlast); memcpy(h + o_xw, Xw, 3 * n4); memcpy(h + o_oct, octave, n4); memcpy(h + o_ang, angle, n4);
    memcpy(h + o_desc, mp_desc, (size_t)nlast * 32);
    { int* qf = reinterpret_cast<int*>(h + o_qf); for (int i = 0; i < nlast; ++i) qf[i] = b; }
    { int* mo = reinterpret_cast<int*>(h + o_mpoff); mo[0] = 0; mo[1] = nlast; }
    { int* km = reinterpret_cast<int*>(h + o_kpmp); for (int k = 0; k < c->g.kp_cap; ++k) km[k] = k < nkp ? kp_mp[k] : -1; }
    HIPCHK(hipMemcpyAsync(p, h, in_bytes, hipMemcpyHostToDevice, s));
    rc = cms_project_last_frame_device(c, nlast, nullptr, p + o_pose, p + o_valid, p + o_xw, p + o_oct, th, p + o_qx, p + o_qy, p + o_qr, p + o_qmin, p + o_qmax);
    if (rc) return rc;
    rc = cms_features_in_area_batch_device(c, nlast, p + o_qf, p + o_qx, p + o_qy, p + o_qr, p + o_qmin, p + o_qmax, p + o_cnt, p + o_off, p + o_idx, cap, p + o_tot);
    if (rc) return rc;
    CmsSearchLocalArgs a;
    a.mp_off = (const int*)(p + o_mpoff); a.mp_desc = (const uint4*)(p + o_desc); a.cand_off = (const int*)(p + o_off); a.cand_idx = (const int*)(p + o_idx);
    a.t_desc = (const uint4*)c->d_desc; a.kp = (const CmsKeyPoint*)c->d_kps; a.kp_cap = c->g.kp_cap;
    a.pair_dist = (uint16_t*)(p + o_pd); a.kp_mp = (int*)(p + o_kpmp) - (size_t)b * c->g.kp_cap; a.mp_match = (int*)(p + o_match); a.rounds = nullptr;
    a.nnratio = -1.0f; a.th_high = th_high; a.frame0 = b; a.total = (const int*)(p + o_tot); a.cap = cap;
    hipLaunchKernelGGL(k_search_local, dim3(1), dim3(1024), 0, s, a);
    CmsRotFilterArgs r;
    r.mp_off = (const int*)(p + o_mpoff); r.last_angle = (const float*)(p + o_ang); r.kp = (const CmsKeyPoint*)c->d_kps; r.kp_mp = a.kp_mp;
    r.mp_match = (int*)(p + o_match); r.n_matches = (int*)(p + o_nm); r.check_orientation = check_orientation; r.total = a.total; r.cap = cap;
    hipLaunchKernelGGL(k_rot_filter, dim3(1), dim3(1024), 0, s, r);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h + out_begin, p + out_begin, out_bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const int tot = *reinterpret_cast<const int*>(h + o_tot);
    if (tot > cap) { cap = tot + 64; continue; }
    const int* m = reinterpret_cast<const int*>(h + o_match);
    if (nkp > 0) memcpy(kp_mp, h + o_kpmp, (size_t)nkp * 4);
    for (int i = 0; i < nlast; ++i) match[i] = m[i] >= 0 ? m[i] - b * c->g.kp_cap : -1;
    if (n_matches) *n_matches = *reinterpret_cast<const int*>(h + o_nm);
    return CMS_OK;
  }
  return cms_fail(CMS_ERR_OVERFLOW, "cms_search_by_projection: candidate lists kept growing");
}

// ORBMatcher::SearchForInitialization(Frame& F1, Frame& F2, vbPrevMatched, vnMatches12, windowSize) (src/ORBMatcher.cpp:676-794): F2 is
// frame slot b2 (key points / descriptors on the device, cms_area_grid first), F1 comes from the caller.  prev_matched: n1 x 2 floats,
// in/out (vbPrevMatched); matches12[i1] = key point of F2 or -1.
extern "C" int cms_search_for_initialization(cms_ctx* c, int b2, int n1, const cms_keypoint* kps1, const uint8_t* desc1, float* prev_matched,
                                             int window_size, float nnratio, int check_orientation, int* matches12, int* n_matches) {
  if (!c || n1 < 0 || (n1 > 0 && (!kps1 || !desc1 || !prev_matched || !matches12)) || window_size < 0)
    return cms_fail(CMS_ERR_ARG, "cms_search_for_initialization: bad argument");
  if (b2 < 0 || b2 >= c->area_frames) return cms_fail(CMS_ERR_ARG, "cms_search_for_initialization: no grid for this frame (cms_area_grid first)");
  if (n_matches) *n_matches = 0;
  if (n1 == 0) return CMS_OK;
  HIPCHK(hipSetDevice(c->device));
  {   // the attribute is per device: one flag per device, like ba_lds_attrs_once / cms_area_reserve
    static std::mutex mu;
    static bool done[64] = {false};
    std::lock_guard<std::mutex> lk(mu);
    if (c->device >= 0 && c->device < 64 && !done[c->device]) {
      HIPCHK(hipFuncSetAttribute((const void*)k_init_greedy, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
      done[c->device] = true;
    }
  }
  const size_t lds = (size_t)c->g.kp_cap * 8;
  if (lds > 160 * 1024 - 1024) return cms_fail(CMS_ERR_UNSUPPORTED, "cms_search_for_initialization: too many key points per frame for the LDS tables");
  hipStream_t s = c->stream;
  std::vector<int> qi;
  for (int i = 0; i < n1; ++i) if (kps1[i].octave <= 0) qi.push_back(i);      // only level 0 (:693-696)
  const int nq = (int)qi.size();
  for (int i = 0; i < n1; ++i) matches12[i] = -1;
  if (nq == 0) return CMS_OK;
  const size_t q4 = (size_t)nq * 4, n4 = (size_t)n1 * 4;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
  const size_t o_qi = take(q4), o_qx = take(q4), o_qy = take(q4), o_qr = take(q4), o_qmin = take(q4), o_qmax = take(q4), o_qf = take(q4),
               o_desc = take((size_t)n1 * 32), o_ang = take(n4), o_prev = take(2 * n4);
  const size_t in_bytes = o;
  const size_t o_m12 = take(n4), o_nm = take(16), o_tot = take(16);
  const size_t out_begin = o_prev, out_bytes = o - o_prev;
  const size_t o_bin = take(n1), o_cnt = take(q4), o_off = take(q4 + 4);
  const size_t fixed = o;
  int cap = 128 * nq + 4096;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const size_t o_idx = fixed, o_pd = fixed + al((size_t)cap * 4);
    int rc = cms_scratch(c, o_pd + al((size_t)cap * 2));
    if (rc) return rc;
    rc = cms_hstage(c, std::max(in_bytes, out_begin + out_bytes));      // the read-back lands at h + out_begin, not at h
    if (rc) return rc;
    uint8_t* p = (uint8_t*)c->d_match;
    uint8_t* h = c->h_stage;
    {
      int* hq = reinterpret_cast<int*>(h + o_qi); float* hx = reinterpret_cast<float*>(h + o_qx); float* hy = reinterpret_cast<float*>(h + o_qy);
      float* hr = reinterpret_cast<float*>(h + o_qr); int* hmin = reinterpret_cast<int*>(h + o_qmin); int* hmax = reinterpret_cast<int*>(h + o_qmax);
      int* hf = reinterpret_cast<int*>(h + o_qf);
      for (int q = 0; q < nq; ++q) {
        hq[q] = qi[q]; hx[q] = prev_matched[2 * qi[q]]; hy[q] = prev_matched[2 * qi[q] + 1]; hr[q] = (float)window_size; hmin[q] = 0; hmax[q] = 0; hf[q] = b2;
      }
      float* ha = reinterpret_cast<float*>(h + o_ang);
      for (int i = 0; i < n1; ++i) ha[i] = kps1[i].angle;
      memcpy(h + o_desc, desc1, (size_t)n1 * 32);
      memcpy(h + o_prev, prev_matched, 2 * n4);
    }
    HIPCHK(hipMemcpyAsync(p, h, in_bytes, hipMemcpyHostToDevice, s));
    rc = cms_features_in_area_batch_device(c, nq, p + o_qf, p + o_qx, p + o_qy, p + o_qr, p + o_qmin, p + o_qmax, p + o_cnt, p + o_off, p + o_idx, cap, p + o_tot);
    if (rc) return rc;
    CmsInitArgs a;
    a.nq = nq; a.q_i1 = (const int*)(p + o_qi); a.cand_off = (const int*)(p + o_off); a.cand_idx = (const int*)(p + o_idx);
    a.desc1 = (const uint4*)(p + o_desc); a.t_desc = (const uint4*)c->d_desc; a.kp2 = (const CmsKeyPoint*)c->d_kps; a.row0 = b2 * c->g.kp_cap; a.kp_cap = c->g.kp_cap;
    a.angle1 = (const float*)(p + o_ang); a.pair_dist = (uint16_t*)(p + o_pd);
    a.n1 = n1; a.matches12 = (int*)(p + o_m12); a.prev_matched = (float*)(p + o_prev); a.n_matches = (int*)(p + o_nm); a.bin_of = (int8_t*)(p + o_bin);
    a.nnratio = nnratio; a.check_orientation = check_orientation; a.total = (const int*)(p + o_tot); a.cap = cap;
    hipLaunchKernelGGL(k_init_dist, dim3(nq), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_init_greedy, dim3(1), dim3(64), lds, s, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h + out_begin, p + out_begin, out_bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const int tot = *reinterpret_cast<const int*>(h + o_tot);
    if (tot > cap) { cap = tot + 64; continue; }
    memcpy(matches12, h + o_m12, n4);
    memcpy(prev_matched, h + o_prev, 2 * n4);
    if (n_matches) *n_matches = *reinterpret_cast<const int*>(h + o_nm);
    return CMS_OK;
  }
  return cms_fail(CMS_ERR_OVERFLOW, "cms_search_for_initialization: candidate lists kept growing");
}
