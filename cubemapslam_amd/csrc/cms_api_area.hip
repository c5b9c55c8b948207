// cms_api_area.hip -- host side of the frame grid + window query (Frame::AssignFeaturesToGrid / GetFeaturesInArea), included by
// cms_lib.hip after cms_api_frames.hip (uses cms_ctx, cms_fail, HIPCHK, cms_scratch).
#include <mutex>
#include <vector>

// the rank sort of k_area_grid keeps 8 bytes per key point in LDS: above the 64 KB default for the 3 x nFeatures extractor of the
// initialisation.  The attribute is per function AND per device; every launcher of the kernel (frame grids, key-frame store) calls this.
static int cms_area_grid_attr(int device) {
  static std::mutex mu;
  static bool done[64] = {false};
  std::lock_guard<std::mutex> lk(mu);
  if (device >= 0 && device < 64 && !done[device]) {
    HIPCHK(hipFuncSetAttribute((const void*)k_area_grid, hipFuncAttributeMaxDynamicSharedMemorySize, (CMS_AREA_MAXKP + 1) * 8));
    done[device] = true;
  }
  return CMS_OK;
}
static int cms_area_reserve(cms_ctx* c) {
  if (c->d_area_sorted) return CMS_OK;
  if (c->g.kp_cap > CMS_AREA_MAXKP) return cms_fail(CMS_ERR_UNSUPPORTED, "frame grid: more than 16383 key points per frame");
  { const int rca = cms_area_grid_attr(c->device); if (rca) return rca; }
  const size_t B = (size_t)c->max_batch;
  HIPCHK(hipMalloc((void**)&c->d_area_sorted, B * c->g.kp_cap * sizeof(uint16_t)));
  HIPCHK(hipMalloc((void**)&c->d_area_cell_start, B * (CMS_AREA_CELLS + 1) * sizeof(int)));
  HIPCHK(hipMalloc((void**)&c->d_area_nvalid, B * sizeof(int)));
  return CMS_OK;
}

// hit buffer of the query kernel's first pass (CMS_AREA_TMP entries per query)
static int cms_area_tmp_reserve(cms_ctx* c, int nq) {
  if ((size_t)nq <= c->area_tmp_cap) return CMS_OK;
  if (c->d_area_tmp) hipFree(c->d_area_tmp);
  c->d_area_tmp = nullptr; c->area_tmp_cap = 0;
  const size_t cap = (size_t)nq + (size_t)nq / 4 + 1024;
  HIPCHK(hipMalloc((void**)&c->d_area_tmp, cap * CMS_AREA_TMP * sizeof(int)));
  c->area_tmp_cap = cap;
  return CMS_OK;
}
static int cms_area_bsum_reserve(cms_ctx* c, int nblk) {
  if (nblk <= c->area_bsum_cap) return CMS_OK;
  if (c->d_area_bsum) hipFree(c->d_area_bsum);
  c->d_area_bsum = nullptr; c->area_bsum_cap = 0;
  HIPCHK(hipMalloc((void**)&c->d_area_bsum, (size_t)(nblk + 64) * sizeof(int)));
  c->area_bsum_cap = nblk + 64;
  return CMS_OK;
}

extern "C" int cms_area_set_keypoints(cms_ctx* c, int b, int n, const cms_keypoint* kps) {
  if (!c || b < 0 || b >= c->max_batch || n < 0 || n > c->g.kp_cap || (n > 0 && !kps)) return cms_fail(CMS_ERR_ARG, "cms_area_set_keypoints: bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (n > 0) HIPCHK(hipMemcpyAsync(c->d_kps + (size_t)b * c->g.kp_cap, kps, (size_t)n * sizeof(cms_keypoint), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_kp_cnt + b, &n, sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return CMS_OK;
}

extern "C" int cms_area_grid(cms_ctx* c, int B) {
  if (!c || B < 1 || B > c->max_batch) return cms_fail(CMS_ERR_ARG, "cms_area_grid: bad batch");
  HIPCHK(hipSetDevice(c->device));
  int rc = cms_area_reserve(c);
  if (rc) return rc;
  const float inv = (float)(3 * CMS_AREA_G) / (float)c->g.W;          // mfGridElementLengthInv (Frame.cpp:149)
  hipLaunchKernelGGL(k_area_grid, dim3(B), dim3(1024), (size_t)(c->g.kp_cap + 1) * 8, c->stream, (const CmsKeyPoint*)c->d_kps, (const int*)c->d_kp_cnt, c->g.kp_cap, c->g.F, inv,
                     c->d_area_sorted, c->d_area_cell_start, c->d_area_nvalid);
  HIPCHK(hipGetLastError());
  c->area_frames = B;
  return CMS_OK;
}

extern "C" int cms_features_in_area_device(cms_ctx* c, int b, int nq, const void* d_qx, const void* d_qy, const void* d_qr, const void* d_qmin,
                                           const void* d_qmax, void* d_cnt_scratch, void* d_cand_off, void* d_cand_idx, int cap, int idx_base,
                                           void* d_total) {
  if (!c || b < 0 || b >= c->area_frames || nq < 0 || cap < 0) return cms_fail(CMS_ERR_ARG, "cms_features_in_area_device: bad argument (cms_area_grid first)");
  if (nq == 0) return CMS_OK;
  HIPCHK(hipSetDevice(c->device));
  CmsAreaArgs a;
  a.kp = (const CmsKeyPoint*)c->d_kps + (size_t)b * c->g.kp_cap;
  a.sorted_idx = c->d_area_sorted + (size_t)b * c->g.kp_cap;
  a.cell_start = c->d_area_cell_start + (size_t)b * (CMS_AREA_CELLS + 1);
  a.qx = (const float*)d_qx; a.qy = (const float*)d_qy; a.qr = (const float*)d_qr; a.qmin = (const int*)d_qmin; a.qmax = (const int*)d_qmax;
  a.q_frame = nullptr; a.kp_cap = c->g.kp_cap;
  a.nq = nq; a.F = c->g.F; a.inv = (float)(3 * CMS_AREA_G) / (float)c->g.W;
  a.cnt = (int*)d_cnt_scratch; a.off = (const int*)d_cand_off; a.idx = (int*)d_cand_idx; a.cap = cap; a.idx_base = idx_base;
  hipStream_t s = c->stream;
  {
    const int nblk = (nq + 1023) / 1024, qgrid = (nq * CMS_AREA_QL + 255) / 256;
    int rcb = cms_area_bsum_reserve(c, nblk);
    if (rcb) return rcb;
    rcb = cms_area_tmp_reserve(c, nq);
    if (rcb) return rcb;
    a.tmp = c->d_area_tmp;
    hipLaunchKernelGGL(k_area_query, dim3(qgrid), dim3(256), 0, s, a, 0);
    hipLaunchKernelGGL(k_area_blocksum, dim3(nblk), dim3(1024), 0, s, (const int*)d_cnt_scratch, nq, c->d_area_bsum);
    hipLaunchKernelGGL(k_area_scan, dim3(nblk), dim3(1024), 0, s, (const int*)d_cnt_scratch, nq, (const int*)c->d_area_bsum, (int*)d_cand_off, (int*)d_total);
    hipLaunchKernelGGL(k_area_query, dim3(qgrid), dim3(256), 0, s, a, 1);
  }
  HIPCHK(hipGetLastError());
  return CMS_OK;
}

// every query names the frame of the batch it searches (d_qframe); candidate indices are rows of the batch (frame * kp_cap + i)
extern "C" int cms_features_in_area_batch_device(cms_ctx* c, int nq, const void* d_qframe, const void* d_qx, const void* d_qy, const void* d_qr,
                                                 const void* d_qmin, const void* d_qmax, void* d_cnt_scratch, void* d_cand_off, void* d_cand_idx,
                                                 int cap, void* d_total) {
  if (!c || nq < 0 || cap < 0 || c->area_frames < 1 || !d_qframe) return cms_fail(CMS_ERR_ARG, "cms_features_in_area_batch_device: bad argument (cms_area_grid first)");
  if (nq == 0) return CMS_OK;
  HIPCHK(hipSetDevice(c->device));
  CmsAreaArgs a;
  a.kp = (const CmsKeyPoint*)c->d_kps; a.sorted_idx = c->d_area_sorted; a.cell_start = c->d_area_cell_start;
  a.qx = (const float*)d_qx; a.qy = (const float*)d_qy; a.qr = (const float*)d_qr; a.qmin = (const int*)d_qmin; a.qmax = (const int*)d_qmax;
  a.q_frame = (const int*)d_qframe; a.kp_cap = c->g.kp_cap;
  a.nq = nq; a.F = c->g.F; a.inv = (float)(3 * CMS_AREA_G) / (float)c->g.W;
  a.cnt = (int*)d_cnt_scratch; a.off = (const int*)d_cand_off; a.idx = (int*)d_cand_idx; a.cap = cap; a.idx_base = 0;
  hipStream_t s = c->stream;
  {
    const int nblk = (nq + 1023) / 1024, qgrid = (nq * CMS_AREA_QL + 255) / 256;
    int rcb = cms_area_bsum_reserve(c, nblk);
    if (rcb) return rcb;
    rcb = cms_area_tmp_reserve(c, nq);
    if (rcb) return rcb;
    a.tmp = c->d_area_tmp;
    hipLaunchKernelGGL(k_area_query, dim3(qgrid), dim3(256), 0, s, a, 0);
    hipLaunchKernelGGL(k_area_blocksum, dim3(nblk), dim3(1024), 0, s, (const int*)d_cnt_scratch, nq, c->d_area_bsum);
    hipLaunchKernelGGL(k_area_scan, dim3(nblk), dim3(1024), 0, s, (const int*)d_cnt_scratch, nq, (const int*)c->d_area_bsum, (int*)d_cand_off, (int*)d_total);
    hipLaunchKernelGGL(k_area_query, dim3(qgrid), dim3(256), 0, s, a, 1);
  }
  HIPCHK(hipGetLastError());
  return CMS_OK;
}

extern "C" int cms_features_in_area(cms_ctx* c, int b, int nq, const float* qx, const float* qy, const float* qr, const int* qmin, const int* qmax,
                                    int* cand_off, int* cand_idx, int cap, int* total) {
  if (!c || nq < 0 || cap < 0 || !cand_off || (nq > 0 && (!qx || !qy || !qr || !qmin || !qmax))) return cms_fail(CMS_ERR_ARG, "cms_features_in_area: bad argument");
  if (b < 0 || b >= c->area_frames) return cms_fail(CMS_ERR_ARG, "cms_features_in_area: no grid for this frame (cms_area_grid first)");
  cand_off[0] = 0;
  if (total) *total = 0;
  if (nq == 0) return CMS_OK;
  HIPCHK(hipSetDevice(c->device));
  const size_t qb = (size_t)nq * 4;
  const size_t bytes = 5 * qb + qb /*cnt*/ + (qb + 4) /*off*/ + (size_t)cap * 4 + 64;
  int rc = cms_scratch(c, bytes);
  if (rc) return rc;
  uint8_t* p = (uint8_t*)c->d_match;
  hipStream_t s = c->stream;
  void* dq[5];
  const void* hq[5] = {qx, qy, qr, qmin, qmax};
  for (int i = 0; i < 5; ++i) { dq[i] = p; p += qb; HIPCHK(hipMemcpyAsync(dq[i], hq[i], qb, hipMemcpyHostToDevice, s)); }
  void* d_cnt = p; p += qb;
  void* d_off = p; p += qb + 4;
  void* d_tot = p; p += 16;
  void* d_idx = p;
  rc = cms_features_in_area_device(c, b, nq, dq[0], dq[1], dq[2], dq[3], dq[4], d_cnt, d_off, d_idx, cap, 0, d_tot);
  if (rc) return rc;
  int tot = 0;
  HIPCHK(hipMemcpyAsync(cand_off, d_off, qb + 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(&tot, d_tot, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (total) *total = tot;
  if (tot > cap) return cms_fail(CMS_ERR_OVERFLOW, "cms_features_in_area: candidate capacity too small");
  if (tot > 0 && cand_idx) HIPCHK(hipMemcpy(cand_idx, d_idx, (size_t)tot * 4, hipMemcpyDeviceToHost));
  return CMS_OK;
}
