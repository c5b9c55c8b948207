// cms_api_pose.hip -- host side of the pose-only optimisation (Optimizer::PoseOptimization, Optimizer.cpp:48-190).
// Persistent device buffers + one stream per handle; a batch of frames is one launch of k_pose_optimize (cms_pose_opt.hip).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../include/cubemapslam_hip.h"

static inline size_t pose_al(size_t v) { return (v + 255) & ~(size_t)255; }
struct cms_pose {
  int device = 0, cap_f = 0, cap_e = 0, nf = 0, ne = 0, max_n = 0;   // max_n: most edges of one uploaded frame
  hipStream_t stream = nullptr;
  int* d_off = nullptr; double* d_Xw = nullptr; double* d_obs = nullptr; double* d_inv = nullptr; int8_t* d_face = nullptr;
  uint8_t* d_out = nullptr; double* d_err = nullptr; double* d_poses = nullptr; double* d_poses0 = nullptr; int* d_res = nullptr;
  double fx = 0, fy = 0, cx = 0, cy = 0;
  uint8_t* h_stage = nullptr; size_t h_stage_bytes = 0;     // pinned staging of cms_pose_optimize_batch (inputs out, results back: no pageable copies)
  uint8_t* h_fetch = nullptr; size_t h_fetch_bytes = 0;     // pinned landing block of cms_pose_fetch
  hipEvent_t ev_fetch = nullptr; bool fetch_queued = false; // the results' copies into h_fetch were enqueued by cms_pose_launch right behind the kernel (ev_fetch: their end)
  uint8_t* h_direct = nullptr; size_t h_direct_bytes = 0;   // pinned block of the direct (few frames) path: its own, so that a staged upload still copying out of h_stage is never overwritten
};

static void cms_pose_free(cms_pose* p) {
  if (!p) return;
  hipSetDevice(p->device);
  void* ptrs[] = {p->d_off, p->d_Xw, p->d_obs, p->d_inv, p->d_face, p->d_out, p->d_err, p->d_poses, p->d_poses0, p->d_res};
  for (void* q : ptrs) if (q) hipFree(q);
  if (p->h_stage) (void)hipHostFree(p->h_stage);
  if (p->h_fetch) (void)hipHostFree(p->h_fetch);
  if (p->h_direct) (void)hipHostFree(p->h_direct);
  if (p->ev_fetch) (void)hipEventDestroy(p->ev_fetch);
  if (p->stream) hipStreamDestroy(p->stream);
  delete p;
}

extern "C" int cms_pose_create(cms_pose** out, int device, int max_frames, int max_edges) {
  if (!out || max_frames < 1 || max_edges < 1) return cms_fail(CMS_ERR_ARG, "cms_pose_create: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev)
    return cms_fail(CMS_ERR_NO_DEVICE, "cms_pose_create: no HIP device (the pose optimisation has no CPU fallback)");
  HIPCHK(hipSetDevice(device));
  cms_pose* p = new cms_pose();
  p->device = device; p->cap_f = max_frames; p->cap_e = max_edges;
#define PALLOC(ptr, bytes) do { if (hipMalloc((void**)&(ptr), (bytes)) != hipSuccess) { cms_pose_free(p); return cms_fail(CMS_ERR_HIP, "hipMalloc " #ptr); } } while (0)
  {
    uint32_t cu_mask[8]; int cu_words = 0;      // (developer A/B, see cms_cu_mask_from_env)
    const hipError_t se = cms_cu_mask_from_env(cu_mask, &cu_words) ? hipExtStreamCreateWithCUMask(&p->stream, (uint32_t)cu_words, cu_mask)
                                                                   : hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (se != hipSuccess) { cms_pose_free(p); return cms_fail(CMS_ERR_HIP, "hipStreamCreate"); }
  }
  PALLOC(p->d_off, ((size_t)max_frames + 1) * sizeof(int));
  PALLOC(p->d_Xw, (size_t)max_edges * 3 * sizeof(double)); PALLOC(p->d_obs, (size_t)max_edges * 2 * sizeof(double));
  PALLOC(p->d_inv, (size_t)max_edges * sizeof(double)); PALLOC(p->d_face, (size_t)max_edges);
  PALLOC(p->d_out, (size_t)max_edges); PALLOC(p->d_err, (size_t)max_edges * 2 * sizeof(double));
  PALLOC(p->d_poses, (size_t)max_frames * 7 * sizeof(double)); PALLOC(p->d_poses0, (size_t)max_frames * 7 * sizeof(double));
  PALLOC(p->d_res, (size_t)max_frames * 8 * sizeof(int));
#undef PALLOC
  *out = p;
  return CMS_OK;
}
extern "C" void cms_pose_destroy(cms_pose* p) { cms_pose_free(p); }
extern "C" void* cms_pose_stream(cms_pose* p) { return p ? (void*)p->stream : nullptr; }

// staged = false: the caller's arrays go as they are (pageable memory: every copy is staged by the runtime and waited for).
// staged = true (cms_pose_optimize_batch): they are packed into the handle's pinned block first, the copies leave from there without a wait --
// the host-buffer entry then costs one synchronisation in all instead of nine staged copies (~100 us of a 290 us call for one frame).
static int cms_pose_upload_impl(cms_pose* p, int nf, const int* edge_off, const double* Xw, const double* obs_uv, const double* inv_sigma2,
                                const int8_t* face, double fx, double fy, double cx, double cy, const double* poses7, bool staged) {
  if (!p || nf < 1 || nf > p->cap_f || !edge_off || !poses7) return cms_fail(CMS_ERR_ARG, "cms_pose_upload: bad argument");
  const int ne = edge_off[nf];
  if (edge_off[0] != 0 || ne < 0 || ne > p->cap_e) return cms_fail(CMS_ERR_ARG, "cms_pose_upload: edge count exceeds the handle's capacity");
  for (int f = 0; f < nf; ++f) if (edge_off[f + 1] < edge_off[f]) return cms_fail(CMS_ERR_ARG, "cms_pose_upload: edge_off must not decrease");
  if (ne > 0 && (!Xw || !obs_uv || !inv_sigma2 || !face)) return cms_fail(CMS_ERR_ARG, "cms_pose_upload: null edge array");
  for (int e = 0; e < ne; ++e) if (face[e] < 0 || face[e] > 4) return cms_fail(CMS_ERR_ARG, "cms_pose_upload: edge on an unknown face");   // the reference exits
  HIPCHK(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  const int* h_off = edge_off; const double* h_X = Xw; const double* h_obs = obs_uv; const double* h_inv = inv_sigma2; const int8_t* h_face = face;
  const double* h_pose = poses7;
  if (staged) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_off = 0, o_X = al(((size_t)nf + 1) * 4), o_obs = o_X + al((size_t)ne * 24), o_inv = o_obs + al((size_t)ne * 16), o_face = o_inv + al((size_t)ne * 8),
                 o_pose = o_face + al((size_t)ne), in_bytes = o_pose + al((size_t)nf * 56);
    const size_t out_bytes = al((size_t)nf * 32) + al((size_t)nf * 56) + al((size_t)ne);
    // the inputs use the first half of the block, the results of cms_pose_optimize_batch the second: EACH must fit its half (a block kept from
    // a smaller call could hold in + out together and still be too short for a result set larger than the inputs)
    if (2 * std::max(in_bytes, out_bytes) > p->h_stage_bytes) {
      if (p->h_stage) (void)hipHostFree(p->h_stage);
      p->h_stage = nullptr; p->h_stage_bytes = 0;
      const size_t want = 4 * std::max(in_bytes, out_bytes);
      HIPCHK(hipHostMalloc((void**)&p->h_stage, want));
      p->h_stage_bytes = want;
    }
    uint8_t* h = p->h_stage;
    memcpy(h + o_off, edge_off, ((size_t)nf + 1) * 4);
    if (ne > 0) { memcpy(h + o_X, Xw, (size_t)ne * 24); memcpy(h + o_obs, obs_uv, (size_t)ne * 16); memcpy(h + o_inv, inv_sigma2, (size_t)ne * 8); memcpy(h + o_face, face, (size_t)ne); }
    memcpy(h + o_pose, poses7, (size_t)nf * 56);
    h_off = (const int*)(h + o_off); h_X = (const double*)(h + o_X); h_obs = (const double*)(h + o_obs); h_inv = (const double*)(h + o_inv);
    h_face = (const int8_t*)(h + o_face); h_pose = (const double*)(h + o_pose);
  }
  HIPCHK(hipMemcpyAsync(p->d_off, h_off, ((size_t)nf + 1) * sizeof(int), hipMemcpyHostToDevice, s));
  if (ne > 0) {
    HIPCHK(hipMemcpyAsync(p->d_Xw, h_X, (size_t)ne * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(p->d_obs, h_obs, (size_t)ne * 2 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(p->d_inv, h_inv, (size_t)ne * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(p->d_face, h_face, (size_t)ne, hipMemcpyHostToDevice, s));
  }
  HIPCHK(hipMemcpyAsync(p->d_poses0, h_pose, (size_t)nf * 7 * sizeof(double), hipMemcpyHostToDevice, s));
  if (!staged) HIPCHK(hipStreamSynchronize(s));       // the caller's arrays may be temporaries
  p->nf = nf; p->ne = ne; p->fx = fx; p->fy = fy; p->cx = cx; p->cy = cy;
  p->fetch_queued = false;                                         // (the landing block holds an earlier batch's results)
  p->max_n = 0;
  for (int f = 0; f < nf; ++f) p->max_n = std::max(p->max_n, edge_off[f + 1] - edge_off[f]);
  return CMS_OK;
}
extern "C" int cms_pose_upload(cms_pose* p, int nf, const int* edge_off, const double* Xw, const double* obs_uv, const double* inv_sigma2,
                               const int8_t* face, double fx, double fy, double cx, double cy, const double* poses7) {
  return cms_pose_upload_impl(p, nf, edge_off, Xw, obs_uv, inv_sigma2, face, fx, fy, cx, cy, poses7, false);
}
extern "C" int cms_pose_launch(cms_pose* p) {
  if (!p || p->nf < 1) return cms_fail(CMS_ERR_ARG, "cms_pose_launch: nothing uploaded");
  HIPCHK(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  HIPCHK(hipMemcpyAsync(p->d_poses, p->d_poses0, (size_t)p->nf * 7 * sizeof(double), hipMemcpyDeviceToDevice, s));
  PoseDev d;
  d.nf = p->nf; d.off = p->d_off; d.Xw = p->d_Xw; d.obs = p->d_obs; d.inv = p->d_inv; d.face = p->d_face; d.outlier = p->d_out;
  d.err = p->d_err; d.poses = p->d_poses; d.result = p->d_res; d.fx = p->fx; d.fy = p->fy; d.cx = p->cx; d.cy = p->cy;
  // edges of a frame in registers when they fit (256 threads x PO_MAXJ edges), otherwise the variant that walks them in memory
  static const bool force_global = getenv("CMS_POSE_GLOBAL") != nullptr;      // developer knob, read once per process
  if (p->max_n <= 256 * PO_MAXJ && !force_global) hipLaunchKernelGGL(k_pose_optimize, dim3(p->nf), dim3(256), 0, s, d);
  else hipLaunchKernelGGL(k_pose_optimize_g, dim3(p->nf), dim3(256), 0, s, d);
  HIPCHK(hipGetLastError());
  // The results start for the handle's pinned block right behind the kernel: cms_pose_fetch then only waits for an event that is long through when a
  // caller launches early and fetches late (bench.py: launch at the top of a step, fetch at its end).  Enqueued by the fetch itself, the three copies
  // each waited for a slot on the busy chip -- 3.5 ms of the step's host thread inside a 12.5 ms step (CMS_BENCH_STEP_TRACE).
  p->fetch_queued = false;
  {
    const size_t o_res = 0, o_pose = pose_al((size_t)p->nf * 32), o_out = o_pose + pose_al((size_t)p->nf * 56), total = o_out + pose_al((size_t)std::max(p->ne, 1));
    if (total > p->h_fetch_bytes) {
      if (p->h_fetch) { HIPCHK(hipStreamSynchronize(s)); (void)hipHostFree(p->h_fetch); }
      p->h_fetch = nullptr; p->h_fetch_bytes = 0;
      HIPCHK(hipHostMalloc((void**)&p->h_fetch, 2 * total));
      p->h_fetch_bytes = 2 * total;
    }
    if (!p->ev_fetch) HIPCHK(hipEventCreateWithFlags(&p->ev_fetch, hipEventDisableTiming));
    uint8_t* h = p->h_fetch;
    HIPCHK(hipMemcpyAsync(h + o_res, p->d_res, (size_t)p->nf * 8 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h + o_pose, p->d_poses, (size_t)p->nf * 7 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (p->ne > 0) HIPCHK(hipMemcpyAsync(h + o_out, p->d_out, (size_t)p->ne, hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(p->ev_fetch, s));
    p->fetch_queued = true;
  }
  return CMS_OK;
}
extern "C" int cms_pose_fetch(cms_pose* p, double* poses7, uint8_t* outlier, int* n_inliers, cms_pose_stats* stats) {
  if (!p || p->nf < 1) return cms_fail(CMS_ERR_ARG, "cms_pose_fetch: nothing launched");
  HIPCHK(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  // results through the handle's pinned block, one synchronisation: three copies into the caller's pageable arrays were three staged,
  // waited-for transfers (1.2 ms for 256 frames next to a busy PCIe link, on the thread that drives the frame path)
  const size_t o_res = 0, o_pose = pose_al((size_t)p->nf * 32), o_out = o_pose + pose_al((size_t)p->nf * 56), total = o_out + pose_al((size_t)std::max(p->ne, 1));
  if (p->fetch_queued) {
    HIPCHK(hipEventSynchronize(p->ev_fetch));        // (cms_pose_launch enqueued the copies: the block holds this launch's results until the next launch)
  } else {
    if (total > p->h_fetch_bytes) {
      if (p->h_fetch) (void)hipHostFree(p->h_fetch);
      p->h_fetch = nullptr; p->h_fetch_bytes = 0;
      HIPCHK(hipHostMalloc((void**)&p->h_fetch, 2 * total));
      p->h_fetch_bytes = 2 * total;
    }
    uint8_t* hq = p->h_fetch;
    HIPCHK(hipMemcpyAsync(hq + o_res, p->d_res, (size_t)p->nf * 8 * sizeof(int), hipMemcpyDeviceToHost, s));
    if (poses7) HIPCHK(hipMemcpyAsync(hq + o_pose, p->d_poses, (size_t)p->nf * 7 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (outlier && p->ne > 0) HIPCHK(hipMemcpyAsync(hq + o_out, p->d_out, (size_t)p->ne, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  uint8_t* h = p->h_fetch;
  const int* res = (const int*)(h + o_res);
  if (poses7) memcpy(poses7, h + o_pose, (size_t)p->nf * 56);
  if (outlier && p->ne > 0) memcpy(outlier, h + o_out, (size_t)p->ne);
  for (int f = 0; f < p->nf; ++f) {
    if (n_inliers) n_inliers[f] = res[8 * f];
    if (stats) { stats[f].n_bad = res[8 * f + 1]; stats[f].rounds = res[8 * f + 2]; for (int i = 0; i < 4; ++i) stats[f].iterations_done[i] = res[8 * f + 4 + i]; }
  }
  return CMS_OK;
}
// A handful of frames (tracking's one call per frame: ~600 edges, 30 KB): the kernel reads its edges straight from the handle's pinned block -- once,
// into registers -- and stores poses, counts and outlier flags straight into it.  No copy at all: the six uploads, the pose copy and the three
// read-backs of the staged path each cost a trip through a copy engine's queue, ~70 us of a 230 us call whose kernel runs for ~140.
static int cms_pose_optimize_direct(cms_pose* p, int nf, const int* edge_off, const double* Xw, const double* obs_uv, const double* inv_sigma2,
                                    const int8_t* face, double fx, double fy, double cx, double cy, double* poses7, uint8_t* outlier,
                                    int* n_inliers, cms_pose_stats* stats) {
  const int ne = edge_off[nf];
  HIPCHK(hipSetDevice(p->device));
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_off = 0, o_X = al(((size_t)nf + 1) * 4), o_obs = o_X + al((size_t)ne * 24), o_inv = o_obs + al((size_t)ne * 16), o_face = o_inv + al((size_t)ne * 8),
               o_pose = o_face + al((size_t)ne), o_res = o_pose + al((size_t)nf * 56), o_out = o_res + al((size_t)nf * 32), total = o_out + al((size_t)ne);
  // (a block of the direct path's own: the handle's resident batch -- cms_pose_upload + cms_pose_launch, possibly still copying out of h_stage --
  // is not touched by this call, and a cms_pose_fetch afterwards still finds it: p->nf / p->ne stay as they are)
  if (total > p->h_direct_bytes) {
    if (p->h_direct) { HIPCHK(hipStreamSynchronize(p->stream)); (void)hipHostFree(p->h_direct); }
    p->h_direct = nullptr; p->h_direct_bytes = 0;
    HIPCHK(hipHostMalloc((void**)&p->h_direct, 4 * total));
    p->h_direct_bytes = 4 * total;
  }
  uint8_t* h = p->h_direct;
  memcpy(h + o_off, edge_off, ((size_t)nf + 1) * 4);
  if (ne > 0) { memcpy(h + o_X, Xw, (size_t)ne * 24); memcpy(h + o_obs, obs_uv, (size_t)ne * 16); memcpy(h + o_inv, inv_sigma2, (size_t)ne * 8); memcpy(h + o_face, face, (size_t)ne); }
  memcpy(h + o_pose, poses7, (size_t)nf * 56);
  PoseDev d;
  d.nf = nf; d.off = (const int*)(h + o_off); d.Xw = (const double*)(h + o_X); d.obs = (const double*)(h + o_obs); d.inv = (const double*)(h + o_inv);
  d.face = (const int8_t*)(h + o_face); d.outlier = h + o_out; d.err = p->d_err; d.poses = (double*)(h + o_pose); d.result = (int*)(h + o_res);
  d.fx = fx; d.fy = fy; d.cx = cx; d.cy = cy;
  hipLaunchKernelGGL(k_pose_optimize, dim3(nf), dim3(256), 0, p->stream, d);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(p->stream));
  const int* res = (const int*)(h + o_res);
  memcpy(poses7, h + o_pose, (size_t)nf * 56);
  if (outlier && ne > 0) memcpy(outlier, h + o_out, (size_t)ne);
  for (int f = 0; f < nf; ++f) {
    if (n_inliers) n_inliers[f] = res[8 * f];
    if (stats) { stats[f].n_bad = res[8 * f + 1]; stats[f].rounds = res[8 * f + 2]; for (int i = 0; i < 4; ++i) stats[f].iterations_done[i] = res[8 * f + 4 + i]; }
  }
  return CMS_OK;
}
extern "C" int cms_pose_optimize_batch(cms_pose* p, int nf, const int* edge_off, const double* Xw, const double* obs_uv, const double* inv_sigma2,
                                       const int8_t* face, double fx, double fy, double cx, double cy, double* poses7, uint8_t* outlier,
                                       int* n_inliers, cms_pose_stats* stats) {
  static const bool copies = getenv("CMS_POSE_COPY_ENGINE") != nullptr;      // developer A/B: the staged copies for small calls too
  if (!copies && p && nf >= 1 && nf <= 8 && nf <= p->cap_f && edge_off && poses7 && edge_off[0] == 0 && edge_off[nf] >= 0 && edge_off[nf] <= p->cap_e &&
      (edge_off[nf] == 0 || (Xw && obs_uv && inv_sigma2 && face))) {
    bool ok = true; int max_n = 0;
    for (int f = 0; f < nf && ok; ++f) { ok = edge_off[f + 1] >= edge_off[f]; max_n = std::max(max_n, edge_off[f + 1] - edge_off[f]); }
    for (int e = 0; e < edge_off[nf] && ok; ++e) ok = face[e] >= 0 && face[e] <= 4;
    if (ok && max_n <= 256 * PO_MAXJ) return cms_pose_optimize_direct(p, nf, edge_off, Xw, obs_uv, inv_sigma2, face, fx, fy, cx, cy, poses7, outlier, n_inliers, stats);
  }                                                                // (anything else: the staged path, which also reports bad arguments)
  int rc = cms_pose_upload_impl(p, nf, edge_off, Xw, obs_uv, inv_sigma2, face, fx, fy, cx, cy, poses7, true);
  if (rc) return rc;
  rc = cms_pose_launch(p);
  if (rc) return rc;
  return cms_pose_fetch(p, poses7, outlier, n_inliers, stats);       // (the launch enqueued the results' copies into the handle's pinned landing block: one wait)
}
extern "C" int cms_pose_optimize(int device, int n, const double* Xw, const double* obs_uv, const double* inv_sigma2, const int8_t* face,
                                 double fx, double fy, double cx, double cy, double* pose7, uint8_t* outlier, int* n_inliers,
                                 cms_pose_stats* stats) {
  if (n < 0 || !pose7) return cms_fail(CMS_ERR_ARG, "cms_pose_optimize: bad argument");
  cms_pose* p = nullptr;
  int rc = cms_pose_create(&p, device, 1, n > 0 ? n : 1);
  if (rc) return rc;
  const int off[2] = {0, n};
  rc = cms_pose_optimize_batch(p, 1, off, Xw, obs_uv, inv_sigma2, face, fx, fy, cx, cy, pose7, outlier, n_inliers, stats);
  cms_pose_free(p);
  return rc;
}
