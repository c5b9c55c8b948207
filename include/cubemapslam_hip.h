/*
 * cubemapslam_hip.h -- C-ABI of libcubemapslam_hip.so: the MI355X (gfx950) implementation of CubemapSLAM's
 * per-frame data-parallel hot path.  Plain C, caller-owned buffers, int status returns (0 = ok, <0 = error, see
 * cms_last_error()); nothing throws across this boundary and no torch / OpenCV / Eigen type appears in it.
 *
 * The reference has no FFI layer: its "boundary" is four C++ interfaces inside libCubemapSLAM.so.
 * Each entry point below names the reference interface it stands under; cubemapslam_amd/host/ holds C++ classes with
 * the reference's own signatures (ORBextractor::operator(), ORBMatcher, Optimizer::LocalBundleAdjustment,
 * System::CvtFisheyeToCubeMap_reverseQuery_withInterpolation) implemented on top of these calls, and INTEGRATION.md
 * shows the binding a maintainer of the reference would add.
 *
 * Threading contract (reference: Tracking thread extracts/matches, LocalMapping thread runs local BA concurrently,
 * System.cpp:108-127): a cms_ctx owns one HIP stream for the frame path; every cms_ba handle owns its own stream.
 * Frame-path calls on one ctx must come from one thread at a time; BA calls on one handle likewise; the two may
 * overlap freely.  The window-query, projection-search, mapping (cms_create_new_map_points, cms_kfstore_*, cms_fuse_search) and
 * map-point entries use the stream and the scratch arena of the ctx they are given: a mapping thread that runs next to a tracking
 * thread creates a context of its own (bench.py does).
 */
#ifndef CUBEMAPSLAM_HIP_H
#define CUBEMAPSLAM_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CMS_OK 0
#define CMS_ERR_ARG (-1)
#define CMS_ERR_HIP (-2)
#define CMS_ERR_UNSUPPORTED (-3)
#define CMS_ERR_OVERFLOW (-4)
#define CMS_ERR_NO_DEVICE (-5)

/* Face ids, include/CamModelGeneral.h:55-62 */
enum { CMS_FACE_UNKNOWN = -1, CMS_FACE_FRONT = 0, CMS_FACE_LEFT = 1, CMS_FACE_RIGHT = 2, CMS_FACE_UPPER = 3, CMS_FACE_LOWER = 4 };

/* What System::System reads from the YAML and hands to CamModelGeneral::SetCamParams (System.cpp:63-89). */
typedef struct {
  double c, d, e, u0, v0;  /* Camera.c/d/e/u0/v0 */
  double invpol[12];       /* Camera.pol0..11, zero padded (System.cpp:70-72) */
  double pol[5];           /* Camera.a0..4 */
  int Iw, Ih;              /* Camera.Iw / Ih */
  int face;                /* CubeFace.w == CubeFace.h; fx = fy = cx = cy = face/2 (System.cpp:83-84) */
  double fov_deg;          /* Camera.fov */
} cms_camera;

/* ORBextractor ctor arguments (include/ORBExtractor.h:55-56, Tracking.cpp:88-96) */
typedef struct { int nfeatures; float scale_factor; int nlevels; int ini_th_fast; int min_th_fast; } cms_orb_params;

/* The cv::KeyPoint fields the extractor fills (ORBExtractor.cpp:811-821, 918-919) */
typedef struct { float x, y, size, angle, response; int octave; } cms_keypoint;

typedef struct {
  int W, F, nlevels, kp_cap, max_batch;
  int level_w[12], level_h[12], level_quota[12], level_cells[12];
  float scale[12], inv_scale[12], sigma2[12], inv_sigma2[12];  /* ORBextractor::GetScaleFactors() etc. (ORBExtractor.h:67-87) */
  size_t pyramid_bytes_per_frame, candidate_entries_per_frame;
  int fisheye_stride;      /* row stride (bytes) of the device fisheye staging buffer */
} cms_geometry;

typedef struct cms_ctx cms_ctx;

const char* cms_last_error(void);
int cms_device_count(void);

/* ---- context: builds the remap LUT (System::CreateUndistortRectifyMap, System.cpp:301-324) and the extractor
 *      tables (ORBextractor::ORBextractor, ORBExtractor.cpp:381-442) once, allocates device buffers for max_batch frames. */
int cms_ctx_create(cms_ctx** out, int device, const cms_camera* cam, const cms_orb_params* orb, int max_batch);
void cms_ctx_destroy(cms_ctx* ctx);
int cms_ctx_geometry(const cms_ctx* ctx, cms_geometry* out);
void* cms_ctx_stream(cms_ctx* ctx);  /* hipStream_t of the frame path (for event timing / interop) */

/* ---- single-frame, host-buffer calls (the literal drop-in boundary)
 * cms_remap        : System::CvtFisheyeToCubeMap_reverseQuery_withInterpolation (include/System.h:106-107, System.cpp:327-355).
 *                    Writes the five face rectangles of the caller's 3F x 3F canvas; corner blocks are left untouched.
 * cms_set_mask     : the `mask` argument of ORBextractor::operator() (ORBExtractor.cpp:846-848), kept on the device.
 * cms_extract      : ORBextractor::operator()(image, mask, keypoints, descriptors) (include/ORBExtractor.h:63-65,
 *                    ORBExtractor.cpp:838-926).  *n receives the key-point count; returns CMS_ERR_OVERFLOW if cap is too small. */
int cms_remap(cms_ctx* ctx, const uint8_t* fisheye, int fstride, uint8_t* cubemap, int cstride);
int cms_set_mask(cms_ctx* ctx, const uint8_t* mask, int mstride);
/* cv::GaussianBlur(7x7, sigma 2) in front of the descriptors (ORBExtractor.cpp:907-908): column_mode 0 (default) = the integer column pass
 * (sum + 32768) >> 16 of OpenCV <= 3.2's generic path, 1 = the float column pass of its SSE2 functor (x86 builds: ties round to even for
 * the columns x < width & ~3).  OpenCV is not vendored with the reference, so both definitions are offered. */
int cms_set_gaussian_mode(cms_ctx* ctx, int column_mode);
int cms_extract(cms_ctx* ctx, const uint8_t* cubemap, int cstride, cms_keypoint* kps, uint8_t* desc, int cap, int* n);
int cms_remap_extract(cms_ctx* ctx, const uint8_t* fisheye, int fstride, cms_keypoint* kps, uint8_t* desc, int cap, int* n);
/* ... and with Frame::mvKeyRays: Frame::ComputeKeyPointRays -> CamModelGeneral::TransformCubemapToRays of every key point (src/Frame.cpp:746-760,
 * include/CamModelGeneral.h:494-513), rays[3 i .. 3 i + 2] = unit bearing vector of key point i in rig axes.  The extraction computes them anyway
 * (every consumer behind it -- PoseOptimization's and LocalBundleAdjustment's edge filter ray.z < cosFovTh, Optimizer.cpp:97 / 323-325, cms_keyframe.rays
 * -- wants them); cms_frames_rays: device pointer [max_batch][kp_cap][3] float of the batched path; cms_frames_fetch_rays: frame b's to the host. */
int cms_remap_extract_rays(cms_ctx* ctx, const uint8_t* fisheye, int fstride, cms_keypoint* kps, uint8_t* desc, float* rays, int cap, int* n);
int cms_frames_rays(cms_ctx* ctx, void** d_rays);
int cms_frames_fetch_rays(cms_ctx* ctx, int b, float* rays, int cap, int* n);

/* ---- batched, device-resident frame path (many frames / streams per launch; inputs already in HBM)
 * cms_frames_input()   : device pointer of the fisheye staging buffer, [max_batch][Ih][fisheye_stride] bytes.
 * cms_frames_upload()  : convenience host->device copy of B fisheye frames into that buffer.
 * cms_frames_process() : enqueue remap + pyramid + FAST + octree + cull + orientation + rBRIEF for frames [0,B) on the
 *                        ctx stream (asynchronous).  from_fisheye = 0 skips the remap (level 0 was uploaded by cms_extract).
 * cms_frames_results() : device pointers of the outputs: kps [max_batch][kp_cap] cms_keypoint, desc [max_batch][kp_cap][32],
 *                        counts [max_batch] int.
 * cms_frames_upload_async() : input streaming (the per-frame imread -> remap -> track loop of cubemap_lafida.cpp:128-154, batched): copies
 *                        B fisheye frames from PINNED host memory (cms_host_alloc / hipHostMalloc / hipHostRegister) on the
 *                        context's copy stream.  The staging buffer is only read by the first kernel of a batch (the remap), so the
 *                        copy of batch s + 1 overlaps the rest of batch s; the next cms_frames_process waits for it on the device.
 *                        The source must stay untouched until cms_frames_upload_wait() returns (or that process call was synced).
 */
void* cms_frames_input(cms_ctx* ctx);
int cms_frames_upload(cms_ctx* ctx, const uint8_t* fisheye, int fstride, size_t frame_pitch, int B);
int cms_frames_upload_async(cms_ctx* ctx, const uint8_t* fisheye, int fstride, size_t frame_pitch, int B);
int cms_frames_upload_wait(cms_ctx* ctx);
/* scheduling aid: `hip_stream` (cms_ctx_stream of a mapping-side context, cms_ba_stream of a window group) waits on the device for the
 * extraction of ctx's last cms_frames_process; the first call only arms the event (takes effect from the next process call on) */
int cms_stream_wait_extracted(cms_ctx* ctx, void* hip_stream);
int cms_frames_upload_device(cms_ctx* ctx, const void* d_src, int B);   /* device -> staging copy, [B][Ih][fisheye_stride], async on the ctx stream */
/* device -> device copy on the ctx stream, asynchronous: e.g. resetting the per-key-point map-point slots of a resident batch before a tracking pass
 * (Frame::Frame leaves mvpMapPoints all NULL, src/Frame.cpp:99-102) without leaving the frame path's queue */
int cms_stream_copy_device(cms_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
int cms_host_alloc(void** out, size_t bytes);   /* pinned host memory for cms_frames_upload_async */
void cms_host_free(void* p);
int cms_frames_process(cms_ctx* ctx, int B, int from_fisheye);
int cms_frames_sync(cms_ctx* ctx);
int cms_frames_results(cms_ctx* ctx, void** d_kps, void** d_desc, void** d_counts);
int cms_frames_fetch(cms_ctx* ctx, int b, cms_keypoint* kps, uint8_t* desc, int cap, int* n);

/* ---- stage-by-stage read-back for the parity tests */
int cms_debug_lut(cms_ctx* ctx, uint32_t* out, int* stride);
int cms_debug_cubemap(cms_ctx* ctx, int b, uint8_t* dst, int dstride);
int cms_debug_level(cms_ctx* ctx, int b, int level, uint8_t* dst, int dstride);
int cms_debug_candidates(cms_ctx* ctx, int b, int level, int* xys, int cap, int* n);   /* (x, y, score) rel. to minBorder, unordered */
int cms_debug_distributed(cms_ctx* ctx, int b, int level, int* xys, int cap, int* n);  /* octree output, list order, level coords */

/* ---- in-library stage timing with HIP events on the ctx stream (ms of the last cms_frames_process call)
 *      stages: 0 remap, 1 pyramid, 2 fast, 3 octree, 4 cull, 5 describe, 6 total */
int cms_profile_enable(cms_ctx* ctx, int on);
int cms_profile_get(cms_ctx* ctx, float* ms7);

/* ---- matching: ORBMatcher::DescriptorDistance (ORBMatcher.cpp:951-967) evaluated over the candidate list of every query
 *      (the inner loops of ORBMatcher::SearchByProjection, ORBMatcher.cpp:84-113 / 186-205).  Candidates are a CSR:
 *      query q scans cand_idx[cand_off[q] .. cand_off[q+1]).  t_excluded (optional) marks target key points that already
 *      hold a map point (ORBMatcher.cpp:91-95).  Outputs follow the sequential scan exactly (first minimum wins).
 *      The *_device variant takes device pointers and is asynchronous on the ctx stream; q_row (optional int[nq])
 *      gathers query q from row q_row[q] of qdesc, so one launch can match many frame pairs straight out of the
 *      extractor's descriptor buffer. */
int cms_hamming_best2(cms_ctx* ctx, const uint8_t* qdesc, int nq, const uint8_t* tdesc, int nt, const int* cand_off,
                      const int* cand_idx, const int* t_level, const uint8_t* t_excluded, int* best_idx, int* best_dist,
                      int* best_level, int* second_dist, int* second_level);
int cms_hamming_best2_device(cms_ctx* ctx, const void* qdesc, const void* q_row, int nq, const void* tdesc, const void* cand_off,
                             const void* cand_idx, const void* t_level, const void* t_excluded, void* best_idx,
                             void* best_dist, void* best_level, void* second_dist, void* second_level);
int cms_hamming_matrix(cms_ctx* ctx, const uint8_t* a, int na, const uint8_t* b, int nb, uint16_t* out);

/* ---- local bundle adjustment: numeric core of Optimizer::LocalBundleAdjustment (include/Optimizer.h:50,
 *      src/Optimizer.cpp:192-451) with EdgeSE3ProjectXYZMultiPinhole (include/g2o_cubemap_vertices_edges.h:90-134).
 *      The caller (the Optimizer mirror in cubemapslam_amd/host/) assembles the window exactly like Optimizer.cpp:194-357:
 *      poses K x 7 (tx,ty,tz,qx,qy,qz,qw; world->camera), fixed[K], points P x 3, and per observation: pose index,
 *      point index, measurement in the face, invSigma2 of the octave, face id.
 *      cms_ba_optimize runs optimize(its_robust) with Huber(sqrt(5.991)) -> chi2/depth classification ->
 *      optimize(its_final) on the inliers without kernel -> final classification (outlier_flags[e] = 1 to erase);
 *      *stop is polled between iterations like g2o's forceStopFlag (Optimizer.cpp:256-257): before anything is launched, then
 *      before every Levenberg trial the host enqueues.  With a stop pointer the grouped driver (optimize_many) queues a trial only
 *      after the previous one has finished, so a request raised mid-way takes effect at the next trial boundary -- where g2o's
 *      terminate() polls -- and the estimate left behind is the last accepted one, as with g2o. */
typedef struct cms_ba cms_ba;
typedef struct {
  int iterations_done[2];
  double chi2_initial[2], chi2_final[2], lambda_final[2];
  int n_outliers_mid, n_outliers_final;
} cms_ba_stats;
int cms_ba_create(cms_ba** out, int device, int K, const double* poses, const uint8_t* fixed, int P, const double* points,
                  int E, const int* e_pose, const int* e_point, const double* e_obs, const double* e_invsig2,
                  const int8_t* e_face, double fx, double fy, double cx, double cy);
/* The set-up of a window GROUP in one call (round 6): n windows as an array of descriptions (cms_ba_create's arguments).  The host parts run on up to
 * `threads` threads of the call (0: four), the set-up kernels of all device-planned windows are one launch per eight windows instead of one each --
 * a process that tracks many camera streams per GPU creates a group's sixteen windows per step (what Optimizer::LocalBundleAdjustment assembles once
 * per key frame, Optimizer.cpp:246-357).  On any error every window of the call is destroyed and out[] is all NULL.  The windows are independent
 * afterwards (cms_ba_set_stream, cms_ba_optimize_many, cms_ba_read / cms_ba_read_many, cms_ba_destroy each). */
typedef struct cms_ba_window {
  int K; const double* poses; const uint8_t* fixed; int P; const double* points; int E; const int* e_pose; const int* e_point;
  const double* e_obs; const double* e_invsig2; const int8_t* e_face; double fx, fy, cx, cy;
  int flags;      /* CMS_BA_INPUTS_PINNED | CMS_BA_PLAN_ON_DEVICE, or 0 */
} cms_ba_window;
/* flags: the window's observation-sized arrays (points, e_pose, e_point, e_obs, e_invsig2, e_face) lie in pinned host memory (cms_host_alloc, or registered with
 * the HIP runtime) and stay unchanged and alive until the window's first cms_ba_optimize* / cms_ba_read has returned or the window is destroyed: they are then
 * copied to the device from where they lie, asynchronously, instead of through the window's staging block -- a host that assembles a window's observations
 * (Optimizer.cpp:246-357) straight into pinned buffers saves one 3 MB memcpy per 80 k-observation window, as much host time as the window's whole plan. */
#define CMS_BA_INPUTS_PINNED 1
/* flags: the window's whole plan (signature groups, runs, internal point order, chunks, the left-over observations' diagonal copies) is made by a kernel
 * (k_ba_plan_many, one workgroup per window, launched per eight windows in front of their expansion) instead of by the calling threads; the call then waits
 * for those kernels (40 bytes of counts per window come back).  Same device arrays, byte for byte.  Worth it for a host short of cores -- the plan is
 * ~0.4 ms of a host thread per 80 k-observation window, half of what a two-core rank spends per step of bench.py; with cores to spare the host plan overlaps
 * the GPU better (bench.py sets the flag when its rank has at most two cores).  Windows the kernel does not take (a point seen twice by a key frame, no
 * signature runs, more than 64 key frames, CMS_BA_DET_POINTS windows) fall back to the host planners inside the call; an index out of range fails the call as ever. */
#define CMS_BA_PLAN_ON_DEVICE 2
int cms_ba_create_many(cms_ba** out, int n, int device, const cms_ba_window* windows, int threads);
/* ... and the read-back of n optimised windows (Optimizer.cpp:419-450): poses[i] / points[i] / outlier_flags[i] as cms_ba_read's (the arrays of
 * pointers, or single entries, may be NULL); one gather kernel per sixteen device-planned windows, one wait. */
int cms_ba_read_many(cms_ba** bas, int n, double** poses, double** points, uint8_t** outlier_flags);
int cms_ba_reset(cms_ba* ba);  /* restore the initial estimate on the device (benchmark loops) */
int cms_ba_optimize(cms_ba* ba, int its_robust, int its_final, const volatile uint8_t* stop, cms_ba_stats* stats);
/* n independent windows (e.g. the LocalMapping threads of n camera streams) advanced in lock-step by ONE host thread, each on
 * its own stream: same results as n cms_ba_optimize calls, without n host threads competing for the HIP runtime. stats[n]. */
int cms_ba_optimize_many(cms_ba** bas, int n, int its_robust, int its_final, const volatile uint8_t* stop, cms_ba_stats* stats);
int cms_ba_read(cms_ba* ba, double* poses, double* points, uint8_t* outlier_flags);
void* cms_ba_stream(cms_ba* ba);
/* run this window on a stream the caller owns (shared by the windows of a group, or the mapping context's own stream): keeps the
 * number of streams of a process below the number of hardware queues */
int cms_ba_set_stream(cms_ba* ba, void* hip_stream);
/* developer aid: 100 MHz wall-clock stamps of the last single-window solve kernel (start, assembled, factorised, solved, end) */
int cms_ba_debug_clocks(cms_ba* ba, long long* out16);
/* measurement aid (bench.py's roofline of the dominant BA kernel): HIP events on the group's stream around ONE kernel of every round the
 * grouped driver enqueues for the group owned by `ba` (the first handle passed to cms_ba_optimize_many).  kernel_id: 0 off, 1 kb_ba_lin,
 * 2 kb_ba_maxdiag, 3 the Schur kernel of the path in use (kb_ba_lin_schur_edges by default: linearisation + Schur complement), 4 its range
 * reduction, 5 kb_ba_trial_solve3, 6 kb_ba_trial_edges, 7 kb_ba_reduce2.
 * cms_ba_profile_get returns the summed duration and the number of launches since cms_ba_profile_kernel was called. */
int cms_ba_profile_kernel(cms_ba* ba, int kernel_id);
int cms_ba_profile_get(cms_ba* ba, double* total_ms, long* launches);
/* developer / test aid, host only (no device needed): the chunk composition cms_ba_create builds for the edge-major Schur kernel -- which
 * points share a wavefront (<= 64 observations), in which order, and which copy of its key frame's diagonal block every observation adds
 * to -- chosen so that the lanes of one LDS addition fall on different banks.  lookahead <= 1: the caller's order.
 * pinv: P entries (internal point -> caller's point); chunk_pt0: room for P + 1, *n_chunks + 1 are written (first internal point of every
 * chunk, P last); rank (may be NULL): E entries, the caller's points in order, observations of a point by ascending key frame.
 * The graph itself is what Optimizer::LocalBundleAdjustment hands to g2o (Optimizer.cpp:192-363); the composition has no counterpart there. */
int cms_ba_debug_compose(int K, const uint8_t* fixed, int P, int E, const int* e_pose, const int* e_point, int lookahead,
                         int* pinv, int* chunk_pt0, int* n_chunks, uint8_t* rank);
/* developer / test aid, host only: the whole plan cms_ba_create makes for a window.  Points seen by the same set of key frames (one observation
 * signature) are grouped; a signature with enough points becomes a RUN whose chunks the run-major Schur kernel sums in registers
 * (cubemapslam_amd/csrc/cms_ba_schur_runs.hip), the rest are the composed chunks of cms_ba_debug_compose.  Outputs (upper-bound sizes): pinv P
 * (internal point -> caller's point), perm E (sorted edge -> caller's edge), info E (per-edge word of the kernels), chunk_pt0 P + 2, rm_chunk
 * 4 ints per run chunk (<= P), run_lane 128 u32 per run (<= P runs), run_mf 64 / run_fl 768 u32 per run (the MFMA variant's tables; may be
 * NULL), counts[8] = chunks, run chunks, runs, free key frames, points inside runs,
 * workgroups of a lone window for the run chunks / the left-over chunks, 1 if the window can use these kernels at all.  No counterpart in the
 * reference: the graph is what Optimizer::LocalBundleAdjustment hands to g2o (Optimizer.cpp:192-363). */
int cms_ba_debug_plan(int K, const uint8_t* fixed, int P, int E, const int* e_pose, const int* e_point, int* pinv, int* perm, uint32_t* info,
                      int* chunk_pt0, int* rm_chunk, uint32_t* run_lane, int* counts, uint32_t* run_mf, uint32_t* run_fl);
/* developer / test aid: the plan arrays as the DEVICE holds them for this window (after its set-up ran), whichever planner made them.  Since round 5
 * cms_ba_create plans most windows with one sequential pass over the observations on the host and per-point work only; the observation-sized
 * arrays (edge permutation, sorted edge arrays, per-edge words, diagonal copies, the runs' tables) are written by kernels
 * (cubemapslam_amd/csrc/cms_api_ba_plan.hip).  For the windows both planners take they give the same arrays: this entry is how the tests
 * compare them with cms_ba_debug_plan.  Sizes: pinv P, perm E, info E, pt_off P + 1, e_pose / e_point / e_face E, chunk_e0 chunks + 1, rm_chunk
 * 4 ints per run chunk, rm_cost chunks + 1, run_mf 64 / run_fl 768 words per run (any may be NULL); counts[8] as cms_ba_debug_plan's, with
 * counts[7] = 1 if the device-side planner made the window.  CMS_BA_HOST_PLAN=1 selects the host planner for every window (A/B). */
int cms_ba_debug_fetch_plan(cms_ba* ba, int* pinv, int* perm, uint32_t* info, int* pt_off, int* e_pose, int* e_point, int8_t* e_face, int* chunk_e0,
                            int* rm_chunk, uint32_t* rm_cost, uint32_t* run_mf, uint32_t* run_fl, int* counts);
/* ... and the same planner run entirely on the host (no device needed; the expansion kernels' bodies over host arrays): outputs as cms_ba_debug_plan's
 * (run_lane excepted); counts[7] = 0 when the window is not one the device-side planner takes (cms_ba_create then uses the host planner). */
int cms_ba_debug_plan_fast(int K, const uint8_t* fixed, int P, int E, const int* e_pose, const int* e_point, int* pinv, int* perm, uint32_t* info,
                           int* chunk_pt0, int* rm_chunk, int* counts, uint32_t* run_mf, uint32_t* run_fl);
/* developer / test aid, host only: the tables of the opt-in one-wavefront run workgroups (CMS_BA_RUN_WG=1, cubemapslam_amd/csrc/cms_ba_schur_runwg.hip:
 * round 6's re-decomposition of the Schur kernel, block_solver.hpp:367-437) as either planner makes them (fast != 0: the device-side planner's host
 * part).  run_fg: 64 x 24 words per run (per lane and MFMA accumulator its offset into the window's global copy of the reduced system, 0xFFFFFFFF: none),
 * rm_cut: 2 x 1025 cut points (per class of signature), counts[4] = runs (-1: the planner does not take the window), run chunks, class-0 run chunks,
 * free key frames. */
int cms_ba_debug_run_fg(int K, const uint8_t* fixed, int P, int E, const int* e_pose, const int* e_point, int fast, uint32_t* run_fg, int* rm_cut, int* counts);
/* developer / test entry, host only: the matching lanes -> copies of one group of 16 left-over observations (which = 0: the host planners' search, 1: the plan
 * kernel's register-only form of it -- the two must agree, tests/test_ba_plan_kernel_cpu.py); slots[n] = the lanes' free-pose slots, n <= 16 */
int cms_ba_debug_match(int which, int n, const int* slots, int np, int* choice_out);
void cms_ba_destroy(cms_ba* ba);
/* Device slabs and pinned blocks of destroyed windows wait in a per-device pool for the next window (CMS_BA_POOL_MB bounds the device part, default
 * 16384; the pool is also emptied and the allocation retried when hipMalloc fails).  cms_ba_pool_trim hands everything cached for `device` back
 * to the runtime -- for callers that share the device with other allocators; *released (may be NULL) receives the bytes. */
int cms_ba_pool_trim(int device, size_t* released);
/* Determinism as a product mode.  The reference optimises with a single-threaded g2o (ThirdParty/g2o/config.h:4: no OpenMP), so two runs on
 * the same window give the same bits.  The default device path adds with FP64 atomics (order varies from run to run: last bits differ, see
 * DESIGN.md section 2); cms_ba_set_deterministic(1) makes every window created AFTERWARDS add in a fixed order: the same kernels as the default
 * path with the Schur kernel's LDS additions performed in an order that follows from the window's plan alone (kb_ba_lin_schur_runs_det /
 * kb_ba_lin_schur_edges_det), the workgroups' sums kept as slices that the solve kernel adds in slice order instead of one global copy, and a cut
 * into workgroups that does not depend on the other windows of the call.  Same input, same bits: run after run, alone or in any group of a
 * cms_ba_optimize_many call, created by cms_ba_create or cms_ba_create_many (plan kernel included); bench.py's step runs at 0.84-0.87 of the default with
 * it (config.deterministic).  Windows the fused chain cannot take (more than 25 free key frames, a point seen twice by a key frame) run rounds 3-5's
 * pair-owner kernel (kb_ba_schur_points, CMS_BA_DET_POINTS=1 selects it for all) -- deterministic as well.  The choice is taken at cms_ba_create and
 * travels with the window; windows of both kinds may be passed to one cms_ba_optimize_many call (they run as separate groups).  Process-wide; the
 * environment variable CMS_BA_DETERMINISTIC=1 gives the initial value.  cms_ba_get_deterministic returns the current setting.
 * on >= 2 also says into how many workgroups such a window is cut (1: the default, 16 -- a group of sixteen windows fills the chip once; at most 128).  The count is
 * fixed when the window is created, because the bits depend on it.  A host that optimises ONE window per call (the reference's LocalMapping thread) should pass 64:
 * 2.4 instead of 3.5 ms per 80 k-observation window (default path 1.7 ms); sixteen windows in lock-step then take 6.1 instead of 4.8 ms. */
int cms_ba_set_deterministic(int on);
int cms_ba_get_deterministic(void);
/* one-shot convenience: create + optimize + read + destroy */
int cms_ba_run(int device, int K, double* poses, const uint8_t* fixed, int P, double* points, int E, const int* e_pose,
               const int* e_point, const double* e_obs, const double* e_invsig2, const int8_t* e_face, double fx, double fy,
               double cx, double cy, int its_robust, int its_final, const volatile uint8_t* stop, uint8_t* outlier_flags,
               cms_ba_stats* stats);
/* one linearisation at the given estimate (parity of the J^T J / J^T r build): Hpp K x 36, bp K x 6, Hll P x 9, bl P x 3,
 * Hpl E x 18 (caller's edge order), err E x 2, robust chi2 sum */
int cms_ba_linearize(int device, int K, const double* poses, const uint8_t* fixed, int P, const double* points, int E,
                     const int* e_pose, const int* e_point, const double* e_obs, const double* e_invsig2,
                     const int8_t* e_face, double fx, double fy, double cx, double cy, int robust, double huber_delta,
                     double* err, double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, double* robust_chi2_sum);

/* ---- frame grid + window query: Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cpp:158-176, 728-744) and
 * Frame::GetFeaturesInArea with AddCells (src/Frame.cpp:36-72, 251-716), the step in front of every Hamming scan of
 * ORBMatcher::SearchByProjection / SearchForInitialization.  The reference's 41 window-unfolding cases (90 AddCells calls) are
 * reproduced literally (cubemapslam_amd/csrc/cms_area_table.h), so the candidate ORDER -- which decides Hamming ties -- is the
 * reference's.  cms_area_grid builds the 5 x 50 x 50 cell lists of frames 0..B-1 from the key points the last
 * cms_frames_process left on the device (or cms_area_set_keypoints put there); a query is (x, y, r, minLevel, maxLevel) in canvas
 * pixels; the result is a CSR list cand_off[nq+1] / cand_idx[] of key-point indices of frame b.  The _device variant takes and
 * returns device pointers (cnt_scratch: nq ints) and adds idx_base to every index, so the lists feed cms_hamming_best2_device
 * without touching the host. */
int cms_area_set_keypoints(cms_ctx* ctx, int b, int n, const cms_keypoint* kps);
int cms_area_grid(cms_ctx* ctx, int B);
int cms_features_in_area(cms_ctx* ctx, int b, int nq, const float* qx, const float* qy, const float* qr, const int* min_level,
                         const int* max_level, int* cand_off, int* cand_idx, int cap, int* total);
int cms_features_in_area_device(cms_ctx* ctx, int b, int nq, const void* d_qx, const void* d_qy, const void* d_qr, const void* d_min_level,
                                const void* d_max_level, void* d_cnt_scratch, void* d_cand_off, void* d_cand_idx, int cap, int idx_base,
                                void* d_total);
/* one launch sequence for a whole batch: d_qframe[q] names the frame (0 .. B-1 of cms_area_grid) query q searches; indices are batch rows */
int cms_features_in_area_batch_device(cms_ctx* ctx, int nq, const void* d_qframe, const void* d_qx, const void* d_qy, const void* d_qr,
                                      const void* d_min_level, const void* d_max_level, void* d_cnt_scratch, void* d_cand_off,
                                      void* d_cand_idx, int cap, void* d_total);

/* ---- track local map: Frame::isInFrustum (src/Frame.cpp:197-249, with MapPoint::PredictScale, src/MapPoint.cpp:404-419, and
 * Get{Min,Max}DistanceInvariance :375-385) for every local map point, then ORBMatcher::SearchByProjection(Frame&, const
 * vector<MapPoint*>&, th) (src/ORBMatcher.cpp:50-128, RadiusByViewingCos :380-386) -- the pair Tracking::SearchLocalPoints
 * (src/Tracking.cpp:794-846) runs per frame with viewingCosLimit 0.5, ORBMatcher(0.8), th 1 (5 after a relocalisation).
 *   pose15      Rcw (9 floats, row major) | tcw (3) | Ow (3): Frame::mRcw / mtcw / mOw as Frame::UpdatePoseMatrices leaves them
 *   pos, normal MapPoint::mWorldPos / mNormalVector (3 floats each); min_dist / max_dist = mfMinDistance / mfMaxDistance
 *   outputs     in_view = mbTrackInView, proj_x/y = mTrackProjX/Y (-1 when not in view), level = mnTrackScaleLevel, view_cos
 * The reference's matching loop is sequential -- a key point taken by one map point is skipped by all later ones -- and the
 * device reproduces exactly that result (cms_track_kernels.hip explains how).  kp_mp holds one int per key point: >= 0 on entry
 * means "already holds a map point with observations" (ORBMatcher.cpp:91-93); a new match stores the map point's list index.
 * cms_search_local_points is the one-frame entry with host buffers: frame b's key points/descriptors are the ones the last
 * cms_frames_process left on the device, or cms_area_set_keypoints + cms_area_set_descriptors put there (cms_area_grid first).
 * The _device entries take device pointers, work on all frames of a batch at once and are asynchronous on the ctx stream:
 * cms_is_in_frustum_device also writes the window of every point (qr < 0: not in view) for cms_features_in_area_batch_device;
 * cms_search_local_points_device takes that CSR (indices = batch rows), mp_off[B+1] (map points grouped by frame, list order
 * inside a frame), scratch pair_dist (2 bytes per candidate) and kp_mp over all batch rows; mp_match = batch row or -1.
 * Limit: at most 4096 key points per frame for the greedy projection searches (CMS_ERR_UNSUPPORTED; the frame grid itself and
 * cms_search_for_initialization take up to 16383, i.e. the 3 x nFeatures extractor); any number of map points per frame (a thread of the greedy kernel
 * takes every 1024th point of its frame). */
int cms_area_set_descriptors(cms_ctx* ctx, int b, int n, const uint8_t* desc);
/* What the caller hands over as min_dist / max_dist of a map point (cms_search_local_points, cms_is_in_frustum_device, cms_fuse_search,
 * cms_kfstore_fuse_search of this context): scaled = 0 (default) MapPoint::mfMinDistance / mfMaxDistance themselves -- private members, a binding
 * needs two one-line accessors; scaled = 1 the public MapPoint::GetMinDistanceInvariance() / GetMaxDistanceInvariance() (src/MapPoint.cpp:375-385:
 * 0.8f / 1.2f already applied), so that the reference's headers stay byte-identical.  With scaled = 1 the bounds of Frame::isInFrustum / Fuse
 * are exactly the getters' values; mfMaxDistance (the numerator of MapPoint::PredictScale, :387-419) is recovered as the float r with
 * 1.2f * r == bound -- where the product crossed a power of two, two neighbouring r share a bound and the smaller is taken: the predicted
 * level can then differ from the reference only if log(ratio) / log(scale factor) sits within an ulp of an integer. */
int cms_set_distance_bounds_mode(cms_ctx* ctx, int scaled);
/* ORBMatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (src/ORBMatcher.cpp:130-251), the matcher of
 * Tracking::TrackWithMotionModel, whole on the device: the last frame's map points are projected with the current pose (Rcw | tcw of
 * CurrentFrame.mTcw; z < cosFovTh and UNKNOWN_FACE are dropped), windows th * scale[octave] over octave -1 .. +1, greedy best match
 * <= TH_HIGH (a key point taken by one map point is skipped by the later ones), then the rotation-consistency histogram
 * (ComputeThreeMaxima, :905-946).  valid[i]: key point i of the last frame holds a map point and is not an outlier.  kp_mp as in
 * cms_search_local_points.  The _device entries are the batched pieces: cms_project_last_frame_device writes the windows of all
 * queries (q_frame = current frame each one searches), cms_features_in_area_batch_device and cms_search_local_points_device
 * (nnratio < 0: no second-best test) follow, cms_rotation_filter_device applies the histogram per frame and counts the matches. */
int cms_search_by_projection(cms_ctx* ctx, int b, const float* pose12, int nlast, const uint8_t* valid, const float* Xw, const int* octave,
                             const float* angle, const uint8_t* mp_desc, float th, int check_orientation, int th_high, int nkp, int* kp_mp,
                             int* match, int* n_matches);
/* ORBMatcher::SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched, vector<int>& vnMatches12, int windowSize)
 * (include/ORBMatcher.h:58, src/ORBMatcher.cpp:676-794), the matcher of Tracking::MonocularInitialization (Tracking.cpp:428-429:
 * ORBMatcher(0.9, true), window 100).  F2 = frame slot b2 on the device (cms_area_grid first), F1 = the caller's key points and
 * descriptors.  Level-0 key points of F1 only; a key point of F2 is taken over by a later strictly better match; rotation histogram
 * over every accepted match; prev_matched (n1 x 2 floats = vbPrevMatched) is updated for the matched key points. */
int cms_search_for_initialization(cms_ctx* ctx, int b2, int n1, const cms_keypoint* kps1, const uint8_t* desc1, float* prev_matched,
                                  int window_size, float nnratio, int check_orientation, int* matches12, int* n_matches);
int cms_project_last_frame_device(cms_ctx* ctx, int n, const void* d_qframe, const void* d_pose12, const void* d_valid, const void* d_Xw,
                                  const void* d_oct, float th, void* d_qx, void* d_qy, void* d_qr, void* d_qmin, void* d_qmax);
int cms_rotation_filter_device(cms_ctx* ctx, int B, const void* d_mp_off, const void* d_last_angle, void* d_kp_mp, void* d_mp_match,
                               void* d_n_matches, int check_orientation);
int cms_search_local_points(cms_ctx* ctx, int b, const float* pose15, int nmp, const float* pos, const float* normal,
                            const float* min_dist, const float* max_dist, const uint8_t* mp_desc, float viewing_cos_limit, float th,
                            float nnratio, int th_high, int nkp, int* kp_mp, uint8_t* in_view, float* proj_x, float* proj_y, int* level,
                            float* view_cos, int* mp_match, int* n_matches, int* rounds);
int cms_is_in_frustum_device(cms_ctx* ctx, int nmp, const void* d_mp_frame, const void* d_pose15, const void* d_pos, const void* d_normal,
                             const void* d_min_dist, const void* d_max_dist, float viewing_cos_limit, float th, void* d_in_view,
                             void* d_proj_x, void* d_proj_y, void* d_level, void* d_view_cos, void* d_qr, void* d_qmin, void* d_qmax);
int cms_search_local_points_device(cms_ctx* ctx, int B, const void* d_mp_off, const void* d_mp_desc, const void* d_cand_off,
                                   const void* d_cand_idx, void* d_pair_dist, float nnratio, int th_high, void* d_kp_mp, void* d_mp_match,
                                   void* d_rounds);

/* ---- mapping thread, either side of the local BA.
 * cms_create_new_map_points: numeric core of LocalMapping::CreateNewMapPoints (src/LocalMapping.cpp:209-386, bearing-vector version):
 * for every neighbour of a current key frame, in the order given (GetBestCovisibilityKeyFrames), the baseline test (:238-247),
 * ComputeE12 (:469-482), ORBMatcher::SearchForTriangulation (src/ORBMatcher.cpp:971-1125; CheckDistEpipolarLine :388-407,
 * CamModelGeneral::GetVectorSigma src/CamModelGeneral.cpp:307-333) and the triangulation of every match on its key rays with all of
 * the reference's tests (:266-357).  A feature triangulated with one neighbour is taken for the following ones, as
 * KeyFrame::AddMapPoint makes it.  njobs current key frames (independent maps / streams) are ONE launch.
 *   cms_keyframe   what these functions read of a KeyFrame: mvKeys, mDescriptors, mvKeyRays, mp[i] >= 0 <=> GetMapPoint(i) != NULL,
 *                  GetRotation / GetTranslation / GetCameraCenter (float), mFeatVec as CSR (node ids ascending; DBoW2 is only
 *                  needed to produce it), ComputeSceneMedianDepth(2)
 *   neigh_off      njobs + 1 offsets into neigh[]
 *   outputs        n_new[j]; records j * cap_per_job + k, k < n_new[j], in creation order: neighbour index (within job j),
 *                  idx1, idx2, x3D.  check_orientation = ORBMatcher's mbCheckOrientation (false at this call site, :217).
 * cms_fuse_search: search half of ORBMatcher::Fuse(pKF, vpMapPoints, th) (src/ORBMatcher.cpp:1127-1226) against key-frame slot b
 * (cms_area_set_keypoints + cms_area_set_descriptors + cms_area_grid first): projection, image / distance / viewing-angle tests,
 * PredictScale, KeyFrame::GetFeaturesInArea(u, v, th * scale), level and reprojection gates, Hamming minimum <= TH_LOW.
 * skip[i] != 0 <=> !pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF).  best_idx[i] = key point to fuse with or -1; the Replace /
 * AddObservation bookkeeping (:1216-1241) stays with the caller, in list order. */
typedef struct {
  int n; const cms_keypoint* kps; const uint8_t* desc; const float* rays; const int* mp;
  float Rcw[9], tcw[3], Ow[3];
  int nnodes; const int* node_id; const int* node_off; const int* node_feat;
  float median_depth;
} cms_keyframe;
int cms_create_new_map_points(cms_ctx* ctx, int njobs, const cms_keyframe* cur, const int* neigh_off, const cms_keyframe* neigh,
                              int check_orientation, int cap_per_job, int* n_new, int* out_neigh, int* out_idx1, int* out_idx2,
                              float* out_x3d);
/* ORBMatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat E12, vector<pair<size_t, size_t>>& vMatchedPairs) (include/ORBMatcher.h:61,
 * src/ORBMatcher.cpp:971-1125) on its own -- the call of LocalMapping.cpp:254 for callers that triangulate themselves: for every feature of kf1
 * without a map point, the best feature of kf2 (same vocabulary node, no map point, Hamming <= TH_LOW, away from the epipole, inside the
 * epipolar gate), then the rotation histogram if check_orientation.  E12: the 3 x 3 essential matrix the caller computed (row major), NULL =
 * LocalMapping::ComputeE12 of the two poses.  matches12[i1] = index in kf2 or -1 (kf1->n entries); *n_matches = their number. */
int cms_search_for_triangulation(cms_ctx* ctx, const cms_keyframe* kf1, const cms_keyframe* kf2, const float* E12, int check_orientation,
                                 int* matches12, int* n_matches);
/* Resident key frames: the map's key frames stay on the device in fixed-size slots, a CreateNewMapPoints call names slots and only
 * ~100 bytes per (current, neighbour) pair travel.  cms_kfstore_put uploads / replaces a key frame; cms_kfstore_update refreshes what
 * changes between calls (pose after a local BA, median depth, map-point slots; NULL = unchanged).  Results as above. */
typedef struct cms_kfstore cms_kfstore;
int cms_kfstore_create(cms_kfstore** out, cms_ctx* ctx, int max_keyframes, int max_features, int max_nodes);
void cms_kfstore_destroy(cms_kfstore* st);
int cms_kfstore_put(cms_kfstore* st, int slot, const cms_keyframe* kf);
int cms_kfstore_update(cms_kfstore* st, int slot, const float* Rcw, const float* tcw, const float* Ow, const float* median_depth, const int* mp);
/* LocalMapping::ProcessNewKeyFrame's device half (src/LocalMapping.cpp:52-117: the frame Tracking turned into a key frame enters the map): the key
 * points, descriptors, key rays (Frame::mvKeyRays) and the frame grid of frame b of `src`'s last batch are already on the device (cms_frames_process
 * + cms_area_grid left them there); one kernel copies them device to device into `slot` -- no trip through the host, no second grid build, no
 * synchronisation -- and reads what only the host has from a pinned block: the FeatureVector (KeyFrame::ComputeBoW, DBoW2 on the host) and the
 * map-point slots mp[n] (NULL: none).  n = the frame's key-point count (what cms_frames_fetch reports).  Asynchronous: the copy is enqueued on `src`'s
 * stream -- behind the work that produced frame b and in front of the next batch, like the KeyFrame constructor's copy on the Tracking thread
 * (Tracking.cpp:1015-1017) -- and the store's stream waits for it on the device; `src` may be the store's own context.  Equivalent to cms_kfstore_put of the
 * fetched frame (tests/test_gpu_parity.py::test_kfstore_put_from_frame_equals_put). */
int cms_kfstore_put_from_frame(cms_kfstore* st, int slot, cms_ctx* src, int b, int n, const float* Rcw, const float* tcw, const float* Ow, float median_depth,
                               const int* mp, int nnodes, const int* node_id, const int* node_off, const int* node_feat);
/* ... several key frames of one batch in one call (one per camera stream and step when a process tracks many streams per GPU) */
typedef struct {
  int slot, b, n;
  const float* Rcw; const float* tcw; const float* Ow; float median_depth;
  const int* mp; int nnodes; const int* node_id; const int* node_off; const int* node_feat;
} cms_kf_from_frame;
int cms_kfstore_put_from_frames(cms_kfstore* st, cms_ctx* src, int n_items, const cms_kf_from_frame* items);
/* the poses of n resident key frames after a local BA (Optimizer.cpp:419-431 writes them back into the KeyFrames): Rcw n x 9, tcw n x 3, Ow n x 3.
 * One kernel reading a pinned block, asynchronous -- cms_kfstore_update per key frame costs a copy and a stream wait each. */
int cms_kfstore_update_poses(cms_kfstore* st, int n, const int* slots, const float* Rcw, const float* tcw, const float* Ow);
/* developer / test aid: what a slot holds on the device (any pointer may be NULL): kps / desc / rays / mp / feat_node / sorted n entries (n = header[1]),
 * node_id nnodes, node_off nnodes + 1, node_feat node_off[nnodes], cell_start 5 x 50 x 50 + 1, header = the slot's 21-word record (f0, n, node0, nnodes,
 * noff0, nfeat0, Rcw, tcw, Ow), misc[2] = valid grid entries, stored key-point count */
int cms_kfstore_debug_fetch(cms_kfstore* st, int slot, cms_keypoint* kps, uint8_t* desc, float* rays, int* mp, int* feat_node, uint16_t* sorted,
                            int* node_id, int* node_off, int* node_feat, int* cell_start, uint32_t* header, int* misc);
/* SearchInNeighbors' Fuse calls (src/LocalMapping.cpp:388-467) on resident key frames in one launch sequence: job j searches the map points
 * [mp_off[j], mp_off[j+1]) of the concatenated host arrays in the key frame of slot job_slot[j]; arguments and results as cms_fuse_search. */
int cms_kfstore_fuse_search(cms_kfstore* st, int njobs, const int* job_slot, const int* mp_off, const uint8_t* skip, const float* pos,
                            const float* normal, const float* min_dist, const float* max_dist, const uint8_t* mp_desc, float th, int* best_idx,
                            int* best_dist);
/* SearchInNeighbors' shape (src/LocalMapping.cpp:388-466): one key frame's map points go into each of its ~20 neighbours, so the same positions /
 * normals / descriptors would travel 20 times.  Here they are uploaded ONCE as sets: set s = map points set_off[s] .. set_off[s + 1] of pos / normal /
 * min_dist / max_dist / mp_desc; job j searches key frame slot job_slot[j] with set job_set[j].  skip (may be NULL), best_idx, best_dist are per
 * ENTRY: job after job, a job's entries in the order of its set.  Results as cms_kfstore_fuse_search's. */
int cms_kfstore_fuse_search_sets(cms_kfstore* st, int nsets, const int* set_off, const float* pos, const float* normal, const float* min_dist,
                                 const float* max_dist, const uint8_t* mp_desc, int njobs, const int* job_slot, const int* job_set, const uint8_t* skip,
                                 float th, int* best_idx, int* best_dist);
int cms_kfstore_create_new_map_points(cms_kfstore* st, int njobs, const int* cur_slot, const int* neigh_off, const int* neigh_slot,
                                      int check_orientation, int cap_per_job, int* n_new, int* out_neigh, int* out_idx1, int* out_idx2,
                                      float* out_x3d);
/* Per-map-point bookkeeping the mapping thread runs over the points it created, fused or optimised, batched:
 * MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cpp:243-308): obs_off[npts+1] into the concatenated descriptors of each
 * point's non-bad observations (std::map order); best_idx[p] = the observation whose descriptor has the least median distance to the
 * rest (first on ties), -1 without observations.  MapPoint::UpdateNormalAndDepth (:332-373): obs_Ow = camera centres of the observing
 * key frames (same order), ref_Ow / ref_level = centre of mpRefKF and octave of its observation; normal / min_dist / max_dist are
 * in/out (points without observations stay untouched). */
int cms_distinctive_descriptors(cms_ctx* ctx, int npts, const int* obs_off, const uint8_t* desc, int* best_idx);
int cms_update_normal_and_depth(cms_ctx* ctx, int npts, const int* obs_off, const float* pos, const float* obs_Ow, const float* ref_Ow,
                                const int* ref_level, float* normal, float* min_dist, float* max_dist);
int cms_fuse_search(cms_ctx* ctx, int b, const float* pose15, int nmp, const uint8_t* skip, const float* pos, const float* normal,
                    const float* min_dist, const float* max_dist, const uint8_t* mp_desc, float th, int* best_idx, int* best_dist);

/* ---- pose-only optimisation: Optimizer::PoseOptimization(Frame*) (src/Optimizer.cpp:48-190), the per-frame solver Tracking calls
 * 1-3 times per frame (Tracking.cpp:585,647,688).  Edge = EdgeSE3ProjectXYZMultiPinholeOnlyPose
 * (include/g2o_cubemap_vertices_edges.h:42-88, src/g2o_cubemap_vertices_edges.cpp:61-134).  One workgroup per frame runs all four
 * rounds (10 Levenberg iterations each, chi2 > 5.991 re-classification in between, Huber dropped for the last round) on the device;
 * a batch of nf frames (camera streams) is ONE launch.  The caller prepares, exactly as Optimizer.cpp:80-127 does per matched
 * map point whose key ray passes the FoV test: Xw (world point), obs_uv = GetPosInFace(kp.pt), face = FaceInCubemap(kp.pt),
 * inv_sigma2 = mvInvLevelSigma2[kp.octave]; edge_off[f]..edge_off[f+1] are frame f's edges.  poses7: nf x (tx,ty,tz,qx,qy,qz,qw),
 * world->camera, in/out.  outlier: one byte per edge (pFrame->mvbOutlier).  n_inliers[f] = nInitialCorrespondences - nBad, or 0
 * and an untouched pose when the frame has fewer than 3 edges. */
typedef struct cms_pose cms_pose;
typedef struct { int rounds, n_bad; int iterations_done[4]; } cms_pose_stats;
int cms_pose_create(cms_pose** out, int device, int max_frames, int max_edges);
void cms_pose_destroy(cms_pose* p);
void* cms_pose_stream(cms_pose* p);
int cms_pose_optimize_batch(cms_pose* p, int nf, const int* edge_off, const double* Xw, const double* obs_uv, const double* inv_sigma2,
                            const int8_t* face, double fx, double fy, double cx, double cy, double* poses7, uint8_t* outlier,
                            int* n_inliers, cms_pose_stats* stats);
/* the same in three steps, for callers that keep the problem resident: upload once, launch (asynchronous, restarts from the
 * uploaded poses every time; the results' copies into the handle's pinned block are enqueued right behind the kernel), fetch (waits for
 * those copies -- nothing is enqueued at fetch time, so a caller that launches early and fetches late does not wait for a slot on a busy
 * device -- and hands the results out; may be called again for the same launch) */
int cms_pose_upload(cms_pose* p, int nf, const int* edge_off, const double* Xw, const double* obs_uv, const double* inv_sigma2,
                    const int8_t* face, double fx, double fy, double cx, double cy, const double* poses7);
int cms_pose_launch(cms_pose* p);
int cms_pose_fetch(cms_pose* p, double* poses7, uint8_t* outlier, int* n_inliers, cms_pose_stats* stats);
/* one frame, one shot (create + optimize + destroy) */
int cms_pose_optimize(int device, int n, const double* Xw, const double* obs_uv, const double* inv_sigma2, const int8_t* face,
                      double fx, double fy, double cx, double cy, double* pose7, uint8_t* outlier, int* n_inliers, cms_pose_stats* stats);

#ifdef __cplusplus
}
#endif
#endif
