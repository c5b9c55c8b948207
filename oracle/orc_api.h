/*
 * oracle/orc_api.h -- CPU ORACLE for the CubemapSLAM hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory is a plain C++ (no OpenCV / Eigen / g2o) restatement of the reference's
 * per-frame hot path, used exclusively as the checker by tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg.  Nothing under cubemapslam_amd/ may include, link or
 * call it; the product path fails loudly when its HIP library is missing.
 *
 * PARITY STATUS: **parity unpinned**.  The reference ships no tests, golden vectors or
 * fixtures, and it cannot be built in this image: it needs OpenCV
 * 2.4.11/3.2, Eigen3 and Pangolin, none of which are present, so no
 * `oracle/_ref` build exists.  Reference-owned logic (the src and include trees, vendored
 * ThirdParty/g2o) is restated from the cited file:line.  OpenCV primitives (remap, resize,
 * FAST, GaussianBlur, fastAtan2, cvRound -- third-party, un-vendored; README.md:59 pins
 * "OpenCV 2.4.11 and 3.2") are restated from the published algorithm of those versions as
 * summarised in SURVEY.md Appendix C.  What pins the oracle instead:
 *   - known answers recorded in SURVEY.md section 8c from a run of the reference's own
 *     CamModelGeneral TU (tests/test_oracle_cam.py),
 *   - independent brute-force numpy re-implementations of every integer primitive
 *     (tests/test_oracle_cv.py), finite-difference Jacobians and a dense numpy LM step for
 *     the bundle adjustment (tests/test_oracle_ba.py).
 */
#ifndef ORC_API_H
#define ORC_API_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Face ids: include/CamModelGeneral.h:55-62 */
enum { ORC_FACE_UNKNOWN = -1, ORC_FACE_FRONT = 0, ORC_FACE_LEFT = 1, ORC_FACE_RIGHT = 2,
       ORC_FACE_UPPER = 3, ORC_FACE_LOWER = 4 };

/* Camera parameters as System.cpp:63-89 hands them to CamModelGeneral::SetCamParams. */
typedef struct {
  double c, d, e, u0, v0;
  double invpol[12]; /* zero padded, degree fixed at 12 (System.cpp:70-72, CamModelGeneral.cpp:85-86) */
  double pol[5];     /* forward polynomial, zero padded to 5 (System.cpp:67-69) */
  int Iw, Ih;        /* fisheye size */
  int face;          /* CubeFace.w == CubeFace.h == F ; fx=fy=cx=cy=F/2 (System.cpp:83-84) */
  double fov_deg;
} orc_camera;

typedef struct {
  int nfeatures; float scale_factor; int nlevels; int ini_th_fast; int min_th_fast;
} orc_orb_params;

/* cv::KeyPoint fields the extractor fills (ORBExtractor.cpp:811-821, 918-919) */
typedef struct { float x, y, size, angle, response; int octave; } orc_keypoint;

/* ---- camera model (CamModelGeneral.h / .cpp) ---- */
void orc_world_to_img(const orc_camera* cam, double x, double y, double z, double* u, double* v);
void orc_img_to_world(const orc_camera* cam, double u, double v, double* x, double* y, double* z);
void orc_cubemap_to_fisheye(const orc_camera* cam, double up, double vp, double* uf, double* vf);
int  orc_face_in_cubemap(const orc_camera* cam, float x, float y);
int  orc_rays_to_cubemap(const orc_camera* cam, float x, float y, float z, float* up, float* vp);
void orc_rays_to_target_face(const orc_camera* cam, float x, float y, float z, int face, float* up, float* vp);
int  orc_cubemap_to_rays(const orc_camera* cam, float px, float py, float* ray3);
float orc_cos_fov_th(const orc_camera* cam);
/* System::CreateUndistortRectifyMap (System.cpp:301-324): two W x W float maps, W = 3F */
void orc_build_lut(const orc_camera* cam, float* map1, float* map2);

/* ---- OpenCV primitive restatements ---- */
int  orc_cv_round(double v);
float orc_fast_atan2(float y, float x);
void orc_remap_bilinear(const uint8_t* src, int sw, int sh, int sstride,
                        const float* map1, const float* map2, int mstride,
                        uint8_t* dst, int dw, int dh, int dstride);
/* System::CvtFisheyeToCubeMap_reverseQuery_withInterpolation (System.cpp:327-355) */
void orc_fisheye_to_cubemap(const orc_camera* cam, const float* map1, const float* map2,
                            const uint8_t* fisheye, int fstride, uint8_t* cubemap, int cstride);
void orc_resize_linear(const uint8_t* src, int sw, int sh, int sstride,
                       uint8_t* dst, int dw, int dh, int dstride);
void orc_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
/* column_mode 0: integer column pass; 1: the SSE2 float column pass of an x86 OpenCV <= 3.2 build (differs on even ties) */
void orc_gaussian_blur7_mode(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, int column_mode);
/* cv::FAST(img, kps, threshold, nonmaxSuppression=true); out triplets (x, y, score) */
int  orc_fast(const uint8_t* img, int w, int h, int stride, int threshold, int* out_xys, int cap);

/* ---- ORBextractor (ORBExtractor.cpp) ---- */
typedef struct orc_orb orc_orb;
orc_orb* orc_orb_create(const orc_orb_params* p);
void orc_orb_set_gaussian_mode(orc_orb* o, int column_mode);
void orc_orb_destroy(orc_orb* o);
int  orc_orb_nlevels(const orc_orb* o);
void orc_orb_tables(const orc_orb* o, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int* features_per_level, int* umax16);
/* operator()(image, mask, keypoints, descriptors)  (ORBExtractor.cpp:838-926). Returns count or <0. */
int  orc_orb_extract(orc_orb* o, const orc_camera* cam, const uint8_t* image, int w, int h, int stride,
                     const uint8_t* mask, int mstride, orc_keypoint* kps, uint8_t* desc, int cap);
/* debug views of the last extract call (for stage-by-stage parity) */
int  orc_orb_level_size(const orc_orb* o, int level, int* w, int* h);
void orc_orb_level_copy(const orc_orb* o, int level, uint8_t* dst, int dstride);
int  orc_orb_level_candidates(const orc_orb* o, int level, int* xys, int cap);      /* vToDistributeKeys, rel. to minBorder */
int  orc_orb_level_distributed(const orc_orb* o, int level, orc_keypoint* kps, int cap); /* after octree + orientation, pre-cull */
/* DistributeOctTree stand-alone (ORBExtractor.cpp:511-737); in/out triplets (x,y,response) */
int  orc_distribute_octree(const int* xys, int n, int min_x, int max_x, int min_y, int max_y, int N, int* out_xys, int cap);

/* ---- Frame grid + window query (Frame.cpp:36-72 AddCells, 158-176 AssignFeaturesToGrid, 251-716 GetFeaturesInArea, 728-744 PosInGrid):
 * nq queries (x, y, r, minLevel, maxLevel) against the n key points of one frame; candidate indices in the reference's order as
 * a CSR list (off[nq+1], idx[cap]).  Returns the total number of candidates (may exceed cap: then only cap were written). */
int  orc_features_in_area(const orc_camera* cam, int n, const float* kx, const float* ky, const int* koct, int nq,
                          const float* qx, const float* qy, const float* qr, const int* qmin, const int* qmax,
                          int* off, int* idx, int cap);

/* ---- track local map: Frame::isInFrustum (Frame.cpp:197-249) + ORBMatcher::SearchByProjection(F, vpMapPoints, th)
 * (ORBMatcher.cpp:50-128); orc_track.cpp.  Rcw (row major), tcw, Ow: the float members Frame::UpdatePoseMatrices leaves;
 * min_dist / max_dist: MapPoint::mfMinDistance / mfMaxDistance (the 0.8 / 1.2 invariance factors are applied inside). */
int  orc_is_in_frustum(const orc_camera* cam, const float* Rcw, const float* tcw, const float* Ow, int n, const float* P,
                       const float* normal, const float* min_dist, const float* max_dist, float viewing_cos_limit,
                       float scale_factor, int nlevels, uint8_t* in_view, float* proj_x, float* proj_y, int* level,
                       float* view_cos);
int  orc_search_local_points(const orc_camera* cam, int nkp, const float* kx, const float* ky, const int* koct,
                             const uint8_t* kdesc, const float* scale_factors, int nmp, const uint8_t* in_view,
                             const float* proj_x, const float* proj_y, const int* level, const float* view_cos,
                             const uint8_t* mp_desc, float th, float nnratio, int th_high, int* kp_mp, int* mp_match);

/* ---- mapping thread, either side of the local BA (orc_tri.cpp): LocalMapping::CreateNewMapPoints (LocalMapping.cpp:209-386),
 * ORBMatcher::SearchForTriangulation (ORBMatcher.cpp:971-1125), search half of ORBMatcher::Fuse (ORBMatcher.cpp:1127-1226).
 * A key frame as these functions see it; node_* is its DBoW2::FeatureVector (node ids ascending, CSR of feature indices). */
typedef struct {
  int n; const orc_keypoint* kps; const uint8_t* desc; const float* rays; const int* mp;   /* mp[i] < 0: no map point */
  float Rcw[9], tcw[3], Ow[3];
  int nnodes; const int* node_id; const int* node_off; const int* node_feat;
  float median_depth;                                                                      /* KeyFrame::ComputeSceneMedianDepth(2) */
} orc_keyframe;
void orc_compute_e12(const float* R1w, const float* t1w, const float* R2w, const float* t2w, float* E12);
int  orc_search_for_triangulation(const orc_camera* cam, const orc_keyframe* kf1, const orc_keyframe* kf2, const float* E12,
                                  const float* scale_factors, const float* level_sigma2, int check_orientation, int* matches12);
int  orc_triangulate_match(const orc_camera* cam, const orc_keyframe* kf1, const orc_keyframe* kf2, int idx1, int idx2,
                           const float* scale_factors, const float* level_sigma2, float ratio_factor, float* x3d_out);
int  orc_create_new_map_points(const orc_camera* cam, const orc_keyframe* cur, int nneigh, const orc_keyframe* neigh,
                               const float* scale_factors, const float* level_sigma2, int* cur_mp_inout, int* out_neigh,
                               int* out_idx1, int* out_idx2, float* out_x3d, int cap);
void orc_fuse_search(const orc_camera* cam, const orc_keyframe* kf, int nmp, const uint8_t* skip, const float* P, const float* normal,
                     const float* min_dist, const float* max_dist, const uint8_t* mp_desc, float th, const float* scale_factors,
                     const float* inv_level_sigma2, int nlevels, int* best_idx, int* best_dist);

/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cpp:243-308) and MapPoint::UpdateNormalAndDepth (:332-373), batched over map points */
void orc_distinctive_descriptors(int npts, const int* obs_off, const uint8_t* desc, int* best_idx);
void orc_update_normal_and_depth(int npts, const int* obs_off, const float* pos, const float* obs_Ow, const float* ref_Ow,
                                 const int* ref_level, const float* scale_factors, int nlevels, float* normal, float* min_dist,
                                 float* max_dist);

/* ORBMatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORBMatcher.cpp:130-251): projection of the last frame's map points with
 * the current pose, windows, greedy best match, rotation-consistency filter; orc_track.cpp */
int  orc_search_by_projection_frames(const orc_camera* cam, const float* Rcw, const float* tcw, int nkp, const float* kx, const float* ky,
                                     const int* koct, const float* kangle, const uint8_t* kdesc, const float* scale_factors, int nlast,
                                     const uint8_t* valid, const float* Xw, const int* loct, const float* langle, const uint8_t* mp_desc,
                                     float th, int check_orientation, int th_high, int* kp_mp, int* match);
/* ORBMatcher::SearchForInitialization (ORBMatcher.cpp:676-794); orc_track.cpp */
int  orc_search_for_initialization(const orc_camera* cam, int n1, const float* k1x, const float* k1y, const int* k1oct, const float* k1angle,
                                   const uint8_t* desc1, int n2, const float* k2x, const float* k2y, const int* k2oct, const float* k2angle,
                                   const uint8_t* desc2, float* prev_matched, int window_size, float nnratio, int check_orientation,
                                   int* matches12);

/* ---- ORBMatcher (ORBMatcher.cpp) ---- */
int  orc_descriptor_distance(const uint8_t* a, const uint8_t* b);
/* best / second-best over CSR candidate lists, the inner loop of SearchByProjection (ORBMatcher.cpp:84-113) */
void orc_hamming_best2(const uint8_t* qdesc, int nq, const uint8_t* tdesc, const int* cand_off, const int* cand_idx,
                       const int* tlevel, int* best_idx, int* best_dist, int* best_level, int* second_dist, int* second_level);
void orc_hamming_matrix(const uint8_t* a, int na, const uint8_t* b, int nb, uint16_t* out);

/* ---- Local bundle adjustment (Optimizer.cpp:192-451 + g2o slice) ---- */
typedef struct {
  int iterations_done[2];
  double chi2_initial[2], chi2_final[2], lambda_final[2];
  int n_outliers_mid, n_outliers_final;
} orc_ba_stats;
/* poses: K x 7 (tx,ty,tz,qx,qy,qz,qw), world->camera; points P x 3; edges: pose idx, point idx, obs in face (u,v),
 * invSigma2, face. its = {5,10}. outlier_flags[E]: bit0 set = erased by the final test. */
int  orc_ba_run(int K, double* poses, const uint8_t* fixed, int P, double* points,
                int E, const int* e_pose, const int* e_point, const double* e_obs, const double* e_invsig2,
                const int8_t* e_face, double fx, double fy, double cx, double cy,
                int its_robust, int its_final, const volatile uint8_t* stop,
                uint8_t* outlier_flags, orc_ba_stats* stats);
/* test hook: the stop flag is raised while trial number stop_after_trials (1-based, over both stages) runs */
int  orc_ba_run_stop_after(int K, double* poses, const uint8_t* fixed, int P, double* points,
                int E, const int* e_pose, const int* e_point, const double* e_obs, const double* e_invsig2,
                const int8_t* e_face, double fx, double fy, double cx, double cy,
                int its_robust, int its_final, int stop_after_trials,
                uint8_t* outlier_flags, orc_ba_stats* stats);
/* one residual + linearisation pass: per-edge error/chi2/Jacobians and the accumulated blocks.
 * Hpp: K x 36 (row major 6x6), bp: K x 6, Hll: P x 9, bl: P x 3, Hpl: E x 18 (6x3 row major per edge), robust=huber */
void orc_ba_linearize(int K, const double* poses, const uint8_t* fixed, int P, const double* points,
                      int E, const int* e_pose, const int* e_point, const double* e_obs, const double* e_invsig2,
                      const int8_t* e_face, double fx, double fy, double cx, double cy, int robust, double huber_delta,
                      double* err, double* chi2, double* Jpose, double* Jpoint,
                      double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, double* robust_chi2_sum);
void orc_se3_exp_apply(const double* upd6, double* pose7);

/* ---- Optimizer::PoseOptimization (Optimizer.cpp:48-190): pose-only LM over n unary multi-pinhole edges, 4 rounds x 10
 * iterations with outlier re-classification.  Xw n x 3 (world), obs_uv n x 2 (measurement in its face), face ids, pose7 in/out.
 * Returns the number of inliers (nInitialCorrespondences - nBad), 0 if n < 3 (pose untouched). */
typedef struct { int rounds, n_bad; int iterations_done[4]; double chi2_final[4]; } orc_pose_stats;
int  orc_pose_optimize(int n, const double* Xw, const double* obs_uv, const double* inv_sigma2, const int8_t* face,
                       double fx, double fy, double cx, double cy, double* pose7, uint8_t* outlier, orc_pose_stats* stats);

#ifdef __cplusplus
}
#endif
#endif
