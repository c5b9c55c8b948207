// CubemapHipBridge.cpp -- see CubemapHipBridge.h.  Compiled inside the reference's tree (OpenCV, Eigen, g2o types); every function gathers
// what the reference function reads into flat arrays, makes ONE call into libcubemapslam_hip.so and writes the result back through the
// reference's own setters, in the reference's order.  Not built in this repository (integration/README.md; syntax-checked by tests/test_integration_syntax.py); the same logic over plain
// structs is cubemapslam_amd/host/cubemap_hot_path.cpp, which is built and tested here.
#include "CubemapHipBridge.h"

#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>

#include "CamModelGeneral.h"
#include "Converter.h"
#include "Frame.h"
#include "KeyFrame.h"
#include "Map.h"
#include "MapPoint.h"
#include "ORBMatcher.h"

namespace Hip {
namespace {
double g_fov_deg = 190.0;
int g_device = 0;

void check(int rc, const char* what) {
  if (rc < 0) throw std::runtime_error(std::string(what) + ": " + cms_last_error());
}
void pose7(const cv::Mat& Tcw, double* p) {                       // Converter::toSE3Quat (Converter.cpp:41-51): t then unit quaternion
  const g2o::SE3Quat T = Converter::toSE3Quat(Tcw);
  const Eigen::Vector3d t = T.translation();
  const Eigen::Quaterniond q = T.rotation();
  p[0] = t[0]; p[1] = t[1]; p[2] = t[2]; p[3] = q.x(); p[4] = q.y(); p[5] = q.z(); p[6] = q.w();
}
cv::Mat toMat(const double* p) {                                   // Converter::toCvMat(SE3Quat) (Converter.cpp:53-104): through float
  return Converter::toCvMat(g2o::SE3Quat(Eigen::Quaterniond(p[6], p[3], p[4], p[5]), Eigen::Vector3d(p[0], p[1], p[2])));
}
void kps_to_abi(const std::vector<cv::KeyPoint>& in, std::vector<cms_keypoint>& out) {
  out.resize(in.size());
  for (size_t i = 0; i < in.size(); ++i) out[i] = {in[i].pt.x, in[i].pt.y, in[i].size, in[i].angle, in[i].response, in[i].octave};
}
}  // namespace

void Configure(double camFovDeg, int device) { g_fov_deg = camFovDeg; g_device = device; }   // System::System, next to SetCosFovTh (System.cpp:86-89)

cms_ctx* CreateContext(const cms_orb_params& orb) {
  CamModelGeneral* cam = CamModelGeneral::GetCamera();
  cms_camera c;
  std::memset(&c, 0, sizeof(c));
  c.c = cam->Get_c(); c.d = cam->Get_d(); c.e = cam->Get_e(); c.u0 = cam->Get_u0(); c.v0 = cam->Get_v0();
  const cv::Mat_<double> invP = cam->Get_invP(), P = cam->Get_P();
  for (int i = 0; i < 12 && i < invP.rows; ++i) c.invpol[i] = invP(i, 0);          // zero padded to 12 (System.cpp:70-72)
  for (int i = 0; i < 5 && i < P.rows; ++i) c.pol[i] = P(i, 0);
  c.Iw = cam->GetFisheyeWidth(); c.Ih = cam->GetFisheyeHeight(); c.face = cam->GetCubeFaceWidth(); c.fov_deg = g_fov_deg;
  cms_ctx* ctx = nullptr;
  check(cms_ctx_create(&ctx, g_device, &c, &orb, 1), "cms_ctx_create");
  check(cms_set_distance_bounds_mode(ctx, 1), "cms_set_distance_bounds_mode");     // map points' bounds come from MapPoint's public getters
  // Which 8-bit GaussianBlur (ORBExtractor.cpp:907-908)?  The reference links OpenCV 2.4.11 / 3.2 (README.md:59); on x86 those run the SSE2 column
  // functor (float sums, ties to even) -- definition 1.  A build of the reference on a machine without SSE2 (or against an OpenCV built with
  // -DENABLE_SSE2=OFF) runs the integer column pass: define CUBEMAP_HIP_INTEGER_GAUSSIAN for that one (definition 0, the library's own default).
#ifdef CUBEMAP_HIP_INTEGER_GAUSSIAN
  check(cms_set_gaussian_mode(ctx, 0), "cms_set_gaussian_mode");
#else
  check(cms_set_gaussian_mode(ctx, 1), "cms_set_gaussian_mode");
#endif
  return ctx;
}

void CvtFisheyeToCubeMap(cms_ctx* ctx, cv::Mat& cubemapImg, const cv::Mat& fisheyeImg) {
  check(cms_remap(ctx, fisheyeImg.data, (int)fisheyeImg.step, cubemapImg.data, (int)cubemapImg.step), "cms_remap");   // corner blocks untouched
}

void FrameGrid(cms_ctx* ctx) { check(cms_area_grid(ctx, 1), "cms_area_grid"); }

int SearchForInitialization(cms_ctx* ctx, Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize,
                            float nnratio, bool checkOrientation) {
  std::vector<cms_keypoint> k1;
  kps_to_abi(F1.mvKeys, k1);
  vnMatches12.assign(F1.mvKeys.size(), -1);
  int n = 0;
  // cv::Point2f is two floats: vbPrevMatched is the n1 x 2 float array the entry updates in place (ORBMatcher.cpp:786-789)
  check(cms_search_for_initialization(ctx, 0, (int)k1.size(), k1.data(), F1.mDescriptors.data, reinterpret_cast<float*>(vbPrevMatched.data()), windowSize, nnratio,
                                      checkOrientation ? 1 : 0, vnMatches12.data(), &n), "cms_search_for_initialization");
  (void)F2;                                                         // F2 is the frame ctx extracted last: already on the device
  return n;
}

int SearchByProjection(cms_ctx* ctx, Frame& CurrentFrame, const Frame& LastFrame, float th, bool checkOrientation) {
  const int nl = LastFrame.N;
  std::vector<uint8_t> valid(nl, 0), desc((size_t)nl * 32, 0);
  std::vector<float> Xw((size_t)nl * 3, 0.f), angle(nl);
  std::vector<int> octave(nl);
  for (int i = 0; i < nl; ++i) {
    MapPoint* pMP = LastFrame.mvpMapPoints[i];
    octave[i] = LastFrame.mvKeys[i].octave; angle[i] = LastFrame.mvKeys[i].angle;
    if (!pMP || LastFrame.mvbOutlier[i]) continue;                 // ORBMatcher.cpp:153-157
    valid[i] = 1;
    const cv::Mat x = pMP->GetWorldPos();
    for (int k = 0; k < 3; ++k) Xw[3 * i + k] = x.at<float>(k);
    const cv::Mat d = pMP->GetDescriptor();
    std::memcpy(&desc[(size_t)i * 32], d.data, 32);
  }
  float pose12[12];
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) pose12[3 * r + c] = CurrentFrame.mTcw.at<float>(r, c); pose12[9 + r] = CurrentFrame.mTcw.at<float>(r, 3); }
  const int N = CurrentFrame.N;
  std::vector<int> kp_mp(N), match(nl, -1);
  for (int i = 0; i < N; ++i) {                                    // a key point that already holds a map point WITH observations is skipped (:191-193)
    MapPoint* p = CurrentFrame.mvpMapPoints[i];
    kp_mp[i] = (p && p->Observations() > 0) ? (1 << 30) : -1;
  }
  int n = 0;
  check(cms_search_by_projection(ctx, 0, pose12, nl, valid.data(), Xw.data(), octave.data(), angle.data(), desc.data(), th, checkOrientation ? 1 : 0,
                                 ORBMatcher::TH_HIGH, N, kp_mp.data(), match.data(), &n), "cms_search_by_projection");
  for (int i = 0; i < nl; ++i) if (match[i] >= 0) CurrentFrame.mvpMapPoints[match[i]] = LastFrame.mvpMapPoints[i];
  return n;
}

int SearchLocalPoints(cms_ctx* ctx, Frame& F, const std::vector<MapPoint*>& vpMapPoints, float th) {
  // the caller (Tracking::SearchLocalPoints, Tracking.cpp:797-822) has removed the points already matched in the frame and the bad ones
  const int n = (int)vpMapPoints.size();
  std::vector<float> pos((size_t)n * 3), nrm((size_t)n * 3), dmin(n), dmax(n), px(n), py(n), vc(n);
  std::vector<uint8_t> desc((size_t)n * 32), inview(n);
  std::vector<int> level(n), match(n);
  for (int i = 0; i < n; ++i) {
    MapPoint* p = vpMapPoints[i];
    const cv::Mat x = p->GetWorldPos(), nn = p->GetNormal();
    for (int k = 0; k < 3; ++k) { pos[3 * i + k] = x.at<float>(k); nrm[3 * i + k] = nn.at<float>(k); }
    dmin[i] = p->GetMinDistanceInvariance();                       // the PUBLIC getters (MapPoint.cpp:375-385: 0.8f / 1.2f applied); the context was put
    dmax[i] = p->GetMaxDistanceInvariance();                       // into cms_set_distance_bounds_mode(ctx, 1) by the bridge's set-up: MapPoint.h stays untouched
    const cv::Mat d = p->GetDescriptor();
    std::memcpy(&desc[(size_t)i * 32], d.data, 32);
  }
  float pose15[15];
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) pose15[3 * r + c] = F.mTcw.at<float>(r, c); pose15[9 + r] = F.mTcw.at<float>(r, 3); }
  const cv::Mat Ow = F.GetCameraCenter();
  for (int k = 0; k < 3; ++k) pose15[12 + k] = Ow.at<float>(k);
  std::vector<int> kp_mp(F.N);
  for (int i = 0; i < F.N; ++i) { MapPoint* p = F.mvpMapPoints[i]; kp_mp[i] = (p && p->Observations() > 0) ? (1 << 30) : -1; }   // ORBMatcher.cpp:91-95
  int nm = 0;
  check(cms_search_local_points(ctx, 0, pose15, n, pos.data(), nrm.data(), dmin.data(), dmax.data(), desc.data(), 0.5f, th, 0.8f, ORBMatcher::TH_HIGH, F.N,
                                kp_mp.data(), inview.data(), px.data(), py.data(), level.data(), vc.data(), match.data(), &nm, nullptr), "cms_search_local_points");
  for (int i = 0; i < n; ++i) {                                    // what Frame::isInFrustum leaves on the MapPoint (Frame.cpp:199-248) ...
    MapPoint* p = vpMapPoints[i];
    p->mbTrackInView = inview[i] != 0;
    if (!inview[i]) continue;
    p->mTrackProjX = px[i]; p->mTrackProjY = py[i]; p->mnTrackScaleLevel = level[i]; p->mTrackViewCos = vc[i];
    p->IncreaseVisible();                                          // Tracking.cpp:829
    if (match[i] >= 0) F.mvpMapPoints[match[i]] = p;               // ... and what the matcher leaves in the frame (ORBMatcher.cpp:121)
  }
  return nm;
}

int PoseOptimization(Frame* pFrame) {
  CamModelGeneral* cam = CamModelGeneral::GetCamera();
  const int N = pFrame->N;
  std::vector<double> Xw, obs, inv;
  std::vector<int8_t> face;
  std::vector<int> index;
  {
    std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex);
    for (int i = 0; i < N; ++i) {
      if (pFrame->mvKeyRays[i](2) < cam->GetCosFovTh()) continue;  // Optimizer.cpp:84-86
      MapPoint* pMP = pFrame->mvpMapPoints[i];
      if (!pMP) continue;
      pFrame->mvbOutlier[i] = false;
      const cv::KeyPoint& kp = pFrame->mvKeys[i];
      double u, v;
      cam->GetPosInFace(u, v, (double)kp.pt.x, (double)kp.pt.y);
      obs.push_back(u); obs.push_back(v);
      face.push_back((int8_t)cam->FaceInCubemap(kp.pt));
      inv.push_back(pFrame->mvInvLevelSigma2[kp.octave]);
      const cv::Mat x = pMP->GetWorldPos();
      for (int k = 0; k < 3; ++k) Xw.push_back(x.at<float>(k));
      index.push_back(i);
    }
  }
  const int n = (int)index.size();
  if (n < 3) return 0;                                             // Optimizer.cpp:133-134
  double p[7];
  pose7(pFrame->mTcw, p);
  std::vector<uint8_t> outlier(n, 0);
  int n_in = 0;
  check(cms_pose_optimize(g_device, n, Xw.data(), obs.data(), inv.data(), face.data(), cam->Get_fx(), cam->Get_fy(), cam->Get_cx(), cam->Get_cy(), p, outlier.data(),
                          &n_in, nullptr), "cms_pose_optimize");
  for (int j = 0; j < n; ++j) pFrame->mvbOutlier[index[j]] = outlier[j] != 0;
  pFrame->SetPose(toMat(p));                                       // Optimizer.cpp:184-187
  return n_in;                                                     // nInitialCorrespondences - nBad
}

void LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap) {
  CamModelGeneral* cam = CamModelGeneral::GetCamera();
  // ---- the window: local key frames, the map points they see, fixed key frames that see those (Optimizer.cpp:194-245)
  std::vector<KeyFrame*> kfs;                                      // local first, then fixed
  std::map<KeyFrame*, int> kf_index;
  auto add_kf = [&](KeyFrame* k) { kf_index[k] = (int)kfs.size(); kfs.push_back(k); };
  add_kf(pKF);
  pKF->mnBALocalForKF = pKF->mnId;
  for (KeyFrame* k : pKF->GetVectorCovisibleKeyFrames()) { k->mnBALocalForKF = pKF->mnId; if (!k->isBad()) add_kf(k); }
  const int n_local = (int)kfs.size();
  std::vector<MapPoint*> mps;
  for (int j = 0; j < n_local; ++j)
    for (MapPoint* p : kfs[j]->GetMapPointMatches())
      if (p && !p->isBad() && p->mnBALocalForKF != pKF->mnId) { mps.push_back(p); p->mnBALocalForKF = pKF->mnId; }
  for (MapPoint* p : mps)
    for (const auto& ob : p->GetObservations()) {
      KeyFrame* k = ob.first;
      if (k->mnBALocalForKF != pKF->mnId && k->mnBAFixedForKF != pKF->mnId) { k->mnBAFixedForKF = pKF->mnId; if (!k->isBad()) add_kf(k); }
    }
  // ---- flat problem (what :262-357 hands g2o): poses, fixed flags (fixed cameras and key frame 0), points, one edge per observation
  const int K = (int)kfs.size(), P = (int)mps.size();
  std::vector<double> poses((size_t)K * 7), points((size_t)P * 3), e_obs, e_inv;
  std::vector<uint8_t> fixed(K);
  std::vector<int> e_pose, e_point;
  std::vector<int8_t> e_face;
  for (int j = 0; j < K; ++j) { pose7(kfs[j]->GetPose(), &poses[(size_t)j * 7]); fixed[j] = (j >= n_local || kfs[j]->mnId == 0) ? 1 : 0; }
  for (int i = 0; i < P; ++i) {
    const cv::Mat x = mps[i]->GetWorldPos();
    for (int k = 0; k < 3; ++k) points[(size_t)i * 3 + k] = x.at<float>(k);
    for (const auto& ob : mps[i]->GetObservations()) {
      KeyFrame* k = ob.first;
      if (k->isBad()) continue;
      if (k->mvKeyRays[ob.second](2) < cam->GetCosFovTh()) continue;            // :323-325
      const cv::KeyPoint& kp = k->mvKeys[ob.second];
      double u, v;
      cam->GetPosInFace(u, v, (double)kp.pt.x, (double)kp.pt.y);
      e_pose.push_back(kf_index[k]); e_point.push_back(i); e_obs.push_back(u); e_obs.push_back(v);
      e_inv.push_back(k->mvInvLevelSigma2[kp.octave]); e_face.push_back((int8_t)cam->FaceInCubemap(kp.pt));
    }
  }
  const int E = (int)e_pose.size();
  if (E == 0) return;
  std::vector<uint8_t> erase(E, 0);
  cms_ba_stats st;
  const int rc = cms_ba_run(g_device, K, poses.data(), fixed.data(), P, points.data(), E, e_pose.data(), e_point.data(), e_obs.data(), e_inv.data(), e_face.data(),
                            cam->Get_fx(), cam->Get_fy(), cam->Get_cx(), cam->Get_cy(), 5, 10, reinterpret_cast<const volatile uint8_t*>(pbStopFlag), erase.data(), &st);
  check(rc, "cms_ba_run");
  if (rc == 1) return;                                              // stop requested before the optimisation started (:359-361)
  // ---- write-back under the map mutex (:419-450)
  std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
  for (int e = 0; e < E; ++e) {
    if (!erase[e]) continue;
    MapPoint* p = mps[e_point[e]];
    if (p->isBad()) continue;
    KeyFrame* k = kfs[e_pose[e]];
    k->EraseMapPointMatch(p);
    p->EraseObservation(k);
  }
  for (int j = 0; j < n_local; ++j) kfs[j]->SetPose(toMat(&poses[(size_t)j * 7]));
  for (int i = 0; i < P; ++i) {
    cv::Mat x(3, 1, CV_32F);
    for (int k = 0; k < 3; ++k) x.at<float>(k) = (float)points[(size_t)i * 3 + k];
    mps[i]->SetWorldPos(x);
    mps[i]->UpdateNormalAndDepth();
  }
}
}  // namespace Hip
