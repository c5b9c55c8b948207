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

void SetDeterministic(bool on) { cms_ba_set_deterministic(on ? 64 : 0); }      // (64 workgroups per window: this bridge optimises one window per call)
bool GetDeterministic() { return cms_ba_get_deterministic() != 0; }

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

// ------------------------------------------------------------------------------------------------ LocalMapping on resident key frames
namespace {
struct StoreBook { std::map<KeyFrame*, int> slot; std::vector<int> free_slots; int max_features = 0; };
std::map<cms_kfstore*, StoreBook> g_books;
std::mutex g_books_mutex;
StoreBook& book(cms_kfstore* st) { std::lock_guard<std::mutex> lk(g_books_mutex); return g_books[st]; }
int slot_of(StoreBook& b, KeyFrame* k) { const auto it = b.slot.find(k); return it == b.slot.end() ? -1 : it->second; }
void pose_floats(KeyFrame* k, float* R9, float* t3, float* O3) {
  const cv::Mat R = k->GetRotation(), t = k->GetTranslation(), O = k->GetCameraCenter();
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R9[3 * r + c] = R.at<float>(r, c); t3[r] = t.at<float>(r); O3[r] = O.at<float>(r); }
}
// the map-point slot per key point as the device holds it: >= 0 where the key point holds a point that is not bad (the id only has to be >= 0)
void mp_slots(KeyFrame* k, std::vector<int>& mp) {
  const std::vector<MapPoint*> v = k->GetMapPointMatches();
  mp.assign(v.size(), -1);
  for (size_t i = 0; i < v.size(); ++i) if (v[i] && !v[i]->isBad()) mp[i] = (int)(v[i]->mnId & 0x3FFFFFFF);
}
// a map point as the Fuse search reads it (ORBMatcher.cpp:1140-1175)
void push_point(MapPoint* p, std::vector<float>& pos, std::vector<float>& nrm, std::vector<float>& dmin, std::vector<float>& dmax, std::vector<uint8_t>& desc) {
  const cv::Mat x = p->GetWorldPos(), n = p->GetNormal(), d = p->GetDescriptor();
  for (int k = 0; k < 3; ++k) { pos.push_back(x.at<float>(k)); nrm.push_back(n.at<float>(k)); }
  dmin.push_back(p->GetMinDistanceInvariance()); dmax.push_back(p->GetMaxDistanceInvariance());      // (the store's context runs in distance-bounds mode 1, see CreateContext)
  desc.insert(desc.end(), d.data, d.data + 32);
}
}  // namespace

cms_kfstore* CreateKeyFrameStore(cms_ctx* mappingCtx, int maxKeyFrames, int maxFeatures) {
  cms_kfstore* st = nullptr;
  check(cms_kfstore_create(&st, mappingCtx, maxKeyFrames, maxFeatures, 4096), "cms_kfstore_create");
  StoreBook& b = book(st);
  b.max_features = maxFeatures;
  for (int s = maxKeyFrames - 1; s >= 0; --s) b.free_slots.push_back(s);
  return st;
}

int ProcessNewKeyFrame(cms_kfstore* store, cms_ctx* frameCtx, KeyFrame* pKF) {
  StoreBook& b = book(store);
  int slot = slot_of(b, pKF);
  if (slot < 0) {
    if (b.free_slots.empty()) throw std::runtime_error("Hip::ProcessNewKeyFrame: the key-frame store is full");
    slot = b.free_slots.back(); b.free_slots.pop_back(); b.slot[pKF] = slot;
  }
  // mFeatVec (KeyFrame::ComputeBoW, LocalMapping.cpp:62): node id -> feature indices, std::map order
  std::vector<int> node_id, node_off(1, 0), node_feat, mp;
  for (const auto& nf : pKF->mFeatVec) {
    node_id.push_back((int)nf.first);
    for (unsigned f : nf.second) node_feat.push_back((int)f);
    node_off.push_back((int)node_feat.size());
  }
  mp_slots(pKF, mp);
  float R[9], t[3], O[3];
  pose_floats(pKF, R, t, O);
  check(cms_kfstore_put_from_frame(store, slot, frameCtx, 0, pKF->N, R, t, O, pKF->ComputeSceneMedianDepth(2), mp.data(), (int)node_id.size(), node_id.data(),
                                   node_off.data(), node_feat.data()), "cms_kfstore_put_from_frame");
  return slot;
}

void ReleaseKeyFrame(cms_kfstore* store, KeyFrame* pKF) {
  StoreBook& b = book(store);
  const auto it = b.slot.find(pKF);
  if (it == b.slot.end()) return;
  b.free_slots.push_back(it->second);
  b.slot.erase(it);
}

int CreateNewMapPoints(cms_kfstore* store, KeyFrame* pCurrentKF, Map* pMap, std::vector<MapPoint*>& newPoints) {
  StoreBook& b = book(store);
  const int cur = slot_of(b, pCurrentKF);
  if (cur < 0) throw std::runtime_error("Hip::CreateNewMapPoints: the current key frame is not resident (Hip::ProcessNewKeyFrame first)");
  std::vector<KeyFrame*> neigh;                                    // GetBestCovisibilityKeyFrames(20), the resident ones, in covisibility order
  std::vector<int> neigh_slot;
  for (KeyFrame* k : pCurrentKF->GetBestCovisibilityKeyFrames(20)) {
    const int s = slot_of(b, k);
    if (s < 0 || k->isBad()) continue;
    // poses, median depths and map-point slots may have changed since the key frame entered the store (local BA, Fuse, culling)
    float R[9], t[3], O[3]; std::vector<int> mp;
    pose_floats(k, R, t, O); mp_slots(k, mp);
    const float md = k->ComputeSceneMedianDepth(2);
    check(cms_kfstore_update(store, s, R, t, O, &md, mp.data()), "cms_kfstore_update");
    neigh.push_back(k); neigh_slot.push_back(s);
  }
  {
    float R[9], t[3], O[3]; std::vector<int> mp;
    pose_floats(pCurrentKF, R, t, O); mp_slots(pCurrentKF, mp);
    check(cms_kfstore_update(store, cur, R, t, O, nullptr, mp.data()), "cms_kfstore_update");
  }
  if (neigh.empty()) return 0;
  const int cap = pCurrentKF->N;
  const int neigh_off[2] = {0, (int)neigh.size()};
  int n_new = 0;
  std::vector<int> o_neigh(cap), o_idx1(cap), o_idx2(cap);
  std::vector<float> o_x((size_t)cap * 3);
  check(cms_kfstore_create_new_map_points(store, 1, &cur, neigh_off, neigh_slot.data(), 0, cap, &n_new, o_neigh.data(), o_idx1.data(), o_idx2.data(), o_x.data()),
        "cms_kfstore_create_new_map_points");
  for (int i = 0; i < n_new; ++i) {                                // LocalMapping.cpp:359-381, in the reference's creation order
    KeyFrame* pKF2 = neigh[o_neigh[i]];
    cv::Mat x3D(3, 1, CV_32F);
    for (int k = 0; k < 3; ++k) x3D.at<float>(k) = o_x[(size_t)i * 3 + k];
    MapPoint* pMP = new MapPoint(x3D, pCurrentKF, pMap);
    pMP->AddObservation(pCurrentKF, o_idx1[i]);
    pMP->AddObservation(pKF2, o_idx2[i]);
    pCurrentKF->AddMapPoint(pMP, o_idx1[i]);
    pKF2->AddMapPoint(pMP, o_idx2[i]);
    pMP->ComputeDistinctiveDescriptors();
    pMP->UpdateNormalAndDepth();
    pMap->AddMapPoint(pMP);
    newPoints.push_back(pMP);
  }
  return n_new;
}

void SearchInNeighbors(cms_kfstore* store, KeyFrame* pCurrentKF) {
  StoreBook& b = book(store);
  const int cur = slot_of(b, pCurrentKF);
  if (cur < 0) throw std::runtime_error("Hip::SearchInNeighbors: the current key frame is not resident");
  // ---- target key frames: neighbours and second neighbours (LocalMapping.cpp:391-412), the resident ones
  std::vector<KeyFrame*> targets;
  for (KeyFrame* k : pCurrentKF->GetBestCovisibilityKeyFrames(20)) {
    if (k->isBad() || k->mnFuseTargetForKF == pCurrentKF->mnId) continue;
    targets.push_back(k);
    k->mnFuseTargetForKF = pCurrentKF->mnId;
    for (KeyFrame* k2 : k->GetBestCovisibilityKeyFrames(5)) {
      if (k2->isBad() || k2->mnFuseTargetForKF == pCurrentKF->mnId || k2->mnId == pCurrentKF->mnId) continue;
      targets.push_back(k2);
    }
  }
  std::vector<KeyFrame*> tk;
  for (KeyFrame* k : targets) if (slot_of(b, k) >= 0) tk.push_back(k);
  if (tk.empty()) return;
  // ---- set 0: the current key frame's map points (into every target); set 1: the targets' map points, first appearance (into the current key frame)
  std::vector<MapPoint*> set0, set1;
  for (MapPoint* p : pCurrentKF->GetMapPointMatches()) if (p) set0.push_back(p);            // (ORBMatcher::Fuse skips null / bad points itself: the skip flags below)
  for (KeyFrame* k : tk)
    for (MapPoint* p : k->GetMapPointMatches()) {
      if (!p || p->isBad() || p->mnFuseCandidateForKF == pCurrentKF->mnId) continue;        // LocalMapping.cpp:432-447
      p->mnFuseCandidateForKF = pCurrentKF->mnId;
      set1.push_back(p);
    }
  std::vector<float> pos, nrm, dmin, dmax;
  std::vector<uint8_t> desc;
  for (MapPoint* p : set0) push_point(p, pos, nrm, dmin, dmax, desc);
  for (MapPoint* p : set1) push_point(p, pos, nrm, dmin, dmax, desc);
  const int set_off[3] = {0, (int)set0.size(), (int)(set0.size() + set1.size())};
  // ---- jobs: set 0 into every target, set 1 into the current key frame; an entry is skipped like ORBMatcher.cpp:1143-1147 skips it
  std::vector<int> job_slot, job_set;
  std::vector<uint8_t> skip;
  for (KeyFrame* k : tk) {
    job_slot.push_back(slot_of(b, k)); job_set.push_back(0);
    for (MapPoint* p : set0) skip.push_back((p->isBad() || p->IsInKeyFrame(k)) ? 1 : 0);
  }
  job_slot.push_back(cur); job_set.push_back(1);
  for (MapPoint* p : set1) skip.push_back((p->isBad() || p->IsInKeyFrame(pCurrentKF)) ? 1 : 0);
  // the searched key frames' map-point slots as they are NOW (the search only needs the key points; the decisions below read the live objects)
  std::vector<int> best_idx(skip.size()), best_dist(skip.size());
  check(cms_kfstore_fuse_search_sets(store, 2, set_off, pos.data(), nrm.data(), dmin.data(), dmax.data(), desc.data(), (int)job_slot.size(), job_slot.data(),
                                     job_set.data(), skip.data(), 3.0f, best_idx.data(), best_dist.data()), "cms_kfstore_fuse_search_sets");
  // ---- the decisions, job after job and point after point in the reference's order (ORBMatcher.cpp:1213-1236).  A point that an earlier job of this
  // call added to / replaced in a later job's key frame is re-checked here exactly like the reference's sequential Fuse calls would see it
  size_t e = 0;
  for (size_t j = 0; j < job_slot.size(); ++j) {
    KeyFrame* pKF = j < tk.size() ? tk[j] : pCurrentKF;
    const std::vector<MapPoint*>& pts = j < tk.size() ? set0 : set1;
    for (size_t i = 0; i < pts.size(); ++i, ++e) {
      MapPoint* pMP = pts[i];
      if (best_idx[e] < 0 || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
      MapPoint* pMPinKF = pKF->GetMapPoint(best_idx[e]);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) {
          if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
          else pMPinKF->Replace(pMP);
        }
      } else {
        pMP->AddObservation(pKF, best_idx[e]);
        pKF->AddMapPoint(pMP, best_idx[e]);
      }
    }
  }
  // ---- update points and connections (LocalMapping.cpp:449-465)
  for (MapPoint* p : pCurrentKF->GetMapPointMatches())
    if (p && !p->isBad()) { p->ComputeDistinctiveDescriptors(); p->UpdateNormalAndDepth(); }
  pCurrentKF->UpdateConnections();
}

void UpdateKeyFramePoses(cms_kfstore* store, const std::vector<KeyFrame*>& vpKFs) {
  StoreBook& b = book(store);
  std::vector<int> slots;
  std::vector<float> R, t, O;
  for (KeyFrame* k : vpKFs) {
    const int s = slot_of(b, k);
    if (s < 0) continue;
    float r9[9], t3[3], o3[3];
    pose_floats(k, r9, t3, o3);
    slots.push_back(s); R.insert(R.end(), r9, r9 + 9); t.insert(t.end(), t3, t3 + 3); O.insert(O.end(), o3, o3 + 3);
  }
  if (!slots.empty()) check(cms_kfstore_update_poses(store, (int)slots.size(), slots.data(), R.data(), t.data(), O.data()), "cms_kfstore_update_poses");
}
}  // namespace Hip
