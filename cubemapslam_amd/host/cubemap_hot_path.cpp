// cubemap_hot_path.cpp -- host-side mirror of the reference interfaces over the C-ABI (see cubemap_hot_path.h).
#include "cubemap_hot_path.h"
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <fstream>
#include <iomanip>
#include <stdexcept>

namespace CubemapSLAM {

// ------------------------------------------------------------------------------------------------ camera singleton
CamModelGeneral* CamModelGeneral::GetCamera() {
  static CamModelGeneral* cam = new CamModelGeneral();   // CamModelGeneral.cpp:31-38
  return cam;
}
void CamModelGeneral::SetCamParams(const double cdeu0v0[5], const std::vector<double>& poly, const std::vector<double>& invpoly,
                                   double Iw, double Ih, double, double, double, double, double width, double height, double camFov) {
  std::memset(&cam_, 0, sizeof(cam_));
  cam_.c = cdeu0v0[0]; cam_.d = cdeu0v0[1]; cam_.e = cdeu0v0[2]; cam_.u0 = cdeu0v0[3]; cam_.v0 = cdeu0v0[4];
  for (size_t i = 0; i < invpoly.size() && i < 12; ++i) cam_.invpol[i] = invpoly[i];
  for (size_t i = 0; i < poly.size() && i < 5; ++i) cam_.pol[i] = poly[i];
  cam_.Iw = (int)Iw; cam_.Ih = (int)Ih;
  if ((int)width != (int)height) throw std::runtime_error("CubeFace.w must equal CubeFace.h");
  cam_.face = (int)width;
  cam_.fov_deg = camFov;
  const float fov = (float)camFov;
  cosFovTh_ = cosf(fov / 2 * (3.1415926535897932384626f / 180));   // CamModelGeneral.h:224-229
  configured_ = true;
}
CamModelGeneral::eFace CamModelGeneral::FaceInCubemap(const cv::Point2f& pixel) const {
  const double i = pixel.x / (float)cam_.face, j = pixel.y / (float)cam_.face;
  if (i >= 0 && i < 1 && j >= 1 && j < 2) return LEFT_FACE;
  if (i >= 1 && i < 2 && j >= 0 && j < 1) return UPPER_FACE;
  if (i >= 1 && i < 2 && j >= 1 && j < 2) return FRONT_FACE;
  if (i >= 1 && i < 2 && j >= 2 && j < 3) return LOWER_FACE;
  if (i >= 2 && i < 3 && j >= 1 && j < 2) return RIGHT_FACE;
  return UNKNOWN_FACE;
}
void CamModelGeneral::GetPosInFace(double& u, double& v, double uCubemap, double vCubemap) const {
  const int i = (int)std::floor(uCubemap / cam_.face), j = (int)std::floor(vCubemap / cam_.face);
  u = uCubemap - i * cam_.face; v = vCubemap - j * cam_.face;
}

// ------------------------------------------------------------------------------------------------ shared device context
static std::mutex g_ctx_mutex;
static cms_ctx* g_ctx = nullptr;
static cms_orb_params g_ctx_orb{};
static cms_camera g_ctx_cam{};      // the camera the shared context was built for (SetCamParams zero-fills the record first: comparable byte for byte)
cms_ctx* SharedContext(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) {
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  cms_orb_params orb{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST};
  if (!CamModelGeneral::GetCamera()->configured()) throw std::runtime_error("CamModelGeneral::SetCamParams was not called");
  // the context belongs to (camera, extractor parameters): a process that configures the camera singleton again (another sequence, another face
  // size) gets a new one -- until round 6 only the extractor parameters were compared, and the mirror kept searching with the OLD camera's LUT and grid
  if (g_ctx && std::memcmp(&orb, &g_ctx_orb, sizeof(orb)) == 0 && std::memcmp(&g_ctx_cam, &CamModelGeneral::GetCamera()->params(), sizeof(cms_camera)) == 0) return g_ctx;
  if (g_ctx) { cms_ctx_destroy(g_ctx); g_ctx = nullptr; }
  const int rc = cms_ctx_create(&g_ctx, 0, &CamModelGeneral::GetCamera()->params(), &orb, 1);
  if (rc != CMS_OK) throw std::runtime_error(std::string("cms_ctx_create: ") + cms_last_error());
  g_ctx_orb = orb; g_ctx_cam = CamModelGeneral::GetCamera()->params();
  return g_ctx;
}

// ------------------------------------------------------------------------------------------------ remap
void System::CreateUndistortRectifyMap() { SharedContext(2000, 1.2f, 8, 20, 7); }
void System::CvtFisheyeToCubeMap_reverseQuery_withInterpolation(cv::Mat& cubemapImg, const cv::Mat& fisheyeImg, int interpolation,
                                                                int borderType, const cv::Scalar&) {
  if (interpolation != cv::INTER_LINEAR || borderType != cv::BORDER_CONSTANT)
    throw std::runtime_error("only INTER_LINEAR / BORDER_CONSTANT(0) -- the mode the reference's examples use (cubemap_lafida.cpp:143)");
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  if (!g_ctx) throw std::runtime_error("CreateUndistortRectifyMap was not called");
  if (cms_remap(g_ctx, fisheyeImg.data, (int)fisheyeImg.step, cubemapImg.data, (int)cubemapImg.step) != CMS_OK)
    throw std::runtime_error(std::string("cms_remap: ") + cms_last_error());
}

// ------------------------------------------------------------------------------------------------ extractor
ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST) {
  ctx_ = SharedContext(nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST);
  cms_geometry g;
  cms_ctx_geometry(ctx_, &g);
  for (int l = 0; l < nlevels; ++l) {
    mvScaleFactor.push_back(g.scale[l]); mvInvScaleFactor.push_back(g.inv_scale[l]);
    mvLevelSigma2.push_back(g.sigma2[l]); mvInvLevelSigma2.push_back(g.inv_sigma2[l]);
  }
}
void ORBextractor::operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints,
                              cv::OutputArray descriptors) {
  if (image.empty()) return;                                        // ORBExtractor.cpp:841
  assert(image.type() == cv::CV_8UC1);                              // :845
  assert(mask.type() == cv::CV_8UC1 && !mask.empty());              // :848
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  cms_geometry g;
  cms_ctx_geometry(ctx_, &g);
  if (image.cols != g.W || image.rows != g.W || mask.cols != g.W || mask.rows != g.W)
    throw std::runtime_error("ORBextractor: image / mask must be the 3F x 3F cubemap canvas");
  if (mask.data != last_mask_) {   // the mask is constant per sequence: upload once per distinct buffer
    if (cms_set_mask(ctx_, mask.data, (int)mask.step) != CMS_OK) throw std::runtime_error(cms_last_error());
    last_mask_ = mask.data;
  }
  std::vector<cms_keypoint> kps(g.kp_cap);
  std::vector<uint8_t> desc((size_t)g.kp_cap * 32);
  int n = 0;
  if (cms_extract(ctx_, image.data, (int)image.step, kps.data(), desc.data(), g.kp_cap, &n) != CMS_OK)
    throw std::runtime_error(std::string("cms_extract: ") + cms_last_error());
  keypoints.clear();
  keypoints.reserve(n);
  for (int i = 0; i < n; ++i) {
    cv::KeyPoint kp;
    kp.pt = cv::Point2f(kps[i].x, kps[i].y); kp.size = kps[i].size; kp.angle = kps[i].angle; kp.response = kps[i].response;
    kp.octave = kps[i].octave;
    keypoints.push_back(kp);
  }
  if (mbKeepImagePyramid) {                                         // ORBExtractor.h:89: mvImagePyramid of the last call
    mvImagePyramid.resize(g.nlevels); mvMaskPyramid.resize(g.nlevels);
    for (int l = 0; l < g.nlevels; ++l) {
      mvImagePyramid[l].create(g.level_h[l], g.level_w[l], cv::CV_8U);
      if (cms_debug_level(ctx_, 0, l, mvImagePyramid[l].data, (int)mvImagePyramid[l].step) != CMS_OK) throw std::runtime_error(cms_last_error());
    }
  }
  if (n == 0) { descriptors.release(); return; }                    // ORBExtractor.cpp:863-864
  descriptors.create(n, 32, cv::CV_8U);
  for (int i = 0; i < n; ++i) std::memcpy(descriptors.ptr<uint8_t>(i), &desc[(size_t)i * 32], 32);
}

// ------------------------------------------------------------------------------------------------ matcher
int ORBMatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
  const uint32_t* pa = a.ptr<uint32_t>();
  const uint32_t* pb = b.ptr<uint32_t>();
  int dist = 0;
  for (int i = 0; i < 8; ++i) dist += __builtin_popcount(pa[i] ^ pb[i]);
  return dist;
}

namespace {
struct KfPack {                                            // flat buffers behind one cms_keyframe
  std::vector<cms_keypoint> kps; std::vector<uint8_t> desc; std::vector<float> rays; std::vector<int> mp, node_id, node_off, node_feat;
};
void pose15_from_Tcw(const cv::Mat& Tcw, float* R, float* t, float* Ow) {
  // KeyFrame::SetPose (KeyFrame.cpp): Rcw, tcw, Ow = -Rcw.t()*tcw (transposed operand: double accumulation, one rounding)
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R[3 * r + c] = Tcw.at<float>(r, c); t[r] = Tcw.at<float>(r, 3); }
  for (int r = 0; r < 3; ++r) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += (double)Tcw.at<float>(k, r) * (double)Tcw.at<float>(k, 3);
    Ow[r] = (float)(-1.0 * s);
  }
}
cms_keyframe pack_keyframe(const KeyFrameView& kf, KfPack& b) {
  const int N = (int)kf.mvKeys.size();
  b.kps.resize(N + 1); b.desc.resize(32 * (size_t)N + 32); b.rays.resize(3 * (size_t)N + 3); b.mp.resize(N + 1);
  for (int i = 0; i < N; ++i) {
    const cv::KeyPoint& k = kf.mvKeys[i];
    b.kps[i].x = k.pt.x; b.kps[i].y = k.pt.y; b.kps[i].size = k.size; b.kps[i].angle = k.angle; b.kps[i].response = k.response; b.kps[i].octave = k.octave;
    std::memcpy(&b.desc[32 * (size_t)i], kf.mDescriptors.ptr<uint8_t>(i), 32);
    for (int c = 0; c < 3; ++c) b.rays[3 * (size_t)i + c] = kf.mvKeyRays[i].v[c];
    b.mp[i] = kf.mvpMapPoints[i] >= 0 ? 1 : -1;
  }
  b.node_off.assign(1, 0);
  for (const auto& e : kf.mFeatVec) {
    b.node_id.push_back((int)e.first);
    for (unsigned f : e.second) b.node_feat.push_back((int)f);
    b.node_off.push_back((int)b.node_feat.size());
  }
  if (b.node_feat.empty()) b.node_feat.push_back(0);
  if (b.node_id.empty()) b.node_id.push_back(0);
  cms_keyframe k{};
  k.n = N; k.kps = b.kps.data(); k.desc = b.desc.data(); k.rays = b.rays.data(); k.mp = b.mp.data();
  pose15_from_Tcw(kf.Tcw, k.Rcw, k.tcw, k.Ow);
  k.nnodes = (int)kf.mFeatVec.size(); k.node_id = b.node_id.data(); k.node_off = b.node_off.data(); k.node_feat = b.node_feat.data();
  k.median_depth = kf.medianDepth;
  return k;
}
}  // namespace

std::vector<NewMapPoint> LocalMapping::CreateNewMapPoints(const KeyFrameView& cur, const std::vector<const KeyFrameView*>& neigh) {
  std::vector<NewMapPoint> out;
  if (neigh.empty() || cur.mvKeys.empty()) return out;
  cms_ctx* ctx = SharedContext(g_ctx_orb.nfeatures, g_ctx_orb.scale_factor, g_ctx_orb.nlevels, g_ctx_orb.ini_th_fast, g_ctx_orb.min_th_fast);
  KfPack pc;
  std::vector<KfPack> pn(neigh.size());
  const cms_keyframe kc = pack_keyframe(cur, pc);
  std::vector<cms_keyframe> kn(neigh.size());
  for (size_t i = 0; i < neigh.size(); ++i) kn[i] = pack_keyframe(*neigh[i], pn[i]);
  const int off[2] = {0, (int)neigh.size()};
  const int cap = (int)cur.mvKeys.size();
  std::vector<int> on(cap), o1(cap), o2(cap);
  std::vector<float> ox(3 * (size_t)cap);
  int n_new = 0;
  {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    if (cms_create_new_map_points(ctx, 1, &kc, off, kn.data(), 0 /* ORBMatcher matcher(0.6,false) */, cap, &n_new, on.data(), o1.data(), o2.data(),
                                  ox.data()) != CMS_OK)
      throw std::runtime_error(std::string("cms_create_new_map_points: ") + cms_last_error());
  }
  out.resize(n_new);
  for (int k = 0; k < n_new; ++k) {
    out[k].neighbour = on[k]; out[k].idx1 = o1[k]; out[k].idx2 = o2[k];
    for (int c = 0; c < 3; ++c) out[k].x3D.v[c] = ox[3 * (size_t)k + c];
  }
  return out;
}

int ORBMatcher::Fuse(KeyFrameView& kf, const std::vector<MapPointView>& mps, const std::vector<uint8_t>& skip, float th, std::vector<int>& fused) {
  const int N = (int)kf.mvKeys.size(), M = (int)mps.size();
  fused.assign(M, -1);
  if (M == 0 || N == 0) return 0;
  cms_ctx* ctx = SharedContext(g_ctx_orb.nfeatures, g_ctx_orb.scale_factor, g_ctx_orb.nlevels, g_ctx_orb.ini_th_fast, g_ctx_orb.min_th_fast);
  float pose15[15];
  pose15_from_Tcw(kf.Tcw, pose15, pose15 + 9, pose15 + 12);
  std::vector<float> pos(3 * (size_t)M), nrm(3 * (size_t)M), dmin(M), dmax(M);
  std::vector<uint8_t> desc(32 * (size_t)M);
  for (int i = 0; i < M; ++i) {
    for (int k = 0; k < 3; ++k) { pos[3 * (size_t)i + k] = mps[i].mWorldPos.at<float>(k, 0); nrm[3 * (size_t)i + k] = mps[i].mNormalVector.at<float>(k, 0); }
    dmin[i] = mps[i].mfMinDistance; dmax[i] = mps[i].mfMaxDistance;
    std::memcpy(&desc[32 * (size_t)i], mps[i].mDescriptor.ptr<uint8_t>(0), 32);
  }
  std::vector<cms_keypoint> kps(N);
  std::vector<uint8_t> tdesc(32 * (size_t)N);
  for (int j = 0; j < N; ++j) {
    const cv::KeyPoint& k = kf.mvKeys[j];
    kps[j].x = k.pt.x; kps[j].y = k.pt.y; kps[j].size = k.size; kps[j].angle = k.angle; kps[j].response = k.response; kps[j].octave = k.octave;
    std::memcpy(&tdesc[32 * (size_t)j], kf.mDescriptors.ptr<uint8_t>(j), 32);
  }
  std::vector<int> bi(M), bd(M);
  {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    int rc = cms_area_set_keypoints(ctx, 0, N, kps.data());
    if (rc == CMS_OK) rc = cms_area_set_descriptors(ctx, 0, N, tdesc.data());
    if (rc == CMS_OK) rc = cms_area_grid(ctx, 1);
    if (rc == CMS_OK) rc = cms_fuse_search(ctx, 0, pose15, M, skip.empty() ? nullptr : skip.data(), pos.data(), nrm.data(), dmin.data(), dmax.data(), desc.data(),
                                           th, bi.data(), bd.data());
    if (rc != CMS_OK) throw std::runtime_error(std::string("cms_fuse_search: ") + cms_last_error());
  }
  // ORBMatcher.cpp:1216-1241 in list order.  A key point that already holds a point means Replace (which of the two survives is the
  // caller's Observations() comparison); a free one receives the observation -- and then holds a point for the rest of the list.
  int nFused = 0;
  for (int i = 0; i < M; ++i) {
    if (bi[i] < 0) continue;
    fused[i] = bi[i];
    if (kf.mvpMapPoints[bi[i]] < 0) kf.mvpMapPoints[bi[i]] = mps[i].mnId;
    ++nFused;
  }
  return nFused;
}

int Tracking::SearchLocalPoints(FrameView& F, std::vector<MapPointView>& mps, float th, float nnratio, float viewingCosLimit) {
  const int N = (int)F.mvKeys.size(), M = (int)mps.size();
  if (M == 0) return 0;
  if (F.mTcw.rows != 4 || F.mTcw.cols != 4 || F.mTcw.type() != cv::CV_32F) throw std::runtime_error("SearchLocalPoints: mTcw must be 4x4 CV_32F");
  cms_ctx* ctx = SharedContext(g_ctx_orb.nfeatures, g_ctx_orb.scale_factor, g_ctx_orb.nlevels, g_ctx_orb.ini_th_fast, g_ctx_orb.min_th_fast);
  // Frame::UpdatePoseMatrices (Frame.cpp:189-195): mRcw, mtcw and mOw = -mRcw.t()*mtcw (cv::gemm with a transposed operand:
  // double accumulation, one rounding to float)
  float pose15[15];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) pose15[3 * r + c] = F.mTcw.at<float>(r, c);
    pose15[9 + r] = F.mTcw.at<float>(r, 3);
  }
  for (int r = 0; r < 3; ++r) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += (double)F.mTcw.at<float>(k, r) * (double)F.mTcw.at<float>(k, 3);
    pose15[12 + r] = (float)(-1.0 * s);
  }
  std::vector<float> pos(3 * (size_t)M), nrm(3 * (size_t)M), dmin(M), dmax(M);
  std::vector<uint8_t> desc(32 * (size_t)M);
  for (int i = 0; i < M; ++i) {
    for (int k = 0; k < 3; ++k) { pos[3 * (size_t)i + k] = mps[i].mWorldPos.at<float>(k, 0); nrm[3 * (size_t)i + k] = mps[i].mNormalVector.at<float>(k, 0); }
    dmin[i] = mps[i].mfMinDistance; dmax[i] = mps[i].mfMaxDistance;
    std::memcpy(&desc[32 * (size_t)i], mps[i].mDescriptor.ptr<uint8_t>(0), 32);
  }
  std::vector<cms_keypoint> kps(N);
  std::vector<uint8_t> tdesc(32 * (size_t)N);
  std::vector<int> kp_mp(N, -1);
  for (int j = 0; j < N; ++j) {
    const cv::KeyPoint& k = F.mvKeys[j];
    kps[j].x = k.pt.x; kps[j].y = k.pt.y; kps[j].size = k.size; kps[j].angle = k.angle; kps[j].response = k.response; kps[j].octave = k.octave;
    std::memcpy(&tdesc[32 * (size_t)j], F.mDescriptors.ptr<uint8_t>(j), 32);
    if (F.mvpMapPoints[j] >= 0) kp_mp[j] = 0x40000000;           // holds a map point already (ORBMatcher.cpp:91-93)
  }
  std::vector<uint8_t> vis(M);
  std::vector<float> px(M), py(M), vc(M);
  std::vector<int> lvl(M), match(M);
  int nm = 0;
  {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    int rc = cms_area_set_keypoints(ctx, 0, N, kps.data());
    if (rc == CMS_OK) rc = cms_area_set_descriptors(ctx, 0, N, tdesc.data());
    if (rc == CMS_OK) rc = cms_area_grid(ctx, 1);
    if (rc == CMS_OK) rc = cms_search_local_points(ctx, 0, pose15, M, pos.data(), nrm.data(), dmin.data(), dmax.data(), desc.data(), viewingCosLimit, th,
                                                   nnratio, ORBMatcher::TH_HIGH, N, kp_mp.data(), vis.data(), px.data(), py.data(), lvl.data(), vc.data(),
                                                   match.data(), &nm, nullptr);
    if (rc != CMS_OK) throw std::runtime_error(std::string("cms_search_local_points: ") + cms_last_error());
  }
  for (int i = 0; i < M; ++i) {
    mps[i].mbTrackInView = vis[i] != 0; mps[i].mTrackProjX = px[i]; mps[i].mTrackProjY = py[i];
    mps[i].mnTrackScaleLevel = lvl[i]; mps[i].mTrackViewCos = vc[i];
    if (match[i] >= 0) F.mvpMapPoints[match[i]] = mps[i].mnId;
  }
  return nm;
}

int ORBMatcher::SearchByProjection(FrameView& F, std::vector<MapPointView>& vpMapPoints, float th) {
  // the reference walks vpMapPoints and skips !mbTrackInView (ORBMatcher.cpp:58-59): the marked ones go to the device as one list, in order
  std::vector<MapPointView> marked;
  std::vector<size_t> src;
  for (size_t i = 0; i < vpMapPoints.size(); ++i)
    if (vpMapPoints[i].mbTrackInView) { marked.push_back(vpMapPoints[i]); src.push_back(i); }
  if (marked.empty()) return 0;
  // viewing-cosine limit below any cosine: the caller's isInFrustum applied its own; every other test of isInFrustum repeats with the same result
  const int n = Tracking::SearchLocalPoints(F, marked, th, mfNNratio, -2.0f);
  for (size_t k = 0; k < src.size(); ++k) {
    MapPointView& d = vpMapPoints[src[k]];
    d.mTrackProjX = marked[k].mTrackProjX; d.mTrackProjY = marked[k].mTrackProjY; d.mnTrackScaleLevel = marked[k].mnTrackScaleLevel; d.mTrackViewCos = marked[k].mTrackViewCos;
  }
  return n;
}

int ORBMatcher::SearchForInitialization(FrameView& F1, FrameView& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize) {
  const int N1 = (int)F1.mvKeys.size(), N2 = (int)F2.mvKeys.size();
  vnMatches12.assign(N1, -1);
  if (N1 == 0 || N2 == 0) return 0;
  if ((int)vbPrevMatched.size() != N1) throw std::runtime_error("SearchForInitialization: vbPrevMatched must have one entry per key point of F1");
  cms_ctx* ctx = SharedContext(g_ctx_orb.nfeatures, g_ctx_orb.scale_factor, g_ctx_orb.nlevels, g_ctx_orb.ini_th_fast, g_ctx_orb.min_th_fast);
  auto pack = [](const FrameView& f, std::vector<cms_keypoint>& k, std::vector<uint8_t>& d) {
    const int n = (int)f.mvKeys.size();
    k.resize(n); d.resize(32 * (size_t)n);
    for (int j = 0; j < n; ++j) {
      const cv::KeyPoint& q = f.mvKeys[j];
      k[j].x = q.pt.x; k[j].y = q.pt.y; k[j].size = q.size; k[j].angle = q.angle; k[j].response = q.response; k[j].octave = q.octave;
      std::memcpy(&d[32 * (size_t)j], f.mDescriptors.ptr<uint8_t>(j), 32);
    }
  };
  std::vector<cms_keypoint> k1, k2;
  std::vector<uint8_t> d1, d2;
  pack(F1, k1, d1); pack(F2, k2, d2);
  std::vector<float> prev(2 * (size_t)N1);
  for (int i = 0; i < N1; ++i) { prev[2 * (size_t)i] = vbPrevMatched[i].x; prev[2 * (size_t)i + 1] = vbPrevMatched[i].y; }
  int nm = 0;
  {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    int rc = cms_area_set_keypoints(ctx, 0, N2, k2.data());
    if (rc == CMS_OK) rc = cms_area_set_descriptors(ctx, 0, N2, d2.data());
    if (rc == CMS_OK) rc = cms_area_grid(ctx, 1);
    if (rc == CMS_OK) rc = cms_search_for_initialization(ctx, 0, N1, k1.data(), d1.data(), prev.data(), windowSize, mfNNratio, mbCheckOrientation ? 1 : 0,
                                                         vnMatches12.data(), &nm);
    if (rc != CMS_OK) throw std::runtime_error(std::string("cms_search_for_initialization: ") + cms_last_error());
  }
  for (int i = 0; i < N1; ++i) vbPrevMatched[i] = cv::Point2f(prev[2 * (size_t)i], prev[2 * (size_t)i + 1]);
  return nm;
}

int ORBMatcher::SearchForTriangulation(const KeyFrameView& pKF1, const KeyFrameView& pKF2, const cv::Mat& E12, std::vector<std::pair<size_t, size_t>>& vMatchedPairs) {
  vMatchedPairs.clear();
  if (pKF1.mvKeys.empty() || pKF2.mvKeys.empty()) return 0;
  if (E12.rows != 3 || E12.cols != 3 || E12.type() != cv::CV_32F) throw std::runtime_error("SearchForTriangulation: E12 must be 3x3 CV_32F");
  cms_ctx* ctx = SharedContext(g_ctx_orb.nfeatures, g_ctx_orb.scale_factor, g_ctx_orb.nlevels, g_ctx_orb.ini_th_fast, g_ctx_orb.min_th_fast);
  KfPack p1, p2;
  const cms_keyframe k1 = pack_keyframe(pKF1, p1), k2 = pack_keyframe(pKF2, p2);
  float e[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) e[3 * r + c] = E12.at<float>(r, c);
  std::vector<int> m12(pKF1.mvKeys.size(), -1);
  int nm = 0;
  {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    if (cms_search_for_triangulation(ctx, &k1, &k2, e, mbCheckOrientation ? 1 : 0, m12.data(), &nm) != CMS_OK)
      throw std::runtime_error(std::string("cms_search_for_triangulation: ") + cms_last_error());
  }
  vMatchedPairs.reserve(nm);
  for (size_t i = 0; i < m12.size(); ++i) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));
  return nm;
}

int ORBMatcher::SearchByProjection(FrameView& Cur, const FrameView& Last, float th, bool) {
  const int N2 = (int)Cur.mvKeys.size();
  if (!Last.mvMapPointPos.empty() && Cur.mTcw.rows == 4 && Cur.mTcw.cols == 4 && N2 > 0) {
    // the whole function on the device (cms_search_by_projection)
    const int NL = (int)Last.mvKeys.size();
    cms_ctx* dctx = SharedContext(g_ctx_orb.nfeatures, g_ctx_orb.scale_factor, g_ctx_orb.nlevels, g_ctx_orb.ini_th_fast, g_ctx_orb.min_th_fast);
    float pose12[12];
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) pose12[3 * r + c] = Cur.mTcw.at<float>(r, c); pose12[9 + r] = Cur.mTcw.at<float>(r, 3); }
    std::vector<uint8_t> valid(NL), mpd(32 * (size_t)NL), tdesc(32 * (size_t)N2);
    std::vector<float> Xw(3 * (size_t)NL), ang(NL);
    std::vector<int> oct(NL), kp_mp(N2, -1), match(NL, -1);
    for (int i = 0; i < NL; ++i) {
      valid[i] = Last.mvpMapPoints[i] >= 0 && !(i < (int)Last.mvbOutlier.size() && Last.mvbOutlier[i]);
      for (int c = 0; c < 3; ++c) Xw[3 * (size_t)i + c] = Last.mvMapPointPos[i].v[c];
      oct[i] = Last.mvKeys[i].octave; ang[i] = Last.mvKeys[i].angle;
      std::memcpy(&mpd[32 * (size_t)i], Last.mMapPointDescriptors.ptr<uint8_t>(i), 32);
    }
    std::vector<cms_keypoint> kps(N2);
    for (int j = 0; j < N2; ++j) {
      const cv::KeyPoint& k = Cur.mvKeys[j];
      kps[j].x = k.pt.x; kps[j].y = k.pt.y; kps[j].size = k.size; kps[j].angle = k.angle; kps[j].response = k.response; kps[j].octave = k.octave;
      std::memcpy(&tdesc[32 * (size_t)j], Cur.mDescriptors.ptr<uint8_t>(j), 32);
      if (Cur.mvpMapPoints[j] >= 0) kp_mp[j] = 0x40000000;
    }
    int nm = 0;
    {
      std::lock_guard<std::mutex> lock(g_ctx_mutex);
      int rc = cms_area_set_keypoints(dctx, 0, N2, kps.data());
      if (rc == CMS_OK) rc = cms_area_set_descriptors(dctx, 0, N2, tdesc.data());
      if (rc == CMS_OK) rc = cms_area_grid(dctx, 1);
      if (rc == CMS_OK) rc = cms_search_by_projection(dctx, 0, pose12, NL, valid.data(), Xw.data(), oct.data(), ang.data(), mpd.data(), th, mbCheckOrientation ? 1 : 0,
                                                     TH_HIGH, N2, kp_mp.data(), match.data(), &nm);
      if (rc != CMS_OK) throw std::runtime_error(std::string("cms_search_by_projection: ") + cms_last_error());
    }
    for (int i = 0; i < NL; ++i) if (match[i] >= 0) Cur.mvpMapPoints[match[i]] = Last.mvpMapPoints[i];
    return nm;
  }
  cms_ctx* ctx = SharedContext(g_ctx_orb.nfeatures, g_ctx_orb.scale_factor, g_ctx_orb.nlevels, g_ctx_orb.ini_th_fast, g_ctx_orb.min_th_fast);
  // candidate windows: CurrentFrame.GetFeaturesInArea(u, v, th * scale[octave], octave - 1, octave + 1) (ORBMatcher.cpp:176-181) for
  // every projected map point of the last frame, answered on the device by the frame grid of the current frame's key points --
  // same unfolding cases, same candidate order as Frame.cpp:251-716
  std::vector<int> qall;
  std::vector<float> qx, qy, qr;
  std::vector<int> qlo, qhi;
  for (int i = 0; i < (int)Last.mvKeys.size(); ++i) {
    if (Last.mvpMapPoints[i] < 0 || (i < (int)Last.mvbOutlier.size() && Last.mvbOutlier[i])) continue;
    const cv::Point2f p = Last.projInCurrent[i];
    if (p.x < 0 || p.y < 0) continue;
    const int oct = Last.mvKeys[i].octave;
    qall.push_back(i); qx.push_back(p.x); qy.push_back(p.y); qr.push_back(th * Cur.mvScaleFactors[oct]);
    qlo.push_back(oct - 1); qhi.push_back(oct + 1);
  }
  std::vector<int> qidx, off(1, 0), cand;
  if (!qall.empty()) {
    std::vector<cms_keypoint> kps(N2);
    for (int j = 0; j < N2; ++j) {
      const cv::KeyPoint& k = Cur.mvKeys[j];
      kps[j].x = k.pt.x; kps[j].y = k.pt.y; kps[j].size = k.size; kps[j].angle = k.angle; kps[j].response = k.response; kps[j].octave = k.octave;
    }
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    std::vector<int> aoff(qall.size() + 1), aidx(64 * qall.size() + 1024);
    int total = 0;
    int rc = cms_area_set_keypoints(ctx, 0, N2, kps.data());
    if (rc == CMS_OK) rc = cms_area_grid(ctx, 1);
    if (rc == CMS_OK) rc = cms_features_in_area(ctx, 0, (int)qall.size(), qx.data(), qy.data(), qr.data(), qlo.data(), qhi.data(), aoff.data(),
                                                aidx.data(), (int)aidx.size(), &total);
    if (rc == CMS_ERR_OVERFLOW) {               // denser than 64 candidates per window: retry with the exact size
      aidx.resize((size_t)total + 1);
      rc = cms_features_in_area(ctx, 0, (int)qall.size(), qx.data(), qy.data(), qr.data(), qlo.data(), qhi.data(), aoff.data(), aidx.data(),
                                (int)aidx.size(), &total);
    }
    if (rc != CMS_OK) throw std::runtime_error(std::string("cms_features_in_area: ") + cms_last_error());
    for (size_t q = 0; q < qall.size(); ++q) {
      if (aoff[q + 1] == aoff[q]) continue;     // vIndices2.empty()
      qidx.push_back(qall[q]);
      cand.insert(cand.end(), aidx.begin() + aoff[q], aidx.begin() + aoff[q + 1]);
      off.push_back((int)cand.size());
    }
  }
  const int nq = (int)qidx.size();
  if (nq == 0) return 0;
  std::vector<uint8_t> qdesc((size_t)nq * 32), excl(N2, 0);
  for (int q = 0; q < nq; ++q) std::memcpy(&qdesc[(size_t)q * 32], Last.mDescriptors.ptr<uint8_t>(qidx[q]), 32);
  for (int j = 0; j < N2; ++j) excl[j] = Cur.mvpMapPoints[j] >= 0;
  std::vector<uint8_t> tdesc((size_t)N2 * 32);
  for (int j = 0; j < N2; ++j) std::memcpy(&tdesc[(size_t)j * 32], Cur.mDescriptors.ptr<uint8_t>(j), 32);
  std::vector<int> bi(nq), bd(nq), sd(nq);
  {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    if (cms_hamming_best2(ctx, qdesc.data(), nq, tdesc.data(), N2, off.data(), cand.data(), nullptr, excl.data(), bi.data(), bd.data(),
                          nullptr, sd.data(), nullptr) != CMS_OK)
      throw std::runtime_error(std::string("cms_hamming_best2: ") + cms_last_error());
  }
  // greedy replay in the reference's order (ORBMatcher.cpp:150-222)
  const int nBins = (int)std::ceil(360.0f / HISTO_LENGTH);
  std::vector<std::vector<int>> rotHist(nBins);
  const float factor = 1.0f / HISTO_LENGTH;
  int nmatches = 0;
  for (int q = 0; q < nq; ++q) {
    int bestIdx = bi[q], bestDist = bd[q];
    if (bestIdx >= 0 && Cur.mvpMapPoints[bestIdx] >= 0 && !excl[bestIdx]) {   // taken earlier in THIS call: rescan on the host
      bestDist = 256; bestIdx = -1;
      for (int c = off[q]; c < off[q + 1]; ++c) {
        const int j = cand[c];
        if (Cur.mvpMapPoints[j] >= 0) continue;
        const int d = DescriptorDistance(Last.mDescriptors.row(qidx[q]), Cur.mDescriptors.row(j));
        if (d < bestDist) { bestDist = d; bestIdx = j; }
      }
    }
    if (bestIdx < 0 || bestDist > TH_HIGH) continue;
    Cur.mvpMapPoints[bestIdx] = Last.mvpMapPoints[qidx[q]];
    ++nmatches;
    if (mbCheckOrientation) {
      float rot = Last.mvKeys[qidx[q]].angle - Cur.mvKeys[bestIdx].angle;
      if (rot < 0.0) rot += 360.0f;
      int bin = (int)std::round(rot * factor);
      if (bin == nBins) bin = 0;
      rotHist[bin].push_back(bestIdx);
    }
  }
  if (mbCheckOrientation) {   // ComputeThreeMaxima (ORBMatcher.cpp:905-946)
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < nBins; ++i) {
      const int s = (int)rotHist[i].size();
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) ind3 = -1;
    for (int i = 0; i < nBins; ++i)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int j : rotHist[i]) { Cur.mvpMapPoints[j] = -1; --nmatches; }
  }
  return nmatches;
}

// ------------------------------------------------------------------------------------------------ local BA
static void R_to_quat(const double m[9], double* q) {  // Eigen::Quaterniond(Matrix3d), as Converter::toSE3Quat uses it
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0); q[i] = 0.5 * t; t = 0.5 / t;
    q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t; q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t; q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
  }
}
void System::SaveKeyFrameTrajectoryTUM(const std::string& filename, const std::vector<TrajectoryKeyFrame>& vpKFs) {
  std::ofstream f(filename.c_str());
  f << std::fixed;
  for (const TrajectoryKeyFrame& kf : vpKFs) {
    if (kf.bad) continue;
    double Rt[9];                                           // R^T = GetRotation().t(), through Converter::toMatrix3d (float -> double)
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rt[3 * r + c] = kf.Tcw.at<float>(c, r);
    double q[4];
    R_to_quat(Rt, q);                                       // Eigen::Quaterniond(Matrix3d), stored as float (Converter.cpp:141-153)
    float t[3];                                             // KeyFrame::GetCameraCenter() = -R^T t in float arithmetic (KeyFrame.cpp SetPose)
    for (int r = 0; r < 3; ++r) {
      float acc = 0.f;
      for (int c = 0; c < 3; ++c) acc += kf.Tcw.at<float>(c, r) * kf.Tcw.at<float>(c, 3);
      t[r] = -acc;
    }
    f << std::setprecision(6) << kf.mTimeStamp << std::setprecision(7) << " " << t[0] << " " << t[1] << " " << t[2] << " " << (float)q[0] << " "
      << (float)q[1] << " " << (float)q[2] << " " << (float)q[3] << std::endl;
  }
}

int Optimizer::PoseOptimization(PoseFrame* fr) {
  CamModelGeneral* cam = CamModelGeneral::GetCamera();
  const int N = fr->N;
  fr->mvbOutlier.resize(N);
  std::vector<double> Xw, obs, inv;
  std::vector<int8_t> face;
  std::vector<int> index;
  int nInitialCorrespondences = 0;
  for (int i = 0; i < N; ++i) {
    if (fr->mvKeyRays[i](2) < cam->GetCosFovTh()) continue;                      // Optimizer.cpp:83-85
    if (!fr->mvbHasMapPoint[i]) continue;
    ++nInitialCorrespondences;
    fr->mvbOutlier[i] = false;
    const cv::KeyPoint& kp = fr->mvKeys[i];
    const int f = cam->FaceInCubemap(kp.pt);                                      // :103
    if (f == CamModelGeneral::UNKNOWN_FACE) throw std::runtime_error("PoseOptimization: key point on an unknown face");   // the reference exits
    double u, v;
    cam->GetPosInFace(u, v, (double)kp.pt.x, (double)kp.pt.y);                     // :104-106
    obs.push_back(u); obs.push_back(v);
    inv.push_back((double)fr->mvInvLevelSigma2[kp.octave]);                        // :107-108
    face.push_back((int8_t)f);
    for (int k = 0; k < 3; ++k) Xw.push_back((double)fr->mvMapPointPos[i](k));     // :118-121
    index.push_back(i);
  }
  if (nInitialCorrespondences < 3) return 0;                                       // :131-132
  double pose[7], R[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[3 * r + c] = fr->mTcw.at<float>(r, c);   // Converter::toSE3Quat
  for (int r = 0; r < 3; ++r) pose[r] = fr->mTcw.at<float>(r, 3);
  R_to_quat(R, pose + 3);
  const int n = (int)index.size();
  std::vector<uint8_t> out(n);
  int ninl = 0;
  const double f = cam->Get_fx();
  const int rc = cms_pose_optimize(0, n, Xw.data(), obs.data(), inv.data(), face.data(), f, f, f, f, pose, out.data(), &ninl, nullptr);
  if (rc < 0) throw std::runtime_error(std::string("cms_pose_optimize: ") + cms_last_error());
  for (int e = 0; e < n; ++e) fr->mvbOutlier[index[e]] = out[e] != 0;
  const double x = pose[3], y = pose[4], z = pose[5], w = pose[6];                  // Converter::toCvMat(SE3Quat) -> float 4x4
  const double Ro[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                        2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) fr->mTcw.at<float>(r, c) = (float)Ro[3 * r + c]; fr->mTcw.at<float>(r, 3) = (float)pose[r]; }
  return ninl;
}

void Optimizer::LocalBundleAdjustment(LocalBAWindow* win, bool* pbStopFlag) {
  CamModelGeneral* cam = CamModelGeneral::GetCamera();
  const int K = (int)win->keyframes.size(), P = (int)win->mappoints.size();
  win->toErase.clear();
  if (K == 0 || P == 0) return;
  std::vector<double> poses((size_t)K * 7), points((size_t)P * 3);
  std::vector<uint8_t> fixed(K);
  for (int k = 0; k < K; ++k) {   // Converter::toSE3Quat (Converter.cpp:41-51): float Tcw -> double R, t -> quaternion
    const cv::Mat& T = win->keyframes[k].Tcw;
    double R[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[3 * r + c] = T.at<float>(r, c);
    for (int r = 0; r < 3; ++r) poses[7 * k + r] = T.at<float>(r, 3);
    R_to_quat(R, &poses[7 * k + 3]);
    fixed[k] = (win->keyframes[k].mnId == 0) || win->keyframes[k].fixed;   // Optimizer.cpp:268, 281
  }
  std::vector<int> e_pose, e_point;
  std::vector<double> e_obs, e_inv;
  std::vector<int8_t> e_face;
  std::vector<std::pair<int, int>> e_src;
  for (int p = 0; p < P; ++p) {
    const LocalBAWindow::MP& mp = win->mappoints[p];
    for (int i = 0; i < 3; ++i) points[3 * p + i] = mp.Xw.at<float>(i, 0);   // Converter::toVector3d
    for (const LocalBAWindow::Obs& ob : mp.observations) {
      if (ob.ray(2) < cam->GetCosFovTh()) continue;                            // Optimizer.cpp:323-325
      const int face = cam->FaceInCubemap(ob.kp.pt);                           // :335
      if (face == CamModelGeneral::UNKNOWN_FACE) continue;                     // would exit() inside g2o (edge h:103-108)
      double u, v;
      cam->GetPosInFace(u, v, (double)ob.kp.pt.x, (double)ob.kp.pt.y);          // :336-338
      e_pose.push_back(ob.kf); e_point.push_back(p); e_obs.push_back(u); e_obs.push_back(v);
      e_inv.push_back((double)win->keyframes[ob.kf].mvInvLevelSigma2[ob.kp.octave]);   // :339-340
      e_face.push_back((int8_t)face);
      e_src.push_back(std::make_pair(ob.kf, p));
    }
  }
  const int E = (int)e_pose.size();
  if (E == 0) return;
  if (pbStopFlag && *pbStopFlag) return;                                       // :359-361
  std::vector<uint8_t> flags(E);
  const double f = cam->Get_fx();
  const int rc = cms_ba_run(0, K, poses.data(), fixed.data(), P, points.data(), E, e_pose.data(), e_point.data(), e_obs.data(), e_inv.data(),
                            e_face.data(), f, f, f, f, 5, 10, reinterpret_cast<const volatile uint8_t*>(pbStopFlag), flags.data(), nullptr);
  if (rc < 0) throw std::runtime_error(std::string("cms_ba_run: ") + cms_last_error());
  if (rc == 1) return;
  for (int e = 0; e < E; ++e) if (flags[e]) win->toErase.push_back(e_src[e]);  // :399-412
  for (int k = 0; k < K; ++k) {   // Converter::toCvMat(SE3Quat) -> float 4x4 (Converter.cpp:53-104)
    if (fixed[k]) continue;       // the reference rewrites local key frames only (:432-438); fixed ones are unchanged anyway
    const double* q = &poses[7 * k + 3];
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                         2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    cv::Mat& T = win->keyframes[k].Tcw;
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T.at<float>(r, c) = (float)R[3 * r + c]; T.at<float>(r, 3) = (float)poses[7 * k + r]; }
  }
  for (int p = 0; p < P; ++p)
    for (int i = 0; i < 3; ++i) win->mappoints[p].Xw.at<float>(i, 0) = (float)points[3 * p + i];
}

}  // namespace CubemapSLAM
