// closed_loop_driver.cpp -- a Python-free, per-frame driver of the drop-in boundary: what Examples/cubemap_lafida.cpp:128-179 does around
// System::TrackCubemap (imread -> fisheye-to-cubemap -> track -> median / mean tracking time), with the hot path of Tracking / LocalMapping
// behind it running on the MI355X through the C-ABI of libcubemapslam_hip.so.
//
//   cubemap_closed_loop <settings.yaml> <image list> <image dir> <mask.pgm> <ground truth poses> [--kf-every 5] [--ba-window 8]
//                       [--new-points 400] [--warmup 6] [--log frames.jsonl] [--trajectory KeyFrameTrajectory.txt] [--perf perf.txt] [--device 0]
//
// Per frame, in the order src/Tracking.cpp runs it (the same order and the same stand-ins as cubemapslam_amd/harness.py, whose product
// run tests/test_gpu_harness.py holds against the CPU oracle frame by frame):
//   frames 0 / 1   3 x nFeatures extractor (Tracking.cpp:95-96, 145-148), MonocularInitialization (:391-465): > 100 key points, then
//                  ORBMatcher(0.9, true).SearchForInitialization(F0, F1, 100) >= 100 matches.  Initializer + GlobalBA are out of scope
//                  (SURVEY.md 8): poses and matched points are seeded from the ground-truth file (a ray cast into the rendered box room).
//   frame t >= 2   TrackWithMotionModel (:620-677): SearchByProjection(Cur, Last, 15 | 30), PoseOptimization, outliers dropped;
//                  TrackLocalMap (:679-719, 794-843): isInFrustum + SearchByProjection over the map, PoseOptimization; velocity (:360-368).
//   key frames     every --kf-every frames: new points behind free key points (ground-truth stand-in for CreateNewMapPoints' BoW pairing),
//                  Optimizer::LocalBundleAdjustment over the last --ba-window key frames, write-back through float (Optimizer.cpp:419-449).
// Image files are binary PGM (P5); the ground-truth file has one line per frame: Rcw (9, row major) tcw (3) as doubles; its first line is
// "room hx hy hz" (half extents of the box room the stream was rendered in).
//
// Host arithmetic follows the reference's types: poses are float 4x4 (cv::Mat CV_32F), products summed left to right in float.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "cubemapslam_hip.h"
#include "io_formats.h"

namespace {

using CubemapSLAM::Settings;

struct Image { int w = 0, h = 0; std::vector<uint8_t> px; };
static bool read_pgm(const std::string& path, Image& im) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::string magic; f >> magic;
  if (magic != "P5") return false;
  auto next_int = [&]() { int v; while (f >> std::ws && f.peek() == '#') { std::string l; std::getline(f, l); } f >> v; return v; };
  im.w = next_int(); im.h = next_int(); const int maxv = next_int();
  f.get();
  if (im.w <= 0 || im.h <= 0 || maxv != 255) return false;
  im.px.resize((size_t)im.w * im.h);
  f.read(reinterpret_cast<char*>(im.px.data()), (std::streamsize)im.px.size());
  return (size_t)f.gcount() == im.px.size();
}

struct Mat4f { float m[16]; };                       // row major 4x4, cv::Mat CV_32F
static Mat4f eye4() { Mat4f T; for (int i = 0; i < 16; ++i) T.m[i] = (i % 5 == 0) ? 1.0f : 0.0f; return T; }
static Mat4f mul4(const Mat4f& A, const Mat4f& B) {  // float products, summed left to right in float
  Mat4f C;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      float s = 0.0f;
      for (int k = 0; k < 4; ++k) s += A.m[4 * r + k] * B.m[4 * k + c];
      C.m[4 * r + c] = s;
    }
  return C;
}
static void camera_centre(const Mat4f& T, float Ow[3]) {      // -R^T t
  for (int i = 0; i < 3; ++i) {
    float s = 0.0f;
    for (int k = 0; k < 3; ++k) s += T.m[4 * k + i] * T.m[4 * k + 3];
    Ow[i] = -s;
  }
}
// float pose -> (t, q) doubles like Converter::toSE3Quat (Converter.cpp:41-51: Eigen quaternion of the rotation)
static void pose7_from_T(const Mat4f& T, double p[7]) {
  double R[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[3 * r + c] = (double)T.m[4 * r + c];
  for (int i = 0; i < 3; ++i) p[i] = (double)T.m[4 * i + 3];
  const double tr = R[0] + R[4] + R[8];
  double q[4];
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0); q[3] = 0.5 * s; s = 0.5 / s;
    q[0] = (R[7] - R[5]) * s; q[1] = (R[2] - R[6]) * s; q[2] = (R[3] - R[1]) * s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[i] = 0.5 * s; s = 0.5 / s;
    q[3] = (R[3 * k + j] - R[3 * j + k]) * s; q[j] = (R[3 * j + i] + R[3 * i + j]) * s; q[k] = (R[3 * k + i] + R[3 * i + k]) * s;
  }
  if (q[3] < 0) for (double& v : q) v = -v;
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) p[3 + i] = q[i] / n;
}
// SE3Quat -> float cv::Mat (Converter::toCvMat, Converter.cpp:53-104)
static Mat4f T_from_pose7(const double p[7]) {
  const double n = std::sqrt(p[3] * p[3] + p[4] * p[4] + p[5] * p[5] + p[6] * p[6]);
  const double x = p[3] / n, y = p[4] / n, z = p[5] / n, w = p[6] / n;
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                       2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
  Mat4f T = eye4();
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T.m[4 * r + c] = (float)R[3 * r + c]; T.m[4 * r + 3] = (float)p[r]; }
  return T;
}

// cubemap pixel -> face and rig-frame ray with unit depth on its face (CamModelGeneral::TransformCubemapToRays); face as in cms_keypoint use
static int pixel_to_ray(int F, double px, double py, double ray[3]) {
  const int i = (int)std::floor(px / F), j = (int)std::floor(py / F);
  int face = -1;
  if (i == 1 && j == 1) face = 0; else if (i == 0 && j == 1) face = 1; else if (i == 2 && j == 1) face = 2; else if (i == 1 && j == 0) face = 3;
  else if (i == 1 && j == 2) face = 4;
  const double lx = (px - std::floor(px / F) * F - F / 2.0) / (F / 2.0), ly = (py - std::floor(py / F) * F - F / 2.0) / (F / 2.0);
  switch (face) {
    case 0: ray[0] = lx; ray[1] = ly; ray[2] = 1; break;
    case 1: ray[0] = -1; ray[1] = ly; ray[2] = lx; break;
    case 2: ray[0] = 1; ray[1] = ly; ray[2] = -lx; break;
    case 4: ray[0] = lx; ray[1] = 1; ray[2] = -ly; break;
    case 3: ray[0] = lx; ray[1] = -1; ray[2] = ly; break;
    default: ray[0] = 0; ray[1] = 0; ray[2] = 1; break;
  }
  return face;
}

struct GroundTruth { double R[9], t[3]; };
struct Room {
  double half[3] = {3.0, 1.5, 2.0};
  // the room point seen at cubemap pixel (px, py) from the ground-truth pose: ray cast from the camera centre to the first wall
  bool point_behind_pixel(int F, const GroundTruth& g, double px, double py, double Xw[3]) const {
    double ray[3];
    if (pixel_to_ray(F, px, py, ray) < 0) return false;
    double cw[3], d[3];
    for (int i = 0; i < 3; ++i) { cw[i] = -(g.R[i] * g.t[0] + g.R[3 + i] * g.t[1] + g.R[6 + i] * g.t[2]); d[i] = g.R[i] * ray[0] + g.R[3 + i] * ray[1] + g.R[6 + i] * ray[2]; }
    double tbest = INFINITY;
    for (int a = 0; a < 3; ++a) {
      if (d[a] == 0) continue;
      const double tt = ((d[a] > 0 ? half[a] : -half[a]) - cw[a]) / d[a];
      if (tt < tbest) tbest = tt;
    }
    for (int i = 0; i < 3; ++i) Xw[i] = cw[i] + d[i] * tbest;
    return true;
  }
};

#define CHECK(call) do { const int rc_ = (call); if (rc_ < 0) { std::fprintf(stderr, "%s failed: %s\n", #call, cms_last_error()); std::exit(2); } } while (0)
// CMS_DRIVER_CALL_TIMES=1: wall time of every boundary call (host work + launches + the wait for its results), summed per entry point and
// printed after the summary -- where a tracked frame's time goes
struct CallTimes {
  struct Slot { const char* name; double s; long n; };
  std::vector<Slot> slots;
  bool on = std::getenv("CMS_DRIVER_CALL_TIMES") != nullptr;
  void add(const char* name, double sec) {
    for (Slot& sl : slots) if (std::strcmp(sl.name, name) == 0) { sl.s += sec; ++sl.n; return; }
    slots.push_back({name, sec, 1});
  }
  void clear() { slots.clear(); }
};
static CallTimes g_calls;
#define TIMED(name, call) do { if (!g_calls.on) { CHECK(call); break; } const auto c0_ = std::chrono::steady_clock::now(); CHECK(call); \
  g_calls.add(name, std::chrono::duration<double>(std::chrono::steady_clock::now() - c0_).count()); } while (0)

struct FrameData {
  Mat4f T = eye4();
  std::vector<cms_keypoint> kps;
  std::vector<uint8_t> desc;          // n x 32
  std::vector<float> rays;            // n x 3: Frame::mvKeyRays, from the device (cms_remap_extract_rays)
  std::vector<int> kp_mp;             // map point per key point or -1
  std::vector<uint8_t> outlier;
  int frame = 0;
};

class Tracker {
 public:
  Tracker(const Settings& st, const Image& mask, const Room& room, int device, int kf_every, int ba_window, int new_pts)
      : room_(room), device_(device), kf_every_(kf_every), ba_window_(ba_window), new_pts_(new_pts) {
    cam_ = st.Camera();
    cms_orb_params orb = st.Orb();
    F_ = cam_.face;
    cms_orb_params ini = orb; ini.nfeatures = 3 * orb.nfeatures;
    CHECK(cms_ctx_create(&ctx_ini_, device, &cam_, &ini, 1));
    CHECK(cms_ctx_create(&ctx_trk_, device, &cam_, &orb, 1));
    CHECK(cms_set_mask(ctx_ini_, mask.px.data(), mask.w));
    CHECK(cms_set_mask(ctx_trk_, mask.px.data(), mask.w));
    cms_geometry g;
    CHECK(cms_ctx_geometry(ctx_trk_, &g));
    nlevels_ = g.nlevels;
    for (int l = 0; l < g.nlevels; ++l) { sf_.push_back(g.scale[l]); inv_sigma2_.push_back(g.inv_sigma2[l]); }
    cap_trk_ = g.kp_cap;
    CHECK(cms_ctx_geometry(ctx_ini_, &g));
    cap_ini_ = g.kp_cap;
    CHECK(cms_pose_create(&pose_, device, 1, std::max(cap_ini_, cap_trk_)));
    const float pif = 3.1415926535897932384626f;
    cos_fov_ = std::cos((float)cam_.fov_deg / 2 * (pif / 180));
  }
  ~Tracker() {
    if (pose_) cms_pose_destroy(pose_);
    if (ctx_ini_) cms_ctx_destroy(ctx_ini_);
    if (ctx_trk_) cms_ctx_destroy(ctx_trk_);
  }
  enum State { NO_IMAGES, NOT_INITIALIZED, OK, LOST };
  State state = NO_IMAGES;
  std::vector<FrameData> kfs;
  std::vector<float> mp_pos, mp_normal, mp_min, mp_max;     // map, parallel arrays
  std::vector<uint8_t> mp_desc;
  std::string last_log;

  void feed(int i, const Image& fisheye, const GroundTruth& gt) {
    std::ostringstream log;
    log << "{\"frame\": " << i;
    if (state == NO_IMAGES || state == NOT_INITIALIZED) { state = NOT_INITIALIZED; initialize(i, fisheye, gt, log); }
    else if (state == OK) track(i, fisheye, gt, log);
    else log << ", \"stage\": \"lost\"";
    log << "}";
    last_log = log.str();
  }

 private:
  cms_camera cam_{};
  Room room_;
  int device_, kf_every_, ba_window_, new_pts_, F_ = 0, nlevels_ = 8, cap_ini_ = 0, cap_trk_ = 0;
  cms_ctx* ctx_ini_ = nullptr; cms_ctx* ctx_trk_ = nullptr; cms_ctx* cur_ = nullptr;
  cms_pose* pose_ = nullptr;
  std::vector<float> sf_, inv_sigma2_;
  float cos_fov_ = 0;
  bool have_velocity_ = false, have_ini_ = false;
  Mat4f velocity_ = eye4();
  FrameData last_, ini_;
  GroundTruth ini_gt_{};
  std::vector<float> ini_prev_;

  size_t n_map() const { return mp_min.size(); }

  void extract(const Image& fisheye, bool init, FrameData& fr) {
    cur_ = init ? ctx_ini_ : ctx_trk_;
    const int cap = init ? cap_ini_ : cap_trk_;
    fr.kps.resize(cap); fr.desc.resize((size_t)cap * 32); fr.rays.resize((size_t)cap * 3);
    int n = 0;
    TIMED("cms_remap_extract", cms_remap_extract_rays(cur_, fisheye.px.data(), fisheye.w, fr.kps.data(), fr.desc.data(), fr.rays.data(), cap, &n));
    fr.kps.resize(n); fr.desc.resize((size_t)n * 32); fr.rays.resize((size_t)n * 3);
    TIMED("cms_area_grid", cms_area_grid(cur_, 1));                       // Frame::AssignFeaturesToGrid
    fr.kp_mp.assign(n, -1); fr.outlier.assign(n, 0);
  }

  // new MapPoints with one observation: MapPoint::UpdateNormalAndDepth (MapPoint.cpp:332-373) for n = 1
  int add_point(const float Xw[3], const uint8_t* desc, int octave, const float Ow[3]) {
    const float PO[3] = {Xw[0] - Ow[0], Xw[1] - Ow[1], Xw[2] - Ow[2]};
    const float dist = (float)std::sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
    const int id = (int)n_map();
    for (int k = 0; k < 3; ++k) { mp_pos.push_back(Xw[k]); mp_normal.push_back(PO[k] / dist); }
    mp_desc.insert(mp_desc.end(), desc, desc + 32);
    const float mx = dist * sf_[octave];
    mp_max.push_back(mx); mp_min.push_back(mx / sf_[nlevels_ - 1]);
    return id;
  }

  // Optimizer::PoseOptimization's edges (Optimizer.cpp:78-129) + the optimisation on the device; returns the inlier count
  int optimize_pose(FrameData& fr) {
    std::vector<int> idx;
    std::vector<double> Xw, obs, inv;
    std::vector<int8_t> face;
    for (size_t i = 0; i < fr.kps.size(); ++i) {
      if (fr.kp_mp[i] < 0) continue;
      double ray[3];
      const double px = fr.kps[i].x, py = fr.kps[i].y;
      const int fc = pixel_to_ray(F_, px, py, ray);                 // (the face; the key ray itself comes from the device)
      if (fr.rays[3 * i + 2] < cos_fov_ || fc < 0) continue;        // Optimizer.cpp:97: key ray outside the field of view / not on a face
      idx.push_back((int)i);
      for (int k = 0; k < 3; ++k) Xw.push_back((double)mp_pos[3 * (size_t)fr.kp_mp[i] + k]);
      obs.push_back(px - std::floor(px / F_) * F_); obs.push_back(py - std::floor(py / F_) * F_);
      inv.push_back((double)inv_sigma2_[fr.kps[i].octave]); face.push_back((int8_t)fc);
    }
    if (idx.size() < 3) return 0;
    double pose7[7];
    pose7_from_T(fr.T, pose7);
    const int off[2] = {0, (int)idx.size()};
    std::vector<uint8_t> out(idx.size());
    int ninl = 0;
    cms_pose_stats st;
    TIMED("cms_pose_optimize_batch", cms_pose_optimize_batch(pose_, 1, off, Xw.data(), obs.data(), inv.data(), face.data(), F_ / 2.0, F_ / 2.0, F_ / 2.0, F_ / 2.0, pose7,
                                  out.data(), &ninl, &st));
    fr.T = T_from_pose7(pose7);
    std::fill(fr.outlier.begin(), fr.outlier.end(), 0);
    for (size_t k = 0; k < idx.size(); ++k) if (out[k]) fr.outlier[idx[k]] = 1;
    return ninl;
  }

  static Mat4f T_of(const GroundTruth& g) {
    Mat4f T = eye4();
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T.m[4 * r + c] = (float)g.R[3 * r + c]; T.m[4 * r + 3] = (float)g.t[r]; }
    return T;
  }

  void initialize(int i, const Image& fisheye, const GroundTruth& gt, std::ostringstream& log) {
    FrameData fr;
    extract(fisheye, true, fr);
    fr.frame = i;
    const int n = (int)fr.kps.size();
    log << ", \"stage\": \"init\", \"nkp\": " << n;
    if (!have_ini_) {
      if (n > 100) {
        ini_ = fr; ini_gt_ = gt; have_ini_ = true;
        ini_prev_.resize(2 * (size_t)n);
        for (int k = 0; k < n; ++k) { ini_prev_[2 * k] = fr.kps[k].x; ini_prev_[2 * k + 1] = fr.kps[k].y; }
      }
      return;
    }
    if (n <= 100) { have_ini_ = false; return; }
    const int n1 = (int)ini_.kps.size();
    std::vector<int> m12(n1, -1);
    int nm = 0;
    TIMED("cms_search_for_initialization", cms_search_for_initialization(cur_, 0, n1, ini_.kps.data(), ini_.desc.data(), ini_prev_.data(), 100, 0.9f, 1, m12.data(), &nm));
    log << ", \"n_init\": " << nm;
    if (nm < 100) { have_ini_ = false; return; }
    // ground-truth stand-in for Initializer + GlobalBA: both poses and the matched points' positions
    const Mat4f T0 = T_of(ini_gt_), T1 = T_of(gt);
    float Ow1[3];
    camera_centre(T1, Ow1);
    ini_.T = T0; fr.T = T1;
    int nmap = 0;
    for (int k = 0; k < n1; ++k) {
      if (m12[k] < 0) continue;
      double Xd[3];
      if (!room_.point_behind_pixel(F_, ini_gt_, (double)ini_.kps[k].x, (double)ini_.kps[k].y, Xd)) continue;
      const float Xf[3] = {(float)Xd[0], (float)Xd[1], (float)Xd[2]};
      const int id = add_point(Xf, &fr.desc[(size_t)m12[k] * 32], fr.kps[m12[k]].octave, Ow1);
      ini_.kp_mp[k] = id; fr.kp_mp[m12[k]] = id;
      ++nmap;
    }
    kfs.clear(); kfs.push_back(ini_); kfs.push_back(fr);
    last_ = fr;
    have_velocity_ = false;
    state = OK;
    log << ", \"n_map\": " << nmap;
  }

  void track(int i, const Image& fisheye, const GroundTruth& gt, std::ostringstream& log) {
    FrameData cur;
    extract(fisheye, false, cur);
    cur.frame = i;
    const int n = (int)cur.kps.size(), nl = (int)last_.kps.size();
    log << ", \"stage\": \"track\", \"nkp\": " << n;
    // TrackWithMotionModel (Tracking.cpp:620-677); the first tracked frame has no velocity yet: a standing camera
    cur.T = have_velocity_ ? mul4(velocity_, last_.T) : last_.T;
    float pose12[12];
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) pose12[3 * r + c] = cur.T.m[4 * r + c]; pose12[9 + r] = cur.T.m[4 * r + 3]; }
    std::vector<uint8_t> valid(nl), mdesc((size_t)nl * 32, 0);
    std::vector<float> Xw((size_t)nl * 3, 0.0f), ang(nl);
    std::vector<int> oct(nl);
    for (int k = 0; k < nl; ++k) {
      valid[k] = last_.kp_mp[k] >= 0 && !last_.outlier[k];
      const int mp = std::max(last_.kp_mp[k], 0);
      if (n_map() > 0) { for (int c = 0; c < 3; ++c) Xw[3 * (size_t)k + c] = mp_pos[3 * (size_t)mp + c]; std::memcpy(&mdesc[(size_t)k * 32], &mp_desc[(size_t)mp * 32], 32); }
      oct[k] = last_.kps[k].octave; ang[k] = last_.kps[k].angle;
    }
    std::vector<int> kp_slot(n, -1), match(nl, -1);
    int nm = 0;
    TIMED("cms_search_by_projection", cms_search_by_projection(cur_, 0, pose12, nl, valid.data(), Xw.data(), oct.data(), ang.data(), mdesc.data(), 15.0f, 1, 100, n, kp_slot.data(),
                                   match.data(), &nm));
    if (nm < 20) {
      std::fill(kp_slot.begin(), kp_slot.end(), -1);
      TIMED("cms_search_by_projection", cms_search_by_projection(cur_, 0, pose12, nl, valid.data(), Xw.data(), oct.data(), ang.data(), mdesc.data(), 30.0f, 1, 100, n, kp_slot.data(),
                                     match.data(), &nm));
    }
    log << ", \"n_mm\": " << nm;
    if (nm < 20) { state = LOST; return; }
    for (int k = 0; k < n; ++k) if (kp_slot[k] >= 0) cur.kp_mp[k] = last_.kp_mp[kp_slot[k]];
    optimize_pose(cur);
    int n_after = 0;
    for (int k = 0; k < n; ++k) { if (cur.outlier[k]) cur.kp_mp[k] = -1; cur.outlier[k] = 0; n_after += cur.kp_mp[k] >= 0; }     // discard outliers (:655-671)
    log << ", \"n_mm_inliers\": " << n_after;
    if (n_after < 10) { state = LOST; return; }
    // TrackLocalMap (:679-719): the local map of this small scene is the whole map; points already matched are skipped (:806-822)
    std::vector<uint8_t> taken(n_map(), 0);
    for (int k = 0; k < n; ++k) if (cur.kp_mp[k] >= 0) taken[cur.kp_mp[k]] = 1;
    std::vector<int> cand;
    for (size_t p = 0; p < n_map(); ++p) if (!taken[p]) cand.push_back((int)p);
    const int nc = (int)cand.size();
    float pose15[15];
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) pose15[3 * r + c] = cur.T.m[4 * r + c]; pose15[9 + r] = cur.T.m[4 * r + 3]; }
    camera_centre(cur.T, pose15 + 12);
    std::vector<float> cpos((size_t)nc * 3), cnrm((size_t)nc * 3), cmin(nc), cmax(nc);
    std::vector<uint8_t> cdesc((size_t)nc * 32), in_view(nc);
    for (int q = 0; q < nc; ++q) {
      const size_t p = cand[q];
      for (int c = 0; c < 3; ++c) { cpos[3 * (size_t)q + c] = mp_pos[3 * p + c]; cnrm[3 * (size_t)q + c] = mp_normal[3 * p + c]; }
      cmin[q] = mp_min[p]; cmax[q] = mp_max[p];
      std::memcpy(&cdesc[(size_t)q * 32], &mp_desc[p * 32], 32);
    }
    std::vector<int> kp_lm(n), lmatch(nc, -1);
    for (int k = 0; k < n; ++k) kp_lm[k] = cur.kp_mp[k] >= 0 ? (1 << 20) : -1;
    int nlm = 0, rounds = 0;
    if (nc > 0)
      TIMED("cms_search_local_points", cms_search_local_points(cur_, 0, pose15, nc, cpos.data(), cnrm.data(), cmin.data(), cmax.data(), cdesc.data(), 0.5f, 1.0f, 0.8f, 100, n,
                                    kp_lm.data(), in_view.data(), nullptr, nullptr, nullptr, nullptr, lmatch.data(), &nlm, &rounds));
    for (int k = 0; k < n; ++k) if (kp_lm[k] >= 0 && kp_lm[k] < (1 << 20)) cur.kp_mp[k] = cand[kp_lm[k]];
    log << ", \"n_lm\": " << nlm;
    optimize_pose(cur);
    int n_track = 0;
    for (int k = 0; k < n; ++k) n_track += cur.kp_mp[k] >= 0 && !cur.outlier[k];
    log << ", \"n_inliers\": " << n_track;
    {
      float Ow[3];
      camera_centre(cur.T, Ow);
      double cw[3];
      for (int c = 0; c < 3; ++c) cw[c] = -(gt.R[c] * gt.t[0] + gt.R[3 + c] * gt.t[1] + gt.R[6 + c] * gt.t[2]);
      const double e = std::sqrt((Ow[0] - cw[0]) * (Ow[0] - cw[0]) + (Ow[1] - cw[1]) * (Ow[1] - cw[1]) + (Ow[2] - cw[2]) * (Ow[2] - cw[2]));
      char buf[64]; std::snprintf(buf, sizeof(buf), ", \"pos_err_m\": %.6f", e);
      log << buf;
    }
    if (n_track < 30) { state = LOST; return; }
    // motion model update (:360-368): mVelocity = mCurrentFrame.mTcw * LastTwc
    Mat4f Twc = eye4();
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Twc.m[4 * r + c] = last_.T.m[4 * c + r];
    float Owl[3];
    camera_centre(last_.T, Owl);
    for (int r = 0; r < 3; ++r) Twc.m[4 * r + 3] = Owl[r];
    velocity_ = mul4(cur.T, Twc); have_velocity_ = true;
    for (int k = 0; k < n; ++k) { if (cur.outlier[k]) cur.kp_mp[k] = -1; cur.outlier[k] = 0; }      // (:377-384)
    last_ = cur;
    if (i - kfs.back().frame >= kf_every_) new_keyframe(i, gt, log);
  }

  // LocalMapping for this key frame: new points (ground-truth stand-in for CreateNewMapPoints), then Optimizer::LocalBundleAdjustment
  void new_keyframe(int i, const GroundTruth& gt, std::ostringstream& log) {
    FrameData& cur = last_;
    const int n = (int)cur.kps.size();
    float Ow[3];
    camera_centre(cur.T, Ow);
    int made = 0, tried = 0;
    for (int k = 0; k < n && tried < new_pts_; ++k) {
      if (cur.kp_mp[k] >= 0 || cur.kps[k].octave > 3) continue;
      ++tried;
      double Xd[3];
      if (!room_.point_behind_pixel(F_, gt, (double)cur.kps[k].x, (double)cur.kps[k].y, Xd)) continue;
      // the seed lives in the ground-truth world; into the estimated one through this frame: Xw_est = Twc_est * Tcw_gt * Xw
      double Xc[3], Xe[3];
      for (int r = 0; r < 3; ++r) Xc[r] = gt.R[3 * r] * Xd[0] + gt.R[3 * r + 1] * Xd[1] + gt.R[3 * r + 2] * Xd[2] + gt.t[r];
      for (int c = 0; c < 3; ++c) {
        Xe[c] = 0;
        for (int r = 0; r < 3; ++r) Xe[c] += (Xc[r] - (double)cur.T.m[4 * r + 3]) * (double)cur.T.m[4 * r + c];
      }
      const float Xf[3] = {(float)Xe[0], (float)Xe[1], (float)Xe[2]};
      cur.kp_mp[k] = add_point(Xf, &cur.desc[(size_t)k * 32], cur.kps[k].octave, Ow);
      ++made;
    }
    kfs.push_back(cur);
    log << ", \"new_points\": " << made;
    // the window (Optimizer.cpp:192-358): the last ba_window key frames are free, older ones that see the same points are fixed
    const int nk = (int)kfs.size(), first_free = nk - std::min(ba_window_, nk);
    std::vector<int> pt_index(n_map(), -1), pts;
    {
      std::vector<uint8_t> in(n_map(), 0);
      for (int j = first_free; j < nk; ++j) for (int mp : kfs[j].kp_mp) if (mp >= 0) in[mp] = 1;
      for (size_t p = 0; p < n_map(); ++p) if (in[p]) { pt_index[p] = (int)pts.size(); pts.push_back((int)p); }
    }
    std::vector<int> kf_ids;
    for (int j = 0; j < nk; ++j) {
      bool sees = false;
      for (int mp : kfs[j].kp_mp) if (mp >= 0 && pt_index[mp] >= 0) { sees = true; break; }
      if (sees) kf_ids.push_back(j);
    }
    std::vector<int> e_pose, e_point;
    std::vector<double> e_obs, e_inv;
    std::vector<int8_t> e_face;
    for (size_t kj = 0; kj < kf_ids.size(); ++kj) {
      const FrameData& kf = kfs[kf_ids[kj]];
      for (size_t k = 0; k < kf.kps.size(); ++k) {
        const int mp = kf.kp_mp[k];
        if (mp < 0 || pt_index[mp] < 0) continue;
        double ray[3];
        const double px = kf.kps[k].x, py = kf.kps[k].y;
        const int fc = pixel_to_ray(F_, px, py, ray);
        if (kf.rays[3 * k + 2] < cos_fov_ || fc < 0) continue;        // Optimizer.cpp:323-325 on the key frame's mvKeyRays
        e_pose.push_back((int)kj); e_point.push_back(pt_index[mp]);
        e_obs.push_back(px - std::floor(px / F_) * F_); e_obs.push_back(py - std::floor(py / F_) * F_);
        e_inv.push_back((double)inv_sigma2_[kf.kps[k].octave]); e_face.push_back((int8_t)fc);
      }
    }
    std::vector<uint8_t> fixed(kf_ids.size());
    bool all_fixed = true;
    for (size_t kj = 0; kj < kf_ids.size(); ++kj) { fixed[kj] = (kf_ids[kj] < first_free || kf_ids[kj] == 0) ? 1 : 0; all_fixed = all_fixed && fixed[kj]; }
    if (all_fixed || pts.size() < 10 || e_pose.empty()) return;
    std::vector<double> poses(7 * kf_ids.size()), points(3 * pts.size());
    for (size_t kj = 0; kj < kf_ids.size(); ++kj) pose7_from_T(kfs[kf_ids[kj]].T, &poses[7 * kj]);
    for (size_t p = 0; p < pts.size(); ++p) for (int c = 0; c < 3; ++c) points[3 * p + c] = (double)mp_pos[3 * (size_t)pts[p] + c];
    std::vector<uint8_t> out(e_pose.size(), 0);
    cms_ba_stats st;
    TIMED("cms_ba_run", cms_ba_run(device_, (int)kf_ids.size(), poses.data(), fixed.data(), (int)pts.size(), points.data(), (int)e_pose.size(), e_pose.data(), e_point.data(),
                     e_obs.data(), e_inv.data(), e_face.data(), F_ / 2.0, F_ / 2.0, F_ / 2.0, F_ / 2.0, 5, 10, nullptr, out.data(), &st));
    int nout = 0;
    for (uint8_t o : out) nout += o;
    log << ", \"ba_edges\": " << e_pose.size() << ", \"ba_iterations\": [" << st.iterations_done[0] << ", " << st.iterations_done[1] << "], \"ba_outliers\": " << nout
        << ", \"ba_kfs\": " << kf_ids.size() << ", \"ba_points\": " << pts.size();
    // write-back through float (Optimizer.cpp:419-449); observations flagged as outliers are erased (:424-434)
    for (size_t kj = 0; kj < kf_ids.size(); ++kj) if (!fixed[kj]) kfs[kf_ids[kj]].T = T_from_pose7(&poses[7 * kj]);
    for (size_t p = 0; p < pts.size(); ++p) for (int c = 0; c < 3; ++c) mp_pos[3 * (size_t)pts[p] + c] = (float)points[3 * p + c];
    for (size_t e = 0; e < out.size(); ++e)
      if (out[e]) {
        FrameData& kf = kfs[kf_ids[e_pose[e]]];
        const int mp = pts[e_point[e]];
        for (int& v : kf.kp_mp) if (v == mp) v = -1;
      }
    // the current frame is the newest key frame: tracking continues from its refined pose
    last_.T = kfs.back().T; last_.kp_mp = kfs.back().kp_mp;
  }
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 6) {
    std::fprintf(stderr, "usage: %s settings.yaml image_list image_dir mask.pgm ground_truth.txt [--kf-every N] [--ba-window N] [--new-points N] "
                         "[--warmup N] [--log file] [--trajectory file] [--perf file] [--device D]\n", argv[0]);
    return 1;
  }
  const std::string settings_path = argv[1], list_path = argv[2], dir = argv[3], mask_path = argv[4], gt_path = argv[5];
  int kf_every = 5, ba_window = 8, new_pts = 400, warmup = 6, device = 0;
  std::string log_path, traj_path, perf_path = "perf.txt";
  for (int a = 6; a + 1 < argc; a += 2) {
    const std::string k = argv[a], v = argv[a + 1];
    if (k == "--kf-every") kf_every = std::atoi(v.c_str()); else if (k == "--ba-window") ba_window = std::atoi(v.c_str());
    else if (k == "--new-points") new_pts = std::atoi(v.c_str()); else if (k == "--warmup") warmup = std::atoi(v.c_str());
    else if (k == "--log") log_path = v; else if (k == "--trajectory") traj_path = v; else if (k == "--perf") perf_path = v;
    else if (k == "--device") device = std::atoi(v.c_str());
    else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 1; }
  }
  Settings st;
  if (!st.Load(settings_path)) { std::fprintf(stderr, "cannot open settings %s\n", settings_path.c_str()); return 1; }
  const CubemapSLAM::ImageList list = CubemapSLAM::LoadImageListLafida(list_path);
  if (list.names.empty()) { std::fprintf(stderr, "empty image list %s\n", list_path.c_str()); return 1; }
  Image mask;
  if (!read_pgm(mask_path, mask)) { std::fprintf(stderr, "cannot read mask %s\n", mask_path.c_str()); return 1; }
  Room room;
  std::vector<GroundTruth> gts;
  {
    std::ifstream f(gt_path);
    std::string word;
    if (!(f >> word) || word != "room" || !(f >> room.half[0] >> room.half[1] >> room.half[2])) { std::fprintf(stderr, "bad ground-truth file %s\n", gt_path.c_str()); return 1; }
    GroundTruth g;
    while (f >> g.R[0]) {
      for (int i = 1; i < 9; ++i) f >> g.R[i];
      for (int i = 0; i < 3; ++i) f >> g.t[i];
      gts.push_back(g);
    }
  }
  if (gts.size() < list.names.size()) { std::fprintf(stderr, "ground truth has %zu poses for %zu images\n", gts.size(), list.names.size()); return 1; }
  // like the reference's main loop, every image is read inside the loop (cubemap_lafida.cpp:128-136); reading is not part of the tracking time
  std::vector<Image> images(list.names.size());
  for (size_t i = 0; i < list.names.size(); ++i)
    if (!read_pgm(dir + "/" + list.names[i], images[i])) { std::fprintf(stderr, "cannot read image %s\n", (dir + "/" + list.names[i]).c_str()); return 1; }
  {   // warm-up on a throw-away tracker: first launches, allocations (the reference's first frames pay its vocabulary load instead)
    Tracker w(st, mask, room, device, kf_every, ba_window, new_pts);
    for (int i = 0; i < warmup && i < (int)images.size(); ++i) w.feed(i, images[i], gts[i]);
  }
  g_calls.clear();
  Tracker trk(st, mask, room, device, kf_every, ba_window, new_pts);
  std::vector<float> vTimesTrack;
  std::ofstream logf;
  if (!log_path.empty()) logf.open(log_path);
  for (size_t i = 0; i < images.size(); ++i) {
    const auto t1 = std::chrono::steady_clock::now();
    trk.feed((int)i, images[i], gts[i]);
    const auto t2 = std::chrono::steady_clock::now();
    const double ttrack = std::chrono::duration_cast<std::chrono::duration<double>>(t2 - t1).count();   // cubemap_lafida.cpp:141-150
    vTimesTrack.push_back((float)ttrack);
    if (logf.is_open()) logf << trk.last_log.substr(0, trk.last_log.size() - 1) << ", \"ms\": " << 1e3 * ttrack << "}\n";
  }
  std::printf("state: %s, key frames: %zu, map points: %zu\n", trk.state == Tracker::OK ? "ok" : trk.state == Tracker::LOST ? "lost" : "not initialized", trk.kfs.size(),
              trk.mp_min.size());
  std::vector<float> times = vTimesTrack;
  const std::string summary = CubemapSLAM::WriteTrackingSummary(perf_path, times, (int)images.size());
  std::fputs(summary.c_str(), stdout);
  for (const CallTimes::Slot& sl : g_calls.slots)
    std::printf("call times: %-30s %6ld calls, %8.3f ms each, %8.3f ms per image\n", sl.name, sl.n, 1e3 * sl.s / sl.n, 1e3 * sl.s / (double)images.size());
  if (!traj_path.empty()) {
    // System::SaveKeyFrameTrajectoryTUM (System.cpp:238-268): "ts tx ty tz qx qy qz qw" of the camera centre and R^T, per key frame
    std::ofstream f(traj_path);
    f.setf(std::ios::fixed);
    for (const FrameData& kf : trk.kfs) {
      float Ow[3];
      camera_centre(kf.T, Ow);
      Mat4f Tt = eye4();
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Tt.m[4 * r + c] = kf.T.m[4 * c + r];
      double p7[7];
      pose7_from_T(Tt, p7);
      f.precision(6); f << list.timestamps[kf.frame];
      f.precision(7); f << " " << Ow[0] << " " << Ow[1] << " " << Ow[2] << " " << (float)p7[3] << " " << (float)p7[4] << " " << (float)p7[5] << " " << (float)p7[6] << "\n";
    }
  }
  return trk.state == Tracker::OK ? 0 : 3;
}
