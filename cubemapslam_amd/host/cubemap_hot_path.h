// cubemap_hot_path.h -- host-side C++ mirror of the reference interfaces on the hot path, implemented on top of the
// C-ABI of libcubemapslam_hip.so (include/cubemapslam_hip.h).  Same class / method names, argument meaning and error
// behaviour as the reference so Tracking / LocalMapping style call sites read unchanged:
//
//   CamModelGeneral::GetCamera()->SetCamParams(...)                 include/CamModelGeneral.h:100-108, src/System.cpp:63-89
//   System::CreateUndistortRectifyMap / CvtFisheyeToCubeMap_reverseQuery_withInterpolation
//                                                                   include/System.h:104-107, src/System.cpp:301-355
//   ORBextractor(int,float,int,int,int) / operator()(image, mask, keypoints, descriptors) / Get* accessors
//                                                                   include/ORBExtractor.h:55-90, src/ORBExtractor.cpp:838-926
//   ORBMatcher(nnratio, checkOri) / DescriptorDistance / SearchByProjection(CurrentFrame, LastFrame, th, mono)
//                                                                   include/ORBMatcher.h:46-84, src/ORBMatcher.cpp:130-251,951-967
//   Optimizer::LocalBundleAdjustment(...)                           include/Optimizer.h:50, src/Optimizer.cpp:192-451
//
// The pointer-graph types the reference passes around (Frame, KeyFrame, MapPoint, Map) are out of scope (SURVEY.md
// section 2); the two methods that take them receive plain views instead (FrameView, LocalBAWindow) holding exactly the
// fields those methods read.
#ifndef CMS_CUBEMAP_HOT_PATH_H
#define CMS_CUBEMAP_HOT_PATH_H
#include <string>
#include <vector>
#include "cubemapslam_hip.h"
#include "mini_cv.h"

namespace CubemapSLAM {

class CamModelGeneral {
 public:
  enum eFace { UNKNOWN_FACE = -1, FRONT_FACE = 0, LEFT_FACE = 1, RIGHT_FACE = 2, UPPER_FACE = 3, LOWER_FACE = 4 };
  static CamModelGeneral* GetCamera();
  // same argument order as the reference overload used by System (CamModelGeneral.h:105-108); poly / invpoly zero padded
  void SetCamParams(const double cdeu0v0[5], const std::vector<double>& poly, const std::vector<double>& invpoly, double Iw,
                    double Ih, double fx, double fy, double cx, double cy, double width, double height, double camFov);
  int GetCubeFaceWidth() const { return cam_.face; }
  int GetCubeFaceHeight() const { return cam_.face; }
  int GetFisheyeWidth() const { return cam_.Iw; }
  int GetFisheyeHeight() const { return cam_.Ih; }
  double Get_fx() const { return cam_.face / 2.0; }
  float GetCosFovTh() const { return cosFovTh_; }
  eFace FaceInCubemap(const cv::Point2f& pixel) const;                          // CamModelGeneral.h:445-456
  void GetPosInFace(double& u, double& v, double uCubemap, double vCubemap) const;  // CamModelGeneral.h:204-209
  const cms_camera& params() const { return cam_; }
  bool configured() const { return configured_; }

 private:
  cms_camera cam_{};
  float cosFovTh_ = 0;
  bool configured_ = false;
};

// Owner of the device context shared by the remap entry point and the extractor (one per process, like the reference's
// camera singleton + System instance).  Throws std::runtime_error when the HIP library cannot create a context.
cms_ctx* SharedContext(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);

// one key frame of the trajectory output: KeyFrame::mTimeStamp and its world->camera pose (KeyFrame::GetPose(), 4x4 CV_32F)
struct TrajectoryKeyFrame { double mTimeStamp; cv::Mat Tcw; bool bad = false; };

class System {
 public:
  // System.cpp:238-268: one line per (non-bad) key frame, "ts tx ty tz qx qy qz qw" with the camera centre -R^T t and the
  // quaternion of R^T through float, fixed notation, 6 digits for the time stamp and 7 for the rest.  The caller passes
  // the key frames already sorted by id (the reference sorts with KeyFrame::lId).
  static void SaveKeyFrameTrajectoryTUM(const std::string& filename, const std::vector<TrajectoryKeyFrame>& vpKFs);
  void CreateUndistortRectifyMap();  // builds the LUT on the device (inside the shared context)
  void CvtFisheyeToCubeMap_reverseQuery_withInterpolation(cv::Mat& cubemapImg, const cv::Mat& fisheyeImg, int interpolation,
                                                          int borderType = cv::BORDER_CONSTANT,
                                                          const cv::Scalar& borderValue = cv::Scalar());
};

class ORBextractor {
 public:
  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
  // image: CV_8UC1 3F x 3F cubemap; mask: CV_8UC1 non-empty (assert in the reference, ORBExtractor.cpp:845-848);
  // empty image -> silent return (ORBExtractor.cpp:841).
  void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors);
  int GetLevels() const { return nlevels; }
  float GetScaleFactor() const { return (float)scaleFactor; }
  std::vector<float> GetScaleFactors() const { return mvScaleFactor; }
  std::vector<float> GetInverseScaleFactors() const { return mvInvScaleFactor; }
  std::vector<float> GetScaleSigmaSquares() const { return mvLevelSigma2; }
  std::vector<float> GetInverseScaleSigmaSquares() const { return mvInvLevelSigma2; }

  // ORBExtractor.h:89-90.  The reference keeps the (19-px padded) level images of the last call here; no stage of the hot path reads
  // them afterwards, so the mirror only fetches them from the device when a caller sets mbKeepImagePyramid -- level l is the
  // cvRound(W / scale_l)^2 image without the border frame ("19-px REFLECT_101 frame").
  std::vector<cv::Mat> mvImagePyramid;
  std::vector<cv::Mat> mvMaskPyramid;      // resized and never filled in the reference either (ORBExtractor.cpp:405)
  bool mbKeepImagePyramid = false;

 protected:
  int nfeatures; double scaleFactor; int nlevels, iniThFAST, minThFAST;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  cms_ctx* ctx_ = nullptr;
  const uint8_t* last_mask_ = nullptr;
};

// What ORBMatcher::SearchByProjection(Frame&, const Frame&, th, mono) reads from a Frame (Frame.h): key points,
// descriptors, the map-point slot per key point (>=0 = id of the assigned map point) and, for the last frame, the
// projection of each of its map points into the current frame (u, v in cubemap pixels, <0 = not visible).
struct FrameView {
  std::vector<cv::KeyPoint> mvKeys;
  cv::Mat mDescriptors;                 // N x 32 CV_8U
  std::vector<long> mvpMapPoints;       // -1 = none
  std::vector<uint8_t> mvbOutlier;
  std::vector<cv::Point2f> projInCurrent;  // last frame only: where map point i projects in the current frame
  std::vector<float> mvScaleFactors;
  cv::Mat mTcw;                         // 4x4 CV_32F (Frame::SetPose): SearchLocalPoints, and SearchByProjection's device path
  // last frame only, optional: world position and descriptor of the map point behind key point i (MapPoint::GetWorldPos / GetDescriptor).
  // When given (and CurrentFrame.mTcw is set) SearchByProjection projects on the device instead of reading projInCurrent.
  std::vector<cv::Vec3f> mvMapPointPos;
  cv::Mat mMapPointDescriptors;         // N x 32 CV_8U
};

struct KeyFrameView;
struct MapPointView;

class ORBMatcher {
 public:
  ORBMatcher(float nnratio = 0.6f, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
  static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);   // ORBMatcher.cpp:951-967 (single pair, host)
  // ORBMatcher.cpp:130-251: windows of radius th*scale[octave] around the projections, octave +-1, best Hamming <= TH_HIGH,
  // rotation-histogram consistency.  With LastFrame.mvMapPointPos + CurrentFrame.mTcw the whole function runs on the device
  // (cms_search_by_projection: projection, windows, greedy, histogram).  Otherwise the projections are read from projInCurrent,
  // distances / arg-min run on the GPU (cms_hamming_best2) and the greedy acceptance is replayed on the host in the reference's
  // order.  Returns the number of matches, fills CurrentFrame.mvpMapPoints.
  int SearchByProjection(FrameView& CurrentFrame, const FrameView& LastFrame, float th, bool bMono);
  // ORBMatcher.cpp:50-128, Tracking.cpp:841: search matches between the frame's key points and the projected map points Frame::isInFrustum marked
  // (mbTrackInView with mTrackProjX / Y, mnTrackScaleLevel, mTrackViewCos).  Points not in view are skipped like the reference does; for the
  // others the device repeats isInFrustum on F.mTcw (the same arithmetic gives the same fields) and runs the greedy search with mfNNratio.
  // Fills F.mvpMapPoints with the matched points' ids, returns the number of matches.
  int SearchByProjection(FrameView& F, std::vector<MapPointView>& vpMapPoints, float th = 3.0f);
  // ORBMatcher.cpp:676-794, Tracking.cpp:428-429: level-0 key points of F1 against F2 inside windows around vbPrevMatched (updated for the matches
  // like the reference), best / second best with mfNNratio, take-over of a key point of F2 by a strictly better match, rotation histogram.
  // vnMatches12[i1] = index in F2 or -1; returns the number of matches.  (cms_search_for_initialization)
  int SearchForInitialization(FrameView& F1, FrameView& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10);
  // ORBMatcher.cpp:971-1125, LocalMapping.cpp:254: features of pKF1 without a map point against those of pKF2 in the same vocabulary node,
  // epipole distance and epipolar gate with E12 (3x3 CV_32F).  vMatchedPairs in ascending idx1.  (cms_search_for_triangulation)
  int SearchForTriangulation(const KeyFrameView& pKF1, const KeyFrameView& pKF2, const cv::Mat& E12, std::vector<std::pair<size_t, size_t>>& vMatchedPairs);
  // ORBMatcher.cpp:1127-1226: search on the device (cms_fuse_search), then the reference's Replace / AddObservation decisions in list
  // order: fused[i] = key point map point i is fused with (or -1); returns nFused.  mvpMapPoints of pKF is updated for additions.
  int Fuse(KeyFrameView& pKF, const std::vector<MapPointView>& vpMapPoints, const std::vector<uint8_t>& skip, float th, std::vector<int>& fused);
  static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 12;

 protected:
  float mfNNratio;
  bool mbCheckOrientation;
};

// What Frame::isInFrustum and ORBMatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) read and write on a MapPoint
// (include/MapPoint.h): position, mean viewing direction, scale-invariance distances, representative descriptor, and the
// mbTrackInView / mTrackProj* / mnTrackScaleLevel / mTrackViewCos fields Tracking uses afterwards.
struct MapPointView {
  long mnId = -1;
  cv::Mat mWorldPos, mNormalVector;     // 3x1 CV_32F
  float mfMinDistance = 0, mfMaxDistance = 0;
  cv::Mat mDescriptor;                  // 1x32 CV_8U
  bool mbTrackInView = false;
  float mTrackProjX = -1, mTrackProjY = -1, mTrackViewCos = 0;
  int mnTrackScaleLevel = -1;
};

// Tracking::SearchLocalPoints (src/Tracking.cpp:794-846): Frame::isInFrustum(pMP, 0.5) for every local map point, then
// ORBMatcher(0.8).SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th).  Both halves run on the device in one call
// (cms_search_local_points); F.mTcw is the 4x4 CV_32F pose (Frame::SetPose), F.mvpMapPoints gets the id of the matched point.
class Tracking {
 public:
  static int SearchLocalPoints(FrameView& F, std::vector<MapPointView>& vpLocalMapPoints, float th = 1.0f, float nnratio = 0.8f,
                               float viewingCosLimit = 0.5f);
};

// The window Optimizer::LocalBundleAdjustment assembles from pKF's covisibility graph (Optimizer.cpp:194-357), as data.
struct LocalBAWindow {
  struct KF { long mnId; cv::Mat Tcw; bool fixed; std::vector<float> mvInvLevelSigma2; };   // Tcw: 4x4 CV_32F
  struct Obs { int kf; cv::KeyPoint kp; cv::Vec3f ray; };                                   // kf = index into keyframes
  struct MP { long mnId; cv::Mat Xw; std::vector<Obs> observations; };                      // Xw: 3x1 CV_32F
  std::vector<KF> keyframes;
  std::vector<MP> mappoints;
  std::vector<std::pair<int, int>> toErase;   // out: (keyframe index, map point index) observations to erase
};

// The members of Frame that Optimizer::PoseOptimization reads and writes (Frame.h: mTcw, N, mvKeys, mvKeyRays, mvpMapPoints,
// mvbOutlier, mvInvLevelSigma2); a map point is represented by its world position (MapPoint::GetWorldPos()).
struct PoseFrame {
  cv::Mat mTcw;                                 // 4x4 CV_32F, in/out (Frame::SetPose)
  int N = 0;
  std::vector<cv::KeyPoint> mvKeys;
  std::vector<cv::Vec3f> mvKeyRays;
  std::vector<bool> mvbHasMapPoint;             // mvpMapPoints[i] != NULL
  std::vector<cv::Vec3f> mvMapPointPos;         // pMP->GetWorldPos() where mvbHasMapPoint[i]
  std::vector<bool> mvbOutlier;                 // out
  std::vector<float> mvInvLevelSigma2;
};

// What LocalMapping::CreateNewMapPoints, ORBMatcher::SearchForTriangulation and ORBMatcher::Fuse read of a KeyFrame
// (include/KeyFrame.h): key points, descriptors, key rays, the map-point slot per key point (>= 0 = id), pose, mFeatVec
// (DBoW2::FeatureVector: node id -> feature indices, std::map order) and ComputeSceneMedianDepth(2).
struct KeyFrameView {
  long mnId = -1;
  std::vector<cv::KeyPoint> mvKeys;
  cv::Mat mDescriptors;                                  // N x 32 CV_8U
  std::vector<cv::Vec3f> mvKeyRays;
  std::vector<long> mvpMapPoints;                        // -1 = none
  cv::Mat Tcw;                                           // 4x4 CV_32F
  std::vector<std::pair<unsigned, std::vector<unsigned>>> mFeatVec;   // ascending node id
  float medianDepth = 1.0f;                              // ComputeSceneMedianDepth(2)
};
struct NewMapPoint { int neighbour; int idx1, idx2; cv::Vec3f x3D; };

class LocalMapping {
 public:
  // LocalMapping.cpp:209-386 with the covisible neighbours already collected (GetBestCovisibilityKeyFrames(20)): returns the points
  // the reference would create, in creation order; the MapPoint bookkeeping (:359-381) stays with the caller.
  static std::vector<NewMapPoint> CreateNewMapPoints(const KeyFrameView& current, const std::vector<const KeyFrameView*>& vpNeighKFs);
};

class Optimizer {
 public:
  // Optimizer.cpp:48-190: edges collected exactly like :80-127 (rays with z < cosFovTh skipped, face + in-face measurement of
  // the key point, invSigma2 of its octave, map point through float), the four optimisation rounds run in one GPU kernel
  // (cms_pose_optimize), the pose is written back through float (Converter::toCvMat).  Returns nInitialCorrespondences - nBad.
  static int PoseOptimization(PoseFrame* pFrame);

  // Optimizer.cpp:192-451 with the graph already collected: builds the edge list exactly like :246-357 (fixed = mnId==0 or
  // fixed KF, rays with z < cosFovTh skipped, face + in-face measurement from the key point, invSigma2 of the octave),
  // runs optimize(5) / classify / optimize(10) on the GPU, writes poses and points back through float (Converter.cpp:53-104)
  // and lists the observations to erase.  *pbStopFlag is polled like g2o's forceStopFlag.
  static void LocalBundleAdjustment(LocalBAWindow* window, bool* pbStopFlag);
};

}  // namespace CubemapSLAM
#endif
