// CubemapHipBridge.h -- adapters between the reference's pointer-graph types and the C-ABI of libcubemapslam_hip.so.
// Written against the reference's real headers (global-namespace Frame / KeyFrame / MapPoint / Map / Converter / CamModelGeneral);
// Not built in this repository (syntax-checked against the reference's headers by tests/test_integration_syntax.py), see integration/README.md.  Each function is the new body of the reference function it names.
#ifndef CUBEMAP_HIP_BRIDGE_H
#define CUBEMAP_HIP_BRIDGE_H
#include <vector>
#include <opencv2/opencv.hpp>
#include "cubemapslam_hip.h"

class Frame;
class KeyFrame;
class MapPoint;
class Map;

namespace Hip {
// once, from System::System next to SetCosFovTh (System.cpp:86-89): Camera.fov of the settings file and the HIP device to use
void Configure(double camFovDeg, int device);
// cms_ctx for the camera the CamModelGeneral singleton holds (System.cpp:63-89 has set it before Tracking creates the extractors)
cms_ctx* CreateContext(const cms_orb_params& orb);

// System::CvtFisheyeToCubeMap_reverseQuery_withInterpolation (System.cpp:327-355)
void CvtFisheyeToCubeMap(cms_ctx* ctx, cv::Mat& cubemapImg, const cv::Mat& fisheyeImg);

// After Frame::Frame ran the extractor: upload nothing (cms_extract left key points + descriptors in slot 0), build the grid
// (Frame::AssignFeaturesToGrid, Frame.cpp:158-176).  Call once per frame before the searches below.
void FrameGrid(cms_ctx* ctx);

// ORBMatcher::SearchForInitialization (ORBMatcher.cpp:676-794); F2 is the frame `ctx` extracted last
int SearchForInitialization(cms_ctx* ctx, Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize,
                            float nnratio, bool checkOrientation);
// ORBMatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (ORBMatcher.cpp:130-251)
int SearchByProjection(cms_ctx* ctx, Frame& CurrentFrame, const Frame& LastFrame, float th, bool checkOrientation);
// Tracking::SearchLocalPoints' two halves (Tracking.cpp:794-846): Frame::isInFrustum(pMP, 0.5) + ORBMatcher(0.8).SearchByProjection(F, vpMapPoints, th)
int SearchLocalPoints(cms_ctx* ctx, Frame& F, const std::vector<MapPoint*>& vpMapPoints, float th);
// Optimizer::PoseOptimization (Optimizer.cpp:48-190)
int PoseOptimization(Frame* pFrame);
// Optimizer::LocalBundleAdjustment (Optimizer.cpp:192-451)
void LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap);
// The reference's g2o is single-threaded (ThirdParty/g2o/config.h:4): the same window gives the same bits every time.  SetDeterministic(true) makes every
// LocalBundleAdjustment call from then on add in a fixed order on the device (cms_ba_set_deterministic: 0.84-0.87 of the default throughput); the default adds with
// FP64 atomics and its last bits vary from run to run (DESIGN.md section 2 says what it guarantees).
void SetDeterministic(bool on);
bool GetDeterministic();

// ---- LocalMapping's per-key-frame sequence on key frames RESIDENT on the device (cms_kfstore_*).  The store maps KeyFrame* -> slot; a key frame enters
// it once (ProcessNewKeyFrame), its pose and map-point slots are refreshed where the reference changes them, and it leaves with ReleaseKeyFrame
// (KeyFrame::SetBadFlag / the end of the map).
cms_kfstore* CreateKeyFrameStore(cms_ctx* mappingCtx, int maxKeyFrames, int maxFeatures);
// LocalMapping::ProcessNewKeyFrame's device half (LocalMapping.cpp:52-117; the KeyFrame constructor's copy of the frame, KeyFrame.cpp:29-55): the frame
// `frameCtx` extracted last (slot 0 of its batch, FrameGrid already called) becomes pKF's resident copy -- key points, descriptors, key rays and grid
// device to device, mFeatVec (pKF->ComputeBoW() must have run) and the map-point slots from the host.  Returns the slot.
int ProcessNewKeyFrame(cms_kfstore* store, cms_ctx* frameCtx, KeyFrame* pKF);
void ReleaseKeyFrame(cms_kfstore* store, KeyFrame* pKF);
// the body of LocalMapping::CreateNewMapPoints (LocalMapping.cpp:209-386) for mpCurrentKeyFrame and its GetBestCovisibilityKeyFrames(20): the search,
// triangulation and every test on the device, the MapPoint bookkeeping (:359-381) here; newPoints receives what the reference pushes to
// mlpRecentAddedMapPoints.  Returns nnew.
int CreateNewMapPoints(cms_kfstore* store, KeyFrame* pCurrentKF, Map* pMap, std::vector<MapPoint*>& newPoints);
// the body of LocalMapping::SearchInNeighbors (LocalMapping.cpp:388-466): both Fuse directions as ONE device call (every set of map points uploaded
// once), then ORBMatcher::Fuse's Replace / AddObservation decisions in the reference's order (ORBMatcher.cpp:1213-1236) and the update of the
// current key frame's points and connections (:449-465)
void SearchInNeighbors(cms_kfstore* store, KeyFrame* pCurrentKF);
// after Optimizer::LocalBundleAdjustment's write-back (Optimizer.cpp:419-431): the resident copies of the local key frames get their new poses
void UpdateKeyFramePoses(cms_kfstore* store, const std::vector<KeyFrame*>& vpKFs);
}  // namespace Hip
#endif
