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
}  // namespace Hip
#endif
