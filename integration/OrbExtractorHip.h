// OrbExtractorHip.h -- drop-in for the reference's include/ORBExtractor.h + src/ORBExtractor.cpp: same class name, constructor,
// operator(), getters and public members (ORBExtractor.h:55-90), implemented on libcubemapslam_hip.so.
// Not BUILT in this repository (needs OpenCV); syntax-checked against the reference's real headers by tests/test_integration_syntax.py;
// see integration/README.md.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H
#include <cstring>
#include <vector>
#include <opencv/cv.h>
#include "cubemapslam_hip.h"
#include "CubemapHipBridge.h"

class ORBextractor {
 public:
  ORBextractor(int nfeatures_, float scaleFactor_, int nlevels_, int iniThFAST_, int minThFAST_)
      : nfeatures(nfeatures_), scaleFactor(scaleFactor_), nlevels(nlevels_), iniThFAST(iniThFAST_), minThFAST(minThFAST_) {
    const cms_orb_params orb = {nfeatures_, scaleFactor_, nlevels_, iniThFAST_, minThFAST_};
    ctx = Hip::CreateContext(orb);                                   // LUT + tables of ORBExtractor.cpp:381-442 on the device
    cms_geometry g;
    cms_ctx_geometry(ctx, &g);
    mvScaleFactor.assign(g.scale, g.scale + nlevels); mvInvScaleFactor.assign(g.inv_scale, g.inv_scale + nlevels);
    mvLevelSigma2.assign(g.sigma2, g.sigma2 + nlevels); mvInvLevelSigma2.assign(g.inv_sigma2, g.inv_sigma2 + nlevels);
    mvImagePyramid.resize(nlevels); mvMaskPyramid.resize(nlevels);
    kp_cap = g.kp_cap;
    for (int l = 0; l < nlevels; ++l) { level_w.push_back(g.level_w[l]); level_h.push_back(g.level_h[l]); }
  }
  ~ORBextractor() { cms_ctx_destroy(ctx); }

  // ORBExtractor.cpp:838-926.  image: CV_8UC1 3F x 3F cubemap, mask: CV_8UC1 (asserts :845-848), empty image -> silent return (:841)
  void operator()(cv::InputArray _image, cv::InputArray _mask, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors) {
    if (_image.empty()) return;
    cv::Mat image = _image.getMat(), mask = _mask.getMat();
    assert(image.type() == CV_8UC1);
    assert(mask.type() == CV_8UC1 && !mask.empty());
    // The mask is the same cv::Mat every frame (Tracking.cpp:131), so its upload (3F x 3F bytes) is skipped while it is recognisably the same one:
    // same buffer, geometry AND a hash of all of its pixels (a caller that REWRITES the same cv::Mat in place is noticed; SetMask / InvalidateMask
    // below force an upload).
    const unsigned long long fp = Fingerprint(mask);
    if (mask.data != last_mask || mask.rows != last_rows || mask.cols != last_cols || (size_t)mask.step != last_step || fp != last_fp) SetMask(mask);
    std::vector<cms_keypoint> k(kp_cap);
    cv::Mat d(kp_cap, 32, CV_8U);
    int n = 0;
    if (cms_extract(ctx, image.data, (int)image.step, k.data(), d.data, kp_cap, &n) != CMS_OK) { _keypoints.clear(); _descriptors.release(); return; }
    _keypoints.resize(n);
    for (int i = 0; i < n; ++i) _keypoints[i] = cv::KeyPoint(k[i].x, k[i].y, k[i].size, k[i].angle, k[i].response, k[i].octave);
    if (n == 0) _descriptors.release(); else d.rowRange(0, n).copyTo(_descriptors);
    if (mbKeepImagePyramid)                                          // the reference keeps the padded pyramid in mvImagePyramid (ORBExtractor.h:89);
      for (int l = 0; l < nlevels; ++l) {                            // nothing on the hot path reads it, so it is fetched only on request
        mvImagePyramid[l].create(level_h[l], level_w[l], CV_8U);
        cms_debug_level(ctx, 0, l, mvImagePyramid[l].data, (int)mvImagePyramid[l].step);
      }
  }

  // upload `mask` now, whatever was cached (call after editing a mask cv::Mat in place); InvalidateMask: the next operator() uploads its mask
  void SetMask(const cv::Mat& mask) {
    assert(mask.type() == CV_8UC1 && !mask.empty());
    cms_set_mask(ctx, mask.data, (int)mask.step);
    last_mask = mask.data; last_rows = mask.rows; last_cols = mask.cols; last_step = (size_t)mask.step; last_fp = Fingerprint(mask);
  }
  void InvalidateMask() { last_mask = nullptr; }

  int inline GetLevels() { return nlevels; }
  float inline GetScaleFactor() { return scaleFactor; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

  std::vector<cv::Mat> mvImagePyramid;        // level images WITHOUT the 19-px border (no consumer reads the border); filled when mbKeepImagePyramid
  std::vector<cv::Mat> mvMaskPyramid;         // resized and never filled in the reference either (ORBExtractor.cpp:405)
  bool mbKeepImagePyramid = false;
  cms_ctx* ctx = nullptr;                     // the frame context: Hip:: matchers take it to search the frame this extractor produced

 protected:
  int nfeatures; double scaleFactor; int nlevels; int iniThFAST; int minThFAST;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  std::vector<int> level_w, level_h;
  int kp_cap = 0;
  const unsigned char* last_mask = nullptr;
  int last_rows = 0, last_cols = 0; size_t last_step = 0; unsigned long long last_fp = 0;
  // A hash of EVERY pixel of the mask (four interleaved multiply-xor lanes over 8-byte words; 2.7 MB at F = 550: ~0.2 ms per frame on one core,
  // against 1-2 ms for the upload it saves).  (Until round 6 this sampled one word in 61 of every 16th row -- about 400 words, 0.1 % of the mask:
  // an in-place edit of the cached cv::Mat was very likely to miss every sample.)
  static unsigned long long Fingerprint(const cv::Mat& m) {
    unsigned long long h[4] = {1469598103934665603ull, 0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull};
    for (int y = 0; y < m.rows; ++y) {
      const unsigned char* row = m.data + (size_t)y * (size_t)m.step;
      int x = 0;
      for (; x + 32 <= m.cols; x += 32)
        for (int l = 0; l < 4; ++l) {
          unsigned long long w;
          std::memcpy(&w, row + x + 8 * l, 8);
          h[l] = (h[l] ^ w) * 1099511628211ull;
        }
      for (; x < m.cols; ++x) h[x & 3] = (h[x & 3] ^ row[x]) * 1099511628211ull;
    }
    return (h[0] ^ (h[1] << 1 | h[1] >> 63)) ^ ((h[2] << 2 | h[2] >> 62) ^ (h[3] << 3 | h[3] >> 61));
  }
};
#endif
