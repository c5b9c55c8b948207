// OrbExtractorHip.h -- drop-in for the reference's include/ORBExtractor.h + src/ORBExtractor.cpp: same class name, constructor,
// operator(), getters and public members (ORBExtractor.h:55-90), implemented on libcubemapslam_hip.so.
// Not BUILT in this repository (needs OpenCV); syntax-checked against the reference's real headers by tests/test_integration_syntax.py;
// see integration/README.md.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H
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
    // same buffer, geometry AND fingerprint (a few thousand sampled words: a caller that REWRITES the same cv::Mat in place is noticed unless the
    // change misses every sample -- SetMask / InvalidateMask below are the guaranteed way after an in-place edit).
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
  // FNV-1a over one 8-byte word in 61 of every 16th row (~10^4 words of a 1650 x 1650 mask: a few microseconds)
  static unsigned long long Fingerprint(const cv::Mat& m) {
    unsigned long long h = 1469598103934665603ull;
    for (int y = 0; y < m.rows; y += 16) {
      const unsigned char* row = m.data + (size_t)y * (size_t)m.step;
      for (int x = 0; x + 8 <= m.cols; x += 8 * 61) {
        unsigned long long w = 0;
        for (int b = 0; b < 8; ++b) w |= (unsigned long long)row[x + b] << (8 * b);
        h = (h ^ w) * 1099511628211ull;
      }
    }
    return h;
  }
};
#endif
