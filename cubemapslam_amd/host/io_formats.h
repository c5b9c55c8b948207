// io_formats.h -- the reference's file formats around the hot path, plain C++ (no OpenCV, no GPU):
//   settings YAML          cv::FileStorage keys read at src/System.cpp:63-91 and src/Tracking.cpp:61-93 (Config/*.yaml)
//   image lists            Examples/cubemap_lafida.cpp:91-107 ("<timestamp> <path>" per line, directory stripped) and
//                          Examples/cubemap_fangshan.cpp:93-101 ("<timestamp>_<suffix>" file names, one per line)
//   tracking-time summary  Examples/cubemap_lafida.cpp:160-179
// (the TUM key-frame trajectory writer, src/System.cpp:238-268, lives in System::SaveKeyFrameTrajectoryTUM.)
#ifndef CUBEMAPSLAM_IO_FORMATS_H
#define CUBEMAPSLAM_IO_FORMATS_H
#include <map>
#include <string>
#include <vector>
#include "cubemapslam_hip.h"

namespace CubemapSLAM {

// The YAML subset cv::FileStorage files of the reference use: "%YAML:1.0" header, comments, flat "Key.sub: scalar" lines.
class Settings {
 public:
  bool Load(const std::string& path);                 // false when the file cannot be opened
  bool LoadFromString(const std::string& text);
  bool Has(const std::string& key) const { return values_.count(key) != 0; }
  double Real(const std::string& key) const;          // cv::FileNode -> double; a missing key reads as 0 like an empty FileNode
  int Int(const std::string& key) const;              // cv::FileNode -> int (rounds a real like cvRound)
  std::string String(const std::string& key) const;
  // System.cpp:63-89: polynomial arrays zero padded to 5 / 12 entries, Camera.{Iw,Ih,c,d,e,u0,v0,fov}, CubeFace.w
  cms_camera Camera() const;
  // Tracking.cpp:88-96
  cms_orb_params Orb() const;
  float Fps() const;                                  // Camera.fps, 30 when 0 or absent (Tracking.cpp:66-68)
  int WithFisheyeMask() const { return Int("Camera.withFisheyeMask"); }
  bool RGB() const { return Int("Camera.RGB") != 0; }

 private:
  std::map<std::string, std::string> values_;
};

struct ImageList { std::vector<std::string> names; std::vector<double> timestamps; };
ImageList LoadImageListLafida(const std::string& path);     // cubemap_lafida.cpp:91-107
ImageList LoadImageListFangshan(const std::string& path);   // cubemap_fangshan.cpp:93-101

// cubemap_lafida.cpp:160-179: sorts the times, median = v[n/2], mean = float sum / n; writes the perf file and returns the text
// the reference prints to stdout.  vTimesTrack is sorted in place like the reference does.
std::string WriteTrackingSummary(const std::string& perfSavingPath, std::vector<float>& vTimesTrack, int frame_counter);

}  // namespace CubemapSLAM
#endif
