#include "io_formats.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace CubemapSLAM {

static std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) ++a;
  while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) --b;
  return s.substr(a, b - a);
}

bool Settings::LoadFromString(const std::string& text) {
  values_.clear();
  std::istringstream in(text);
  std::string line;
  while (std::getline(in, line)) {
    std::string t = trim(line);
    if (t.empty() || t[0] == '#' || t[0] == '%' || t.compare(0, 3, "---") == 0) continue;
    const size_t colon = t.find(':');
    if (colon == std::string::npos) continue;
    std::string key = trim(t.substr(0, colon)), val = t.substr(colon + 1);
    if (!val.empty() && val[0] != ' ' && val[0] != '\t') continue;            // "a:b" is not a mapping entry
    bool quoted = false;
    std::string v = trim(val);
    if (!v.empty() && (v[0] == '"' || v[0] == '\'')) {
      const size_t e = v.find(v[0], 1);
      v = v.substr(1, e == std::string::npos ? std::string::npos : e - 1);
      quoted = true;
    }
    if (!quoted) {
      const size_t hash = v.find(" #");
      if (hash != std::string::npos) v = trim(v.substr(0, hash));
    }
    values_[key] = v;
  }
  return true;
}

bool Settings::Load(const std::string& path) {
  std::ifstream f(path.c_str());
  if (!f.is_open()) return false;
  std::stringstream ss;
  ss << f.rdbuf();
  return LoadFromString(ss.str());
}

double Settings::Real(const std::string& key) const {
  const auto it = values_.find(key);
  if (it == values_.end() || it->second.empty()) return 0.0;
  return std::strtod(it->second.c_str(), nullptr);
}

int Settings::Int(const std::string& key) const {
  const auto it = values_.find(key);
  if (it == values_.end() || it->second.empty()) return 0;
  const std::string& v = it->second;
  if (v.find_first_of(".eE") == std::string::npos) return (int)std::strtol(v.c_str(), nullptr, 10);
  return (int)std::nearbyint(std::strtod(v.c_str(), nullptr));               // FileNode::operator int on a real node: cvRound
}

std::string Settings::String(const std::string& key) const {
  const auto it = values_.find(key);
  return it == values_.end() ? std::string() : it->second;
}

cms_camera Settings::Camera() const {
  cms_camera c;
  std::memset(&c, 0, sizeof(c));
  const int nrpol = Int("Camera.nrpol"), nrinvpol = Int("Camera.nrinvpol");
  for (int i = 0; i < nrpol && i < 5; ++i) c.pol[i] = Real("Camera.a" + std::to_string(i));            // zero padded (System.cpp:67-69)
  for (int i = 0; i < nrinvpol && i < 12; ++i) c.invpol[i] = Real("Camera.pol" + std::to_string(i));   // (System.cpp:70-72)
  c.Iw = Int("Camera.Iw"); c.Ih = Int("Camera.Ih");
  c.c = Real("Camera.c"); c.d = Real("Camera.d"); c.e = Real("Camera.e"); c.u0 = Real("Camera.u0"); c.v0 = Real("Camera.v0");
  c.face = Int("CubeFace.w");
  c.fov_deg = Real("Camera.fov");
  return c;
}

cms_orb_params Settings::Orb() const {
  cms_orb_params o;
  o.nfeatures = Int("ORBextractor.nFeatures");
  o.scale_factor = (float)Real("ORBextractor.scaleFactor");
  o.nlevels = Int("ORBextractor.nLevels");
  o.ini_th_fast = Int("ORBextractor.iniThFAST");
  o.min_th_fast = Int("ORBextractor.minThFAST");
  return o;
}

float Settings::Fps() const {
  float fps = (float)Real("Camera.fps");
  if (fps == 0) fps = 30;
  return fps;
}

ImageList LoadImageListLafida(const std::string& path) {
  ImageList out;
  std::ifstream fin(path.c_str());
  std::string line;
  while (std::getline(fin, line)) {
    std::stringstream ss(line);
    double ts = 0;
    ss >> ts;
    std::string name;
    ss >> name;
    const size_t p = name.find_last_of("/");
    name = name.substr(p + 1, name.length());                       // npos + 1 == 0: a bare file name stays whole
    out.timestamps.push_back(ts);
    out.names.push_back(name);
  }
  return out;
}

ImageList LoadImageListFangshan(const std::string& path) {
  ImageList out;
  std::ifstream fin(path.c_str());
  std::string line;
  while (std::getline(fin, line)) {
    out.names.push_back(line);
    const size_t p = line.find_last_of("_");
    std::stringstream ss(line.substr(0, p));
    double ts = 0;
    ss >> ts;
    out.timestamps.push_back(ts);
  }
  return out;
}

std::string WriteTrackingSummary(const std::string& perfSavingPath, std::vector<float>& v, int frame_counter) {
  const int imageCnt = (int)v.size();
  std::sort(v.begin(), v.end());
  float totaltime = 0;
  for (int i = 0; i < imageCnt; ++i) totaltime += v[i];
  const float median = imageCnt > 0 ? v[imageCnt / 2] : 0.0f, mean = imageCnt > 0 ? totaltime / imageCnt : 0.0f;
  std::ostringstream console;
  console << "-------" << std::endl << std::endl;
  console << "median tracking time: " << median << std::endl;
  console << "mean tracking time: " << mean << std::endl;
  if (!perfSavingPath.empty()) {
    std::ofstream f(perfSavingPath.c_str());
    f << std::fixed;
    f << "-------" << std::endl << std::endl;
    f << "median tracking time: " << median << std::endl;
    f << "mean tracking time: " << mean << std::endl;
    f << "tracking frames/ total frames: " << frame_counter << "/ " << imageCnt << " " << static_cast<float>(frame_counter) / imageCnt << std::endl;
  }
  return console.str();
}

}  // namespace CubemapSLAM
