// oracle/orc_cam.cpp -- CPU oracle (TEST INFRASTRUCTURE ONLY, parity unpinned: see orc_api.h).
// Restates the Scaramuzza omnidirectional model + 5-face cubemap geometry of the reference:
//   include/CamModelGeneral.h:43-50 (horner), 262-280 (ImgToWorld), 359-374 (WorldToImg),
//   388-443 (cvtFacesToRig / cvtRigToFaces), 445-470 (FaceInCubemap), 494-513 (TransformCubemapToRays)
//   src/CamModelGeneral.cpp:95-154 (TransformRaysToCubemap), 228-263 (TransformRaysToTargetFace),
//   265-290 (CubemapToFisheye);  src/System.cpp:301-324 (CreateUndistortRectifyMap).
// The float/double mix of the reference is kept expression by expression.
#include "orc_api.h"
#include <cmath>

namespace {
inline double horner(const double* c, int n, double x) {
  double r = 0.0;
  for (int i = n - 1; i >= 0; --i) r = r * x + c[i];
  return r;
}
struct V3d { double x, y, z; };
struct V3f { float x, y, z; };

template <class T, class V> inline V faces_to_rig(const V& l, int face) {
  V r;
  switch (face) {  // CamModelGeneral.h:388-414
    case ORC_FACE_FRONT: r.x = l.x;  r.y = l.y;  r.z = l.z;  break;
    case ORC_FACE_LEFT:  r.x = -l.z; r.y = l.y;  r.z = l.x;  break;
    case ORC_FACE_RIGHT: r.x = l.z;  r.y = l.y;  r.z = -l.x; break;
    case ORC_FACE_LOWER: r.x = l.x;  r.y = l.z;  r.z = -l.y; break;
    case ORC_FACE_UPPER: r.x = l.x;  r.y = -l.z; r.z = l.y;  break;
    default: r.x = 0; r.y = 0; r.z = 0;
  }
  return r;
}
template <class V> inline V rig_to_faces(const V& g, int face) {
  V l;
  switch (face) {  // CamModelGeneral.h:417-443
    case ORC_FACE_FRONT: l.x = g.x;  l.y = g.y;  l.z = g.z;  break;
    case ORC_FACE_LEFT:  l.x = g.z;  l.y = g.y;  l.z = -g.x; break;
    case ORC_FACE_RIGHT: l.x = -g.z; l.y = g.y;  l.z = g.x;  break;
    case ORC_FACE_LOWER: l.x = g.x;  l.y = -g.z; l.z = g.y;  break;
    case ORC_FACE_UPPER: l.x = g.x;  l.y = g.z;  l.z = -g.y; break;
    default: l.x = 0; l.y = 0; l.z = 0;
  }
  return l;
}
template <class T> inline int face_of(T i, T j) {  // CamModelGeneral.h:445-470 (same tests for both overloads)
  if (i >= 0 && i < 1 && j >= 1 && j < 2) return ORC_FACE_LEFT;
  if (i >= 1 && i < 2 && j >= 0 && j < 1) return ORC_FACE_UPPER;
  if (i >= 1 && i < 2 && j >= 1 && j < 2) return ORC_FACE_FRONT;
  if (i >= 1 && i < 2 && j >= 2 && j < 3) return ORC_FACE_LOWER;
  if (i >= 2 && i < 3 && j >= 1 && j < 2) return ORC_FACE_RIGHT;
  return ORC_FACE_UNKNOWN;
}
}  // namespace

extern "C" {

void orc_world_to_img(const orc_camera* cam, double x, double y, double z, double* u, double* v) {
  double norm = std::sqrt(x * x + y * y);  // CamModelGeneral.h:359-374
  if (norm == 0.0) norm = 1e-14;
  const double theta = std::atan(-z / norm);
  const double rho = horner(cam->invpol, 12, theta);
  const double uu = x / norm * rho;
  const double vv = y / norm * rho;
  *u = uu * cam->c + vv * cam->d + cam->u0;
  *v = uu * cam->e + vv + cam->v0;
}

void orc_img_to_world(const orc_camera* cam, double u, double v, double* x, double* y, double* z) {
  const double invAffine = cam->c - cam->d * cam->e;  // CamModelGeneral.cpp:66 / h:262-280
  const double u_t = u - cam->u0, v_t = v - cam->v0;
  double X = (u_t - cam->d * v_t) / invAffine;
  double Y = (-cam->e * u_t + cam->c * v_t) / invAffine;
  const double X2 = X * X, Y2 = Y * Y;
  double Z = -horner(cam->pol, 5, std::sqrt(X2 + Y2));
  const double n = std::sqrt(X2 + Y2 + Z * Z);
  *x = X / n; *y = Y / n; *z = Z / n;
}

int orc_face_in_cubemap(const orc_camera* cam, float x, float y) {
  // FaceInCubemap(const cv::Point2f&): double i = pixel.x / mWCubeFace  (float / int, then widened)
  const float fi = x / (float)cam->face, fj = y / (float)cam->face;
  return face_of<double>((double)fi, (double)fj);
}

void orc_cubemap_to_fisheye(const orc_camera* cam, double up, double vp, double* uf, double* vf) {
  const int F = cam->face;  // CamModelGeneral.cpp:265-290
  const double fx = F / 2.0, fy = F / 2.0, cx = F / 2.0, cy = F / 2.0;
  float i = (float)up, j = (float)vp;
  *uf = -1; *vf = -1;
  const int face = face_of<float>(i / (float)F, j / (float)F);
  if (face == ORC_FACE_UNKNOWN) return;
  const double z = 1.0;
  i = i - (float)(static_cast<int>(i / (float)F) * F);
  j = j - (float)(static_cast<int>(j / (float)F) * F);
  V3d l;
  l.x = ((double)i - cx) * z / fx;
  l.y = ((double)j - cy) * z / fy;
  l.z = z;
  const V3d r = faces_to_rig<double>(l, face);
  orc_world_to_img(cam, r.x, r.y, r.z, uf, vf);
  if (*uf < 0 || *uf >= cam->Iw || *vf < 0 || *vf >= cam->Ih) { *uf = -1; *vf = -1; }
}

static inline bool project_face(const orc_camera* cam, const V3f& rig, int face, float* up, float* vp) {
  const int F = cam->face;
  const double fx = F / 2.0, fy = F / 2.0, cx = F / 2.0, cy = F / 2.0;
  const V3f l = rig_to_faces(rig, face);
  *up = (float)((double)l.x * fx / (double)l.z + cx);
  *vp = (float)((double)l.y * fy / (double)l.z + cy);
  return !(*up < 0 || *up >= F || *vp < 0 || *vp >= F);
}

int orc_rays_to_cubemap(const orc_camera* cam, float x, float y, float z, float* up, float* vp) {
  const int F = cam->face;  // CamModelGeneral.cpp:95-154
  const V3f rig = {x, y, z};
  if (z > 0 && x / z <= 1 && x / z >= -1 && y / z <= 1 && y / z >= -1) {
    if (!project_face(cam, rig, ORC_FACE_FRONT, up, vp)) return ORC_FACE_UNKNOWN;
    *up += F; *vp += F; return ORC_FACE_FRONT;
  } else if (x > 0 && y / x <= 1 && y / x >= -1 && z / x <= 1 && z / x >= -1) {
    if (!project_face(cam, rig, ORC_FACE_RIGHT, up, vp)) return ORC_FACE_UNKNOWN;
    *up += 2 * F; *vp += F; return ORC_FACE_RIGHT;
  } else if (x < 0 && y / (-x) <= 1 && y / (-x) >= -1 && z / (-x) <= 1 && z / (-x) >= -1) {
    if (!project_face(cam, rig, ORC_FACE_LEFT, up, vp)) return ORC_FACE_UNKNOWN;
    *vp += F; return ORC_FACE_LEFT;
  } else if (y > 0 && x / y <= 1 && x / y >= -1 && z / y <= 1 && z / y >= -1) {
    if (!project_face(cam, rig, ORC_FACE_LOWER, up, vp)) return ORC_FACE_UNKNOWN;
    *up += F; *vp += 2 * F; return ORC_FACE_LOWER;
  } else if (y < 0 && x / (-y) <= 1 && x / (-y) >= -1 && z / (-y) <= 1 && z / (-y) >= -1) {
    if (!project_face(cam, rig, ORC_FACE_UPPER, up, vp)) return ORC_FACE_UNKNOWN;
    *up += F; return ORC_FACE_UPPER;
  }
  *up = -1; *vp = -1;
  return ORC_FACE_UNKNOWN;
}

void orc_rays_to_target_face(const orc_camera* cam, float x, float y, float z, int face, float* up, float* vp) {
  if (face < ORC_FACE_FRONT || face > ORC_FACE_LOWER) { *up = -1.0f; *vp = -1.0f; return; }  // cpp:228-263
  const V3f rig = {x, y, z};
  project_face(cam, rig, face, up, vp);  // no bounds test in the reference here
}

int orc_cubemap_to_rays(const orc_camera* cam, float px, float py, float* ray3) {
  const int F = cam->face;  // CamModelGeneral.h:494-513
  const double fx = F / 2.0, fy = F / 2.0, cx = F / 2.0, cy = F / 2.0;
  const int face = orc_face_in_cubemap(cam, px, py);
  if (face == ORC_FACE_UNKNOWN) return face;
  const double z = 1.0;
  double i = px, j = py;
  i = i - static_cast<int>(i / F) * F;
  j = j - static_cast<int>(j / F) * F;
  V3f l;
  l.x = (float)((i - cx) * z / fx);
  l.y = (float)((j - cy) * z / fy);
  l.z = (float)z;
  V3f r = faces_to_rig<float>(l, face);
  const double n = std::sqrt((double)r.x * r.x + (double)r.y * r.y + (double)r.z * r.z);
  const double s = n > 0 ? 1. / n : 0.;
  ray3[0] = (float)(r.x * s); ray3[1] = (float)(r.y * s); ray3[2] = (float)(r.z * s);
  return face;
}

float orc_cos_fov_th(const orc_camera* cam) {
  const float fov = (float)cam->fov_deg;  // CamModelGeneral.h:224-229
  const float M_PIf_ = 3.1415926535897932384626f;
  return cosf(fov / 2 * (M_PIf_ / 180));
}

void orc_build_lut(const orc_camera* cam, float* map1, float* map2) {
  const int W = 3 * cam->face;  // System.cpp:301-324
  for (int y = 0; y < W; ++y)
    for (int x = 0; x < W; ++x) {
      map1[(size_t)y * W + x] = 0.f;
      map2[(size_t)y * W + x] = 0.f;
      double u, v;
      orc_cubemap_to_fisheye(cam, (double)x, (double)y, &u, &v);
      if (u < 0 || v < 0 || u >= cam->Iw || v >= cam->Ih) continue;
      map1[(size_t)y * W + x] = (float)u;
      map2[(size_t)y * W + x] = (float)v;
    }
}

}  // extern "C"
