// oracle/orc_ba.cpp -- CPU oracle (TEST INFRASTRUCTURE ONLY, parity unpinned: see orc_api.h).
// FP64 restatement of the slice of g2o that Optimizer::LocalBundleAdjustment (src/Optimizer.cpp:192-451)
// exercises, with the cubemap multi-pinhole edge:
//   residual      include/g2o_cubemap_vertices_edges.h:100-112, src/g2o_cubemap_vertices_edges.cpp:225-233,
//                 src/CamModelGeneral.cpp:228-263 (float round trip of the camera-frame point)
//   Jacobians     src/g2o_cubemap_vertices_edges.cpp:164-223
//   quadratic form / Huber   ThirdParty/g2o/g2o/core/base_binary_edge.hpp:54-120, robust_kernel_impl.cpp:78-91,
//                 base_edge.h:58-61,96-102
//   block solver  ThirdParty/g2o/g2o/core/block_solver.hpp:353-486 (Schur), 501-560 (buildSystem), 563-604 (lambda)
//   Levenberg     ThirdParty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-189
//   optimize loop ThirdParty/g2o/g2o/core/sparse_optimizer.cpp:354-419 ; update 422-435
//   SE3           ThirdParty/g2o/g2o/types/se3quat.h:104-121 (product), 217-257 (map, exp), 280-285 (normalise)
// g2o needs Eigen3 (absent from this image) so the reference itself cannot be built: see orc_api.h.
// The reduced pose system is solved with a dense LDL^T instead of Eigen::SimplicialLDLT (same solution up to
// rounding; tolerance for this path is 1e-4 relative, BASELINE.json north_star).
#include "orc_api.h"
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {
struct Pose { double t[3]; double q[4]; };  // q = (x, y, z, w)

inline void quat_to_R(const double* q, double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
               tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
inline void R_to_quat(const double m[9], double* q) {  // Eigen::Quaterniond(Matrix3d)
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
    q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
    q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
  }
}
inline void normalize_rot(double* q) {
  if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] /= n;
}
inline void map_point(const Pose& T, const double* X, double* out) {
  double R[9];
  quat_to_R(T.q, R);
  for (int i = 0; i < 3; ++i) out[i] = R[3 * i] * X[0] + R[3 * i + 1] * X[1] + R[3 * i + 2] * X[2] + T.t[i];
}
inline void se3_exp_mul(const double* u, Pose& T) {  // T <- exp(u) * T   (types_six_dof_expmap.h:73-76)
  const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
  const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  const double Om[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double Om2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += Om[3 * i + k] * Om[3 * k + j];
      Om2[3 * i + j] = s;
    }
  double R[9], V[9];
  if (theta < 0.00001) {
    for (int i = 0; i < 9; ++i) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + Om[i] + Om2[i]; V[i] = R[i]; }
  } else {
    const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta),
                 c = (theta - std::sin(theta)) / (std::pow(theta, 3));
    for (int i = 0; i < 9; ++i) {
      const double I = (i % 4 == 0 ? 1.0 : 0.0);
      R[i] = I + a * Om[i] + b * Om2[i];
      V[i] = I + b * Om[i] + c * Om2[i];
    }
  }
  Pose E;
  R_to_quat(R, E.q);
  for (int i = 0; i < 3; ++i) E.t[i] = V[3 * i] * up[0] + V[3 * i + 1] * up[1] + V[3 * i + 2] * up[2];
  normalize_rot(E.q);
  // result = E * T
  double RE[9];
  quat_to_R(E.q, RE);
  Pose Rs;
  for (int i = 0; i < 3; ++i)
    Rs.t[i] = E.t[i] + RE[3 * i] * T.t[0] + RE[3 * i + 1] * T.t[1] + RE[3 * i + 2] * T.t[2];
  const double* a = E.q; const double* b = T.q;
  Rs.q[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  Rs.q[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  Rs.q[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  Rs.q[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  normalize_rot(Rs.q);
  T = Rs;
}

const double kRface[5][9] = {  // g2o_cubemap_vertices_edges.cpp:173-205, indexed by face id
    {1, 0, 0, 0, 1, 0, 0, 0, 1},    // FRONT
    {0, 0, 1, 0, 1, 0, -1, 0, 0},   // LEFT
    {0, 0, -1, 0, 1, 0, 1, 0, 0},   // RIGHT
    {1, 0, 0, 0, 0, 1, 0, -1, 0},   // UPPER
    {1, 0, 0, 0, 0, -1, 0, 1, 0}};  // LOWER

struct Problem {
  int K, P, E;
  std::vector<Pose> poses;
  std::vector<double> pts;
  const uint8_t* fixed;
  const int *e_pose, *e_point;
  const double *e_obs, *e_invsig2;
  const int8_t* e_face;
  double fx, fy, cx, cy;
  std::vector<double> err;      // persistent _error per edge (only refreshed while the edge is active)
  std::vector<uint8_t> level;   // g2o edge level (0 active / 1 excluded)
  bool robust;
  double delta;
};

inline void edge_error(const Problem& pb, int e, double* r) {
  double Xc[3];
  map_point(pb.poses[pb.e_pose[e]], &pb.pts[3 * pb.e_point[e]], Xc);
  const float xf = (float)Xc[0], yf = (float)Xc[1], zf = (float)Xc[2];  // cv::Vec3f rigPt (cpp:227)
  const double* Rf = kRface[pb.e_face[e]];
  const float lx = (float)(Rf[0] * xf + Rf[1] * yf + Rf[2] * zf), ly = (float)(Rf[3] * xf + Rf[4] * yf + Rf[5] * zf),
              lz = (float)(Rf[6] * xf + Rf[7] * yf + Rf[8] * zf);  // exact: signed permutation
  const float u = (float)((double)lx * pb.fx / (double)lz + pb.cx);
  const float v = (float)((double)ly * pb.fy / (double)lz + pb.cy);
  r[0] = pb.e_obs[2 * e] - (double)u;
  r[1] = pb.e_obs[2 * e + 1] - (double)v;
}
inline void edge_jacobians(const Problem& pb, int e, double* Jp /*2x6*/, double* Jl /*2x3*/) {
  const Pose& T = pb.poses[pb.e_pose[e]];
  double Xc[3];
  map_point(T, &pb.pts[3 * pb.e_point[e]], Xc);
  const double* Rf = kRface[pb.e_face[e]];
  double l[3];
  for (int i = 0; i < 3; ++i) l[i] = Rf[3 * i] * Xc[0] + Rf[3 * i + 1] * Xc[1] + Rf[3 * i + 2] * Xc[2];
  const double G[6] = {pb.fx / l[2], 0, -pb.fx * l[0] / (l[2] * l[2]), 0, pb.fy / l[2], -pb.fy * l[1] / (l[2] * l[2])};
  double M[6];  // -G * Rface
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j)
      M[3 * i + j] = -1.0 * (G[3 * i] * Rf[j] + G[3 * i + 1] * Rf[3 + j] + G[3 * i + 2] * Rf[6 + j]);
  const double S[9] = {0, Xc[2], -Xc[1], -Xc[2], 0, Xc[0], Xc[1], -Xc[0], 0};  // -[Xc]x
  double R[9];
  quat_to_R(T.q, R);
  for (int i = 0; i < 2; ++i) {
    for (int j = 0; j < 3; ++j) {
      Jp[6 * i + j] = M[3 * i] * S[j] + M[3 * i + 1] * S[3 + j] + M[3 * i + 2] * S[6 + j];
      Jp[6 * i + 3 + j] = M[3 * i + j];
      Jl[3 * i + j] = M[3 * i] * R[j] + M[3 * i + 1] * R[3 + j] + M[3 * i + 2] * R[6 + j];
    }
  }
}
inline void huber(double e, double delta, double* rho) {
  const double dsqr = delta * delta;
  if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
  else { const double s = std::sqrt(e); rho[0] = 2 * s * delta - dsqr; rho[1] = delta / s; rho[2] = -0.5 * rho[1] / e; }
}
inline bool inv3(const double* A, double* Ai) {
  const double a = A[0], b = A[1], c = A[2], d = A[3], e = A[4], f = A[5], g = A[6], h = A[7], i = A[8];
  const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  const double id = 1.0 / det;
  Ai[0] = (e * i - f * h) * id; Ai[1] = (c * h - b * i) * id; Ai[2] = (b * f - c * e) * id;
  Ai[3] = (f * g - d * i) * id; Ai[4] = (a * i - c * g) * id; Ai[5] = (c * d - a * f) * id;
  Ai[6] = (d * h - e * g) * id; Ai[7] = (b * g - a * h) * id; Ai[8] = (a * e - b * d) * id;
  return std::isfinite(id);
}
// dense LDL^T, in place on the full symmetric matrix; returns false on a zero / non-finite pivot
bool ldlt_solve(std::vector<double>& A, int n, std::vector<double>& b) {
  std::vector<double> D(n);
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * D[k];
    if (!(std::isfinite(d)) || d == 0.0) return false;
    D[j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * D[k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; ++i) for (int k = 0; k < i; ++k) b[i] -= A[(size_t)i * n + k] * b[k];
  for (int i = 0; i < n; ++i) b[i] /= D[i];
  for (int i = n - 1; i >= 0; --i) for (int k = i + 1; k < n; ++k) b[i] -= A[(size_t)k * n + i] * b[k];
  return true;
}

struct System {
  std::vector<int> pose_slot, point_slot;  // index in the reduced ordering or -1
  int np = 0, nl = 0;
  std::vector<int> act;                    // active edge ids
  std::vector<double> Hpp, bp, Hll, bl, Hpl;  // Hpp np x 36, Hll nl x 9, Hpl per active edge 18 (6x3)
  std::vector<double> x;                   // 6np + 3nl
};

void compute_active_errors(Problem& pb, const System& s) {
  for (int e : s.act) edge_error(pb, e, &pb.err[2 * e]);
}
double active_robust_chi2(const Problem& pb, const System& s) {
  double chi = 0;
  for (int e : s.act) {
    const double c2 = pb.e_invsig2[e] * (pb.err[2 * e] * pb.err[2 * e] + pb.err[2 * e + 1] * pb.err[2 * e + 1]);
    if (pb.robust) { double rho[3]; huber(c2, pb.delta, rho); chi += rho[0]; }
    else chi += c2;
  }
  return chi;
}
void build_system(const Problem& pb, System& s) {
  std::fill(s.Hpp.begin(), s.Hpp.end(), 0.0); std::fill(s.bp.begin(), s.bp.end(), 0.0);
  std::fill(s.Hll.begin(), s.Hll.end(), 0.0); std::fill(s.bl.begin(), s.bl.end(), 0.0);
  std::fill(s.Hpl.begin(), s.Hpl.end(), 0.0);
  for (size_t a = 0; a < s.act.size(); ++a) {
    const int e = s.act[a];
    double Jp[12], Jl[6];
    edge_jacobians(pb, e, Jp, Jl);
    const double* r = &pb.err[2 * e];
    const double om = pb.e_invsig2[e];
    double w = 1.0;
    if (pb.robust) { double rho[3]; huber(om * (r[0] * r[0] + r[1] * r[1]), pb.delta, rho); w = rho[1]; }
    const double ow = w * om;
    const double omr[2] = {-om * r[0] * w, -om * r[1] * w};
    const int ps = s.pose_slot[pb.e_pose[e]], ls = s.point_slot[pb.e_point[e]];
    if (ls >= 0) {
      for (int i = 0; i < 3; ++i) {
        s.bl[3 * ls + i] += Jl[i] * omr[0] + Jl[3 + i] * omr[1];
        for (int j = 0; j < 3; ++j) s.Hll[9 * ls + 3 * i + j] += ow * (Jl[i] * Jl[j] + Jl[3 + i] * Jl[3 + j]);
      }
    }
    if (ps >= 0) {
      for (int i = 0; i < 6; ++i) {
        s.bp[6 * ps + i] += Jp[i] * omr[0] + Jp[6 + i] * omr[1];
        for (int j = 0; j < 6; ++j) s.Hpp[36 * ps + 6 * i + j] += ow * (Jp[i] * Jp[j] + Jp[6 + i] * Jp[6 + j]);
      }
      if (ls >= 0)
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 3; ++j) s.Hpl[18 * a + 3 * i + j] = ow * (Jp[i] * Jl[j] + Jp[6 + i] * Jl[3 + j]);
    }
  }
}
bool schur_solve(const Problem& pb, System& s, double lambda) {
  const int n = 6 * s.np;
  std::vector<double> H((size_t)n * n, 0.0), b(n, 0.0);
  for (int p = 0; p < s.np; ++p)
    for (int i = 0; i < 6; ++i) {
      b[6 * p + i] = s.bp[6 * p + i];
      for (int j = 0; j < 6; ++j) H[(size_t)(6 * p + i) * n + 6 * p + j] = s.Hpp[36 * p + 6 * i + j] + (i == j ? lambda : 0.0);
    }
  // edges grouped per point
  std::vector<std::vector<int>> per_point(s.nl);
  for (size_t a = 0; a < s.act.size(); ++a) {
    const int e = s.act[a];
    if (s.pose_slot[pb.e_pose[e]] >= 0 && s.point_slot[pb.e_point[e]] >= 0) per_point[s.point_slot[pb.e_point[e]]].push_back((int)a);
  }
  std::vector<double> Dinv((size_t)9 * s.nl);
  for (int l = 0; l < s.nl; ++l) {
    double D[9];
    for (int i = 0; i < 9; ++i) D[i] = s.Hll[9 * l + i] + (i % 4 == 0 ? lambda : 0.0);
    inv3(D, &Dinv[9 * l]);
    const double* Di = &Dinv[9 * l];
    double db[3];
    for (int i = 0; i < 3; ++i) db[i] = Di[3 * i] * s.bl[3 * l] + Di[3 * i + 1] * s.bl[3 * l + 1] + Di[3 * i + 2] * s.bl[3 * l + 2];
    for (int a1 : per_point[l]) {
      const int p1 = s.pose_slot[pb.e_pose[s.act[a1]]];
      const double* B1 = &s.Hpl[18 * a1];
      double BD[18];
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j) BD[3 * i + j] = B1[3 * i] * Di[j] + B1[3 * i + 1] * Di[3 + j] + B1[3 * i + 2] * Di[6 + j];
      for (int i = 0; i < 6; ++i) b[6 * p1 + i] -= B1[3 * i] * db[0] + B1[3 * i + 1] * db[1] + B1[3 * i + 2] * db[2];
      for (int a2 : per_point[l]) {
        const int p2 = s.pose_slot[pb.e_pose[s.act[a2]]];
        const double* B2 = &s.Hpl[18 * a2];
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 6; ++j)
            H[(size_t)(6 * p1 + i) * n + 6 * p2 + j] -= BD[3 * i] * B2[3 * j] + BD[3 * i + 1] * B2[3 * j + 1] + BD[3 * i + 2] * B2[3 * j + 2];
      }
    }
  }
  std::fill(s.x.begin(), s.x.end(), 0.0);
  if (n > 0) {
    if (!ldlt_solve(H, n, b)) return false;
    for (int i = 0; i < n; ++i) s.x[i] = b[i];
  }
  for (int l = 0; l < s.nl; ++l) {
    double cl[3] = {s.bl[3 * l], s.bl[3 * l + 1], s.bl[3 * l + 2]};
    for (int a1 : per_point[l]) {
      const int p1 = s.pose_slot[pb.e_pose[s.act[a1]]];
      const double* B1 = &s.Hpl[18 * a1];
      for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 6; ++i) cl[j] -= B1[3 * i + j] * s.x[6 * p1 + i];
    }
    const double* Di = &Dinv[9 * l];
    for (int i = 0; i < 3; ++i) s.x[n + 3 * l + i] = Di[3 * i] * cl[0] + Di[3 * i + 1] * cl[1] + Di[3 * i + 2] * cl[2];
  }
  return true;
}

// Test hook for the stop flag (Optimizer.cpp:256-257 hands g2o a bool another thread sets; g2o polls it in terminate(), i.e. at every
// trial boundary -- optimization_algorithm_levenberg.cpp:127 -- and at every iteration -- sparse_optimizer.cpp:376): a trial budget makes
// "the flag was raised while trial number n was running" reproducible.  -1 = no budget.
static thread_local int g_trial_budget = -1, g_trials = 0;
static bool budget_spent() { return g_trial_budget >= 0 && g_trials >= g_trial_budget; }

// SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg + BlockSolver_6_3 (Schur)
int optimize(Problem& pb, int iterations, const volatile uint8_t* stop, double* chi_ini, double* chi_fin, double* lam_fin) {
  System s;
  s.pose_slot.assign(pb.K, -1); s.point_slot.assign(pb.P, -1);
  std::vector<uint8_t> pose_act(pb.K, 0), point_act(pb.P, 0);
  for (int e = 0; e < pb.E; ++e)
    if (pb.level[e] == 0) { s.act.push_back(e); pose_act[pb.e_pose[e]] = 1; point_act[pb.e_point[e]] = 1; }
  for (int k = 0; k < pb.K; ++k) if (pose_act[k] && !pb.fixed[k]) s.pose_slot[k] = s.np++;
  for (int p = 0; p < pb.P; ++p) if (point_act[p]) s.point_slot[p] = s.nl++;
  *chi_ini = *chi_fin = 0; *lam_fin = 0;
  if (s.np + s.nl == 0) return -1;
  s.Hpp.resize((size_t)36 * s.np); s.bp.resize((size_t)6 * s.np); s.Hll.resize((size_t)9 * s.nl); s.bl.resize((size_t)3 * s.nl);
  s.Hpl.resize((size_t)18 * s.act.size()); s.x.resize((size_t)6 * s.np + 3 * s.nl);
  double lambda = -1, ni = 2;
  int nBad = 0, done = 0;
  auto stopped = [&]() { return (stop && *stop) || budget_spent(); };
  for (int it = 0; it < iterations && !stopped(); ++it) {
    compute_active_errors(pb, s);
    double currentChi = active_robust_chi2(pb, s);
    double tempChi = currentChi;
    const double iniChi = currentChi;
    if (it == 0) *chi_ini = iniChi;
    build_system(pb, s);
    if (it == 0) {
      double maxDiag = 0;
      for (int p = 0; p < s.np; ++p) for (int j = 0; j < 6; ++j) maxDiag = std::max(std::fabs(s.Hpp[36 * p + 7 * j]), maxDiag);
      for (int l = 0; l < s.nl; ++l) for (int j = 0; j < 3; ++j) maxDiag = std::max(std::fabs(s.Hll[9 * l + 4 * j]), maxDiag);
      lambda = 1e-5 * maxDiag; ni = 2; nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      const std::vector<Pose> poses_bak = pb.poses;
      const std::vector<double> pts_bak = pb.pts;
      const bool ok2 = schur_solve(pb, s, lambda);
      for (int k = 0; k < pb.K; ++k) if (s.pose_slot[k] >= 0) se3_exp_mul(&s.x[6 * s.pose_slot[k]], pb.poses[k]);
      for (int p = 0; p < pb.P; ++p)
        if (s.point_slot[p] >= 0) for (int i = 0; i < 3; ++i) pb.pts[3 * p + i] += s.x[6 * s.np + 3 * s.point_slot[p] + i];
      compute_active_errors(pb, s);
      tempChi = active_robust_chi2(pb, s);
      if (!ok2) tempChi = DBL_MAX;
      rho = (currentChi - tempChi);
      double scale = 0;
      for (int p = 0; p < s.np; ++p) for (int j = 0; j < 6; ++j) scale += s.x[6 * p + j] * (lambda * s.x[6 * p + j] + s.bp[6 * p + j]);
      for (int l = 0; l < s.nl; ++l)
        for (int j = 0; j < 3; ++j) scale += s.x[6 * s.np + 3 * l + j] * (lambda * s.x[6 * s.np + 3 * l + j] + s.bl[3 * l + j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        const double scaleFactor = std::max(1. / 3., alpha);
        lambda *= scaleFactor; ni = 2; currentChi = tempChi;
      } else {
        lambda *= ni; ni *= 2;
        pb.poses = poses_bak; pb.pts = pts_bak;  // pop(); edge errors stay those of the rejected trial, as in g2o
      }
      ++qmax;
      ++g_trials;
    } while (rho < 0 && qmax < 10 && !stopped());
    ++done;
    *chi_fin = currentChi; *lam_fin = lambda;
    if (qmax == 10 || rho == 0) break;  // Terminate
    if ((iniChi - currentChi) * 1e3 < iniChi) ++nBad; else nBad = 0;
    if (nBad >= 3) break;
  }
  return done;
}

void load(Problem& pb, int K, const double* poses, const uint8_t* fixed, int P, const double* points, int E,
          const int* e_pose, const int* e_point, const double* e_obs, const double* e_invsig2, const int8_t* e_face,
          double fx, double fy, double cx, double cy) {
  pb.K = K; pb.P = P; pb.E = E; pb.fixed = fixed; pb.e_pose = e_pose; pb.e_point = e_point; pb.e_obs = e_obs;
  pb.e_invsig2 = e_invsig2; pb.e_face = e_face; pb.fx = fx; pb.fy = fy; pb.cx = cx; pb.cy = cy;
  pb.poses.resize(K);
  for (int k = 0; k < K; ++k) {
    for (int i = 0; i < 3; ++i) pb.poses[k].t[i] = poses[7 * k + i];
    for (int i = 0; i < 4; ++i) pb.poses[k].q[i] = poses[7 * k + 3 + i];
    normalize_rot(pb.poses[k].q);  // SE3Quat ctor (se3quat.h:58-64)
  }
  pb.pts.assign(points, points + 3 * (size_t)P);
  pb.err.assign(2 * (size_t)E, 0.0);
  pb.level.assign(E, 0);
}
}  // namespace

extern "C" {

void orc_se3_exp_apply(const double* upd6, double* pose7) {
  Pose T;
  for (int i = 0; i < 3; ++i) T.t[i] = pose7[i];
  for (int i = 0; i < 4; ++i) T.q[i] = pose7[3 + i];
  se3_exp_mul(upd6, T);
  for (int i = 0; i < 3; ++i) pose7[i] = T.t[i];
  for (int i = 0; i < 4; ++i) pose7[3 + i] = T.q[i];
}

int orc_ba_run(int K, double* poses, const uint8_t* fixed, int P, double* points, int E, const int* e_pose,
               const int* e_point, const double* e_obs, const double* e_invsig2, const int8_t* e_face, double fx,
               double fy, double cx, double cy, int its_robust, int its_final, const volatile uint8_t* stop,
               uint8_t* outlier_flags, orc_ba_stats* st) {
  orc_ba_stats dummy;
  if (!st) st = &dummy;
  memset(st, 0, sizeof(*st));
  if (outlier_flags) memset(outlier_flags, 0, E);
  if (stop && *stop) return 1;  // Optimizer.cpp:359-361
  Problem pb;
  load(pb, K, poses, fixed, P, points, E, e_pose, e_point, e_obs, e_invsig2, e_face, fx, fy, cx, cy);
  pb.robust = true; pb.delta = std::sqrt(5.991);  // thHuberMono (Optimizer.cpp:303)
  st->iterations_done[0] = optimize(pb, its_robust, stop, &st->chi2_initial[0], &st->chi2_final[0], &st->lambda_final[0]);
  auto is_outlier = [&](int e) {
    const double c2 = pb.e_invsig2[e] * (pb.err[2 * e] * pb.err[2 * e] + pb.err[2 * e + 1] * pb.err[2 * e + 1]);
    double Xc[3];
    map_point(pb.poses[pb.e_pose[e]], &pb.pts[3 * pb.e_point[e]], Xc);
    return c2 > 5.991 || !(Xc[2] > 0.0);
  };
  if (!(stop && *stop) && !budget_spent()) {  // Optimizer.cpp:366-397
    for (int e = 0; e < E; ++e) if (is_outlier(e)) { pb.level[e] = 1; ++st->n_outliers_mid; }
    pb.robust = false;
    st->iterations_done[1] = optimize(pb, its_final, stop, &st->chi2_initial[1], &st->chi2_final[1], &st->lambda_final[1]);
  }
  for (int e = 0; e < E; ++e)
    if (is_outlier(e)) { if (outlier_flags) outlier_flags[e] = 1; ++st->n_outliers_final; }
  for (int k = 0; k < K; ++k) {
    for (int i = 0; i < 3; ++i) poses[7 * k + i] = pb.poses[k].t[i];
    for (int i = 0; i < 4; ++i) poses[7 * k + 3 + i] = pb.poses[k].q[i];
  }
  memcpy(points, pb.pts.data(), sizeof(double) * 3 * (size_t)P);
  return 0;
}

// orc_ba_run with the stop flag raised by "another thread" while trial number stop_after_trials (counted over both stages, from 1) was
// running: that trial completes, then every terminate() poll sees the flag.
int orc_ba_run_stop_after(int K, double* poses, const uint8_t* fixed, int P, double* points, int E, const int* e_pose,
                          const int* e_point, const double* e_obs, const double* e_invsig2, const int8_t* e_face, double fx,
                          double fy, double cx, double cy, int its_robust, int its_final, int stop_after_trials,
                          uint8_t* outlier_flags, orc_ba_stats* st) {
  g_trial_budget = stop_after_trials; g_trials = 0;
  const int rc = orc_ba_run(K, poses, fixed, P, points, E, e_pose, e_point, e_obs, e_invsig2, e_face, fx, fy, cx, cy, its_robust, its_final,
                            nullptr, outlier_flags, st);
  g_trial_budget = -1; g_trials = 0;
  return rc;
}

// ---- Optimizer::PoseOptimization (src/Optimizer.cpp:48-190): one free SE3 vertex, N unary multi-pinhole edges
// (EdgeSE3ProjectXYZMultiPinholeOnlyPose: include/g2o_cubemap_vertices_edges.h:42-88, src/g2o_cubemap_vertices_edges.cpp:61-134;
// quadratic form ThirdParty/g2o/g2o/core/base_unary_edge.hpp:43-72), BlockSolver_6_3 without Schur (no marginalised vertex:
// block_solver.hpp:354-365) + LinearSolverDense (solvers/linear_solver_dense.h:64-118), Levenberg as in optimize() above.
// Four rounds of optimize(10) from the SAME initial pose; after each round every edge is re-classified with chi2 > 5.991
// (Optimizer.cpp:138-176), edges that were outliers get their error recomputed first (:147-150), active ones keep the
// error of the last computeActiveErrors (the rejected trial's if the last trial was rejected); the Huber kernel is dropped
// after the third round (:171-172); the loop stops early when the graph holds fewer than 10 edges (:175-176).
int orc_pose_optimize(int n, const double* Xw, const double* obs_uv, const double* inv_sigma2, const int8_t* face, double fx,
                      double fy, double cx, double cy, double* pose7, uint8_t* outlier, orc_pose_stats* st) {
  orc_pose_stats dummy;
  if (!st) st = &dummy;
  memset(st, 0, sizeof(*st));
  if (outlier) memset(outlier, 0, n);
  if (n < 3) return 0;                                   // Optimizer.cpp:131-132
  // reuse the edge arithmetic of the binary edge: one pose, n "points" that never move
  Problem pb;
  std::vector<int> e_pose(n, 0), e_point(n);
  for (int i = 0; i < n; ++i) e_point[i] = i;
  const uint8_t fixed0 = 0;
  load(pb, 1, pose7, &fixed0, n, Xw, n, e_pose.data(), e_point.data(), obs_uv, inv_sigma2, face, fx, fy, cx, cy);
  pb.delta = std::sqrt(5.991);
  const Pose T0 = pb.poses[0];
  std::vector<uint8_t> isout(n, 0);
  int nBad = 0;
  for (int round = 0; round < 4; ++round) {
    pb.robust = round < 3;
    pb.poses[0] = T0;                                    // vSE3->setEstimate(toSE3Quat(pFrame->mTcw)) every round
    std::vector<int> act;
    for (int e = 0; e < n; ++e) if (pb.level[e] == 0) act.push_back(e);
    int done = 0;
    double chi_fin = 0;
    if (!act.empty()) {                                  // no active edge -> no active vertex -> optimize() returns at once
      double H[36], b[6], lambda = -1, ni = 2;
      int nBadIt = 0;
      for (int it = 0; it < 10; ++it) {
        double currentChi = 0;
        for (int e : act) edge_error(pb, e, &pb.err[2 * e]);
        auto chi_of = [&]() {
          double c = 0;
          for (int e : act) {
            const double c2 = pb.e_invsig2[e] * (pb.err[2 * e] * pb.err[2 * e] + pb.err[2 * e + 1] * pb.err[2 * e + 1]);
            if (pb.robust) { double rho[3]; huber(c2, pb.delta, rho); c += rho[0]; } else c += c2;
          }
          return c;
        };
        currentChi = chi_of();
        double tempChi = currentChi;
        const double iniChi = currentChi;
        for (int i = 0; i < 36; ++i) H[i] = 0;
        for (int i = 0; i < 6; ++i) b[i] = 0;
        for (int e : act) {
          double Jp[12], Jl[6];
          edge_jacobians(pb, e, Jp, Jl);
          const double* r = &pb.err[2 * e];
          const double om = pb.e_invsig2[e];
          double w = 1.0;
          if (pb.robust) { double rho[3]; huber(om * (r[0] * r[0] + r[1] * r[1]), pb.delta, rho); w = rho[1]; }
          for (int i = 0; i < 6; ++i) {
            b[i] -= w * om * (Jp[i] * r[0] + Jp[6 + i] * r[1]);
            for (int j = 0; j < 6; ++j) H[6 * i + j] += w * om * (Jp[i] * Jp[j] + Jp[6 + i] * Jp[6 + j]);
          }
        }
        if (it == 0) {
          double maxDiag = 0;
          for (int j = 0; j < 6; ++j) maxDiag = std::max(std::fabs(H[7 * j]), maxDiag);
          lambda = 1e-5 * maxDiag; ni = 2; nBadIt = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
          const Pose bak = pb.poses[0];
          std::vector<double> A(H, H + 36), x(b, b + 6);
          for (int j = 0; j < 6; ++j) A[7 * j] += lambda;
          const bool ok2 = ldlt_solve(A, 6, x);
          if (!ok2) std::fill(x.begin(), x.end(), 0.0);
          se3_exp_mul(x.data(), pb.poses[0]);
          for (int e : act) edge_error(pb, e, &pb.err[2 * e]);
          tempChi = chi_of();
          if (!ok2) tempChi = DBL_MAX;
          rho = currentChi - tempChi;
          double scale = 0;
          for (int j = 0; j < 6; ++j) scale += x[j] * (lambda * x[j] + b[j]);
          scale += 1e-3;
          rho /= scale;
          if (rho > 0 && std::isfinite(tempChi)) {
            double alpha = 1. - std::pow((2 * rho - 1), 3);
            alpha = std::min(alpha, 2. / 3.);
            lambda *= std::max(1. / 3., alpha); ni = 2; currentChi = tempChi;
          } else {
            lambda *= ni; ni *= 2;
            pb.poses[0] = bak;
          }
          ++qmax;
        } while (rho < 0 && qmax < 10);
        ++done;
        chi_fin = currentChi;
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) ++nBadIt; else nBadIt = 0;
        if (nBadIt >= 3) break;
      }
    }
    st->iterations_done[round] = done; st->chi2_final[round] = chi_fin;
    nBad = 0;
    for (int e = 0; e < n; ++e) {
      if (isout[e]) edge_error(pb, e, &pb.err[2 * e]);
      const float chi2 = (float)(pb.e_invsig2[e] * (pb.err[2 * e] * pb.err[2 * e] + pb.err[2 * e + 1] * pb.err[2 * e + 1]));  // const float chi2 = e->chi2()
      if (chi2 > 5.991f) { isout[e] = 1; pb.level[e] = 1; ++nBad; } else { isout[e] = 0; pb.level[e] = 0; }
    }
    st->rounds = round + 1;
    if (n < 10) break;
  }
  for (int i = 0; i < 3; ++i) pose7[i] = pb.poses[0].t[i];
  for (int i = 0; i < 4; ++i) pose7[3 + i] = pb.poses[0].q[i];
  if (outlier) memcpy(outlier, isout.data(), n);
  st->n_bad = nBad;
  return n - nBad;
}

void orc_ba_linearize(int K, const double* poses, const uint8_t* fixed, int P, const double* points, int E,
                      const int* e_pose, const int* e_point, const double* e_obs, const double* e_invsig2,
                      const int8_t* e_face, double fx, double fy, double cx, double cy, int robust, double huber_delta,
                      double* err, double* chi2, double* Jpose, double* Jpoint, double* Hpp, double* bp, double* Hll,
                      double* bl, double* Hpl, double* robust_chi2_sum) {
  Problem pb;
  load(pb, K, poses, fixed, P, points, E, e_pose, e_point, e_obs, e_invsig2, e_face, fx, fy, cx, cy);
  pb.robust = robust != 0; pb.delta = huber_delta;
  System s;
  s.pose_slot.assign(K, -1); s.point_slot.assign(P, -1);
  for (int e = 0; e < E; ++e) s.act.push_back(e);
  for (int k = 0; k < K; ++k) s.pose_slot[k] = fixed[k] ? -1 : k;  // identity slots: outputs indexed by pose / point id
  for (int p = 0; p < P; ++p) s.point_slot[p] = p;
  s.np = K; s.nl = P;
  s.Hpp.assign((size_t)36 * K, 0); s.bp.assign((size_t)6 * K, 0); s.Hll.assign((size_t)9 * P, 0); s.bl.assign((size_t)3 * P, 0);
  s.Hpl.assign((size_t)18 * E, 0);
  compute_active_errors(pb, s);
  if (robust_chi2_sum) *robust_chi2_sum = active_robust_chi2(pb, s);
  build_system(pb, s);
  for (int e = 0; e < E; ++e) {
    if (err) { err[2 * e] = pb.err[2 * e]; err[2 * e + 1] = pb.err[2 * e + 1]; }
    if (chi2) chi2[e] = e_invsig2[e] * (pb.err[2 * e] * pb.err[2 * e] + pb.err[2 * e + 1] * pb.err[2 * e + 1]);
    if (Jpose || Jpoint) {
      double Jp[12], Jl[6];
      edge_jacobians(pb, e, Jp, Jl);
      if (Jpose) memcpy(Jpose + 12 * (size_t)e, Jp, sizeof(Jp));
      if (Jpoint) memcpy(Jpoint + 6 * (size_t)e, Jl, sizeof(Jl));
    }
  }
  if (Hpp) memcpy(Hpp, s.Hpp.data(), sizeof(double) * s.Hpp.size());
  if (bp) memcpy(bp, s.bp.data(), sizeof(double) * s.bp.size());
  if (Hll) memcpy(Hll, s.Hll.data(), sizeof(double) * s.Hll.size());
  if (bl) memcpy(bl, s.bl.data(), sizeof(double) * s.bl.size());
  if (Hpl) memcpy(Hpl, s.Hpl.data(), sizeof(double) * s.Hpl.size());
}

}  // extern "C"
