// cms_ba_kernels.hip -- FP64 local-bundle-adjustment kernels for gfx950 (Optimizer::LocalBundleAdjustment numeric core).
//
// Replaces, for the cubemap multi-pinhole edge (g2o_cubemap_vertices_edges.h:90-134, .cpp:164-233):
//   computeActiveErrors + activeRobustChi2   (sparse_optimizer.cpp:61-114)                 -> k_ba_errors
//   linearizeOplus + constructQuadraticForm  (base_binary_edge.hpp:54-120, Huber robust_kernel_impl.cpp:78-91)
//                                                                  -> k_ba_lin_points (Hll, bl, Hpl) + k_ba_lin_poses (Hpp, bp)
//   BlockSolver::setLambda / Schur complement / back substitution (block_solver.hpp:367-485, 563-589)
//                                            -> k_ba_dinv, k_ba_schur_init, k_ba_schur_chunks/_finish, k_ba_solve_r192 / k_ba_solve (windows beyond the fused trial of cms_ba_fused.hip), k_ba_backsub
//   vertex oplus (types_six_dof_expmap.h:73-76, types_sba.h:51-55, se3quat.h:217-257) -> k_ba_update_poses / k_ba_backsub
// Edges are stored sorted by point (CSR) so Hll / bl need no atomics; per-pose blocks are reduced by workgroups over
// slices of that pose's edge list; every 6x6 block of the reduced (Schur) system is owned by one workgroup that sums
// over the host-built co-visibility tuple list of its pose pair (deterministic, no atomics); the reduced system is
// factorised by one workgroup with the trailing matrix in registers.  All state stays on the device across Levenberg-Marquardt trials; the host only reads three
// scalars per trial (chi2, gain denominator, solver status).
#include <hip/hip_runtime.h>
#include <stdint.h>

struct BaDev {
  int K, P, E, np;                 // np = number of free poses (reduced system is 6np x 6np)
  const uint8_t* fixed;
  const int* pose_slot;            // K: slot among free poses or -1
  const int* e_pose; const int* e_point;   // E (edges sorted by point)
  const double* e_obs; const double* e_inv; const int8_t* e_face;
  const int* pt_off;               // P+1 CSR over the sorted edges
  const int* pose_off; const int* pose_edges;  // K+1 / E : edge ids per pose
  uint8_t* level;                  // E: 0 active, 1 excluded
  double* err;                     // E x 2 (persistent, refreshed only for active edges)
  double* ow;                      // E: robust weight x information of the last linearisation (0 for excluded edges): with it the 6x3
                                   // pose-point block of an edge can be rebuilt from the estimate instead of being stored (144 B / edge)
  double fx, fy, cx, cy;
};
// The same view with every pointer typed as a GLOBAL-memory pointer (address space 1).  The batched kernels (kb_ba_*) read a window's pointers
// from a BaItem in memory; for such pointers the compiler cannot know the address space and emits flat_load / flat_store -- and a flat
// operation counts in BOTH wait counters (vmcnt and lgkmcnt): every wait for LDS data then also waits for the outstanding prefetches from
// HBM, i.e. nothing a wavefront requests ahead overlaps its LDS work (round 3's run-major Schur kernel spent a quarter of its life in such
// waits).  All device bodies take this view; the plain BaDev (what the host fills in) converts implicitly.
#define BA_AS1 __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ BA_AS1 T* ba_g(T* p) { return (BA_AS1 T*)p; }
struct BaDevG {
  int K, P, E, np;
  const BA_AS1 uint8_t* fixed;
  const BA_AS1 int* pose_slot;
  const BA_AS1 int* e_pose; const BA_AS1 int* e_point;
  const BA_AS1 double* e_obs; const BA_AS1 double* e_inv; const BA_AS1 int8_t* e_face;
  const BA_AS1 int* pt_off;
  const BA_AS1 int* pose_off; const BA_AS1 int* pose_edges;
  BA_AS1 uint8_t* level;
  BA_AS1 double* err;
  BA_AS1 double* ow;
  double fx, fy, cx, cy;
  __device__ __forceinline__ BaDevG() {}
  __device__ __forceinline__ BaDevG(const BaDev& d)
      : K(d.K), P(d.P), E(d.E), np(d.np), fixed(ba_g(d.fixed)), pose_slot(ba_g(d.pose_slot)), e_pose(ba_g(d.e_pose)), e_point(ba_g(d.e_point)),
        e_obs(ba_g(d.e_obs)), e_inv(ba_g(d.e_inv)), e_face(ba_g(d.e_face)), pt_off(ba_g(d.pt_off)), pose_off(ba_g(d.pose_off)),
        pose_edges(ba_g(d.pose_edges)), level(ba_g(d.level)), err(ba_g(d.err)), ow(ba_g(d.ow)), fx(d.fx), fy(d.fy), cx(d.cx), cy(d.cy) {}
};
// 16-byte accesses through global-memory pointers (the HIP vector classes only bind to generic references: native vectors carry the address space)
typedef double ba_nv2d __attribute__((ext_vector_type(2)));
typedef int ba_nv4i __attribute__((ext_vector_type(4)));
typedef unsigned int ba_nv2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ba_ld2(const BA_AS1 double* p) { const ba_nv2d v = *reinterpret_cast<const BA_AS1 ba_nv2d*>(p); return make_double2(v.x, v.y); }
__device__ __forceinline__ void ba_st2(BA_AS1 double* p, double a, double b) { ba_nv2d v; v.x = a; v.y = b; *reinterpret_cast<BA_AS1 ba_nv2d*>(p) = v; }
__device__ __forceinline__ int4 ba_ld4i(const BA_AS1 int4* p) { const ba_nv4i v = *reinterpret_cast<const BA_AS1 ba_nv4i*>(p); return make_int4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 ba_ld2u(const BA_AS1 uint2* p) { const ba_nv2u v = *reinterpret_cast<const BA_AS1 ba_nv2u*>(p); return make_uint2(v.x, v.y); }
// FP64 addition to global memory without a return value (global_atomic_add_f64)
__device__ __forceinline__ void ba_gadd(BA_AS1 double* p, double v) { (void)__builtin_amdgcn_global_atomic_fadd_f64(p, v); }
// an edge's measurement / stored residual as one 16-byte access
#define BA_OBS2(d, e) ba_ld2((d).e_obs + 2 * (size_t)(e))
#define BA_ERR2_LD(d, e) ba_ld2((d).err + 2 * (size_t)(e))
#define BA_ERR2_ST(d, e, a, b) ba_st2((d).err + 2 * (size_t)(e), a, b)

__device__ __forceinline__ void quat_to_R(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
               tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void face_local(int face, const double* X, double* l) {  // cvtRigToFaces (CamModelGeneral.h:417-443)
  switch (face) {
    case 0: l[0] = X[0]; l[1] = X[1]; l[2] = X[2]; break;
    case 1: l[0] = X[2]; l[1] = X[1]; l[2] = -X[0]; break;
    case 2: l[0] = -X[2]; l[1] = X[1]; l[2] = X[0]; break;
    case 3: l[0] = X[0]; l[1] = X[2]; l[2] = -X[1]; break;
    default: l[0] = X[0]; l[1] = -X[2]; l[2] = X[1]; break;
  }
}
// rows of R_face (g2o_cubemap_vertices_edges.cpp:173-205): l = Rf * X
__device__ __forceinline__ void face_R(int face, double* Rf) {
  for (int i = 0; i < 9; ++i) Rf[i] = 0;
  switch (face) {
    case 0: Rf[0] = 1; Rf[4] = 1; Rf[8] = 1; break;
    case 1: Rf[2] = 1; Rf[4] = 1; Rf[6] = -1; break;
    case 2: Rf[2] = -1; Rf[4] = 1; Rf[6] = 1; break;
    case 3: Rf[0] = 1; Rf[5] = 1; Rf[7] = -1; break;
    default: Rf[0] = 1; Rf[5] = -1; Rf[7] = 1; break;
  }
}
__device__ __forceinline__ void cam_point(const double* pose, const double* R, const double* X, double* Xc) {
  for (int i = 0; i < 3; ++i) Xc[i] = R[3 * i] * X[0] + R[3 * i + 1] * X[1] + R[3 * i + 2] * X[2] + pose[i];
}
template <class BAD> __device__ __forceinline__ void edge_error_v(const BAD& d, int face, double o0, double o1, const double* Xc, double* r) {
  // multipinhole_project: the camera-frame point is cast to float first (cv::Vec3f), projection stored in float
  const double Xf[3] = {(double)(float)Xc[0], (double)(float)Xc[1], (double)(float)Xc[2]};
  double l[3];
  face_local(face, Xf, l);
  const float u = (float)(l[0] * d.fx / l[2] + d.cx);
  const float v = (float)(l[1] * d.fy / l[2] + d.cy);
  r[0] = o0 - (double)u;
  r[1] = o1 - (double)v;
}
template <class BAD> __device__ __forceinline__ void edge_error(const BAD& d, int e, const double* Xc, double* r) {
  edge_error_v(d, d.e_face[e], d.e_obs[2 * e], d.e_obs[2 * e + 1], Xc, r);
}
__device__ __forceinline__ double huber_w(double e2, double delta, double* rho0) {
  const double dsqr = delta * delta;
  if (e2 <= dsqr) { *rho0 = e2; return 1.0; }
  const double s = sqrt(e2);
  *rho0 = 2 * s * delta - dsqr;
  return delta / s;
}
// Jacobians of the multi-pinhole reprojection (g2o_cubemap_vertices_edges.cpp:164-233): Jp (2x6, [rotation | translation]) and
// Jl (2x3).  The reference multiplies dense 3x3 matrices; R_face is a signed permutation, so its products are picked apart here, the
// four divisions by z share one reciprocal and mul+add pairs may contract to FMAs.  The linearisation is outside the bit-exact part
// (updates within 1e-4); the residual (edge_error) is not touched by any of this.
// (edge_jac_local: the same from the face-local point l the caller computed -- the run-major Schur body replaces l for excluded edges)
template <class BAD> __device__ __forceinline__ void edge_jac_local(const BAD& d, int face, const double* l, const double* Xc, const double* R, double* Jp, double* Jl);
template <class BAD> __device__ __forceinline__ void edge_jac_face(const BAD& d, int face, const double* Xc, const double* R, double* Jp, double* Jl) {
  double l[3];
  face_local(face, Xc, l);
  edge_jac_local(d, face, l, Xc, R, Jp, Jl);
}
template <class BAD> __device__ __forceinline__ void edge_jac_local(const BAD& d, int face, const double* l, const double* Xc, const double* R, double* Jp, double* Jl) {
#pragma clang fp contract(fast)
  const double iz = 1.0 / l[2];
  const double g00 = d.fx * iz, g11 = d.fy * iz, g02 = -(g00 * l[0]) * iz, g12 = -(g11 * l[1]) * iz;   // G = [g00 0 g02; 0 g11 g12]
  // M = -(G * R_face): the face-frame gradient carried back to the camera frame
  double M[6];
  switch (face) {
    case 0: M[0] = -g00; M[1] = 0.0; M[2] = -g02; M[3] = 0.0; M[4] = -g11; M[5] = -g12; break;
    case 1: M[0] = g02; M[1] = 0.0; M[2] = -g00; M[3] = g12; M[4] = -g11; M[5] = 0.0; break;
    case 2: M[0] = -g02; M[1] = 0.0; M[2] = g00; M[3] = -g12; M[4] = -g11; M[5] = 0.0; break;
    case 3: M[0] = -g00; M[1] = g02; M[2] = 0.0; M[3] = 0.0; M[4] = g12; M[5] = -g11; break;
    default: M[0] = -g00; M[1] = -g02; M[2] = 0.0; M[3] = 0.0; M[4] = -g12; M[5] = g11; break;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double m0 = M[3 * i], m1 = M[3 * i + 1], m2 = M[3 * i + 2];
    Jp[6 * i] = m2 * Xc[1] - m1 * Xc[2];            // M * [Xc]x, the skew matrix written out
    Jp[6 * i + 1] = m0 * Xc[2] - m2 * Xc[0];
    Jp[6 * i + 2] = m1 * Xc[0] - m0 * Xc[1];
    Jp[6 * i + 3] = m0; Jp[6 * i + 4] = m1; Jp[6 * i + 5] = m2;
#pragma unroll
    for (int j = 0; j < 3; ++j) Jl[3 * i + j] = m0 * R[j] + m1 * R[3 + j] + m2 * R[6 + j];
  }
}

template <class BAD> __device__ __forceinline__ void edge_jac(const BAD& d, int e, const double* Xc, const double* R, double* Jp, double* Jl) {
  edge_jac_face(d, d.e_face[e], Xc, R, Jp, Jl);
}
// Hpl block of edge e (6x3, row major) rebuilt from the estimate the system was linearised at and the stored weight ow[e]:
// B = ow * Jp^T Jl.  Zero for excluded edges and fixed poses, like the stored block.
template <class BAD> __device__ __forceinline__ void edge_block(const BAD& d, int e, const double* __restrict__ poses, const double* __restrict__ pts, double* Bv) {
#pragma clang fp contract(fast)
  const int k = d.e_pose[e];
  const double ow = d.ow[e];
  if (d.pose_slot[k] < 0 || ow == 0.0) {
#pragma unroll
    for (int i = 0; i < 18; ++i) Bv[i] = 0.0;
    return;
  }
  const double* pose = poses + 7 * k;
  double R[9], Xc[3], Jp[12], Jl[6];
  quat_to_R(pose + 3, R);
  cam_point(pose, R, pts + 3 * (size_t)d.e_point[e], Xc);
  edge_jac(d, e, Xc, R, Jp, Jl);
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Bv[3 * i + j] = ow * (Jp[i] * Jl[j] + Jp[6 + i] * Jl[3 + j]);
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double s = 0;
  for (int i = 0; i < nw; ++i) s += sh[i];
  return s;
}

// ---- residuals + robust chi2 partial sums (one partial per workgroup, reduced in fixed order by k_ba_reduce)
__device__ __forceinline__ void ba_errors_body(int BX, int GX, BaDevG d, const double* __restrict__ poses, const double* __restrict__ pts, int robust, double delta,
            double* __restrict__ partial) {
  __shared__ double sh[16];
  const int e = BX * blockDim.x + threadIdx.x;
  double rho0 = 0;
  if (e < d.E && d.level[e] == 0) {
    const double* pose = poses + 7 * d.e_pose[e];
    double R[9], Xc[3], r[2];
    quat_to_R(pose + 3, R);
    cam_point(pose, R, pts + 3 * d.e_point[e], Xc);
    edge_error(d, e, Xc, r);
    BA_ERR2_ST(d, e, r[0], r[1]);
    const double c2 = d.e_inv[e] * (r[0] * r[0] + r[1] * r[1]);
    if (robust) huber_w(c2, delta, &rho0); else rho0 = c2;
  }
  const double s = block_sum(rho0, sh);
  if (threadIdx.x == 0) partial[BX] = s;
}
__device__ __forceinline__ void ba_reduce_body(int BX, int GX, const double* __restrict__ partial, int n, double* __restrict__ out, int add) {
  __shared__ double sh[16];
  double v = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) v += partial[i];
  const double s = block_sum(v, sh);
  if (threadIdx.x == 0) *out = add ? *out + s : s;
}

// ---- per-point blocks: Hll (3x3), bl, Hpl per edge (6x3); one thread per point over its (point-sorted) edges
__device__ __forceinline__ void ba_lin_points_body(int BX, int GX, BaDevG d, const double* __restrict__ poses, const double* __restrict__ pts, int robust, double delta,
                double* __restrict__ Hll, double* __restrict__ bl, double* __restrict__ Hpl) {
#pragma clang fp contract(fast)                       // block accumulation: FMAs allowed (not part of the bit-exact surface)
  const int p = BX * blockDim.x + threadIdx.x;
  if (p >= d.P) return;
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
  for (int e = d.pt_off[p]; e < d.pt_off[p + 1]; ++e) {
    // the 6x3 block leaves as nine 16-byte stores: a lane's blocks are 144 B apart from its neighbour's, so the request count, not the
    // byte count, bounds these kernels (8-byte accesses doubled it)
    double2* B2v = reinterpret_cast<double2*>(Hpl + 18 * (size_t)e);
    if (d.level[e] != 0) {
      d.ow[e] = 0.0;
      if (Hpl) for (int i = 0; i < 9; ++i) B2v[i] = make_double2(0.0, 0.0);
      continue;
    }
    const double* pose = poses + 7 * d.e_pose[e];
    double R[9], Xc[3], Jp[12], Jl[6];
    quat_to_R(pose + 3, R);
    cam_point(pose, R, pts + 3 * p, Xc);
    edge_jac(d, e, Xc, R, Jp, Jl);
    const double2 rr = BA_ERR2_LD(d, e);
    const double r0 = rr.x, r1 = rr.y, om = d.e_inv[e];
    double w = 1.0, rho0;
    if (robust) w = huber_w(om * (r0 * r0 + r1 * r1), delta, &rho0);
    const double ow = w * om;
    const double o0 = -om * r0 * w, o1 = -om * r1 * w;
    for (int i = 0; i < 3; ++i) {
      b[i] += Jl[i] * o0 + Jl[3 + i] * o1;
      for (int j = 0; j < 3; ++j) H[3 * i + j] += ow * (Jl[i] * Jl[j] + Jl[3 + i] * Jl[3 + j]);
    }
    d.ow[e] = ow;
    if (Hpl) {                                    // Hpl == nullptr: the trial kernels rebuild the block from ow and the estimate instead of reading it
      const bool free_pose = d.pose_slot[d.e_pose[e]] >= 0;
      double Bv[18];
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j) Bv[3 * i + j] = free_pose ? ow * (Jp[i] * Jl[j] + Jp[6 + i] * Jl[3 + j]) : 0.0;
      for (int i = 0; i < 9; ++i) B2v[i] = make_double2(Bv[2 * i], Bv[2 * i + 1]);
    }
  }
  for (int i = 0; i < 9; ++i) Hll[9 * (size_t)p + i] = H[i];
  for (int i = 0; i < 3; ++i) bl[3 * (size_t)p + i] = b[i];
}

// ---- per-pose blocks: Hpp (6x6) and bp = Gram of [J_pose sqrt(w) | r sqrt(w)] over the pose's edge list.
// grid (K, BA_POSE_CHUNKS): every workgroup reduces one slice of the list into 27 partial sums (21 unique Hpp + 6 bp),
// k_ba_pose_finish adds the slices in fixed order (deterministic, no atomics).
#define BA_POSE_CHUNKS 8
template <int N, int H> __device__ __forceinline__ void rs_step(const double* in, double* out, bool hi, int off);   // defined below
__device__ __forceinline__ void ba_lin_poses_body(int BX, int GX, BaDevG d, const double* __restrict__ poses, const double* __restrict__ pts, int robust, double delta,
               double* __restrict__ pose_partial, int ch_arg = -1) {
#pragma clang fp contract(fast)
  __shared__ double sh[16][27];
  const int k = BX, ch = ch_arg >= 0 ? ch_arg : (int)blockIdx.y;
  __syncthreads();                                  // a previous virtual block of the same workgroup may still read sh
  const int slot = d.pose_slot[k];
  if (slot < 0) return;
  const double* pose = poses + 7 * k;
  double R[9];
  quat_to_R(pose + 3, R);
  double acc[28];
  for (int i = 0; i < 28; ++i) acc[i] = 0;
  const int e0 = d.pose_off[k], e1 = d.pose_off[k + 1];
  const int per = (e1 - e0 + BA_POSE_CHUNKS - 1) / BA_POSE_CHUNKS;
  const int t0 = e0 + ch * per, t1 = min(e1, t0 + per);
  for (int t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
    const int e = d.pose_edges[t];
    if (d.level[e] != 0) continue;
    double Xc[3], Jp[12], Jl[6];
    cam_point(pose, R, pts + 3 * d.e_point[e], Xc);
    edge_jac(d, e, Xc, R, Jp, Jl);
    const double2 rr = BA_ERR2_LD(d, e);
    const double r0 = rr.x, r1 = rr.y, om = d.e_inv[e];
    double w = 1.0, rho0;
    if (robust) w = huber_w(om * (r0 * r0 + r1 * r1), delta, &rho0);
    const double ow = w * om;
    int c = 0;
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 6; ++j) acc[c++] += ow * (Jp[i] * Jp[j] + Jp[6 + i] * Jp[6 + j]);
    for (int i = 0; i < 6; ++i) acc[21 + i] += -ow * (Jp[i] * r0 + Jp[6 + i] * r1);
  }
  // 28 -> 1 value per lane with a reduce-scatter butterfly (29 double shuffles instead of 27 x 6), then the four waves in LDS
  {
    const int lane = threadIdx.x & 63;
    double v14[14], v7[7], v4[4], v2[2], v1[1], v0[1];
    rs_step<28, 14>(acc, v14, (lane & 32) != 0, 32);
    rs_step<14, 7>(v14, v7, (lane & 16) != 0, 16);
    rs_step<7, 4>(v7, v4, (lane & 8) != 0, 8);
    rs_step<4, 2>(v4, v2, (lane & 4) != 0, 4);
    rs_step<2, 1>(v2, v1, (lane & 2) != 0, 2);
    rs_step<1, 1>(v1, v0, (lane & 1) != 0, 1);
    int idx = 0, sz = 28;
    { const bool h = lane & 32; idx += h ? 14 : 0; sz = h ? max(0, sz - 14) : min(14, sz); }
    { const bool h = lane & 16; idx += h ? 7 : 0; sz = h ? max(0, sz - 7) : min(7, sz); }
    { const bool h = lane & 8; idx += h ? 4 : 0; sz = h ? max(0, sz - 4) : min(4, sz); }
    { const bool h = lane & 4; idx += h ? 2 : 0; sz = h ? max(0, sz - 2) : min(2, sz); }
    { const bool h = lane & 2; idx += h ? 1 : 0; sz = h ? max(0, sz - 1) : min(1, sz); }
    { const bool h = lane & 1; idx += h ? 1 : 0; sz = h ? max(0, sz - 1) : min(1, sz); }
    if (sz > 0 && idx < 27) sh[threadIdx.x >> 6][idx] = v0[0];
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    double v = 0;
    for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) v += sh[wv][threadIdx.x];
    pose_partial[((size_t)slot * BA_POSE_CHUNKS + ch) * 27 + threadIdx.x] = v;
  }
}
__device__ __forceinline__ void ba_pose_finish_body(int BX, int GX, int np, const double* __restrict__ pose_partial, double* __restrict__ Hpp, double* __restrict__ bp) {
  const int slot = BX, t = threadIdx.x;
  if (slot >= np || t >= 27) return;
  double s = 0;
  for (int ch = 0; ch < BA_POSE_CHUNKS; ++ch) s += pose_partial[((size_t)slot * BA_POSE_CHUNKS + ch) * 27 + t];
  if (t < 21) {
    int i = 0, rem = t;
    while (rem >= 6 - i) { rem -= 6 - i; ++i; }
    const int j = i + rem;
    Hpp[36 * slot + 6 * i + j] = s; Hpp[36 * slot + 6 * j + i] = s;
  } else {
    bp[6 * slot + (t - 21)] = s;
  }
}

// ---- max |diag| of the assembled system (computeLambdaInit, optimization_algorithm_levenberg.cpp:166-180).
// Non-negative doubles order like their bit patterns, so the cross-workgroup maximum is one u64 atomicMax (*out zeroed first).
__device__ __forceinline__ void ba_maxdiag_body(int BX, int GX, int np, int P, const double* __restrict__ Hpp, const double* __restrict__ Hll, double* __restrict__ out) {
  double m = 0;
  const int gid = BX * blockDim.x + threadIdx.x, gs = GX * blockDim.x;
  for (int i = gid; i < 6 * np; i += gs) m = fmax(m, fabs(Hpp[36 * (i / 6) + 7 * (i % 6)]));
  for (int i = gid; i < 3 * P; i += gs) m = fmax(m, fabs(Hll[9 * (size_t)(i / 3) + 4 * (i % 3)]));
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(m));
}

// ---- reduced system: Hs = blockdiag(Hpp + lambda I), bs = bp ; then minus the Schur products
extern "C" __global__ void __launch_bounds__(256)
k_ba_schur_init(int np, const double* __restrict__ Hpp, const double* __restrict__ bp, double lambda,
                double* __restrict__ Hs, double* __restrict__ bs) {
  const int n = 6 * np;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n * n; idx += gridDim.x * blockDim.x) {
    const int r = idx / n, c = idx - r * n;
    double v = 0;
    if (r / 6 == c / 6) v = Hpp[36 * (r / 6) + 6 * (r % 6) + (c % 6)] + (r == c ? lambda : 0.0);
    Hs[idx] = v;
    if (c == 0) bs[r] = bp[r];
  }
}
__device__ __forceinline__ void inv3(const double* A, double* Ai) {
  const double a = A[0], b = A[1], c = A[2], d = A[3], e = A[4], f = A[5], g = A[6], h = A[7], i = A[8];
  const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  const double id = 1.0 / det;
  Ai[0] = (e * i - f * h) * id; Ai[1] = (c * h - b * i) * id; Ai[2] = (b * f - c * e) * id;
  Ai[3] = (f * g - d * i) * id; Ai[4] = (a * i - c * g) * id; Ai[5] = (c * d - a * f) * id;
  Ai[6] = (d * h - e * g) * id; Ai[7] = (b * g - a * h) * id; Ai[8] = (a * e - b * d) * id;
}
// per point: Dinv = (Hll + lambda I)^-1 and db = Dinv * bl
__device__ __forceinline__ void ba_dinv_body(int BX, int GX, int P, const double* __restrict__ Hll, const double* __restrict__ bl, double lambda, double* __restrict__ Dinv,
          double* __restrict__ db) {
  const int p = BX * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double D[9], Di[9];
  for (int i = 0; i < 9; ++i) D[i] = Hll[9 * (size_t)p + i] + ((i & 3) == 0 ? lambda : 0.0);
  inv3(D, Di);
  for (int i = 0; i < 9; ++i) Dinv[9 * (size_t)p + i] = Di[i];
  const double b0 = bl[3 * (size_t)p], b1 = bl[3 * (size_t)p + 1], b2 = bl[3 * (size_t)p + 2];
  db[3 * (size_t)p] = Di[0] * b0 + Di[1] * b1 + Di[2] * b2;
  db[3 * (size_t)p + 1] = Di[3] * b0 + Di[4] * b1 + Di[5] * b2;
  db[3 * (size_t)p + 2] = Di[6] * b0 + Di[7] * b1 + Di[8] * b2;
}
// Schur complement without atomics: the host lists, per co-visible pose pair (s1 <= s2), the (edge, edge) tuples of the
// points both poses observe, cut into chunks of <= BA_TUP_CHUNK tuples.  One thread evaluates one tuple (all of its 45
// operand loads are independent, so a wave has thousands of loads in flight), the 42 partial values per thread (36 for
// B1 Dinv B2^T + 6 for B1 Dinv bl on diagonal pairs) are folded across the wavefront with a reduce-scatter butterfly
// (44 shuffles instead of 42 x 6), the four wave results are added in LDS.  k_ba_schur_finish adds the chunk sums of a
// pair in fixed order and subtracts them from its 6x6 block of Hs (and the mirrored block) / bs.  Deterministic.
#define BA_TUP_CHUNK 1024    /* tuples per chunk = per workgroup of 256 threads */
template <int N, int H>
__device__ __forceinline__ void rs_step(const double* in, double* out, bool hi, int off) {
  // lanes with the bit clear keep in[0..H), lanes with it set keep in[H..N) (zero padded to H); partner's copy is added
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const double upper = (H + i < N) ? in[H + i] : 0.0;
    const double send = hi ? in[i] : upper;
    const double keep = hi ? upper : in[i];
    out[i] = keep + __shfl_xor(send, off);
  }
}
__device__ __forceinline__ void ba_schur_chunks_body(int BX, int GX, BaDevG d, const int2* __restrict__ chunk_range, const int2* __restrict__ tup, const double* __restrict__ Hpl,
                  const double* __restrict__ Dinv, const double* __restrict__ db, double* __restrict__ chunk_sum) {
  __shared__ double sh[16][42];
  const int2 rg = chunk_range[BX];
  __syncthreads();                                  // a previous virtual block of the same workgroup may still read sh
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc[42];
#pragma unroll
  for (int i = 0; i < 42; ++i) acc[i] = 0;
  for (int t = rg.x + threadIdx.x; t < rg.y; t += blockDim.x) {
    const int2 aa = tup[t];
    if (d.level[aa.x] == 0 && d.level[aa.y] == 0) {
      const int p = d.e_point[aa.x];
      const double* Di = Dinv + 9 * (size_t)p;
      double B1[18], B2[18];
      {
        const double2* q1 = reinterpret_cast<const double2*>(Hpl + 18 * (size_t)aa.x);
        const double2* q2 = reinterpret_cast<const double2*>(Hpl + 18 * (size_t)aa.y);
#pragma unroll
        for (int i = 0; i < 9; ++i) { const double2 u = q1[i], v = q2[i]; B1[2 * i] = u.x; B1[2 * i + 1] = u.y; B2[2 * i] = v.x; B2[2 * i + 1] = v.y; }
      }
      double BD[18];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) BD[3 * i + j] = B1[3 * i] * Di[j] + B1[3 * i + 1] * Di[3 + j] + B1[3 * i + 2] * Di[6 + j];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[6 * i + j] += BD[3 * i] * B2[3 * j] + BD[3 * i + 1] * B2[3 * j + 1] + BD[3 * i + 2] * B2[3 * j + 2];
      if (aa.x == aa.y) {
        const double* dbp = db + 3 * (size_t)p;
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[36 + i] += B1[3 * i] * dbp[0] + B1[3 * i + 1] * dbp[1] + B1[3 * i + 2] * dbp[2];
      }
    }
  }
  double v21[21], v11[11], v6[6], v3[3], v2[2], v1[1];
  rs_step<42, 21>(acc, v21, (lane & 32) != 0, 32);
  rs_step<21, 11>(v21, v11, (lane & 16) != 0, 16);
  rs_step<11, 6>(v11, v6, (lane & 8) != 0, 8);
  rs_step<6, 3>(v6, v3, (lane & 4) != 0, 4);
  rs_step<3, 2>(v3, v2, (lane & 2) != 0, 2);
  rs_step<2, 1>(v2, v1, (lane & 1) != 0, 1);
  // which of the 42 sums this lane ended up with (nested split sizes; padded slots are invalid)
  int s = 21, idx = (lane & 32) ? 21 : 0;
  { const bool h = lane & 16; idx += h ? 11 : 0; s = h ? max(0, s - 11) : min(11, s); }
  { const bool h = lane & 8; idx += h ? 6 : 0; s = h ? max(0, s - 6) : min(6, s); }
  { const bool h = lane & 4; idx += h ? 3 : 0; s = h ? max(0, s - 3) : min(3, s); }
  { const bool h = lane & 2; idx += h ? 2 : 0; s = h ? max(0, s - 2) : min(2, s); }
  { const bool h = lane & 1; idx += h ? 1 : 0; s = h ? max(0, s - 1) : min(1, s); }
  if (s > 0) sh[wave][idx] = v1[0];
  __syncthreads();
  if (threadIdx.x < 42) {
    const int k = threadIdx.x;
    double v = 0;
    for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) v += sh[wv][k];
    chunk_sum[(size_t)BX * 42 + k] = v;
  }
}
extern "C" __global__ void __launch_bounds__(64)
k_ba_schur_finish(int np, const int* __restrict__ pair_s1, const int* __restrict__ pair_s2, const int* __restrict__ pair_chunk_off,
                  const double* __restrict__ chunk_sum, double* __restrict__ Hs, double* __restrict__ bs) {
  const int pr = blockIdx.x, t = threadIdx.x;
  if (t >= 42) return;
  const int s1 = pair_s1[pr], s2 = pair_s2[pr], n = 6 * np;
  double v = 0;
  for (int c = pair_chunk_off[pr]; c < pair_chunk_off[pr + 1]; ++c) v += chunk_sum[(size_t)c * 42 + t];
  if (t < 36) {
    const int i = t / 6, j = t % 6;
    Hs[(size_t)(6 * s1 + i) * n + 6 * s2 + j] -= v;
    if (s1 != s2) Hs[(size_t)(6 * s2 + j) * n + 6 * s1 + i] -= v;
  } else if (s1 == s2) {
    bs[6 * s1 + (t - 36)] -= v;
  }
}

// ---- dense LDL^T of the reduced pose system, one workgroup of 16 x 32 threads, trailing matrix held in REGISTERS
// (2-D cyclic: thread (tx, ty) owns entries (i, k), i = ty + 16 r, k = tx + 32 c, k <= i).  Per column: the owners publish
// column j through a double-buffered LDS vector, one barrier, then every thread updates its tile with the rank-1 term
// a_ij a_kj / d_j.  The forward substitution rides along in the tx == 0 threads, the finished column l_ij = a_ij / d_j is
// stored to a packed LDS triangle, and a single wavefront runs the back substitution (LDS ops of one wave are in order).
template <int NR, int NC>
__device__ __forceinline__ void ba_solve_reg(int n, const double* __restrict__ A, const double* __restrict__ b,
                                             double* __restrict__ x, int* __restrict__ status, double* sm) {
  constexpr int NP = 16 * NR;
  double* Lp = sm;
  double* colbuf = Lp + (size_t)n * (n + 1) / 2;
  double* y = colbuf + 2 * NP;
  double* idg = y + NP;
  double* yjb = idg + NP;
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  double a[NR][NC], yreg[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int i = ty + 16 * r;
    yreg[r] = (tx == 0 && i < n) ? b[i] : 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int k = tx + 32 * c;
      a[r][c] = (i < n && k <= i) ? A[(size_t)i * n + k] : 0.0;
    }
  }
  bool bad = false;
  for (int j = 0; j < n; ++j) {
    double* buf = colbuf + (j & 1) * NP;
    if (tx == (j & 31)) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int i = ty + 16 * r;
        if (i >= j && i < n) {
#pragma unroll
          for (int c = 0; c < NC; ++c) if (c == (j >> 5)) buf[i] = a[r][c];
        }
      }
    }
    if (tx == 0 && ty == (j & 15)) {
#pragma unroll
      for (int r = 0; r < NR; ++r) if (r == (j >> 4)) yjb[j & 1] = yreg[r];
    }
    __syncthreads();
    const double dj = buf[j];
    if (!(isfinite(dj)) || dj == 0.0) { bad = true; break; }   // same value in every thread: uniform exit
    const double idj = 1.0 / dj, yj = yjb[j & 1];
    if (tid == 0) idg[j] = idj;
    double li[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int i = ty + 16 * r;
      li[r] = (i > j && i < n) ? buf[i] * idj : 0.0;
      if (tx == 0) {
        if (i > j && i < n) Lp[(size_t)i * (i + 1) / 2 + j] = li[r];
        yreg[r] -= li[r] * yj;
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int k = tx + 32 * c;
      const double ak = (k > j && k < n) ? buf[k] : 0.0;
#pragma unroll
      for (int r = 0; r < NR; ++r) a[r][c] -= li[r] * ak;
    }
  }
  if (tx == 0) {
#pragma unroll
    for (int r = 0; r < NR; ++r) { const int i = ty + 16 * r; if (i < n) y[i] = yreg[r]; }
  }
  __syncthreads();
  if (!bad && tid < 64) {
    for (int i = tid; i < n; i += 64) y[i] *= idg[i];           // w = D^-1 (L^-1 b)
    __builtin_amdgcn_wave_barrier();
    for (int j = n - 1; j > 0; --j) {                           // L^T x = w, column oriented
      const double xj = y[j];
      const double* row = Lp + (size_t)j * (j + 1) / 2;
      for (int i = tid; i < j; i += 64) y[i] -= row[i] * xj;
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += 512) x[i] = bad ? 0.0 : y[i];
  if (tid == 0) *status = bad ? 0 : 1;
}
extern "C" __global__ void __launch_bounds__(512)
k_ba_solve_r192(int n, const double* __restrict__ A, const double* __restrict__ b, double* __restrict__ x, int* __restrict__ status) {
  extern __shared__ __align__(16) double sm[];
  ba_solve_reg<12, 6>(n, A, b, x, status, sm);
}

extern "C" __global__ void __launch_bounds__(256)
k_ba_solve(int n, double* __restrict__ A, double* __restrict__ b, double* __restrict__ x, double* __restrict__ Dg,
           int* __restrict__ status) {
  __shared__ int bad;
  const int tid = threadIdx.x, T = blockDim.x;
  if (tid == 0) bad = 0;
  __syncthreads();
  for (int j = 0; j < n; ++j) {
    const double dj = A[(size_t)j * n + j];
    if (tid == 0) { Dg[j] = dj; if (!(isfinite(dj)) || dj == 0.0) bad = 1; }
    __syncthreads();
    if (bad) break;
    // column j of L (stored below the diagonal), keep the unscaled column in the upper triangle for the update
    for (int i = j + 1 + tid; i < n; i += T) {
      const double a = A[(size_t)i * n + j];
      A[(size_t)j * n + i] = a;          // a_ij (unscaled)
      A[(size_t)i * n + j] = a / dj;     // l_ij
    }
    __syncthreads();
    const int rem = n - j - 1;
    for (int idx = tid; idx < rem * rem; idx += T) {
      const int i = j + 1 + idx / rem, k = j + 1 + idx % rem;
      if (k <= i) A[(size_t)i * n + k] -= A[(size_t)i * n + j] * A[(size_t)j * n + k];   // l_ij * a_kj
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (!bad) {
      for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * x[k]; x[i] = s; }
      for (int i = 0; i < n; ++i) x[i] /= Dg[i];
      for (int i = n - 1; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * x[k]; x[i] = s; }
    } else {
      for (int i = 0; i < n; ++i) x[i] = 0;
    }
    *status = bad ? 0 : 1;
  }
}

// ---- landmark back-substitution + point update + gain-denominator partial sums (levenberg.cpp:182-189)
extern "C" __global__ void __launch_bounds__(128)
k_ba_backsub(BaDev d, const double* __restrict__ bl, const double* __restrict__ Hpl, const double* __restrict__ Dinv,
             const double* __restrict__ xp, double lambda, const double* __restrict__ pts, double* __restrict__ pts_new,
             double* __restrict__ partial) {
  __shared__ double sh[16];
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  double sc = 0;
  if (p < d.P) {
    double cl[3] = {bl[3 * (size_t)p], bl[3 * (size_t)p + 1], bl[3 * (size_t)p + 2]};
    int nact = 0;
    for (int a = d.pt_off[p]; a < d.pt_off[p + 1]; ++a) {
      if (d.level[a] != 0) continue;
      ++nact;
      const int s = d.pose_slot[d.e_pose[a]];
      if (s < 0) continue;
      double B[18];
      {
        const double2* q = reinterpret_cast<const double2*>(Hpl + 18 * (size_t)a);
        for (int i = 0; i < 9; ++i) { const double2 u = q[i]; B[2 * i] = u.x; B[2 * i + 1] = u.y; }
      }
      for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 6; ++i) cl[j] -= B[3 * i + j] * xp[6 * s + i];
    }
    const double* Di = Dinv + 9 * (size_t)p;
    double xl[3] = {0, 0, 0};
    if (nact > 0)
      for (int i = 0; i < 3; ++i) xl[i] = Di[3 * i] * cl[0] + Di[3 * i + 1] * cl[1] + Di[3 * i + 2] * cl[2];
    for (int i = 0; i < 3; ++i) {
      pts_new[3 * (size_t)p + i] = pts[3 * (size_t)p + i] + xl[i];
      sc += xl[i] * (lambda * xl[i] + bl[3 * (size_t)p + i]);
    }
  }
  const double s = block_sum(sc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__device__ __forceinline__ void R_to_quat(const double* m, double* q) {  // Eigen::Quaterniond(Matrix3d)
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t; t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
  } else {
    // largest diagonal entry i, then j = i + 1, k = i + 2 (mod 3) -- the three cases written out: indexing m and q with i put both into scratch memory
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > (i ? m[4] : m[0])) i = 2;
    if (i == 0) {
      t = sqrt(m[0] - m[4] - m[8] + 1.0);
      q[0] = 0.5 * t; t = 0.5 / t;
      q[3] = (m[7] - m[5]) * t; q[1] = (m[3] + m[1]) * t; q[2] = (m[6] + m[2]) * t;
    } else if (i == 1) {
      t = sqrt(m[4] - m[8] - m[0] + 1.0);
      q[1] = 0.5 * t; t = 0.5 / t;
      q[3] = (m[2] - m[6]) * t; q[2] = (m[7] + m[5]) * t; q[0] = (m[1] + m[3]) * t;
    } else {
      t = sqrt(m[8] - m[0] - m[4] + 1.0);
      q[2] = 0.5 * t; t = 0.5 / t;
      q[3] = (m[3] - m[1]) * t; q[0] = (m[2] + m[6]) * t; q[1] = (m[5] + m[7]) * t;
    }
  }
}
__device__ __forceinline__ void normalize_rot(double* q) {
  if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] /= n;
}
// T <- exp(x) * T for the free poses, copy for the fixed ones; adds the pose part of the gain denominator to *scale
extern "C" __global__ void __launch_bounds__(64)
k_ba_update_poses(BaDev d, const double* __restrict__ xp, const double* __restrict__ bp, double lambda,
                  const double* __restrict__ poses, double* __restrict__ poses_new, double* __restrict__ scale_out) {
  double sc = 0;
  for (int k = threadIdx.x; k < d.K; k += blockDim.x) {
    const double* T = poses + 7 * k;
    double* Tn = poses_new + 7 * k;
    const int s = d.pose_slot[k];
    if (s < 0) { for (int i = 0; i < 7; ++i) Tn[i] = T[i]; continue; }
    const double* u = xp + 6 * s;
    for (int i = 0; i < 6; ++i) sc += u[i] * (lambda * u[i] + bp[6 * s + i]);
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double Om[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double Om2[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Om2[3 * i + j] = Om[3 * i] * Om[j] + Om[3 * i + 1] * Om[3 + j] + Om[3 * i + 2] * Om[6 + j];
    double R[9], V[9];
    if (theta < 0.00001) {
      for (int i = 0; i < 9; ++i) { R[i] = ((i & 3) == 0 ? 1.0 : 0.0) + Om[i] + Om2[i]; V[i] = R[i]; }
    } else {
      const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / (theta * theta * theta);
      for (int i = 0; i < 9; ++i) {
        const double I = ((i & 3) == 0 ? 1.0 : 0.0);
        R[i] = I + a * Om[i] + b * Om2[i];
        V[i] = I + b * Om[i] + c * Om2[i];
      }
    }
    double Eq[4], Et[3], RE[9];
    R_to_quat(R, Eq);
    for (int i = 0; i < 3; ++i) Et[i] = V[3 * i] * up[0] + V[3 * i + 1] * up[1] + V[3 * i + 2] * up[2];
    normalize_rot(Eq);
    quat_to_R(Eq, RE);
    for (int i = 0; i < 3; ++i) Tn[i] = Et[i] + RE[3 * i] * T[0] + RE[3 * i + 1] * T[1] + RE[3 * i + 2] * T[2];
    const double* A = Eq; const double* B = T + 3;
    double q[4];
    q[3] = A[3] * B[3] - A[0] * B[0] - A[1] * B[1] - A[2] * B[2];
    q[0] = A[3] * B[0] + A[0] * B[3] + A[1] * B[2] - A[2] * B[1];
    q[1] = A[3] * B[1] + A[1] * B[3] + A[2] * B[0] - A[0] * B[2];
    q[2] = A[3] * B[2] + A[2] * B[3] + A[0] * B[1] - A[1] * B[0];
    normalize_rot(q);
    for (int i = 0; i < 4; ++i) Tn[3 + i] = q[i];
  }
  for (int o = 32; o > 0; o >>= 1) sc += __shfl_xor(sc, o);
  if (threadIdx.x == 0) *scale_out = sc;
}

// ---- outlier test of Optimizer.cpp:376-382 / 404-410: chi2 of the STORED error > th or depth <= 0
__device__ __forceinline__ void ba_classify_body(int BX, int GX, BaDevG d, const double* __restrict__ poses, const double* __restrict__ pts, double chi2_th, int set_level,
              uint8_t* __restrict__ flags) {
  const int e = BX * blockDim.x + threadIdx.x;
  if (e >= d.E) return;
  const double c2 = d.e_inv[e] * (d.err[2 * e] * d.err[2 * e] + d.err[2 * e + 1] * d.err[2 * e + 1]);
  const double* pose = poses + 7 * d.e_pose[e];
  double R[9], Xc[3];
  quat_to_R(pose + 3, R);
  cam_point(pose, R, pts + 3 * d.e_point[e], Xc);
  const bool out = c2 > chi2_th || !(Xc[2] > 0.0);
  flags[e] = out ? 1 : 0;
  if (set_level && out) d.level[e] = 1;
}
