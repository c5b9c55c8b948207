// cms_tri_kernels.hip -- the mapping thread's steps either side of the local bundle adjustment:
//   k_create_new_map_points   LocalMapping::CreateNewMapPoints (src/LocalMapping.cpp:209-386): for every neighbour key frame, in
//                             covisibility order, ORBMatcher::SearchForTriangulation (src/ORBMatcher.cpp:971-1125, the epipolar gate
//                             CheckDistEpipolarLine :388-407 with CamModelGeneral::GetVectorSigma, src/CamModelGeneral.cpp:307-333), then
//                             the ray triangulation of every match (4x4 system, cv::SVD) and its parallax / FoV / reprojection / scale
//                             tests.  One workgroup per current key frame; the neighbours are walked one after the other because a
//                             feature triangulated with neighbour i is taken when neighbour i+1 is searched (KeyFrame::AddMapPoint,
//                             LocalMapping.cpp:370) -- inside a neighbour every feature of the current key frame is independent
//                             (the reference never sets vbMatched2), so the search and the triangulation run one thread per feature.
//   k_fuse_project + k_fuse_scan   search half of ORBMatcher::Fuse(pKF, vpMapPoints, th) (src/ORBMatcher.cpp:1127-1226): projection and
//                             visibility tests per map point, window query (cms_area_kernels.hip, no level filter), then the
//                             level / reprojection-gated Hamming minimum.  The map surgery (Replace / AddObservation) stays on the host.
// Float arithmetic follows cv::Mat / cv::Matx, operation for operation (DESIGN.md section 2 lists the assumptions): 3x3 products
// without transposed operands = cv::gemm's small path (float products summed left to right, + C through double); Mat::dot and
// cv::norm accumulate in double, Matx::dot in float; `a*row + b*row` = cv::addWeighted in float; cv::SVD = one-sided Jacobi with
// double dot products and float rotations.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CMS_TRI_MAXF 4096           /* features per key frame the workgroup can hold in LDS */

struct CmsTriKF {                    // one key frame on the device; offsets into the concatenated feature / node arrays
  int f0, n;                         // features [f0, f0 + n)
  int node0, nnodes;                 // FeatureVector entries: node_id[node0 + e] ascending, e < nnodes
  int noff0, nfeat0;                 // node_off[noff0 + e] .. node_off[noff0 + e + 1] index node_feat[nfeat0 + ...] (feature indices local to the key frame)
  float Rcw[9], tcw[3], Ow[3];
};
struct CmsTriPair { int kf2; int skip; float E12[9]; float ex, ey; };
struct CmsTriJob { int kf1; int pair0, npairs; };
struct CmsTriArgs {
  const CmsTriKF* kf; const CmsTriPair* pair; const CmsTriJob* job;
  const CmsKeyPoint* kp; const uint4* desc; const float* rays; const int* mp;      // per feature (concatenated)
  const int* feat_node;              // per feature: index of its FeatureVector entry inside its key frame, -1 = in none
  const int* node_id; const int* node_off; const int* node_feat;
  int F; float cos_fov, ratio_factor; int check_orientation;
  float sf[16], sigma2[16];
  int cap;                           // output records per job
  int* n_new; int* out_neigh; int* out_idx1; int* out_idx2; float* out_x3d;
};

__device__ __forceinline__ int tri_hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}
// FaceInCubemap(const cv::Point2f&): float / int, widened (include/CamModelGeneral.h:445-470)
__device__ __forceinline__ int tri_face_in_cubemap(int F, float x, float y) {
  const double i = (double)(x / (float)F), j = (double)(y / (float)F);
  if (i >= 0 && i < 1 && j >= 1 && j < 2) return 1;
  if (i >= 1 && i < 2 && j >= 0 && j < 1) return 3;
  if (i >= 1 && i < 2 && j >= 1 && j < 2) return 0;
  if (i >= 1 && i < 2 && j >= 2 && j < 3) return 4;
  if (i >= 2 && i < 3 && j >= 1 && j < 2) return 2;
  return -1;
}
__device__ __forceinline__ float tri_gemm3(float a0, float a1, float a2, float b0, float b1, float b2) {   // cv::gemm small path: float, left to right
  float t = __fmul_rn(a0, b0);
  t = __fadd_rn(t, __fmul_rn(a1, b1));
  return __fadd_rn(t, __fmul_rn(a2, b2));
}
__device__ __forceinline__ double tri_ddot3(const float* a, const float* b) {
  double s = __dmul_rn((double)a[0], (double)b[0]);
  s = __dadd_rn(s, __dmul_rn((double)a[1], (double)b[1]));
  return __dadd_rn(s, __dmul_rn((double)a[2], (double)b[2]));
}
__device__ __forceinline__ double tri_dnorm3(const float* a) { return sqrt(tri_ddot3(a, a)); }
__device__ __forceinline__ void tri_mat3_vec(const float* A, const float* x, const float* c, float* out) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float t = tri_gemm3(A[3 * r], A[3 * r + 1], A[3 * r + 2], x[0], x[1], x[2]);
    out[r] = c ? (float)((double)t * 1.0 + (double)c[r] * 1.0) : t;
  }
}

// CamModelGeneral::GetVectorSigma(key, normalRig, 1.0f)
__device__ float tri_vector_sigma(int F, float kx, float ky, float na, float nb, float nc_) {
  const double fx = F / 2.0;
  float n0, n1;
  switch (tri_face_in_cubemap(F, kx, ky)) {       // cvtRigToFaces: only the local x, y of the normal are used
    case 0: n0 = na; n1 = nb; break;
    case 1: n0 = nc_; n1 = nb; break;
    case 2: n0 = -nc_; n1 = nb; break;
    case 4: n0 = na; n1 = -nc_; break;
    case 3: n0 = na; n1 = nc_; break;
    default: n0 = 0.0f; n1 = 0.0f;
  }
  const float epi[3] = {n1, -n0, 0.0f}, ver[3] = {n0, n1, 0.0f};
  const int i = (int)floorf(kx / (float)F), j = (int)floorf(ky / (float)F);
  const float u = __fsub_rn(kx, (float)(i * F)), v = __fsub_rn(ky, (float)(j * F));
  const float OP[3] = {(float)((double)u - fx), (float)((double)v - fx), 0.0f};
  auto fdot = [](const float* a, const float* b) {
    float s = __fadd_rn(0.0f, __fmul_rn(a[0], b[0]));
    s = __fadd_rn(s, __fmul_rn(a[1], b[1]));
    return __fadd_rn(s, __fmul_rn(a[2], b[2]));
  };
  float OO1 = (float)((double)fdot(OP, epi) / tri_dnorm3(epi)); if (OO1 < 0) OO1 = -OO1;
  const float CO1 = (float)sqrt(__dadd_rn((double)__fmul_rn(OO1, OO1), __dmul_rn(fx, fx)));
  float PO1 = (float)((double)fdot(OP, ver) / tri_dnorm3(ver)); if (PO1 < 0) PO1 = -PO1;
  const float tan1 = PO1 / CO1;
  const float tan2 = __fadd_rn(PO1, 1.0f) / CO1;
  const float tan3 = __fsub_rn(tan2, tan1) / __fadd_rn(1.0f, __fmul_rn(tan1, tan2));
  return 1.0f / sqrtf(__fadd_rn(1.0f / __fmul_rn(tan3, tan3), 1.0f));
}

// cv::SVD::compute on a 4x4 float matrix, last row of vt: one-sided Jacobi (JacobiSVDImpl_<float>); everything indexed statically
__device__ void tri_svd4_last_row(const float A[16], float out[4]) {
  float At[16], Vt[16];
  double W[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) { At[4 * i + k] = A[4 * k + i]; Vt[4 * i + k] = i == k ? 1.0f : 0.0f; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) sd = __dadd_rn(sd, __dmul_rn((double)At[4 * i + k], (double)At[4 * i + k]));
    W[i] = sd;
  }
  const float eps = 1.1920929e-07f * 2;
  for (int iter = 0; iter < 30; ++iter) {
    bool changed = false;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i + 1; j < 4; ++j) {
        double a = W[i], p = 0, b = W[j];
#pragma unroll
        for (int k = 0; k < 4; ++k) p = __dadd_rn(p, __dmul_rn((double)At[4 * i + k], (double)At[4 * j + k]));
        if (!(fabs(p) <= __dmul_rn((double)eps, sqrt(__dmul_rn(a, b))))) {
          p = __dmul_rn(p, 2.0);
          const double beta = __dsub_rn(a, b), gamma = hypot(p, beta);
          float c, s;
          if (beta < 0) {
            const double delta = __dmul_rn(__dsub_rn(gamma, beta), 0.5);
            s = (float)sqrt(delta / gamma);
            c = (float)(p / __dmul_rn(__dmul_rn(gamma, (double)s), 2.0));
          } else {
            c = (float)sqrt(__dadd_rn(gamma, beta) / __dmul_rn(gamma, 2.0));
            s = (float)(p / __dmul_rn(__dmul_rn(gamma, (double)c), 2.0));
          }
          a = 0; b = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float ai = At[4 * i + k], aj = At[4 * j + k];
            const float t0 = __fadd_rn(__fmul_rn(c, ai), __fmul_rn(s, aj));
            const float t1 = __fadd_rn(__fmul_rn(-s, ai), __fmul_rn(c, aj));
            At[4 * i + k] = t0; At[4 * j + k] = t1;
            a = __dadd_rn(a, __dmul_rn((double)t0, (double)t0)); b = __dadd_rn(b, __dmul_rn((double)t1, (double)t1));
          }
          W[i] = a; W[j] = b;
          changed = true;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float vi = Vt[4 * i + k], vj = Vt[4 * j + k];
            Vt[4 * i + k] = __fadd_rn(__fmul_rn(c, vi), __fmul_rn(s, vj));
            Vt[4 * j + k] = __fadd_rn(__fmul_rn(-s, vi), __fmul_rn(c, vj));
          }
        }
      }
    if (!changed) break;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) sd = __dadd_rn(sd, __dmul_rn((double)At[4 * i + k], (double)At[4 * i + k]));
    W[i] = sqrt(sd);
  }
  // selection sort, descending, first maximum wins; only W and Vt matter from here on
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int j = i;                                       // (compare against the running maximum instead of indexing W[j] dynamically)
    double wmax = W[i];
#pragma unroll
    for (int k = i + 1; k < 4; ++k) if (wmax < W[k]) { wmax = W[k]; j = k; }
#pragma unroll
    for (int k = i + 1; k < 4; ++k)
      if (j == k) {
        const double tw = W[i]; W[i] = W[k]; W[k] = tw;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float tv = Vt[4 * i + q]; Vt[4 * i + q] = Vt[4 * k + q]; Vt[4 * k + q] = tv; }
      }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) out[k] = Vt[12 + k];
}

// r_a*(T.row(ia)+T.row(ib)) - (r_b+r_c)*T.row(ic) as two cv::addWeighted passes in float
__device__ __forceinline__ void tri_row(const float* R, const float* t, int ia, int ib, int ic, float ra, float rb, float rc, float* out) {
  const float g = -__fadd_rn(rb, rc);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float va = k < 3 ? R[3 * ia + k] : t[ia], vb = k < 3 ? R[3 * ib + k] : t[ib], vc = k < 3 ? R[3 * ic + k] : t[ic];
    const float tmp = __fadd_rn(__fmul_rn(va, ra), __fmul_rn(vb, ra));
    out[k] = __fadd_rn(__fmul_rn(tmp, 1.0f), __fmul_rn(vc, g));
  }
}

// the inner loop body of CreateNewMapPoints for one match (LocalMapping.cpp:266-357)
__device__ bool tri_triangulate(const CmsTriArgs& a, const CmsTriKF& k1, const CmsTriKF& k2, const CmsKeyPoint& kp1, const CmsKeyPoint& kp2,
                                const float* r1, const float* r2, float* x3D) {
  float ray1[3], ray2[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {                     // Rwc = Rcw.t(): row r of Rwc is column r of Rcw
    ray1[r] = tri_gemm3(k1.Rcw[r], k1.Rcw[3 + r], k1.Rcw[6 + r], r1[0], r1[1], r1[2]);
    ray2[r] = tri_gemm3(k2.Rcw[r], k2.Rcw[3 + r], k2.Rcw[6 + r], r2[0], r2[1], r2[2]);
  }
  const float cosPar = (float)(tri_ddot3(ray1, ray2) / __dmul_rn(tri_dnorm3(ray1), tri_dnorm3(ray2)));
  const float cosStereo = __fadd_rn(cosPar, 1.0f);
  if (!(cosPar < cosStereo && cosPar > 0 && (double)cosPar < 0.9998)) return false;
  float A[16], v4[4];
  tri_row(k1.Rcw, k1.tcw, 1, 2, 0, r1[0], r1[1], r1[2], A);
  tri_row(k1.Rcw, k1.tcw, 0, 2, 1, r1[1], r1[0], r1[2], A + 4);
  tri_row(k2.Rcw, k2.tcw, 1, 2, 0, r2[0], r2[1], r2[2], A + 8);
  tri_row(k2.Rcw, k2.tcw, 0, 2, 1, r2[1], r2[0], r2[2], A + 12);
  tri_svd4_last_row(A, v4);
  if (v4[3] == 0) return false;
  const float inv_w = (float)(1.0 / (double)v4[3]);
#pragma unroll
  for (int k = 0; k < 3; ++k) x3D[k] = __fmul_rn(v4[k], inv_w);
  float xc[3];
  tri_mat3_vec(k1.Rcw, x3D, k1.tcw, xc);
  if ((float)((double)xc[2] / tri_dnorm3(xc)) <= a.cos_fov) return false;
  tri_mat3_vec(k2.Rcw, x3D, k2.tcw, xc);
  if ((float)((double)xc[2] / tri_dnorm3(xc)) <= a.cos_fov) return false;
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const CmsTriKF& k = v ? k2 : k1;
    const CmsKeyPoint& kp = v ? kp2 : kp1;
    const float x = (float)__dadd_rn(tri_ddot3(k.Rcw, x3D), (double)k.tcw[0]), y = (float)__dadd_rn(tri_ddot3(k.Rcw + 3, x3D), (double)k.tcw[1]),
                z = (float)__dadd_rn(tri_ddot3(k.Rcw + 6, x3D), (double)k.tcw[2]);
    float u, w;
    track_rays_to_cubemap(a.F, x, y, z, u, w);
    const float eX = __fsub_rn(u, kp.x), eY = __fsub_rn(w, kp.y);
    if ((double)__fadd_rn(__fmul_rn(eX, eX), __fmul_rn(eY, eY)) > __dmul_rn(5.991, (double)a.sigma2[kp.octave])) return false;
  }
  const float n1[3] = {__fsub_rn(x3D[0], k1.Ow[0]), __fsub_rn(x3D[1], k1.Ow[1]), __fsub_rn(x3D[2], k1.Ow[2])};
  const float n2[3] = {__fsub_rn(x3D[0], k2.Ow[0]), __fsub_rn(x3D[1], k2.Ow[1]), __fsub_rn(x3D[2], k2.Ow[2])};
  const float dist1 = (float)tri_dnorm3(n1), dist2 = (float)tri_dnorm3(n2);
  if (dist1 == 0 || dist2 == 0) return false;
  const float ratioDist = dist2 / dist1;
  const float ratioOctave = a.sf[kp1.octave] / a.sf[kp2.octave];
  if (__fmul_rn(ratioDist, a.ratio_factor) < ratioOctave || ratioDist > __fmul_rn(ratioOctave, a.ratio_factor)) return false;
  return true;
}

// ORBMatcher::SearchForTriangulation for ONE feature of the current key frame (ORBMatcher.cpp:1009-1058): the key point of the
// neighbour it pairs with, or -1.  The caller has checked that the feature holds no map point.
__device__ int tri_search_feature(const CmsTriArgs& a, const CmsTriKF& k1, const CmsTriKF& k2, const CmsTriPair& pr, int idx1) {
  int best2 = -1;
  const int e1 = a.feat_node[k1.f0 + idx1];
  if (e1 >= 0) {
    const int node = a.node_id[k1.node0 + e1];
    int lo = 0, hi = k2.nnodes;                  // lower_bound in the neighbour's node list
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.node_id[k2.node0 + mid] < node) lo = mid + 1; else hi = mid; }
    if (lo < k2.nnodes && a.node_id[k2.node0 + lo] == node) {
      const uint4 d0 = a.desc[2 * (size_t)(k1.f0 + idx1)], d1 = a.desc[2 * (size_t)(k1.f0 + idx1) + 1];
      const float* ray1 = a.rays + 3 * (size_t)(k1.f0 + idx1);
      // epipolar plane normal of this feature in the neighbour's frame: l = ray1' E12 (CheckDistEpipolarLine)
      const float la = __fadd_rn(__fadd_rn(__fmul_rn(ray1[0], pr.E12[0]), __fmul_rn(ray1[1], pr.E12[3])), __fmul_rn(ray1[2], pr.E12[6]));
      const float lb = __fadd_rn(__fadd_rn(__fmul_rn(ray1[0], pr.E12[1]), __fmul_rn(ray1[1], pr.E12[4])), __fmul_rn(ray1[2], pr.E12[7]));
      const float lc = __fadd_rn(__fadd_rn(__fmul_rn(ray1[0], pr.E12[2]), __fmul_rn(ray1[1], pr.E12[5])), __fmul_rn(ray1[2], pr.E12[8]));
      const float den = __fadd_rn(__fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb)), __fmul_rn(lc, lc));
      int bestDist = 50;                         // TH_LOW
      const int nb0 = a.node_off[k2.noff0 + lo], nb1 = a.node_off[k2.noff0 + lo + 1];
      for (int b = nb0; b < nb1; ++b) {
        const int idx2 = a.node_feat[k2.nfeat0 + b];
        const size_t g2 = (size_t)(k2.f0 + idx2);
        if (a.mp[g2] >= 0) continue;
        const int dist = tri_hamming256(d0, d1, a.desc[2 * g2], a.desc[2 * g2 + 1]);
        if (dist > 50 || dist > bestDist) continue;
        const CmsKeyPoint kp2 = a.kp[g2];
        const float dex = __fsub_rn(pr.ex, kp2.x), dey = __fsub_rn(pr.ey, kp2.y);
        if (__fadd_rn(__fmul_rn(dex, dex), __fmul_rn(dey, dey)) < __fmul_rn(100.0f, a.sf[kp2.octave])) continue;
        if (den == 0) continue;
        const float* ray2 = a.rays + 3 * g2;
        const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, ray2[0]), __fmul_rn(lb, ray2[1])), __fmul_rn(lc, ray2[2]));
        const float sigma = tri_vector_sigma(a.F, kp2.x, kp2.y, la, lb, lc);
        const float dsqr = __fmul_rn(num, num) / __fmul_rn(__fmul_rn(den, __fmul_rn(sigma, sigma)), a.sigma2[kp2.octave]);
        if ((double)dsqr < 3.84) { best2 = idx2; bestDist = dist; }
      }
    }
  }
  return best2;
}

// n16 x 16 bytes from src to dst by one launch: the small host -> device hand-overs of the mapping side (src: pinned host memory the device
// can read) that must not queue behind the copy engines' large transfers
extern "C" __global__ void __launch_bounds__(64) k_copy16(uint4* __restrict__ dst, const uint4* __restrict__ src, int n16) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
extern "C" __global__ void __launch_bounds__(512) k_create_new_map_points(CmsTriArgs a) {
  __shared__ uint8_t s_taken[CMS_TRI_MAXF];        // KeyFrame::GetMapPoint(idx1) != NULL for the current key frame, updated as points are created
  __shared__ int s_match[CMS_TRI_MAXF];
  __shared__ int s_hist[32];
  __shared__ int s_keep[3];
  __shared__ int s_wsum[8];
  __shared__ int s_base;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6;
  const CmsTriJob jb = a.job[blockIdx.x];
  const CmsTriKF k1 = a.kf[jb.kf1];
  for (int i = tid; i < k1.n; i += nt) s_taken[i] = a.mp[k1.f0 + i] >= 0;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int pi = 0; pi < jb.npairs; ++pi) {
    const CmsTriPair pr = a.pair[jb.pair0 + pi];
    if (pr.skip) continue;                           // baseline / median depth < 0.01 (LocalMapping.cpp:243-247)
    const CmsTriKF k2 = a.kf[pr.kf2];
    if (tid < 32) s_hist[tid] = 0;
    __syncthreads();
    // ---- SearchForTriangulation: one thread per feature of the current key frame
    for (int idx1 = tid; idx1 < k1.n; idx1 += nt) {
      const int best2 = s_taken[idx1] ? -1 : tri_search_feature(a, k1, k2, pr, idx1);
      s_match[idx1] = best2;
      if (best2 >= 0 && a.check_orientation) {
        float rot = __fsub_rn(a.kp[k1.f0 + idx1].angle, a.kp[k2.f0 + best2].angle);
        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
        int bin = (int)roundf(__fmul_rn(rot, 1.0f / 12));
        if (bin == 30) bin = 0;
        atomicAdd(&s_hist[bin], 1);
      }
    }
    __syncthreads();
    if (a.check_orientation) {                       // ComputeThreeMaxima (ORBMatcher.cpp:905-946) on the bin sizes
      if (tid == 0) {
        int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
        for (int i = 0; i < 30; ++i) {
          const int s = s_hist[i];
          if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
          else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
          else if (s > max3) { max3 = s; i3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) i3 = -1;
        s_keep[0] = i1; s_keep[1] = i2; s_keep[2] = i3;
      }
      __syncthreads();
      for (int idx1 = tid; idx1 < k1.n; idx1 += nt) {
        const int m = s_match[idx1];
        if (m < 0) continue;
        float rot = __fsub_rn(a.kp[k1.f0 + idx1].angle, a.kp[k2.f0 + m].angle);
        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
        int bin = (int)roundf(__fmul_rn(rot, 1.0f / 12));
        if (bin == 30) bin = 0;
        if (bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) s_match[idx1] = -1;
      }
      __syncthreads();
    }
    // ---- triangulate the matches in ascending idx1 (the order of vMatchedPairs), nt features per pass, ordered compaction
    for (int c0 = 0; c0 < k1.n; c0 += nt) {
      const int idx1 = c0 + tid;
      bool ok = false;
      float x3D[3] = {0, 0, 0};
      int idx2 = -1;
      if (idx1 < k1.n) {
        idx2 = s_match[idx1];
        if (idx2 >= 0)
          ok = tri_triangulate(a, k1, k2, a.kp[k1.f0 + idx1], a.kp[k2.f0 + idx2], a.rays + 3 * (size_t)(k1.f0 + idx1), a.rays + 3 * (size_t)(k2.f0 + idx2), x3D);
      }
      const unsigned long long bal = __ballot(ok);
      if (lane == 0) s_wsum[wv] = __popcll(bal);
      __syncthreads();
      int before = s_base, total = 0;
      for (int w = 0; w < (nt >> 6); ++w) { if (w < wv) before += s_wsum[w]; total += s_wsum[w]; }
      if (ok) {
        const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
        if (pos < a.cap) {
          const size_t o = (size_t)blockIdx.x * a.cap + pos;
          a.out_neigh[o] = pi; a.out_idx1[o] = idx1; a.out_idx2[o] = idx2;
          a.out_x3d[3 * o] = x3D[0]; a.out_x3d[3 * o + 1] = x3D[1]; a.out_x3d[3 * o + 2] = x3D[2];
        }
        s_taken[idx1] = 1;                           // mpCurrentKeyFrame->AddMapPoint(pMP, idx1)
      }
      __syncthreads();
      if (tid == 0) s_base += total;
      __syncthreads();
    }
  }
  if (tid == 0) a.n_new[blockIdx.x] = s_base;
}

// ORBMatcher::SearchForTriangulation alone (cms_search_for_triangulation): job 0's key frame against the one neighbour of pair 0; matches12[idx1] =
// index in the neighbour or -1, *n_matches = the count (after the rotation histogram when it is on).  One workgroup: the histogram is per call.
extern "C" __global__ void __launch_bounds__(1024) k_tri_search(CmsTriArgs a, int* __restrict__ matches12, int* __restrict__ n_matches) {
  __shared__ int s_hist[32];
  __shared__ int s_keep[3];
  __shared__ int s_cnt;
  const int tid = threadIdx.x, nt = blockDim.x;
  const CmsTriJob jb = a.job[0];
  const CmsTriKF k1 = a.kf[jb.kf1];
  const CmsTriPair pr = a.pair[jb.pair0];
  const CmsTriKF k2 = a.kf[pr.kf2];
  if (tid < 32) s_hist[tid] = 0;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  auto bin_of = [&](int idx1, int idx2) {
    float rot = __fsub_rn(a.kp[k1.f0 + idx1].angle, a.kp[k2.f0 + idx2].angle);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, 1.0f / 12));
    return bin == 30 ? 0 : bin;
  };
  for (int idx1 = tid; idx1 < k1.n; idx1 += nt) {
    const int best2 = a.mp[k1.f0 + idx1] >= 0 ? -1 : tri_search_feature(a, k1, k2, pr, idx1);
    matches12[idx1] = best2;
    if (best2 >= 0 && a.check_orientation) atomicAdd(&s_hist[bin_of(idx1, best2)], 1);
  }
  __syncthreads();
  if (a.check_orientation) {                         // ComputeThreeMaxima (ORBMatcher.cpp:905-946) on the bin sizes
    if (tid == 0) {
      int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
      for (int i = 0; i < 30; ++i) {
        const int s = s_hist[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
        else if (s > max3) { max3 = s; i3 = i; }
      }
      if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) i3 = -1;
      s_keep[0] = i1; s_keep[1] = i2; s_keep[2] = i3;
    }
    __syncthreads();
  }
  int mine = 0;
  for (int idx1 = tid; idx1 < k1.n; idx1 += nt) {
    int m = matches12[idx1];
    if (m >= 0 && a.check_orientation) {
      const int bin = bin_of(idx1, m);
      if (bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) { m = -1; matches12[idx1] = -1; }
    }
    mine += m >= 0;
  }
  if (mine) atomicAdd(&s_cnt, mine);
  __syncthreads();
  if (tid == 0) *n_matches = s_cnt;
}

// ---- the same result without walking the neighbours one after the other (check_orientation == 0, the reference's call site).
// Neither the search nor the triangulation of (feature, neighbour) looks at what other features did; the only coupling is "a feature
// that got its point from an earlier neighbour is skipped by the later ones".  So every (neighbour, feature) pair is evaluated at once
// (k_tri_candidates, one thread each), and k_tri_resolve keeps for every feature the FIRST neighbour whose candidate survived, then
// writes the records in the reference's creation order (neighbour ascending, feature ascending inside a neighbour).
struct CmsTriCand { int idx2; float x, y, z; };       // idx2 < 0: no surviving candidate
extern "C" __global__ void __launch_bounds__(256) k_tri_candidates(CmsTriArgs a, const int* __restrict__ pair_job, CmsTriCand* __restrict__ cand, int maxf) {
  const int pg = blockIdx.y;                           // global pair index
  const CmsTriPair pr = a.pair[pg];
  const CmsTriKF k1 = a.kf[a.job[pair_job[pg]].kf1];
  const int idx1 = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx1 >= k1.n) return;
  CmsTriCand c; c.idx2 = -1; c.x = 0; c.y = 0; c.z = 0;
  if (!pr.skip && a.mp[k1.f0 + idx1] < 0) {
    const CmsTriKF k2 = a.kf[pr.kf2];
    const int idx2 = tri_search_feature(a, k1, k2, pr, idx1);
    if (idx2 >= 0) {
      float x3D[3];
      if (tri_triangulate(a, k1, k2, a.kp[k1.f0 + idx1], a.kp[k2.f0 + idx2], a.rays + 3 * (size_t)(k1.f0 + idx1), a.rays + 3 * (size_t)(k2.f0 + idx2), x3D)) {
        c.idx2 = idx2; c.x = x3D[0]; c.y = x3D[1]; c.z = x3D[2];
      }
    }
  }
  cand[(size_t)pg * maxf + idx1] = c;
}

extern "C" __global__ void __launch_bounds__(1024) k_tri_resolve(CmsTriArgs a, const CmsTriCand* __restrict__ cand, int maxf) {
  __shared__ short s_win[CMS_TRI_MAXF];                // neighbour that creates the point of feature i, -1 = none
  __shared__ int s_cnt[64], s_base[65];
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
  const CmsTriJob jb = a.job[blockIdx.x];
  const CmsTriKF k1 = a.kf[jb.kf1];
  for (int p = tid; p < 64; p += nt) s_cnt[p] = 0;
  __syncthreads();
  for (int i = tid; i < k1.n; i += nt) {
    int w = -1;
    for (int p = 0; p < jb.npairs; ++p)
      if (cand[(size_t)(jb.pair0 + p) * maxf + i].idx2 >= 0) { w = p; break; }
    s_win[i] = (short)w;
    if (w >= 0) atomicAdd(&s_cnt[w & 63], 1);          // (npairs <= 64, checked on the host)
  }
  __syncthreads();
  if (tid == 0) { int t = 0; for (int p = 0; p < 64; ++p) { s_base[p] = t; t += s_cnt[p]; } s_base[64] = t; a.n_new[blockIdx.x] = t; }
  __syncthreads();
  // one wavefront per neighbour: its features in ascending order
  for (int p = wv; p < jb.npairs; p += nw) {
    int pos = s_base[p];
    for (int i0 = 0; i0 < k1.n; i0 += 64) {
      const int i = i0 + lane;
      const bool mine = i < k1.n && s_win[i] == p;
      const unsigned long long bal = __ballot(mine);
      if (mine) {
        const int o = pos + __popcll(bal & ((1ull << lane) - 1ull));
        if (o < a.cap) {
          const CmsTriCand c = cand[(size_t)(jb.pair0 + p) * maxf + i];
          const size_t r = (size_t)blockIdx.x * a.cap + o;
          a.out_neigh[r] = p; a.out_idx1[r] = i; a.out_idx2[r] = c.idx2;
          a.out_x3d[3 * r] = c.x; a.out_x3d[3 * r + 1] = c.y; a.out_x3d[3 * r + 2] = c.z;
        }
      }
      pos += __popcll(bal);
    }
  }
}

// ---- Fuse: projection half.  Writes the window (qx, qy, qr; qr < 0 = rejected) and the predicted level per map point.
struct CmsFuseArgs {
  const float* pose15; const int* mp_frame; int n;
  const uint8_t* skip; const float* P; const float* normal; const float* min_dist; const float* max_dist;
  float th, log_scale; int nlevels, F; float sf[16];
  float* qx; float* qy; float* qr; int* qmin; int* qmax; int* level;
  int bounds_scaled;           // cms_set_distance_bounds_mode
  const int* src;              // NULL, or per entry the map point whose P / normal / min_dist / max_dist it stands for (jobs that share a set of map points)
};
extern "C" __global__ void __launch_bounds__(256) k_fuse_project(CmsFuseArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const float* ps = a.pose15 + 15 * (size_t)(a.mp_frame ? a.mp_frame[i] : 0);
  float u = -1.0f, v = -1.0f, r = -1.0f; int lvl = -1;
  do {
    if (a.skip && a.skip[i]) break;
    const size_t m = a.src ? (size_t)a.src[i] : (size_t)i;
    const float p[3] = {a.P[3 * m], a.P[3 * m + 1], a.P[3 * m + 2]};
    float pc[3];
    tri_mat3_vec(ps, p, ps + 9, pc);
    track_rays_to_cubemap(a.F, pc[0], pc[1], pc[2], u, v);       // the face is not looked at (ORBMatcher.cpp:1151)
    const float mx = (float)(3 * a.F);
    if (!(u >= 0.0f && u < mx && v >= 0.0f && v < mx)) break;      // KeyFrame::IsInImage
    float maxd, maxDistance, minDistance;
    track_distance_bounds(a.bounds_scaled, a.min_dist[m], a.max_dist[m], minDistance, maxDistance, maxd);
    const float PO[3] = {__fsub_rn(p[0], ps[12]), __fsub_rn(p[1], ps[13]), __fsub_rn(p[2], ps[14])};
    const float dist3D = (float)tri_dnorm3(PO);
    if (dist3D < minDistance || dist3D > maxDistance) break;
    if (tri_ddot3(PO, a.normal + 3 * m) < __dmul_rn(0.5, (double)dist3D)) break;
    const float ratio = maxd / dist3D;
    int ns = (int)ceilf((float)log((double)ratio) / a.log_scale);
    if (ns < 0) ns = 0; else if (ns >= a.nlevels) ns = a.nlevels - 1;
    lvl = ns; r = __fmul_rn(a.th, a.sf[ns]);
  } while (false);
  a.qx[i] = u; a.qy[i] = v; a.qr[i] = r; a.qmin[i] = -1; a.qmax[i] = -1; a.level[i] = lvl;
}

// ---- Fuse: gated Hamming minimum over the window (ORBMatcher.cpp:1186-1214); eight lanes per map point
struct CmsFuseScanArgs {
  int n; const float* qx; const float* qy; const int* level; const uint4* mp_desc; const int* cand_off; const int* cand_idx;
  const CmsKeyPoint* kp; const uint4* t_desc; float inv_sigma2[16]; int* best_idx; int* best_dist;
  int cap;                      // entries of cand_idx that exist (0: all of them): a scan enqueued before the host has seen the total never reads beyond them
  const int* src;               // NULL, or per entry the map point whose descriptor it scans with (see CmsFuseArgs::src)
  const int* row_slot; int maxf; // NULL, or per entry the store slot its candidates' rows belong to: the result is then the key-point index inside that key frame (row - slot x maxf)
};
extern "C" __global__ void __launch_bounds__(256) k_fuse_scan(CmsFuseScanArgs a) {
  const int n = a.n;
  const float* qx = a.qx; const float* qy = a.qy; const int* level = a.level; const uint4* mp_desc = a.mp_desc;
  const int* cand_off = a.cand_off; const int* cand_idx = a.cand_idx; const CmsKeyPoint* kp = a.kp; const uint4* t_desc = a.t_desc;
  int* best_idx = a.best_idx; int* best_dist = a.best_dist;
  const int gl = threadIdx.x & 7;
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const bool live = i < n;
  const int ii = live ? i : 0;
  const int c0 = live ? cand_off[ii] : 0, c1 = live ? (a.cap > 0 ? min(cand_off[ii + 1], a.cap) : cand_off[ii + 1]) : 0;
  const int lvl = level[ii];
  const float u = qx[ii], v = qy[ii];
  uint32_t key = 0xFFFFFFFFu;                        // dist << 20 | position: the first minimum in list order wins
  if (c1 > c0) {
    const size_t mi = a.src ? (size_t)a.src[ii] : (size_t)ii;
    const uint4 d0 = mp_desc[2 * mi], d1 = mp_desc[2 * mi + 1];
    for (int c = c0 + gl; c < c1; c += 8) {
      const size_t row = (size_t)cand_idx[c];
      const CmsKeyPoint k = kp[row];
      if (k.octave < lvl - 1 || k.octave > lvl) continue;
      const float ex = __fsub_rn(u, k.x), ey = __fsub_rn(v, k.y);
      const float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
      if ((double)__fmul_rn(e2, a.inv_sigma2[k.octave & 15]) > 5.99) continue;
      const uint32_t d = (uint32_t)tri_hamming256(d0, d1, t_desc[2 * row], t_desc[2 * row + 1]);
      const uint32_t kk = (d << 20) | (uint32_t)(c - c0);
      key = min(key, kk);
    }
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) key = min(key, (uint32_t)__shfl_xor((int)key, o, 8));
  if (live && gl == 0) {
    const int d = key == 0xFFFFFFFFu ? 256 : (int)(key >> 20);
    const bool hit = d <= 50;                        // TH_LOW
    best_idx[i] = hit ? cand_idx[c0 + (int)(key & 0xFFFFFu)] - (a.row_slot ? a.row_slot[i] * a.maxf : 0) : -1;
    best_dist[i] = hit ? d : 256;
  }
}

// ---- MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cpp:243-308), one wavefront per map point.  Lane i owns observation i:
// the median of its row of the distance matrix is found by bisection on the value (distances are integers in [0, 256]; the smallest v
// with #(d <= v) > idx is sorted[idx]), each probe recomputing the row's Hamming distances from the descriptors (L1 resident);
// then the wave keeps the smallest (median, i) -- "first one on ties", as `median < BestMedian` does.
extern "C" __global__ void __launch_bounds__(256) k_distinctive(int npts, const int* __restrict__ obs_off, const uint4* __restrict__ desc,
                                                                int* __restrict__ best_idx) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= npts) return;
  const int o0 = obs_off[p], N = obs_off[p + 1] - o0;
  if (N <= 0) { if (lane == 0) best_idx[p] = -1; return; }
  const int idx = (int)(0.5 * (N - 1));
  uint32_t best = 0xFFFFFFFFu;
  for (int i = lane; i < N; i += 64) {
    const uint4 a0 = desc[2 * (size_t)(o0 + i)], a1 = desc[2 * (size_t)(o0 + i) + 1];
    int lo = 0, hi = 256;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      int cnt = 0;
      for (int k = 0; k < N; ++k) {
        const int d = k == i ? 0 : tri_hamming256(a0, a1, desc[2 * (size_t)(o0 + k)], desc[2 * (size_t)(o0 + k) + 1]);
        cnt += d <= mid;
      }
      if (cnt > idx) hi = mid; else lo = mid + 1;
    }
    best = min(best, ((uint32_t)lo << 16) | (uint32_t)i);
  }
  for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o));
  if (lane == 0) best_idx[p] = (int)(best & 0xFFFFu);
}

// ---- MapPoint::UpdateNormalAndDepth (src/MapPoint.cpp:332-373), one thread per map point, observations in the caller's (std::map) order
extern "C" __global__ void __launch_bounds__(256)
k_update_normal_depth(int npts, const int* __restrict__ obs_off, const float* __restrict__ pos, const float* __restrict__ obs_Ow,
                      const float* __restrict__ ref_Ow, const int* __restrict__ ref_level, const float* __restrict__ sf16, int nlevels,
                      float* __restrict__ normal, float* __restrict__ min_dist, float* __restrict__ max_dist) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npts) return;
  const int o0 = obs_off[p], o1 = obs_off[p + 1];
  if (o1 <= o0) return;
  const float P[3] = {pos[3 * (size_t)p], pos[3 * (size_t)p + 1], pos[3 * (size_t)p + 2]};
  float nrm[3] = {0.0f, 0.0f, 0.0f};
  for (int o = o0; o < o1; ++o) {
    const float ni[3] = {__fsub_rn(P[0], obs_Ow[3 * (size_t)o]), __fsub_rn(P[1], obs_Ow[3 * (size_t)o + 1]), __fsub_rn(P[2], obs_Ow[3 * (size_t)o + 2])};
    const float beta = (float)(1.0 / tri_dnorm3(ni));
#pragma unroll
    for (int k = 0; k < 3; ++k) nrm[k] = __fadd_rn(__fmul_rn(nrm[k], 1.0f), __fmul_rn(ni[k], beta));   // cv::addWeighted(normal, 1, normali, 1/norm)
  }
  const float PC[3] = {__fsub_rn(P[0], ref_Ow[3 * (size_t)p]), __fsub_rn(P[1], ref_Ow[3 * (size_t)p + 1]), __fsub_rn(P[2], ref_Ow[3 * (size_t)p + 2])};
  const float dist = (float)tri_dnorm3(PC);
  const float mx = __fmul_rn(dist, sf16[ref_level[p]]);
  max_dist[p] = mx;
  min_dist[p] = mx / sf16[nlevels - 1];
  const float inv_n = (float)(1.0 / (double)(o1 - o0));
#pragma unroll
  for (int k = 0; k < 3; ++k) normal[3 * (size_t)p + k] = __fmul_rn(nrm[k], inv_n);
}
