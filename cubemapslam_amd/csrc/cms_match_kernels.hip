// cms_match_kernels.hip -- Hamming matching kernels for gfx950.
//
// ORBMatcher::DescriptorDistance (ORBMatcher.cpp:951-967) + the best / second-best scan that every guided matcher
// runs over the candidate list of a query (ORBMatcher.cpp:84-113, 186-205).  The sequential scan keeps the two
// smallest distances with "first one wins" on ties, i.e. the first two elements of the candidate list stably sorted
// by distance.  With the key (distance << 22 | position-in-list) that is simply the two smallest keys, so the scan
// parallelises: one wavefront per query (lane id from v_mbcnt), lanes stride the candidates (8 x xor + v_bcnt_u32_b32 each), and the two
// smallest keys of the wave are merged without a trip through LDS memory: DPP quad permutes and row mirrors inside a row of 16 lanes
// (v_mov_b32 dpp, no LDS pipe at all), one ds_swizzle stage across the two rows of a half (the crossbar only, no bank access), and a
// v_readlane for the other half -- lane 0 is the only one that needs the answer.  (Round 4 used six __shfl_xor stages = twelve
// ds_bpermute_b32 per query.)  The greedy assignment loops stay on the host.
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1) {
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

extern "C" __global__ void __launch_bounds__(256)
k_hamming_best2(const uint4* __restrict__ qdesc, const int* __restrict__ q_row, int nq, const uint4* __restrict__ tdesc,
                const int* __restrict__ cand_off, const int* __restrict__ cand_idx, const int* __restrict__ tlevel,
                const uint8_t* __restrict__ texcl, int* __restrict__ best_idx, int* __restrict__ best_dist,
                int* __restrict__ best_level, int* __restrict__ second_dist, int* __restrict__ second_level) {
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));      // lane of the wavefront (workgroups are whole wavefronts)
  const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (q >= nq) return;
  const size_t qr = q_row ? (size_t)q_row[q] : (size_t)q;   // optional gather: query q lives in row q_row[q]
  const uint4 q0 = qdesc[2 * qr], q1 = qdesc[2 * qr + 1];
  const int start = cand_off[q], end = cand_off[q + 1];
  uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
  for (int c = start + lane; c < end; c += 64) {
    const int idx = cand_idx[c];
    if (texcl && texcl[idx]) continue;
    const int d = hamming256(q0, q1, tdesc[2 * (size_t)idx], tdesc[2 * (size_t)idx + 1]);
    const uint32_t key = ((uint32_t)d << 22) | (uint32_t)(c - start);
    if (key < k1) { k2 = k1; k1 = key; }
    else if (key < k2) k2 = key;
  }
  // merge of two (smallest, second smallest) pairs over disjoint lane sets; every stage pairs a group of lanes with a disjoint one, so that
  // after it both groups hold the merged pair: quads (xor 1, xor 2), the two quads of a half row (mirror of 8), the halves of a row (mirror of
  // 16), the two rows of a half wave (ds_swizzle xor 16), and finally lane 0 with lane 32
#define CMS_MERGE2(o1_, o2_) do { const uint32_t a1_ = (o1_), a2_ = (o2_); const uint32_t hi_ = max(k1, a1_); k1 = min(k1, a1_); k2 = min(hi_, min(k2, a2_)); } while (0)
#define CMS_DPP(v_, ctrl_) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v_), (ctrl_), 0xF, 0xF, false))
  { const uint32_t o1 = CMS_DPP(k1, 0xB1), o2 = CMS_DPP(k2, 0xB1); CMS_MERGE2(o1, o2); }        // quad_perm [1, 0, 3, 2]
  { const uint32_t o1 = CMS_DPP(k1, 0x4E), o2 = CMS_DPP(k2, 0x4E); CMS_MERGE2(o1, o2); }        // quad_perm [2, 3, 0, 1]
  { const uint32_t o1 = CMS_DPP(k1, 0x141), o2 = CMS_DPP(k2, 0x141); CMS_MERGE2(o1, o2); }      // row_half_mirror: lane i <-> 7 - i of its group of eight
  { const uint32_t o1 = CMS_DPP(k1, 0x140), o2 = CMS_DPP(k2, 0x140); CMS_MERGE2(o1, o2); }      // row_mirror: lane i <-> 15 - i of its row
  { const uint32_t o1 = (uint32_t)__builtin_amdgcn_ds_swizzle((int)k1, 0x401F), o2 = (uint32_t)__builtin_amdgcn_ds_swizzle((int)k2, 0x401F); CMS_MERGE2(o1, o2); }   // bit mode: and 0x1F, or 0, xor 0x10
  { const uint32_t o1 = (uint32_t)__builtin_amdgcn_readlane((int)k1, 32), o2 = (uint32_t)__builtin_amdgcn_readlane((int)k2, 32); CMS_MERGE2(o1, o2); }             // (only lanes 0 .. 31 merge something new: lane 0 is read)
#undef CMS_MERGE2
#undef CMS_DPP
  if (lane == 0) {
    int bi = -1, bd = 256, bl = -1, sd = 256, sl = -1;
    if (k1 != 0xFFFFFFFFu) {
      bi = cand_idx[start + (int)(k1 & 0x3FFFFFu)];
      bd = (int)(k1 >> 22);
      bl = tlevel ? tlevel[bi] : 0;
    }
    if (k2 != 0xFFFFFFFFu) {
      sd = (int)(k2 >> 22);
      sl = tlevel ? tlevel[cand_idx[start + (int)(k2 & 0x3FFFFFu)]] : 0;
    }
    best_idx[q] = bi; best_dist[q] = bd; second_dist[q] = sd;
    if (best_level) best_level[q] = bl;
    if (second_level) second_level[q] = sl;
  }
}

// dense all-pairs distances (SearchForInitialization / triangulation style windows): out[i*nb + j]
extern "C" __global__ void __launch_bounds__(256)
k_hamming_matrix(const uint4* __restrict__ a, int na, const uint4* __restrict__ b, int nb, uint16_t* __restrict__ out) {
  __shared__ uint4 sa[16][2];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.y * 16, j = blockIdx.x * 16 + tx;
  if (threadIdx.x < 32) {
    const int r = threadIdx.x >> 1, h = threadIdx.x & 1;
    if (i0 + r < na) sa[r][h] = a[2 * (size_t)(i0 + r) + h];
  }
  __syncthreads();
  if (i0 + ty < na && j < nb) {
    const uint4 b0 = b[2 * (size_t)j], b1 = b[2 * (size_t)j + 1];
    out[(size_t)(i0 + ty) * nb + j] = (uint16_t)hamming256(sa[ty][0], sa[ty][1], b0, b1);
  }
}
