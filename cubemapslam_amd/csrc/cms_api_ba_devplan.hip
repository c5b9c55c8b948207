// cms_api_ba_devplan.hip -- the WHOLE plan of a local-BA window on the device (included by cms_api_ba.hip behind cms_api_ba_plan.hip).
//
// cms_api_ba_plan.hip left the host one pass over the observations and ~0.5 ms of per-point work per 80 k-observation window (signature groups,
// runs, internal point order, chunks, the matching of the left-over observations' diagonal copies): 32 windows per 12-ms step of bench.py are half
// of what a rank confined to two host cores can do at all (profiles/r06_bench_runs.txt: 16.8 k frames/s there, 19.7 k with the plans taken from a cache).
// k_ba_plan_many does all of it with ONE workgroup of 1024 threads per window (blockIdx.x = window of a cms_ba_create_many call) and gives, byte for
// byte, the arrays ba_plan_fast + the staging copy would have put on the device (tests/test_gpu_parity.py::test_ba_plan_kernel_equals_the_host_plan):
//
//   pass      per observation: validation; "grouped" = the caller's points never decrease; where a point's observations begin and end.  Then a thread per
//             point ORs its observations' key frames into the point's 64-bit set (grouped order: no atomics; otherwise an atomic OR per observation)
//   groups    open addressing on the set (a slot is read before it is claimed with atomicCAS), per slot the group's size and first point in LDS -> which
//             groups are runs; the run order of the host (first appearance inside a class, class 0 first) is the rank of a run's first point among the candidates'
//   order     a point's ordinal inside its run in the caller's order (the host's `seen` counter): in a tile of 64 points the lanes of the same run find each
//             other with one ballot per bit of the run's number; a wavefront walks a contiguous segment of tiles and keeps 16-bit per-run counters for it in
//             LDS; a prefix over the 16 segments makes them bases.  Left-over points keep the caller's order (a prefix sum over "not inside a run")
//   chunks    run chunks by binary search in the runs' chunk offsets; the left-over points are packed greedily by one thread per segment (<= 8
//             segments, the same cut as the host's), their counts in LDS
//   tables    CSR offsets, chunk first edges, run chunk descriptors, the running chunk cost, the points nobody observes
// ... and k_ba_match_copies_many behind it: per (left-over chunk, 16 lanes) the matching lanes -> LDS banks (BaDiagMatch without recursion, state in nibbles).
// Every loop reads a batch of values before it works on them -- a thread's iteration is otherwise one memory round trip -- and nothing is per-observation
// atomic: the first version (serial probes, an atomic per observation, one round trip per iteration) took 2.4 ms per batch of eight windows, this one 0.36.
//
// What the host still needs -- the counts that size the Levenberg launches -- comes back through 40 bytes of pinned memory per window; the expansion
// kernel (k_ba_expand_edges_many) reads the same counts from device memory, so it is enqueued right behind this kernel without a host round trip.
// Windows this kernel gives up on (status 0: a point seen twice by a key frame, no runs / mostly left-over points, more runs or chunks than its
// tables hold) are planned again by the host (ba_plan_fast or ba_plan); status -1 = an index out of range, reported like the host planners do.
#include <stdint.h>

#define BA_DP_THREADS 1024
#define BA_DP_RUN_CAP 1023            /* runs per window at most: one per thread of the plan kernel with the total behind them (tracked configs[3] windows have ~150) */
#define BA_DP_HCAP 8192               /* slots of the signature table (counts and first points in LDS): more than 4096 different key-frame sets -> the host plans */
#define BA_DP_LCNT_CAP 32768          /* left-over points at most (their observation counts sit in LDS for the greedy packing) */
#define BA_DP_COUNTS 10               /* status, chunks, run chunks, class-0 run chunks, runs, points inside runs, lone points, grouped, first left-over edge */
#define BA_DP_LDS (2 * BA_DP_HCAP * 4 + 3 * 1024 * 4 + 1024)      /* dynamic LDS of the kernel, bytes: the two slot tables, three run tables, 104 small words */

struct BaDevPlan {
  int K, P, E, np, chunk_cap, em_cost_a, em_cost_b, pad_;
  unsigned long long free_mask;
  const int* e_pose; const int* e_point; const int8_t* e_face; const int* pose_slot;
  // scratch (this kernel initialises what it needs)
  int* cnt; unsigned long long* sig; unsigned long long* sig_i; unsigned long long* hkey; int* gslot; int* ord; int* scan; int* cnt_i; int* seg_tmp; int* run_tab; int* chunk_pt0; int* chunk_run;
  // the plan
  int* cpo; int* cedge; int* prank; int* pinv; int* pt_off; uint8_t* pcopy; uint8_t* lo_copy; int* ce0; int4* rm_chunk; uint32_t* rm_cost; unsigned long long* run_sig; int* lone;
  int* counts; int* h_counts;
  long long* clk;                                                   // developer: 100-MHz stamps of the kernel's phases (CMS_BA_DP_CLK=1), else NULL
};
#define BA_DP_BATCH 8
struct BaDevPlanBatch { BaDevPlan x[BA_DP_BATCH]; };

// loads of words other threads changed with atomics: past this CU's vector cache
__device__ __forceinline__ int ba_dp_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ba_dp_ld(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// exclusive prefix sum over the workgroup's threads of one value each (s_part: >= 16 ints of LDS); *total = the sum
__device__ __forceinline__ int ba_dp_block_exscan(int v, int* s_part, int* total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  if (lane == 63) s_part[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < BA_DP_THREADS / 64; ++w) { const int t = s_part[w]; base += w < wave ? t : 0; tot += t; }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}
// exclusive prefix sum of n values by the whole workgroup (a strip of consecutive values per thread, eight of them in flight): out[i] = val(0) + ... + val(i - 1),
// out[n] = the total if asked for; val must not change between its two evaluations (out may be the array val reads: a batch is read before it is written)
template <class F> __device__ __forceinline__ int ba_dp_exscan(int n, F&& val, int* out, bool with_total, int* s_part) {
  const int T = BA_DP_THREADS, tid = threadIdx.x;
  const int S = (n + T - 1) / T;
  const int lo = min(n, tid * S), hi = min(n, lo + S);
  int sum = 0;
  for (int i0 = lo; i0 < hi; i0 += 8) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = i0 + u < hi ? val(i0 + u) : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += v[u];
  }
  int total;
  int run = ba_dp_block_exscan(sum, s_part, &total);
  for (int i0 = lo; i0 < hi; i0 += 8) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = i0 + u < hi ? val(i0 + u) : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) if (i0 + u < hi) { out[i0 + u] = run; run += v[u]; }
  }
  if (with_total && tid == 0) out[n] = total;
  __syncthreads();
  return total;
}

// BaDiagMatch::run (cms_api_ba.hip) for one group of 16 lanes, without recursion and without arrays: the same lanes in the same order, the same choices.
// Lane i's candidate banks are (slot_i + r np) mod 16 for its copies r = 0..3 (BA_SE_DSTRIDE = 33 = 1 mod 16), so a lane is the low four bits of its free-pose
// slot; owners, choices, the path's frames and the banks' loads are nibbles / bytes of 64-bit words (private arrays of a GPU thread live in memory).
static_assert((BA_SE_DSTRIDE & 15) == 1 && BA_SE_DCOPIES == 4, "ba_dp_match: the banks of a lane's copies");
__host__ __device__ __forceinline__ int ba_dp_nib(unsigned long long w, int i) { return (int)((w >> (4 * i)) & 15ull); }
__host__ __device__ __forceinline__ void ba_dp_set_nib(unsigned long long& w, int i, int v) { w = (w & ~(15ull << (4 * i))) | ((unsigned long long)(v & 15) << (4 * i)); }
// slots: nibble i = slot of lane i mod 16; returns the choices (nibble i = copy of lane i)
__host__ __device__ inline unsigned long long ba_dp_match(int n, unsigned long long slots, int np) {
  unsigned long long owner = 0, choice = 0, load_lo = 0, load_hi = 0;      // owner nibbles (valid where `owned` has the bank's bit), choice nibbles, loads as bytes (banks 0-7 | 8-15)
  unsigned owned = 0, chosen = 0;
  auto bank = [&](int i, int r) { return (ba_dp_nib(slots, i) + r * np) & 15; };
  for (int i = 0; i < n; ++i) {
    bool done = false;
    for (int r = 0; r < 4 && !done; ++r) {
      const int c = bank(i, r);
      if (!(owned & (1u << c))) { owned |= 1u << c; ba_dp_set_nib(owner, c, i); ba_dp_set_nib(choice, i, r); chosen |= 1u << i; done = true; }
    }
    if (done) continue;
    // the augmenting path of go(i): frames (lane, copy being tried); owners change only when a free bank has been found.  A frame is pushed per newly seen
    // owned bank, the owners are other lanes: at most 16 frames
    unsigned seen = 0;
    unsigned long long st_i = (unsigned long long)i, st_r = 0;
    int sp = 0;
    for (;;) {
      bool pushed = false, found = false;
      while (ba_dp_nib(st_r, sp) < 4) {
        const int fi = ba_dp_nib(st_i, sp), r = ba_dp_nib(st_r, sp), c = bank(fi, r);
        if (seen & (1u << c)) { ba_dp_set_nib(st_r, sp, r + 1); continue; }
        seen |= 1u << c;
        if (!(owned & (1u << c))) {
          owned |= 1u << c; ba_dp_set_nib(owner, c, fi); ba_dp_set_nib(choice, fi, r); chosen |= 1u << fi;
          while (sp > 0) {
            --sp;
            const int f2 = ba_dp_nib(st_i, sp), r2 = ba_dp_nib(st_r, sp), c2 = bank(f2, r2);
            ba_dp_set_nib(owner, c2, f2); ba_dp_set_nib(choice, f2, r2); chosen |= 1u << f2;
          }
          found = true;
          break;
        }
        ba_dp_set_nib(st_i, sp + 1, ba_dp_nib(owner, c)); ba_dp_set_nib(st_r, sp + 1, 0); ++sp;      // (the frame below stays at copy r until this one has failed)
        pushed = true;
        break;
      }
      if (found) break;
      if (pushed) continue;
      if (sp == 0) break;                                        // go(i) == false
      --sp; ba_dp_set_nib(st_r, sp, ba_dp_nib(st_r, sp) + 1);
    }
  }
  auto load_of = [&](int c) { return (int)(((c < 8 ? load_lo : load_hi) >> (8 * (c & 7))) & 255ull); };
  auto load_inc = [&](int c) { if (c < 8) load_lo += 1ull << (8 * c); else load_hi += 1ull << (8 * (c - 8)); };
  for (int i = 0; i < n; ++i) if (chosen & (1u << i)) load_inc(bank(i, ba_dp_nib(choice, i)));
  for (int i = 0; i < n; ++i)
    if (!(chosen & (1u << i))) {
      int best = 0;
      for (int r = 1; r < 4; ++r) if (load_of(bank(i, r)) < load_of(bank(i, best))) best = r;
      ba_dp_set_nib(choice, i, best); load_inc(bank(i, best));
    }
  return choice;
}

#define BA_DP_STAMP(i) do { if (x.clk && tid == 0) x.clk[i] = (long long)wall_clock64(); } while (0)
#define BA_DP_GIVE_UP(code) do { if (tid == 0) { x.counts[0] = (code); x.h_counts[0] = (code); } return; } while (0)
extern "C" __global__ void __launch_bounds__(BA_DP_THREADS) k_ba_plan_many(BaDevPlanBatch batch) {
  const BaDevPlan& x = batch.x[blockIdx.x];
  const int T = BA_DP_THREADS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = x.K, P = x.P, E = x.E, np = x.np;
  // Every loop below reads a batch of values first and works on them afterwards: a thread's iterations are otherwise one memory round trip (1-2 us) each.
  extern __shared__ int s_dyn[];
  unsigned* s_hcnt = reinterpret_cast<unsigned*>(s_dyn);              // per slot: points of the set -> its run + 1; later the left-over points' counts (bytes)
  unsigned* s_hfirst = s_hcnt + BA_DP_HCAP;                          // per slot: the set's first point; later 16 x 1024 16-bit counters (segment of tiles, run)
  int* s_run_pt0 = reinterpret_cast<int*>(s_hfirst + BA_DP_HCAP);    // 1024
  int* s_run_c0 = s_run_pt0 + 1024;                                  // 1024
  int* s_rq = s_run_c0 + 1024;                                       // 1024: the run candidates' first points (class in bit 30)
  int* s_part = s_rq + 1024;                                         // 16
  int* s_flag = s_part + 16;                                         // 8: bad index, not grouped, give up, chunk too long, different sets, run candidates, points nobody observes
  int* s_seg = s_flag + 8;                                           // 16
  int* s_pslot = s_seg + 16;                                         // 64: free-pose slot of a key frame
  if (tid < 8) s_flag[tid] = 0;
  if (tid < 64) s_pslot[tid] = tid < K ? x.pose_slot[tid] : -1;
  BA_DP_STAMP(0);
  // ---- clear the tables
  int* first_e = x.scan; int* last_e = x.ord;                      // (borrowed until the groups are made)
  for (int p = tid; p < P; p += T) first_e[p] = -1;
  for (int j = tid; j < BA_DP_HCAP; j += T) { x.hkey[j] = 0ull; s_hcnt[j] = 0u; s_hfirst[j] = 0xFFFFFFFFu; }
  __syncthreads();
  BA_DP_STAMP(1);
  // ---- the pass over the observations: validation; "grouped" = the caller's points never decrease; where a point's observations begin and end if they do
  for (int e0 = tid; e0 < E; e0 += 4 * T) {
    int k[4], p[4], q[4], nx[4], f[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * T;
      k[u] = e < E ? x.e_pose[e] : 0; p[u] = e < E ? x.e_point[e] : 0; q[u] = e < E && e > 0 ? x.e_point[e - 1] : -1; nx[u] = e + 1 < E ? x.e_point[e + 1] : -2;
      f[u] = e < E ? (int)x.e_face[e] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * T;
      if (e >= E) continue;
      if ((unsigned)k[u] >= (unsigned)K || (unsigned)p[u] >= (unsigned)P || (unsigned)f[u] > 4u) { s_flag[0] = 1; continue; }
      if (p[u] < q[u]) s_flag[1] = 1;
      if (p[u] != q[u]) first_e[p[u]] = e;
      if (p[u] != nx[u]) last_e[p[u]] = e;
    }
  }
  __syncthreads();
  if (s_flag[0]) BA_DP_GIVE_UP(-1);
  const bool grouped = s_flag[1] == 0;
  BA_DP_STAMP(12);
  // ---- a point's key-frame set and its size.  Grouped (what a host that walks its map points produces): a thread per point ORs the bits of the point's
  // own observations -- no atomics (one atomic OR per observation from one CU is 2.2 ns each: 0.36 ms per 80 k-observation window).  A point seen twice by a
  // key frame has fewer bits than observations: the pair-owner kernel's case, the host plans the window
  if (grouped) {
    for (int p0 = tid; p0 < P; p0 += 4 * T) {
      int fe[4], n[4], kk[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int p = p0 + u * T; fe[u] = p < P ? first_e[p] : -1; n[u] = p < P && fe[u] >= 0 ? last_e[p] - fe[u] + 1 : 0; }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) kk[u][j] = j < n[u] ? x.e_pose[fe[u] + j] : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = p0 + u * T;
        if (p >= P) continue;
        unsigned long long v = 0ull;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j < n[u]) v |= 1ull << kk[u][j];
        if (n[u] > 31) s_flag[2] = 1;
        else for (int j = 8; j < n[u]; ++j) v |= 1ull << x.e_pose[fe[u] + j];
        if (__popcll(v) != n[u]) s_flag[2] = 1;
        if (n[u] == 0) s_flag[6] = 1;
        x.sig[p] = v; x.cnt[p] = n[u];
      }
    }
  } else {
    for (int p = tid; p < P; p += T) x.sig[p] = 0ull;
    __syncthreads();
    for (int e = tid; e < E; e += T) atomicOr(&x.sig[x.e_point[e]], 1ull << x.e_pose[e]);
    __syncthreads();
    for (int p = tid; p < P; p += T) {
      const int c = __popcll(ba_dp_ld(&x.sig[p]));
      x.cnt[p] = c;
      if (c > 31) s_flag[2] = 1;
      if (c == 0) s_flag[6] = 1;
    }
  }
  __syncthreads();
  BA_DP_STAMP(2);
  // ---- offsets of the caller's points (a repeated key frame lost a bit: the sizes then do not sum to E)
  const int tot_obs = ba_dp_exscan(P, [&](int i) { return x.cnt[i]; }, x.cpo, true, s_part);
  if (s_flag[2] || tot_obs != E) BA_DP_GIVE_UP(0);               // (a point seen twice by a key frame: the pair-owner kernel's case)
  if (!grouped) {                                                 // the caller's edges grouped by point (the order inside a point does not matter: the expansion ranks by key frame)
    for (int p = tid; p < P; p += T) x.scan[p] = 0;
    __syncthreads();
    for (int e = tid; e < E; e += T) {
      const int p = x.e_point[e];
      x.cedge[x.cpo[p] + atomicAdd(&x.scan[p], 1)] = e;
    }
    __syncthreads();
  }
  BA_DP_STAMP(3);
  // ---- signature groups: open addressing on the set (a slot is read before it is claimed: most points find their set there), size and first point in LDS
  for (int p0 = tid; p0 < P; p0 += 4 * T) {
    unsigned long long key[4], cur[4]; int j[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * T;
      key[u] = p < P ? ba_dp_ld(&x.sig[p]) : 0ull;
      j[u] = (int)((key[u] * 0x9E3779B97F4A7C15ull) >> 40) & (BA_DP_HCAP - 1);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) cur[u] = key[u] != 0ull ? ba_dp_ld(&x.hkey[j[u]]) : 0ull;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * T;
      if (p >= P) continue;
      if (key[u] == 0ull) { x.gslot[p] = -1; continue; }          // (nobody observes the point: never inside a run)
      int jj = j[u];
      unsigned long long c = cur[u];
      bool ok = false;
      for (int step = 0; step < BA_DP_HCAP; ++step) {
        if (c == 0ull) {
          c = atomicCAS(&x.hkey[jj], 0ull, key[u]);
          if (c == 0ull) { if (atomicAdd(&s_flag[4], 1) + 1 > BA_DP_HCAP / 2) s_flag[2] = 1; c = key[u]; }
        }
        if (c == key[u]) { ok = true; break; }
        jj = (jj + 1) & (BA_DP_HCAP - 1);
        c = ba_dp_ld(&x.hkey[jj]);
      }
      if (!ok) { s_flag[2] = 1; continue; }                        // (the table is full)
      atomicAdd(&s_hcnt[jj], 1u);
      atomicMin(&s_hfirst[jj], (unsigned)p);
      x.gslot[p] = jj;
    }
  }
  __syncthreads();
  if (s_flag[2]) BA_DP_GIVE_UP(0);                                // more different sets than the table is made for
  BA_DP_STAMP(4);
  // ---- which groups are runs; their order = first appearance inside a class, class 0 first (the host's): the rank of a run's first point among the run
  // candidates' first points.  A thread keeps its eight slots' decisions in registers across the ranking.
  int sk[8], skf[8], sg_[8], sq[8];
  {
    unsigned long long key[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = tid + u * T;
      sg_[u] = (int)s_hcnt[j]; sq[u] = (int)s_hfirst[j];
      key[u] = sg_[u] > 0 ? ba_dp_ld(&x.hkey[j]) : 0ull;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { sk[u] = sg_[u] > 0 ? x.cnt[sq[u]] : 0; skf[u] = __popcll(key[u] & x.free_mask); }
  }
  __syncthreads();                                                // (every thread has read its slots' counts: s_hcnt becomes "run + 1")
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int j = tid + u * T, k = sk[u], kf = skf[u];
    bool run = sg_[u] > 0 && !(k < 1 || k > 9 || kf < 1 || kf * (kf + 1) / 2 > 64 || 6 * kf + 1 > 48);
    if (run && sg_[u] < min(64 / k, BA_RM_PTS)) run = false;      // at least one full chunk
    if (run) {
      const int i = atomicAdd(&s_flag[5], 1);
      if (i < 1024) s_rq[i] = sq[u] | (ba_rw_class(kf) << 30);
    } else sk[u] = 0;                                             // (sk == 0: not a run)
    s_hcnt[j] = 0u;
  }
  __syncthreads();
  const int n_runs = s_flag[5];
  if (n_runs == 0 || n_runs > BA_DP_RUN_CAP) BA_DP_GIVE_UP(0);
  int nA = 0;
  for (int i = 0; i < n_runs; ++i) nA += (s_rq[i] >> 30) == 0 ? 1 : 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (sk[u] == 0) continue;
    const int j = tid + u * T, k = sk[u], kf = skf[u], g = sg_[u], q = sq[u], cls = ba_rw_class(kf);
    int r = cls == 0 ? 0 : nA;
    for (int i = 0; i < n_runs; ++i) { const int w = s_rq[i]; r += ((w >> 30) == cls && (w & 0x3FFFFFFF) < q) ? 1 : 0; }
    const int m = min(64 / k, BA_RM_PTS), full = g / m, tail = g - full * m;
    const bool keep_tail = tail > 0 && 2 * tail >= m;
    int* rt = x.run_tab + 8 * (size_t)r;
    rt[0] = j; rt[1] = q; rt[2] = k; rt[3] = kf; rt[4] = m; rt[5] = full * m + (keep_tail ? tail : 0); rt[6] = full + (keep_tail ? 1 : 0); rt[7] = 0;
    x.run_sig[r] = ba_dp_ld(&x.hkey[j]);
    s_hcnt[j] = (unsigned)(r + 1);
  }
  __syncthreads();
  // first internal point and first chunk of every run (<= 1023 runs: one value per thread)
  for (int half = 0; half < 2; ++half) {
    const int v = tid < n_runs ? x.run_tab[8 * (size_t)tid + (half == 0 ? 5 : 6)] : 0;
    int total;
    const int ex = ba_dp_block_exscan(v, s_part, &total);
    if (tid <= n_runs) (half == 0 ? s_run_pt0 : s_run_c0)[tid] = ex;
    __syncthreads();
  }
  const int P_rm = s_run_pt0[n_runs], n_rm = s_run_c0[n_runs], n_rmA = s_run_c0[nA], PL = P - P_rm;
  if (P_rm == 0 || 3 * PL > P || PL > BA_DP_LCNT_CAP || n_rm > x.chunk_cap) BA_DP_GIVE_UP(0);      // no runs, or mostly left-over points: the look-ahead composition pays there (ba_plan)
  BA_DP_STAMP(5);
  // ---- a point's ordinal inside its run group, caller's order (the host's `seen` counter).  Tiles of 64 points; a wavefront takes a contiguous SEGMENT of
  // tiles in order and keeps per run how many of its points it has met (16 x 1024 16-bit counters in LDS): inside a tile the lanes of the same run find each
  // other with one ballot per bit of the run's number, the set's first lane adds the count.  A prefix over the 16 segments per run then makes the segment counts bases.
  unsigned short* s_segc = reinterpret_cast<unsigned short*>(s_hfirst);
  for (int i = tid; i < 16 * 1024 / 2; i += T) s_hfirst[i] = 0u;
  const int ntiles = (P + 63) / 64, tps = (ntiles + T / 64 - 1) / (T / 64);
  __syncthreads();
  {
    unsigned short* my = s_segc + 1024 * wave;
    const int t_lo = min(ntiles, wave * tps), t_hi = min(ntiles, t_lo + tps);
    int gs_nx = t_lo < t_hi && 64 * t_lo + lane < P ? x.gslot[64 * t_lo + lane] : -1;
    for (int t = t_lo; t < t_hi; ++t) {
      const int p = 64 * t + lane, gs = gs_nx;
      if (t + 1 < t_hi) gs_nx = 64 * (t + 1) + lane < P ? x.gslot[64 * (t + 1) + lane] : -1;
      const int r = gs >= 0 ? (int)s_hcnt[gs] - 1 : -1;
      unsigned long long eq = __ballot(r >= 0);                    // the lanes whose point is in the same run: one ballot per bit of the run's number
#pragma unroll
      for (int bit = 0; bit < 10; ++bit) { const unsigned long long bm = __ballot((r >> bit) & 1); eq &= ((r >> bit) & 1) ? bm : ~bm; }
      const int rank = __popcll(eq & ((1ull << lane) - 1ull)), cnt = __popcll(eq);
      const int lead = r >= 0 ? __ffsll((long long)eq) - 1 : lane;
      int base = 0;
      if (r >= 0 && lead == lane) { base = my[r]; my[r] = (unsigned short)(base + cnt); }      // the sets' first lanes, all at once (different runs: different counters)
      base = __shfl(base, lead);
      const int o = r >= 0 ? base + rank : -1;
      if (p < P) x.ord[p] = o;                                    // ordinal inside (segment, run), -1: not a run group's point
    }
  }
  __syncthreads();
  if (tid < n_runs) {
    int acc = 0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) { const int c = s_segc[1024 * w + tid]; s_segc[1024 * w + tid] = (unsigned short)acc; acc += c; }
  }
  __syncthreads();
  BA_DP_STAMP(6);
  // ---- internal point order: the runs' points (run after run, caller's order inside a run, `take` of them), then the left-over points in the caller's order
  for (int p0 = tid; p0 < P; p0 += 4 * T) {
    int o[4], gs[4], r[4], take[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int p = p0 + u * T; o[u] = p < P ? x.ord[p] : -1; gs[u] = p < P ? x.gslot[p] : -1; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { r[u] = o[u] >= 0 ? (int)s_hcnt[gs[u]] - 1 : -1; take[u] = r[u] >= 0 ? x.run_tab[8 * (size_t)r[u] + 5] : 0; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * T;
      if (p >= P) continue;
      int w = -1;
      if (r[u] >= 0) { const int oo = o[u] + s_segc[1024 * ((p >> 6) / tps) + r[u]]; if (oo < take[u]) w = oo | (r[u] << 16); }
      x.ord[p] = w;                                               // ordinal | run << 16, -1: a left-over point
    }
  }
  __syncthreads();
  ba_dp_exscan(P, [&](int i) { return x.ord[i] < 0 ? 1 : 0; }, x.scan, false, s_part);
  for (int p0 = tid; p0 < P; p0 += 4 * T) {
    int w[4], lo[4], c[4], m[4]; unsigned long long sg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * T;
      w[u] = p < P ? x.ord[p] : -1; lo[u] = p < P ? x.scan[p] : 0; c[u] = p < P ? x.cnt[p] : 0; sg[u] = p < P ? ba_dp_ld(&x.sig[p]) : 0ull;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) m[u] = w[u] >= 0 ? x.run_tab[8 * (size_t)(w[u] >> 16) + 4] : 1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * T;
      if (p >= P) continue;
      int ip; uint8_t copy = 0xFF;
      if (w[u] >= 0) { const int o = w[u] & 0xFFFF; ip = s_run_pt0[w[u] >> 16] + o; copy = (uint8_t)((o % m[u]) & (BA_SE_DCOPIES - 1)); }
      else ip = P_rm + lo[u];
      x.prank[p] = ip; x.pinv[ip] = p; x.pcopy[ip] = copy; x.cnt_i[ip] = c[u]; x.sig_i[ip] = sg[u];
    }
  }
  __syncthreads();
  ba_dp_exscan(P, [&](int i) { return x.cnt_i[i]; }, x.pt_off, true, s_part);
  BA_DP_STAMP(7);
  // ---- chunks: the runs' (binary search in the runs' first chunks), then the left-over points packed greedily, one thread per segment (the host's segments)
  for (int c = tid; c < n_rm; c += T) {
    int lo = 0, hi = n_runs;                                      // the last run whose first chunk is <= c
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_run_c0[mid] <= c) lo = mid; else hi = mid; }
    x.chunk_run[c] = lo;
    x.chunk_pt0[c] = s_run_pt0[lo] + (c - s_run_c0[lo]) * x.run_tab[8 * (size_t)lo + 4];
  }
  uint8_t* s_lcnt = reinterpret_cast<uint8_t*>(s_hcnt);          // (the slots' runs are through)
  for (int i0 = tid; i0 < PL; i0 += 4 * T) {
    int c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = i0 + u * T < PL ? x.cnt_i[P_rm + i0 + u * T] : 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i0 + u * T < PL) s_lcnt[i0 + u * T] = (uint8_t)c[u];
  }
  const int nseg = max(1, min(8, PL / 512));
  __syncthreads();
  if (tid < nseg) {
    const int pb = (int)((long long)PL * tid / nseg), pe = (int)((long long)PL * (tid + 1) / nseg);
    int n = 0;
    if (pe > pb) {
      x.seg_tmp[pb + n++] = P_rm + pb;
      int cur = 0;
      int i = pb;
      for (; i < pe && (i & 3); ++i) {
        const int k = s_lcnt[i];
        if (cur + k > 64 && cur > 0) { x.seg_tmp[pb + n++] = P_rm + i; cur = 0; }
        cur += k;
      }
      for (; i + 4 <= pe; i += 4) {                               // (four counts per LDS read)
        const unsigned w4 = *reinterpret_cast<const unsigned*>(s_lcnt + i);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int k = (int)((w4 >> (8 * v)) & 255u);
          if (cur + k > 64 && cur > 0) { x.seg_tmp[pb + n++] = P_rm + i + v; cur = 0; }
          cur += k;
        }
      }
      for (; i < pe; ++i) {
        const int k = s_lcnt[i];
        if (cur + k > 64 && cur > 0) { x.seg_tmp[pb + n++] = P_rm + i; cur = 0; }
        cur += k;
      }
    }
    s_seg[tid] = n;
  }
  __syncthreads();
  int nchunks = n_rm;
  for (int t = 0; t < nseg; ++t) nchunks += s_seg[t];
  if (nchunks > x.chunk_cap) BA_DP_GIVE_UP(0);
  {
    int off = n_rm;
    for (int t = 0; t < nseg; ++t) {
      const int pb = (int)((long long)PL * t / nseg), n = s_seg[t];
      for (int j = tid; j < n; j += T) x.chunk_pt0[off + j] = x.seg_tmp[pb + j];
      off += n;
    }
    if (tid == 0) x.chunk_pt0[nchunks] = P;
  }
  __syncthreads();
  BA_DP_STAMP(8);
  // ---- first edges, run chunk descriptors, the running cost
  const int e_lo0 = x.pt_off[P_rm];
  for (int c = tid; c <= nchunks; c += T) x.ce0[c] = x.pt_off[x.chunk_pt0[c]];
  for (int i = tid; 4 * i < E - e_lo0; i += T) reinterpret_cast<uint32_t*>(x.lo_copy)[i] = 0u;      // (E bytes rounded up to the allocation's granule)
  __syncthreads();
  for (int c = tid; c < nchunks; c += T) {
    const int p0 = x.chunk_pt0[c], pn = x.chunk_pt0[c + 1];
    if (x.ce0[c + 1] - x.ce0[c] > 64) s_flag[3] = 1;              // (cannot happen with <= 31 observations per point)
    int cost;
    if (c < n_rm) {
      const int r = x.chunk_run[c];
      const int* rt = x.run_tab + 8 * (size_t)r;
      const int p1 = min(pn, s_run_pt0[r + 1]);
      x.rm_chunk[c] = make_int4(x.pt_off[p0], (x.pt_off[p1] - x.pt_off[p0]) | (rt[2] << 8) | ((p1 - p0) << 16), r, p0);
      cost = (int)ba_rm_chunk_cost(rt[2], rt[3], p1 - p0);
    } else {
      int kmax = 1;
      for (int p = p0; p < pn; ++p) kmax = max(kmax, (int)s_lcnt[p - P_rm]);
      cost = x.em_cost_a + x.em_cost_b * (kmax / 2);
    }
    x.chunk_run[c] = cost;                                         // (the chunk's run is not needed any more)
  }
  __syncthreads();
  ba_dp_exscan(nchunks, [&](int i) { return x.chunk_run[i]; }, reinterpret_cast<int*>(x.rm_cost), true, s_part);
  if (s_flag[3]) BA_DP_GIVE_UP(0);
  BA_DP_STAMP(9);
  // (the left-over observations' copies of their key frames' diagonal blocks: k_ba_match_copies_many, behind this kernel)
  BA_DP_STAMP(10);
  // ---- points nobody observes (internal order)
  int nlone = 0;
  if (s_flag[6]) {
    nlone = ba_dp_exscan(P, [&](int i) { return x.cnt_i[i] == 0 ? 1 : 0; }, x.scan, false, s_part);
    for (int p0 = tid; p0 < P; p0 += 8 * T) {
      int c[8], w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int p = p0 + u * T; c[u] = p < P ? x.cnt_i[p] : 1; w[u] = p < P ? x.scan[p] : 0; }
#pragma unroll
      for (int u = 0; u < 8; ++u) if (c[u] == 0) x.lone[w[u]] = p0 + u * T;
    }
  }
  BA_DP_STAMP(11);
  if (tid == 0) {
    if (nlone == 0) x.lone[0] = 0;
    const int out[BA_DP_COUNTS] = {1, nchunks, n_rm, n_rmA, n_runs, P_rm, nlone, grouped ? 1 : 0, e_lo0, 0};
    for (int i = 1; i < BA_DP_COUNTS; ++i) { x.counts[i] = out[i]; x.h_counts[i] = out[i]; }
    __threadfence_system();
    x.counts[0] = 1; x.h_counts[0] = 1;
  }
}

// The left-over observations' copies of their key frames' diagonal blocks: per chunk and group of 16 lanes a matching lanes -> LDS banks (every lane may use
// any of the BA_SE_DCOPIES copies = banks; a point's observations in ascending key-frame order are the set bits of its signature).  A kernel of its own behind
// k_ba_plan_many: the matchings are sequential searches whose lengths differ from lane to lane -- inside the one-workgroup plan kernel they were 0.25 of its
// 0.6 ms; here every wavefront runs 16 of them on a SIMD of its own.  blockIdx.y = window of the batch.
#define BA_DP_MATCH_BLOCKS 96
extern "C" __global__ void __launch_bounds__(64) k_ba_match_copies_many(BaDevPlanBatch batch) {
  const BaDevPlan& x = batch.x[blockIdx.y];
  if (x.counts[0] != 1) return;                                   // (the plan kernel gave the window up)
  const int nchunks = x.counts[1], n_rm = x.counts[2], e_lo0 = x.counts[8], np = x.np;
  __shared__ int s_pslot[64];
  s_pslot[threadIdx.x] = (int)threadIdx.x < x.K ? x.pose_slot[threadIdx.x] : -1;
  __syncthreads();
  const int lanes = x.pad_ > 0 ? x.pad_ : 16;                       // matchings per wavefront
  if ((int)threadIdx.x >= lanes) return;
  for (int u = blockIdx.x * lanes + threadIdx.x; u < 4 * (nchunks - n_rm); u += lanes * gridDim.x) {
    const int c = n_rm + (u >> 2), g = u & 3;
    const int p0 = x.chunk_pt0[c], pn = x.chunk_pt0[c + 1], e0 = x.ce0[c], ne = x.ce0[c + 1] - e0;
    if (16 * g >= ne) continue;
    int lo = p0, hi = pn;                                          // the point that holds lane 16 g: the last one whose first observation is <= that lane
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (x.pt_off[mid] - e0 <= 16 * g) lo = mid; else hi = mid; }
    int skip = 16 * g - (x.pt_off[lo] - e0);                      // observations of that point in front of the group
    unsigned long long slots = 0, lane_of = 0;
    int n = 0, L = 16 * g;
    const int Lend = min(16 * g + 16, ne);
    for (int pb = lo; pb < pn && L < Lend; pb += 8) {
      unsigned long long bits8[8];
#pragma unroll
      for (int v = 0; v < 8; ++v) bits8[v] = pb + v < pn ? x.sig_i[pb + v] : 0ull;
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        unsigned long long bits = bits8[v];
        for (; skip > 0 && bits; --skip) bits &= bits - 1;
        while (bits && L < Lend) {
          const int k = __ffsll((long long)bits) - 1;
          bits &= bits - 1;
          const int sl = s_pslot[k];
          if (sl >= 0) { ba_dp_set_nib(slots, n, sl); ba_dp_set_nib(lane_of, n, L); ++n; }
          ++L;
        }
      }
    }
    if (n == 0) continue;
    const unsigned long long choice = ba_dp_match(n, slots, np);
    for (int i = 0; i < n; ++i) x.lo_copy[(size_t)(e0 - e_lo0) + 16 * g + ba_dp_nib(lane_of, i)] = (uint8_t)ba_dp_nib(choice, i);
  }
}

// ---- host side
// May k_ba_plan_many plan this window?  Everything ba_plan_fast decides before it looks at the observations; sets the sizes that follow from the
// window's dimensions like ba_plan_fast does.  (CMS_BA_NO_DEV_PLAN=1: never -- the A/B switch.)
static bool ba_dev_plan_prepare(cms_ba* b, int K, const uint8_t* fixed, int P, int E, std::vector<int>& pose_slot, unsigned long long& free_mask) {
  const BaKnobs& kn = ba_knobs();
  static const bool off = getenv("CMS_BA_NO_DEV_PLAN") != nullptr;
  if (off || K > 64 || b->det_points || !ba_fast_plan_allowed(kn) || ba_want_rw_tables() || P > 65535) return false;      // (ordinals and segment counters are 16-bit)
  pose_slot.assign(K, -1);
  int np = 0;
  free_mask = 0;
  for (int k = 0; k < K; ++k) if (!fixed[k]) { pose_slot[k] = np++; free_mask |= 1ull << k; }
  ba_plan_sizes(b, K, P, E, np);
  const size_t se_fixed_lds = ba_se_fixed_lds(K, np), se_wave_lds = (size_t)64 * 18 * sizeof(double) + 64 * sizeof(int);
  int se_nw = BA_SE_THREADS / 64;
  while (se_nw > 2 && se_fixed_lds + se_nw * se_wave_lds > BA_LDS_CEILING) se_nw -= 2;
  const size_t rm_lds = se_fixed_lds + (size_t)BA_RM_PAIRS * 2 * BA_RM_BUF * sizeof(double);
  if (!(np >= 1 && se_fixed_lds + se_nw * se_wave_lds <= BA_LDS_CEILING && np <= 62 && b->solve_blk && b->solve_blk3 && rm_lds <= BA_LDS_CEILING)) return false;
  b->se_lds_fixed = se_fixed_lds; b->se_waves = se_nw; b->rm_lds = rm_lds;
  b->se.npairs2 = np * (np + 1) / 2;
  return true;
}
// ... and once the kernel is through: its counts -> the window.  1: planned; 0: the kernel gave the window up; -1: an index out of range
static int ba_dev_plan_finish(cms_ba* b) {
  const int* c = b->h_plan_counts;
  if (!c || c[0] != 1) return c && c[0] < 0 ? -1 : 0;
  const BaKnobs& kn = ba_knobs();
  BaSe& se = b->se;
  se.nchunks = c[1]; se.n_rm = c[2]; se.n_rmA = c[3]; se.npairs2 = b->np * (b->np + 1) / 2;
  se.cpw_t = (BA_TE_THREADS / 64) * kn.te_chunks;
  se.Rt = (se.nchunks + se.cpw_t - 1) / se.cpw_t;
  se.nlone = c[6];
  ba_se_split(se, BA_SE_RANGES);
  b->n_runs = c[4]; b->rm_points = c[5];
  if (b->d_plan_clk) {                                              // developer: the plan kernel's phases in microseconds
    long long t[13];
    if (hipMemcpy(t, b->d_plan_clk, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess) {
      fprintf(stderr, "[k_ba_plan_many] P %d E %d runs %d chunks %d us:", b->P, b->E, c[4], c[1]);
      static const char* nm[11] = {"clear", "pass", "cpo", "groups", "runs", "tiles", "order", "chunks", "tables", "-", "lone"};
      for (int i = 0; i < 11; ++i) fprintf(stderr, " %s %.1f", nm[i], (t[i + 1] - t[i]) / 100.0);
      fprintf(stderr, " | total %.1f (validation %.1f)\n", (t[11] - t[0]) / 100.0, (t[12] - t[1]) / 100.0);
    }
  }
  return 1;
}
static_assert(sizeof(BaDevPlanBatch) <= 4096, "k_ba_plan_many: the batch must fit the kernel arguments");
// the kernel's dynamic LDS is above the default ceiling: raised once per device
static hipError_t ba_dev_plan_attr_once(int device) {
  static std::mutex mu;
  static bool done[64] = {false};
  std::lock_guard<std::mutex> lk(mu);
  if (device < 0 || device >= 64 || done[device]) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute((const void*)k_ba_plan_many, hipFuncAttributeMaxDynamicSharedMemorySize, BA_DP_LDS);
  if (e == hipSuccess) done[device] = true;
  return e;
}

// developer / test entry, host only (no device needed): one group's matching by the plan path's register-only search (which = 1) or by the host planners'
// BaDiagMatch (which = 0).  slots[n]: the lanes' free-pose slots (n <= 16), np: free key frames; choice_out[n]: the copy every lane got
extern "C" int cms_ba_debug_match(int which, int n, const int* slots, int np, int* choice_out) {
  if (n < 0 || n > 16 || np < 1 || !slots || !choice_out) return cms_fail(CMS_ERR_ARG, "cms_ba_debug_match: bad argument");
  if (which) {
    unsigned long long w = 0;
    for (int i = 0; i < n; ++i) ba_dp_set_nib(w, i, slots[i]);
    const unsigned long long c = ba_dp_match(n, w, np);
    for (int i = 0; i < n; ++i) choice_out[i] = ba_dp_nib(c, i);
  } else {
    BaDiagMatch m;
    m.n = n;
    for (int i = 0; i < n; ++i)
      for (int r = 0; r < BA_SE_DCOPIES; ++r) m.bank[i][r] = (uint8_t)((BA_SE_DSTRIDE * (r * np + slots[i])) & 15);
    m.run();
    for (int i = 0; i < n; ++i) choice_out[i] = m.choice[i];
  }
  return CMS_OK;
}
