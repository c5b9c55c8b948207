// cms_api_ba.hip -- host driver of the device-resident local bundle adjustment (included by cms_lib.hip).
// Control flow restated from OptimizationAlgorithmLevenberg::solve (optimization_algorithm_levenberg.cpp:61-164),
// SparseOptimizer::optimize (sparse_optimizer.cpp:354-419) and the two-stage schedule of
// Optimizer::LocalBundleAdjustment (Optimizer.cpp:359-412).  All vectors and matrices live on the device; per
// Levenberg trial the host reads back three scalars (new chi2, gain denominator, LDL^T status).
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <atomic>
#include <cmath>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <pthread.h>
#include <unordered_map>

// Every A/B switch of the local-BA host code, read ONCE (first use) so that the choices made when a window is built (which work lists exist)
// and the choices made when it is optimised (which kernels run) can never disagree.  Tests that flip a switch use a child process.
struct BaKnobs {
  bool deterministic, det_points, host_lm, single_host_lm, schur_chunks, schur_points, all_lists, want_all_lists;
  bool no_fused, solve1, trial_points, fixed_ranges, no_permute, create_timing, compose_timing, runs, runs_as_edges, separate_reduce, separate_reduce2, separate_first_pass, rm_valu;
  int solve_reduce_max, leftover_lookahead;
  bool global_sum;
  int lookahead, compose_segments, dup, run_min_chunks, rm_weight, se_waves_cap, em_cost_a, em_cost_b, te_chunks, run_min_pct; bool unified;
  char stream_priority;
  bool run_wg; int rw_waves_cu0, rw_waves_cu1;
};
static const BaKnobs& ba_knobs() {
  static const BaKnobs k = [] {
    BaKnobs q;
    auto on = [](const char* n) { return getenv(n) != nullptr; };
    auto num = [](const char* n, int dflt) { const char* v = getenv(n); return v ? atoi(v) : dflt; };
    q.deterministic = on("CMS_BA_DETERMINISTIC"); q.host_lm = on("CMS_BA_HOST_LM"); q.single_host_lm = on("CMS_BA_SINGLE_HOST_LM");
    q.schur_chunks = on("CMS_BA_SCHUR_CHUNKS"); q.schur_points = on("CMS_BA_SCHUR_POINTS"); q.all_lists = on("CMS_BA_ALL_LISTS");
    // the pair-owner / tuple-chunk kernels' work lists and the stored 6x3 blocks: only built when a switch selects those kernels
    // CMS_BA_DET_POINTS=1: deterministic windows on the pair-owner kernel with the host plan and every work list (rounds 3-5's deterministic path; the
    // default since round 6 is the fused chain with its LDS additions in a fixed order, kb_ba_lin_schur_runs_det / _edges_det)
    q.det_points = on("CMS_BA_DET_POINTS");
    q.want_all_lists = (q.deterministic && q.det_points) || q.host_lm || q.single_host_lm || q.schur_chunks || q.schur_points || q.all_lists;
    q.no_fused = on("CMS_BA_NO_FUSED_LIN"); q.solve1 = on("CMS_BA_SOLVE1"); q.trial_points = on("CMS_BA_TRIAL_POINTS");
    q.fixed_ranges = on("CMS_BA_FIXED_RANGES"); q.no_permute = on("CMS_BA_NO_PERMUTE");
    q.create_timing = on("CMS_BA_CREATE_TIMING"); q.compose_timing = on("CMS_BA_COMPOSE_TIMING");
    q.runs = !on("CMS_BA_NO_RUNS");                      // signature runs (cms_ba_schur_runs.hip); off = every point through the edge-major kernel
    q.runs_as_edges = on("CMS_BA_RUNS_AS_EDGES");        // keep the run order of the points but let the edge-major body take the run chunks too
    q.rm_valu = on("CMS_BA_RM_VALU");                    // the runs' tuple products on the vector ALU (producer / consumer pairs) instead of MFMA tiles
    q.separate_first_pass = on("CMS_BA_SEPARATE_FIRST_PASS");      // kb_ba_errors / reduce / lin / maxdiag in front of a stage's first trial instead of kb_ba_first_pass (developer A/B)
    q.separate_reduce2 = on("CMS_BA_SEPARATE_REDUCE2");  // kb_ba_reduce2 as its own launch behind the trial kernel (developer A/B)
    q.separate_reduce = on("CMS_BA_SEPARATE_REDUCE");    // kb_ba_schur_edges_reduce as its own launch instead of inside the solve kernel
    // ... which only pays while a window has few slices to sum (one workgroup reads them all): with more than this many the sum stays a launch
    { const char* v = getenv("CMS_BA_SOLVE_REDUCE_MAX"); q.solve_reduce_max = v ? atoi(v) : 24; }
    // the Schur kernel's workgroups add their copies of the reduced system to ONE global copy (FP64 atomics) instead of writing a slice each that
    // somebody has to sum: tools/probe/global_atomics.hip -- 256 workgroups x 6512 additions cost 10 us against 4 us for the stores
    q.global_sum = !on("CMS_BA_NO_GLOBAL_SUM");
    q.leftover_lookahead = num("CMS_BA_LEFTOVER_LOOKAHEAD", 0);      // 0: automatic (see ba_plan)
    q.lookahead = num("CMS_BA_LOOKAHEAD", 24); q.compose_segments = num("CMS_BA_COMPOSE_SEGMENTS", 0); q.dup = num("CMS_BA_DUP", 0);
    q.run_min_chunks = std::max(1, num("CMS_BA_RUN_MIN_CHUNKS", q.rm_valu ? 2 : 1));
    q.run_min_pct = std::max(10, std::min(100, num("CMS_BA_RUN_MIN_PCT", 100)));      // ... or this share of one chunk (a signature with half a chunk of points as a one-chunk run)
    q.rm_weight = std::max(10, std::min(400, num("CMS_BA_RM_WEIGHT", 58)));
    q.unified = !on("CMS_BA_SPLIT_WORKGROUPS");          // 1: every Schur workgroup of a window takes run chunks and left-over chunks (cut by cost); 0: separate workgroups
    q.em_cost_a = num("CMS_BA_EM_COST_A", 40); q.em_cost_b = num("CMS_BA_EM_COST_B", 30);      // cost of a left-over chunk: a + b x (edges of its largest point / 2), units of ba_rm_chunk_cost
    q.te_chunks = std::max(1, std::min(16, num("CMS_BA_TE_CHUNKS", 2)));      // chunks per wavefront of kb_ba_trial_edges
    q.se_waves_cap = num("CMS_BA_SE_WAVES", 0);          // developer A/B: fewer wavefronts per Schur workgroup (how much of the kernel is latency?)
    // CMS_BA_RUN_WG=1: the runs through one-wavefront workgroups that add to the global copy (cms_ba_schur_runwg.hip), the left-over chunks through
    // kb_ba_lin_schur_edges behind them -- round 6's re-decomposition of kb_ba_lin_schur_runs, parity-green and SLOWER (the kernel is bound
    // by FP64 issue, not by latency; 160 us against 90 for 16 windows): an opt-in experiment.  Units per launch = wavefronts per CU x CUs / windows
    q.run_wg = on("CMS_BA_RUN_WG");
    q.rw_waves_cu0 = std::max(1, std::min(32, num("CMS_BA_RW_WAVES0", 12))); q.rw_waves_cu1 = std::max(1, std::min(32, num("CMS_BA_RW_WAVES1", 8)));
    const char* pr = getenv("CMS_BA_STREAM_PRIORITY");
    q.stream_priority = pr ? pr[0] : 0;
    return q;
  }();
  return k;
}

struct BaExpand;
static thread_local BaExpand* ba_tl_defer_expand = nullptr;      // cms_ba_create_many: where cms_ba_create leaves the description of a device-planned window's expansion instead of launching it
struct BaDevPlan;
static thread_local BaDevPlan* ba_tl_dev_plan = nullptr;      // cms_ba_create_many, CMS_BA_PLAN_ON_DEVICE: where cms_ba_create leaves the description of the window's plan kernel (taken: set to nullptr)
static thread_local hipStream_t ba_tl_setup_stream = nullptr;      // cms_ba_create_many: the stream the calling thread's windows of this call are set up on (ba_setup_stream_take)
static thread_local bool ba_tl_inputs_pinned = false;      // cms_ba_create_many, CMS_BA_INPUTS_PINNED: the caller's arrays are pinned and stay alive -- they are copied from where they lie
static thread_local bool ba_force_rw_tables = false;      // cms_ba_debug_run_fg: build the one-wavefront workgroups' tables whatever the knob says
static inline bool ba_want_rw_tables() { return ba_knobs().run_wg || ba_force_rw_tables; }
struct BaBlock { void* p; size_t bytes; };      // a device slab / pinned block of the per-device pool (below)
struct cms_ba {
  int device = 0;
  hipStream_t stream = nullptr; bool own_stream = true, pooled_stream = false;
  int K = 0, P = 0, E = 0, np = 0, nblk_e = 0, nblk_p = 0;
  std::vector<int> perm;       // sorted position -> caller's edge index
  std::vector<int> pinv;       // internal point id -> caller's point index (points are permuted into collision-free chunks, see below)
  std::vector<int> se_chunk_pt0;   // first internal point of every chunk of the edge-major Schur kernel
  BaDev d;
  // device memory
  uint8_t* d_fixed = nullptr; int* d_pose_slot = nullptr; int* d_e_pose = nullptr; int* d_e_point = nullptr;
  double* d_e_obs = nullptr; double* d_e_inv = nullptr; int8_t* d_e_face = nullptr; int* d_pt_off = nullptr;
  double* d_raw_obs = nullptr; double* d_raw_inv = nullptr; double* d_raw_pts = nullptr; int* d_perm = nullptr; int* d_pinv = nullptr;   // the caller's arrays and orders (k_ba_gather)
  int* d_pose_off = nullptr; int* d_pose_edges = nullptr; uint8_t* d_level = nullptr; double* d_err = nullptr; double* d_ow = nullptr;
  double* d_poses[2] = {nullptr, nullptr}; double* d_pts[2] = {nullptr, nullptr};
  double* d_poses0 = nullptr; double* d_pts0 = nullptr;
  double* d_Hpp = nullptr; double* d_bp = nullptr; double* d_Hll = nullptr; double* d_bl = nullptr; double* d_Hpl = nullptr;
  double* d_Dinv = nullptr; double* d_Hs = nullptr; double* d_bs = nullptr; double* d_x = nullptr; double* d_Dg = nullptr;
  double* d_partial = nullptr; double* d_scal = nullptr; int* d_status = nullptr; uint8_t* d_flags = nullptr;
  double* d_pose_partial = nullptr; double* d_db = nullptr;
  int* d_pair_s1 = nullptr; int* d_pair_s2 = nullptr; int* d_pair_off = nullptr; int2* d_tup = nullptr;
  int* d_pair_chunk_off = nullptr; int2* d_chunk_range = nullptr; double* d_chunk_sum = nullptr; int* d_pair_of_block = nullptr;
  int npairs = 0, nchunks = 0; size_t solve_lds = 0, blk_lds = 0, blk3_lds = 0; bool solve_in_lds = false, solve_blk = false, solve_blk3 = false;
  // per-point Schur work lists (sp.R == 0: not available for this window, the tuple-chunk kernel is used)
  BaSp sp = {};
  int* d_sp_bat_e0 = nullptr; uint32_t* d_sp_off = nullptr; uint32_t* d_sp_list = nullptr; int* d_sp_slot_pair = nullptr; int* d_sp_tup_base = nullptr;
  int* d_sp_pair_slots = nullptr; double* d_sp_partial = nullptr; double* d_sp_sum = nullptr; int* d_sp_chunk_off = nullptr;
  int sp_threads = 0; size_t sp_lds = 0;
  // edge-major Schur work list (se.R == 0: not available: too many free key frames for the LDS copy of the reduced system)
  BaSe se = {};
  int* d_se_chunk_e0 = nullptr; uint32_t* d_se_info = nullptr; double* d_se_partial = nullptr; double* d_se_bp_partial = nullptr; double* d_se_sum = nullptr; int* d_se_pob = nullptr; int* d_se_chunk_off = nullptr;
  int* d_se_lone = nullptr; int4* d_rm_chunk = nullptr; uint2* d_run_lane = nullptr; uint32_t* d_run_mf = nullptr; uint32_t* d_run_fl = nullptr; uint32_t* d_rm_cost = nullptr;
  uint32_t* d_run_fg = nullptr; int* d_rm_cut = nullptr;      // one-wavefront workgroups (cms_ba_schur_runwg.hip)
  size_t se_lds_fixed = 0; int se_waves = 0;      // LDS of the edge-major kernel without the per-wavefront part; wavefronts per workgroup that fit
  size_t rm_lds = 0; int n_runs = 0, rm_points = 0;   // run-major part (cms_ba_schur_runs.hip): LDS it needs (0: the window has no runs), runs, points inside runs
  char* h_stage = nullptr; size_t h_stage_bytes = 0;  // pinned block the window's uploads went through; cms_ba_read's read-back reuses it
  int cur = 0;
  double* h_pin = nullptr; size_t h_pin_bytes = 0;     // pinned host mirror of d_scal (one small D2H per Levenberg trial)
  // group resources (owned by the first window of a cms_ba_optimize_many call, grown on demand)
  void* grp_items_dev = nullptr; void* grp_items_host = nullptr; double* grp_scal_dev = nullptr; double* grp_scal_host = nullptr;
  void* grp_lm_dev = nullptr; void* grp_lm_host = nullptr;   // BaLmDev per window (device-side Levenberg state) and its pinned mirror
  int grp_cap = 0;
  // optional HIP-event bracket around ONE kernel of the grouped driver's rounds (bench.py's roofline of the dominant BA kernel):
  // kernel ids 1 lin, 2 maxdiag, 3 the Schur kernel of the path in use (kb_ba_lin_schur_edges by default), 4 its range reduction, 5 the trial solve,
  // 6 the trial kernel (kb_ba_trial_edges), 7 reduce2
  int prof_kernel = 0; std::vector<hipEvent_t> prof_ev; double prof_ms = 0; long prof_launches = 0;
  bool fast_plan = false;    // planned by ba_plan_fast (cms_api_ba_plan.hip): the permutations and the per-edge arrays exist on the device only
  int* d_raw_pose = nullptr; int* d_raw_point = nullptr; int8_t* d_raw_face = nullptr; int* d_prank = nullptr; int* d_cpo = nullptr; int* d_cedge = nullptr;
  uint8_t* d_pcopy = nullptr; uint8_t* d_lo_copy = nullptr; uint64_t* d_run_sig = nullptr; int* d_iperm = nullptr;
  bool dev_plan = false;     // ... and planned by k_ba_plan_many (cms_api_ba_devplan.hip): the host learns the counts from h_plan_counts once the kernel is through
  int* d_plan_counts = nullptr; int* h_plan_counts = nullptr; size_t h_plan_counts_bytes = 0;
  long long* d_plan_clk = nullptr;      // developer: CMS_BA_DP_CLK=1
  bool se_only = false;      // only the edge-major work list was built (see cms_ba_create)
  bool deterministic = false;   // created under cms_ba_set_deterministic(1): sums in a fixed order, bit-identical from run to run
  int det_ranges = 0;           // ... cut into this many workgroups whatever the group (fixed at creation: the bits depend on it)
  bool det_points = false;      // ... through the pair-owner Schur kernel on all work lists (CMS_BA_DET_POINTS, or a window the fused chain cannot take);
                                // otherwise through the fused chain's fixed-order kernels (kb_ba_lin_schur_runs_det / kb_ba_lin_schur_edges_det, slices)
  hipEvent_t ev_setup = nullptr;      // cms_ba_set_stream: marks the end of the window's set-up on the stream it was created on
  bool setup_wait_pending = false;    // ... and b->stream has not been told to wait for it yet (ba_order_behind_setup, at the window's first use there)
  hipStream_t grp_stream = nullptr;   // the stream of the group whose rounds may still be in flight for this window (ba_optimize_group; cleared at its successful end)
  bool async_pending = false;   // something asynchronous (upload, reset, a group's rounds) was enqueued on `stream` and nothing has waited for it yet
  bool gsum_clean = false;      // slice 0 of the Schur partial sums (the ONE global copy the workgroups add to, BaSe::gsum) is all zero: true after
                                // k_ba_gather / k_ba_reset and after every consuming solve kernel, false once a slice-STORING group ran on the window
  BaSe grp_se = {};          // the window's share of the current group's Schur launch (ba_upload_items)
  std::vector<BaBlock> slabs; size_t slab_off = 0;      // device memory of the window: carved from pooled slabs (ba_alloc)
  size_t grp_pin_bytes[3] = {0, 0, 0};                  // sizes of grp_items_host, grp_scal_host, grp_lm_host (pooled pinned blocks)
};

// Dynamic-LDS ceilings of the BA kernels: hipFuncAttributeMaxDynamicSharedMemorySize is a per-function (per device) global, so it is
// raised ONCE to the largest size any window can ask for and never lowered per handle (another window / host thread may be launching).
#define BA_LDS_CEILING (160 * 1024 - 512)
static int ba_lds_attrs_once(int device) {
  static std::mutex mu;
  static bool done[64] = {false};
  std::lock_guard<std::mutex> lk(mu);
  if (device < 0 || device >= 64 || done[device]) return CMS_OK;
  const void* fns[] = {(const void*)k_ba_schur_points, (const void*)kb_ba_schur_points, (const void*)kb_ba_schur_edges, (const void*)kb_ba_lin_schur_edges,
                       (const void*)kb_ba_lin_schur_runs, (const void*)kb_ba_lin_schur_runs_det, (const void*)kb_ba_lin_schur_edges_det, (const void*)kb_ba_lin_schur_runs_valu, (const void*)kb_ba_lin_schur_run_wg0, (const void*)kb_ba_lin_schur_run_wg1, (const void*)kb_ba_trial_solve3r, (const void*)kb_ba_trial_solve3rp, (const void*)k_ba_trial_solve,
                       (const void*)kb_ba_trial_solve, (const void*)kb_ba_trial_solve3, (const void*)k_ba_solve_r192};
  for (const void* f : fns) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, BA_LDS_CEILING);
    if (e != hipSuccess) return cms_fail(CMS_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)", e);
  }
  done[device] = true;
  return CMS_OK;
}

// Streams of destroyed windows are kept for the next window of the device: creating a stream costs ~10 ms of host time (measured with
// CMS_BA_CREATE_TIMING), a window's whole Levenberg run ~3.
struct BaStreamPool { std::mutex mu; std::vector<hipStream_t> idle[64]; };
static BaStreamPool& ba_stream_pool() { static BaStreamPool* p = new BaStreamPool; return *p; }      // never destroyed: no HIP calls at process exit
static hipStream_t ba_stream_take(int device) {
  BaStreamPool& pl = ba_stream_pool();
  std::lock_guard<std::mutex> lk(pl.mu);
  if (device < 0 || device >= 64 || pl.idle[device].empty()) return nullptr;
  hipStream_t s = pl.idle[device].back();
  pl.idle[device].pop_back();
  return s;
}
// hipStreamSynchronize that can give the core away (CMS_BA_RELAXED_WAIT=1): a short spin for waits of a few microseconds, then queries between 40-us sleeps.  The host threads
// that build, drive and finish windows outnumber the cores a rank gets (16 window threads + the group threads next to a 16-core quota), and the
// runtime's own wait spins: measured 1.1 ms of CPU per window in cms_ba_read alone, most of it waiting for a 0.6 MB read-back to get its turn.
static hipError_t ba_wait_stream(hipStream_t s) {
  // CMS_BA_RELAXED_WAIT=1 turns it on.  Off by default: on a host with cores to spare the sleeping waits cost 3-16 % of the bench's throughput
  // (windows are ready later, the chain's kernels measure slower) for three cores saved: an option for hosts short of cores
  static const bool relaxed = getenv("CMS_BA_RELAXED_WAIT") != nullptr;
  if (!relaxed) return hipStreamSynchronize(s);
  for (int i = 0; i < 64; ++i) {
    const hipError_t q = hipStreamQuery(s);
    if (q != hipErrorNotReady) return q == hipSuccess ? hipStreamSynchronize(s) : q;
  }
  for (;;) {
    std::this_thread::sleep_for(std::chrono::microseconds(40));
    const hipError_t q = hipStreamQuery(s);
    if (q != hipErrorNotReady) return q == hipSuccess ? hipStreamSynchronize(s) : q;
  }
}
// (a stream may come back with work still queued on it -- cms_ba_set_stream hands a window's set-up over to the group's stream with an event, not a
// host wait: whoever takes the stream next simply queues behind that work.  Only a stream in an error state is destroyed.)
static void ba_stream_give(int device, hipStream_t s) {
  BaStreamPool& pl = ba_stream_pool();
  {
    const hipError_t q = hipStreamQuery(s);
    std::lock_guard<std::mutex> lk(pl.mu);
    if (device >= 0 && device < 64 && pl.idle[device].size() < 256 && (q == hipSuccess || q == hipErrorNotReady)) { pl.idle[device].push_back(s); return; }
  }
  hipStreamDestroy(s);
}
// set-up streams of cms_ba_create_many: a pool of their own -- such a stream never becomes a window's own stream, windows only reference it (own_stream = false)
// until cms_ba_set_stream moves them, so it may be handed to the next call while earlier windows still point at it.  Never destroyed (no HIP calls at exit).
static BaStreamPool& ba_setup_stream_pool() { static BaStreamPool* p = new BaStreamPool; return *p; }
static hipStream_t ba_setup_stream_take(int device) {
  BaStreamPool& pl = ba_setup_stream_pool();
  {
    std::lock_guard<std::mutex> lk(pl.mu);
    if (device >= 0 && device < 64 && !pl.idle[device].empty()) { hipStream_t s = pl.idle[device].back(); pl.idle[device].pop_back(); return s; }
  }
  hipStream_t s = nullptr;
  // CMS_BA_SETUP_PRIORITY=low / high: the set-up streams in another priority class than the frame path's and the groups' (A/B)
  static const char* pr = getenv("CMS_BA_SETUP_PRIORITY");
  int lo = 0, hi = 0;
  if (pr && (pr[0] == 'l' || pr[0] == 'h') && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
    return hipStreamCreateWithPriority(&s, hipStreamNonBlocking, pr[0] == 'l' ? lo : hi) == hipSuccess ? s : nullptr;
  return hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess ? s : nullptr;
}
static void ba_setup_stream_give(int device, hipStream_t s) {
  BaStreamPool& pl = ba_setup_stream_pool();
  std::lock_guard<std::mutex> lk(pl.mu);
  if (device >= 0 && device < 64) pl.idle[device].push_back(s);
}
// events for such hand-overs, pooled like the streams
struct BaEventPool { std::mutex mu; std::vector<hipEvent_t> idle[64]; };
static BaEventPool& ba_event_pool() { static BaEventPool* p = new BaEventPool; return *p; }
static hipEvent_t ba_event_take(int device) {
  BaEventPool& pl = ba_event_pool();
  {
    std::lock_guard<std::mutex> lk(pl.mu);
    if (device >= 0 && device < 64 && !pl.idle[device].empty()) { hipEvent_t e = pl.idle[device].back(); pl.idle[device].pop_back(); return e; }
  }
  hipEvent_t e = nullptr;
  return hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess ? e : nullptr;
}
static void ba_event_give(int device, hipEvent_t e) {
  BaEventPool& pl = ba_event_pool();
  {
    std::lock_guard<std::mutex> lk(pl.mu);
    if (device >= 0 && device < 64 && pl.idle[device].size() < 1024) { pl.idle[device].push_back(e); return; }
  }
  hipEventDestroy(e);
}

// ---- memory of a window comes from a per-device pool.  A window of configs[3] size needs ~50 device buffers: allocated and freed one by
// one (hipMalloc is cheap, hipFree is not: ~40 us each) a LocalBundleAdjustment call spent 2 ms of its 17 in cms_ba_destroy, and the
// grouped driver's pinned blocks (hipHostMalloc: ~0.3 ms each) another millisecond.  Windows now carve their buffers out of a few slabs;
// slabs and pinned blocks of destroyed windows wait in the pool for the next window of the device (bounded: 16 GB of slabs -- a step of bench.py has ~100
// windows of 45 MB alive, and a hipFree in the middle of a step synchronises the device --, 512 pinned blocks).
struct BaMemPool {
  std::mutex mu;
  std::vector<BaBlock> dev[64], pin[64], stage[64];
  size_t dev_cached[64] = {0}, pin_cached[64] = {0}, stage_cached[64] = {0};      // bytes held per kind (pinned kinds: at most 2 GB each)
};
static BaMemPool& ba_pool() { static BaMemPool* p = new BaMemPool; return *p; }      // never destroyed: no HIP calls at process exit
static void* ba_pool_take(std::vector<BaBlock>& v, size_t bytes, size_t* got) {      // smallest cached block that is large enough (and not absurdly larger)
  int best = -1;
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i].bytes >= bytes && v[i].bytes <= 4 * bytes + (1u << 20) && (best < 0 || v[i].bytes < v[best].bytes)) best = (int)i;
  if (best < 0) return nullptr;
  void* p = v[best].p; *got = v[best].bytes;
  v[best] = v.back(); v.pop_back();
  return p;
}
// give the cached blocks of a device back to the runtime (all kinds, or only the device slabs); returns the bytes released
static size_t ba_pool_trim_device(int device, bool pinned_too) {
  BaMemPool& pl = ba_pool();
  if (device < 0 || device >= 64) return 0;
  std::vector<BaBlock> dv, pn, st;
  {
    std::lock_guard<std::mutex> lk(pl.mu);
    dv.swap(pl.dev[device]); pl.dev_cached[device] = 0;
    if (pinned_too) { pn.swap(pl.pin[device]); st.swap(pl.stage[device]); pl.pin_cached[device] = 0; pl.stage_cached[device] = 0; }
  }
  size_t n = 0;
  for (const BaBlock& b : dv) { hipFree(b.p); n += b.bytes; }
  for (const BaBlock& b : pn) { hipHostFree(b.p); n += b.bytes; }
  for (const BaBlock& b : st) { hipHostFree(b.p); n += b.bytes; }
  return n;
}
// The pool keeps slabs and pinned blocks of destroyed windows for the next window (a hipFree in the middle of a step synchronises the device).
// Callers that share the device with other allocators can hand them back: bytes released are returned in *released (may be NULL).
extern "C" int cms_ba_pool_trim(int device, size_t* released) {
  if (device < 0 || device >= 64) return cms_fail(CMS_ERR_ARG, "cms_ba_pool_trim: bad device");
  HIPCHK(hipSetDevice(device));
  const size_t n = ba_pool_trim_device(device, true);
  if (released) *released = n;
  return CMS_OK;
}
static size_t ba_pool_cap_bytes() {      // CMS_BA_POOL_MB: upper bound of the cached device slabs per device (default 16 GB)
  static const size_t cap = [] { const char* v = getenv("CMS_BA_POOL_MB"); return v ? (size_t)std::max(0, atoi(v)) << 20 : (size_t)16 << 30; }();
  return cap;
}
static hipError_t ba_dev_take(int device, size_t bytes, void** p, size_t* got) {
  BaMemPool& pl = ba_pool();
  if (device >= 0 && device < 64) {
    std::lock_guard<std::mutex> lk(pl.mu);
    if ((*p = ba_pool_take(pl.dev[device], bytes, got)) != nullptr) { pl.dev_cached[device] -= *got; return hipSuccess; }
  }
  *got = bytes;
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {      // gigabytes may sit idle in the cache: release them and try once more
    (void)hipGetLastError();
    if (ba_pool_trim_device(device, false) > 0) e = hipMalloc(p, bytes);
  }
  return e;
}
static void ba_dev_give(int device, void* p, size_t bytes) {
  BaMemPool& pl = ba_pool();
  if (device >= 0 && device < 64) {
    std::lock_guard<std::mutex> lk(pl.mu);
    if (pl.dev_cached[device] + bytes <= ba_pool_cap_bytes()) { pl.dev[device].push_back({p, bytes}); pl.dev_cached[device] += bytes; return; }
  }
  hipFree(p);
}
// pinned, device-visible, explicitly coherent host memory (the grouped driver's mirrors are written by running kernels)
static hipError_t ba_pin_take(int device, size_t bytes, void** p, size_t* got) {
  BaMemPool& pl = ba_pool();
  if (device >= 0 && device < 64) {
    std::lock_guard<std::mutex> lk(pl.mu);
    if ((*p = ba_pool_take(pl.pin[device], bytes, got)) != nullptr) { pl.pin_cached[device] -= *got; return hipSuccess; }
  }
  *got = bytes;
  hipError_t e = hipHostMalloc(p, bytes, hipHostMallocMapped | hipHostMallocCoherent);
  if (e != hipSuccess) { (void)hipGetLastError(); if (ba_pool_trim_device(device, true) > 0) e = hipHostMalloc(p, bytes, hipHostMallocMapped | hipHostMallocCoherent); }
  return e;
}
// plain pinned host memory (not device-coherent): staging blocks of uploads and read-backs.  A copy from / to such memory goes through the DMA
// engines; the coherent, device-mapped kind above made the runtime copy with a shader kernel (__amd_rocclr_copyBuffer: 1.8 ms of kernel time
// per bench.py step for the 32 window uploads) -- CMS_BA_STAGE_COHERENT=1 brings that back (A/B)
static hipError_t ba_stage_take(int device, size_t bytes, void** p, size_t* got) {
  static const bool coherent = getenv("CMS_BA_STAGE_COHERENT") != nullptr;
  BaMemPool& pl = ba_pool();
  if (device >= 0 && device < 64) {
    std::lock_guard<std::mutex> lk(pl.mu);
    if ((*p = ba_pool_take(pl.stage[device], bytes, got)) != nullptr) { pl.stage_cached[device] -= *got; return hipSuccess; }
  }
  *got = bytes;
  const unsigned flags = coherent ? (hipHostMallocMapped | hipHostMallocCoherent) : hipHostMallocDefault;
  hipError_t e = hipHostMalloc(p, bytes, flags);
  if (e != hipSuccess) { (void)hipGetLastError(); if (ba_pool_trim_device(device, true) > 0) e = hipHostMalloc(p, bytes, flags); }
  return e;
}
static void ba_stage_give(int device, void* p, size_t bytes) {
  BaMemPool& pl = ba_pool();
  if (device >= 0 && device < 64) {
    std::lock_guard<std::mutex> lk(pl.mu);
    if (pl.stage[device].size() < 512 && pl.stage_cached[device] + bytes <= ((size_t)2 << 30)) { pl.stage[device].push_back({p, bytes}); pl.stage_cached[device] += bytes; return; }
  }
  hipHostFree(p);
}
static void ba_pin_give(int device, void* p, size_t bytes) {
  BaMemPool& pl = ba_pool();
  if (device >= 0 && device < 64) {
    std::lock_guard<std::mutex> lk(pl.mu);
    if (pl.pin[device].size() < 512 && pl.pin_cached[device] + bytes <= ((size_t)2 << 30)) { pl.pin[device].push_back({p, bytes}); pl.pin_cached[device] += bytes; return; }
  }
  hipHostFree(p);
}

template <class T> static int ba_alloc(cms_ba* b, T** p, size_t n) {
  // carve from the window's current slab (256-byte granules); the first slab is sized for the window (~170 B per edge, ~400 B per point, the
  // edge-major kernel's partial sums), further ones are 4 MB or what the request needs
  const size_t need = (((n > 0 ? n : 1) * sizeof(T)) + 255) & ~(size_t)255;
  if (b->slabs.empty() || b->slab_off + need > b->slabs.back().bytes) {
    const size_t kk = (size_t)std::min(b->K, 26);       // (the edge-major partial sums exist for up to 25 free key frames)
    const size_t first = (size_t)b->E * 176 + (size_t)b->P * 416 + kk * kk * 22000 + ((size_t)1 << 20) +
                         (b->dev_plan ? (size_t)b->E * 8 + (size_t)b->P * 96 + ((size_t)5 << 20) : 0);      // (the plan kernel's scratch and its tables sized by bounds)
    size_t want = std::max(need, b->slabs.empty() ? first : (size_t)4 << 20);
    void* sp = nullptr; size_t got = 0;
    hipError_t e = ba_dev_take(b->device, want, &sp, &got);
    if (e != hipSuccess) return cms_fail(CMS_ERR_HIP, "hipMalloc (BA)", e);
    b->slabs.push_back({sp, got});
    b->slab_off = 0;
  }
  *p = reinterpret_cast<T*>(static_cast<char*>(b->slabs.back().p) + b->slab_off);
  b->slab_off += need;
  return CMS_OK;
}

extern "C" void cms_ba_destroy(cms_ba* b) {
  if (!b) return;
  hipSetDevice(b->device);
  // nothing of this window may still be running when its memory goes back to the pool.  A window on a stream of its own waits for that
  // stream; a window on a stream it shares (cms_ba_set_stream: the group's stream, busy with the NEXT windows by now) only when it has
  // something of its own pending there -- optimise / read return with the window's work complete
  if (b->setup_wait_pending) { (void)hipEventSynchronize(b->ev_setup); b->setup_wait_pending = false; b->async_pending = false; }      // (never used on its new stream: only its set-up can be in flight)
  if (b->stream && (b->own_stream || b->async_pending)) (void)ba_wait_stream(b->stream);
  // ... and a window of a group whose call ended early (an error between two rounds): the group's kernels run on the group OWNER's stream
  if (b->async_pending && b->grp_stream && b->grp_stream != b->stream) (void)ba_wait_stream(b->grp_stream);
  for (const BaBlock& sl : b->slabs) ba_dev_give(b->device, sl.p, sl.bytes);
  if (b->h_pin) ba_pin_give(b->device, b->h_pin, b->h_pin_bytes);
  if (b->h_plan_counts) ba_pin_give(b->device, b->h_plan_counts, b->h_plan_counts_bytes);
  if (b->h_stage) ba_stage_give(b->device, b->h_stage, b->h_stage_bytes);
  if (b->grp_items_host) ba_pin_give(b->device, b->grp_items_host, b->grp_pin_bytes[0]);
  if (b->grp_scal_host) ba_pin_give(b->device, b->grp_scal_host, b->grp_pin_bytes[1]);
  if (b->grp_lm_host) ba_pin_give(b->device, b->grp_lm_host, b->grp_pin_bytes[2]);
  for (hipEvent_t e : b->prof_ev) hipEventDestroy(e);
  if (b->ev_setup) ba_event_give(b->device, b->ev_setup);      // (the waits on it are long through: the window's work on the new stream was waited for above)
  if (b->stream && b->own_stream) { if (b->pooled_stream) ba_stream_give(b->device, b->stream); else hipStreamDestroy(b->stream); }
  delete b;
}
// first use of a window on the stream cms_ba_set_stream gave it: that stream waits (on the device) for the window's set-up
static hipError_t ba_order_behind_setup(cms_ba* b) {
  if (!b->setup_wait_pending) return hipSuccess;
  b->setup_wait_pending = false;
  return hipStreamWaitEvent(b->stream, b->ev_setup, 0);
}
extern "C" void* cms_ba_stream(cms_ba* b) { return b ? (void*)b->stream : nullptr; }
// Let the window run on a stream the caller owns (e.g. cms_ctx_stream of the mapping thread's context, shared by all windows of a group):
// a process that creates one stream per window ends up with more streams than hardware queues, and streams that share a queue
// serialise behind each other whether or not they depend on each other.
extern "C" int cms_ba_set_stream(cms_ba* b, void* hip_stream) {
  if (!b || !hip_stream) return cms_fail(CMS_ERR_ARG, "cms_ba_set_stream: bad argument");
  HIPCHK(hipSetDevice(b->device));
  // The window's set-up (one upload, one kernel) may still be queued on the stream it was created on: the NEW stream waits for it on the device
  // (an event), the calling thread does not -- it used to spin here for the ~1-2 ms the set-up takes to get its turn inside a busy step, a host
  // core per window-building thread (CMS_BA_SET_STREAM_WAIT=1: that host wait, A/B).
  // The wait is put into the new stream only when the window is first USED there (ba_order_behind_setup): a window is usually handed over
  // steps ahead of its optimisation, and a wait inserted now would hold up whatever the group's stream is running in the meantime.
  static const bool host_wait = getenv("CMS_BA_SET_STREAM_WAIT") != nullptr;
  if (host_wait || !b->async_pending || (hipStream_t)hip_stream == b->stream) {
    if (b->setup_wait_pending) { HIPCHK(hipEventSynchronize(b->ev_setup)); b->setup_wait_pending = false; }      // (an earlier hand-over's set-up, never waited for)
    HIPCHK(ba_wait_stream(b->stream));
    b->async_pending = false;
  } else {
    // a second hand-over before the window was ever used on the first new stream: that stream never waited for the set-up, so an event recorded
    // on it now would mark nothing -- put the pending wait into it first, then record behind it
    if (b->setup_wait_pending) HIPCHK(ba_order_behind_setup(b));
    if (!b->ev_setup) b->ev_setup = ba_event_take(b->device);
    if (!b->ev_setup) return cms_fail(CMS_ERR_HIP, "cms_ba_set_stream: no event");
    HIPCHK(hipEventRecord(b->ev_setup, b->stream));
    b->setup_wait_pending = true;
    // (async_pending stays set: the set-up counts as pending work of the new stream; optimise / read / reset / destroy order themselves behind it)
  }
  if (b->own_stream) { if (b->pooled_stream) ba_stream_give(b->device, b->stream); else HIPCHK(hipStreamDestroy(b->stream)); }
  b->stream = (hipStream_t)hip_stream; b->own_stream = false;
  return CMS_OK;
}
extern "C" int cms_ba_profile_kernel(cms_ba* b, int kernel_id) {
  if (!b || kernel_id < 0 || kernel_id > 7) return cms_fail(CMS_ERR_ARG, "cms_ba_profile_kernel: bad argument");
  b->prof_kernel = kernel_id; b->prof_ms = 0; b->prof_launches = 0;
  return CMS_OK;
}
extern "C" int cms_ba_profile_get(cms_ba* b, double* total_ms, long* launches) {
  if (!b || !total_ms || !launches) return cms_fail(CMS_ERR_ARG, "cms_ba_profile_get: bad argument");
  *total_ms = b->prof_ms; *launches = b->prof_launches;
  return CMS_OK;
}
#ifdef BA_RW_TS
extern "C" int cms_ba_debug_rw_ts(long long* out) {      // 2 classes x 4096 units x (start, end)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ba_rw_ts), sizeof(long long) * 2 * 2 * 4096) == hipSuccess ? CMS_OK : CMS_ERR_HIP;
}
#endif
#ifdef BA_RM_CLK
extern "C" int cms_ba_debug_rm_clocks(long long* out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(ba_rm_clk), 16 * sizeof(long long)) == hipSuccess ? CMS_OK : CMS_ERR_HIP;
}
#endif
#ifdef BA_S3_CLK
extern "C" int cms_ba_debug_s3_clocks(long long* out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(ba_s3_clk), 16 * sizeof(long long)) == hipSuccess ? CMS_OK : CMS_ERR_HIP;
}
#endif
extern "C" int cms_ba_debug_clocks(cms_ba* b, long long* out8) {
  if (!b || !out8) return CMS_ERR_ARG;
  return hipMemcpy(out8, b->d_scal + 8, 16 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? CMS_OK : CMS_ERR_HIP;
}

// Determinism as a product mode (cms_ba_set_deterministic; the environment variable CMS_BA_DETERMINISTIC gives the initial value): windows
// created while it is on carry the pair-owner kernel's work lists and run kb_ba_schur_points -- fixed summation order, bit-identical runs, like the
// reference's single-threaded g2o (ThirdParty/g2o/config.h:4).  The choice is taken when a window is CREATED and travels with it.
static std::atomic<int>& ba_det_mode() { static std::atomic<int> m{ba_knobs().deterministic ? 1 : 0}; return m; }
// (on >= 2: the number of workgroups such windows are cut into -- see the header; 1: the default count)
extern "C" int cms_ba_set_deterministic(int on) { ba_det_mode().store(on < 0 ? 1 : std::min(on, (int)BA_SE_RANGES)); return CMS_OK; }
extern "C" int cms_ba_get_deterministic(void) { return ba_det_mode().load(); }
static thread_local bool ba_tl_force_host_plan = false;      // cms_ba_linearize: the host plan for the window this thread creates next
static std::atomic<int> ba_plans_in_flight{0};      // windows being planned right now (cms_ba_create / cms_ba_debug_plan calls of all host threads)
// lanes of one group of 16 -> LDS banks, every lane with four candidate banks: augmenting-path matching, the rest on their least-used bank
struct BaDiagMatch {
  int n = 0; uint8_t bank[64][BA_SE_DCOPIES]; int choice[64]; int owner[16]; bool seen[16];
  bool go(int i) {
    for (int r = 0; r < BA_SE_DCOPIES; ++r) {
      const int c = bank[i][r];
      if (seen[c]) continue;
      seen[c] = true;
      if (owner[c] < 0 || go(owner[c])) { owner[c] = i; choice[i] = r; return true; }
    }
    return false;
  }
  void run() {
    int load[16];
    for (int c = 0; c < 16; ++c) { owner[c] = -1; load[c] = 0; }
    for (int i = 0; i < n; ++i) choice[i] = -1;
    for (int i = 0; i < n; ++i) {                               // a free bank among the lane's four, else an augmenting path
      bool done = false;
      for (int r = 0; r < BA_SE_DCOPIES && !done; ++r) if (owner[bank[i][r]] < 0) { owner[bank[i][r]] = i; choice[i] = r; done = true; }
      if (!done) { for (int c = 0; c < 16; ++c) seen[c] = false; go(i); }
    }
    for (int i = 0; i < n; ++i) if (choice[i] >= 0) ++load[bank[i][choice[i]]];
    for (int i = 0; i < n; ++i)
      if (choice[i] < 0) {
        int best = 0;
        for (int r = 1; r < BA_SE_DCOPIES; ++r) if (load[bank[i][r]] < load[bank[i][best]]) best = r;
        choice[i] = best; ++load[bank[i][best]];
      }
  }
};
// Chunk composition of the edge-major Schur kernel (see cms_ba_create): pure host code.  Outputs: prank / pinv (caller point <-> internal
// point), chunk_pt0 (first internal point of every chunk, P at the end), cp_off / cp_pose (CSR of the caller's points over their edges in
// pose order) and cp_rank (per such edge: the copy of the diagonal blocks its (a, a) tuple goes to).  lookahead <= 1: the caller's order.
static void ba_compose_chunks(int K, const uint8_t* fixed, int P, int E, const int* e_pose, const int* e_point, int lookahead,
                              std::vector<int>& prank, std::vector<int>& pinv, std::vector<int>& chunk_pt0, std::vector<int>& cp_off,
                              std::vector<int>& cp_pose, std::vector<uint8_t>& cp_rank) {
  const bool ctiming = ba_knobs().compose_timing;
  auto c_last = std::chrono::steady_clock::now();
  auto ctick = [&](const char* what) {
    const auto now = std::chrono::steady_clock::now();
    if (ctiming) fprintf(stderr, "[compose] %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - c_last).count());
    c_last = now;
  };
  // (scratch vectors are per host thread and keep their capacity from call to call: a pool of threads builds windows side by side, and
  // fresh multi-megabyte allocations -- page faults under the process-wide memory-map lock -- were most of a window's set-up time there)
#define BA_TLV(type, name) static thread_local std::vector<type> tl_##name; std::vector<type>& name = tl_##name
  prank.assign(P, 0); cp_off.assign(P + 1, 0); cp_pose.assign(E, 0); cp_rank.assign(E, 0);
  for (int e = 0; e < E; ++e) ++cp_off[e_point[e] + 1];
  for (int p = 0; p < P; ++p) cp_off[p + 1] += cp_off[p];
  BA_TLV(int, fill); fill.assign(cp_off.begin(), cp_off.end() - 1);
  for (int e = 0; e < E; ++e) cp_pose[fill[e_point[e]]++] = e_pose[e];
  ctick("csr");
  std::vector<int> gslot(K, -1);
  int gnp = 0;
  for (int k = 0; k < K; ++k) if (!fixed[k]) gslot[k] = gnp++;
  pinv.resize(P);
  const bool greedy = lookahead > 1 && gnp >= 2 && gnp <= 62;
  const int LA = std::max(lookahead, 1), MAXD = 4;
  BA_TLV(int, nxt); nxt.assign(P + 1, 0);            // singly linked list of the points not placed yet, in the caller's order
  for (int p = 0; p <= P; ++p) nxt[p] = p + 1;
  struct DiagLane { int cp; int16_t s; int16_t g; };  // diagonal tuples of the open chunk: position in cp_rank, free-pose slot, group of 16 lanes
  auto opair = [&](int s1, int s2) { return s1 * gnp - (s1 * (s1 + 1)) / 2 + (s2 - s1 - 1); };
  BA_TLV(int, tp_beg); BA_TLV(int, tp_end); tp_beg.assign(P, 0); tp_end.assign(P, 0);                  // a point's tuples in its segment's table
  BA_TLV(uint16_t, pm); BA_TLV(uint16_t, pm2); pm.assign((size_t)P * MAXD, 0); pm2.assign((size_t)P * MAXD, 0);   // per step: banks a point touches / touches twice
  BA_TLV(uint8_t, twice); twice.assign(P, 0);                                                          // ... three times: exact count below
  // ---- the composition proper, for the points [p_begin, p_end) of the caller's order: they fill the internal positions [p_begin, p_end).
  // Large windows are cut into segments composed by as many host threads (the segments are fixed by P alone: the result does not depend on
  // the machine); a segment ends with a chunk that may be short.
  auto compose_range = [&](int p_begin, int p_end, std::vector<int>& chunk_starts) {
  // the key frames of a point in ascending order, and every point's off-diagonal tuples, once: step - 1 | edge within the point << 2 | bank
  // class << 7; per step the set of banks they touch (pm), with a flag for points that touch a bank three times in one step (those, and points
  // that straddle two groups of lanes, take the exact count below; for all others the masks decide)
  static thread_local std::vector<uint16_t> tp;      // (per thread that runs a segment)
  tp.clear();
  tp.reserve((size_t)(cp_off[p_end] - cp_off[p_begin]) * 2);
  for (int p = p_begin; p < p_end; ++p) {
    std::sort(cp_pose.begin() + cp_off[p], cp_pose.begin() + cp_off[p + 1]);
    tp_beg[p] = (int)tp.size();
    const int k = cp_off[p + 1] - cp_off[p];
    const int* ps = &cp_pose[cp_off[p]];
    for (int dd = 1; greedy && dd <= k / 2 && dd <= MAXD; ++dd)
      for (int a1 = 0; a1 < k && a1 < 32; ++a1) {
        if (2 * dd == k && a1 >= dd) break;
        const int s1 = gslot[ps[a1]], s2 = gslot[ps[(a1 + dd) % k]];
        if (s1 < 0 || s2 < 0 || s1 == s2) continue;
        const int c = opair(std::min(s1, s2), std::max(s1, s2)) & 15;
        tp.push_back((uint16_t)((dd - 1) | (a1 << 2) | (c << 7)));
        uint16_t& m = pm[(size_t)p * MAXD + dd - 1];
        uint16_t& m2 = pm2[(size_t)p * MAXD + dd - 1];
        if (m2 & (1u << c)) twice[p] = 1;
        if (m & (1u << c)) m2 |= (uint16_t)(1u << c);
        m |= (uint16_t)(1u << c);
      }
    tp_end[p] = (int)tp.size();
  }
  int head = p_begin, placed = p_begin, cur_edges = 0;
  // per open chunk: [step][group of 16 lanes][bank class] lanes so far; a point may straddle two groups
  uint8_t cls[MAXD][4][16], mx[MAXD][4];             // mx: lanes on the fullest bank = what the group costs in that step (start: see the scan below)
  memset(cls, 0, sizeof(cls)); memset(mx, 1, sizeof(mx)); memset(mx[0], 2, sizeof(mx[0]));
  std::vector<DiagLane> dlanes;
  uint16_t fullmask[MAXD][4], nearmask[MAXD][4];        // banks that hold as many lanes as the group's fullest (one more lane there costs a
  memset(fullmask, 0, sizeof(fullmask));                // slot), and banks one lane short of that (two more lanes do)
  memset(nearmask, 0xFF, sizeof(nearmask));
  // how many LDS slots point p would add to the open chunk if it were placed at lane cur_edges: a group of 16 lanes costs, per step, as
  // many slots as its fullest bank holds lanes (0 = the point's tuples leave every group's maximum where it is); the tuples are counted in
  // as they go (a point's own tuples compete too) and taken out again unless the point is placed
  auto cost_of = [&](int p, bool mark, int limit) {
    int cost = 0, t = tp_beg[p];
    const int t1 = tp_end[p];
    const int kk = cp_off[p + 1] - cp_off[p], g0 = (cur_edges >> 4) & 3;
    if (!mark && !twice[p] && kk > 0 && ((cur_edges + kk - 1) >> 4) == (cur_edges >> 4)) {
      const uint16_t* m4 = &pm[(size_t)p * MAXD];
      const uint16_t* d4 = &pm2[(size_t)p * MAXD];
      int c4 = 0;
      for (int st = 0; st < MAXD; ++st)
        c4 += (int)(((m4[st] & fullmask[st][g0]) | (d4[st] & nearmask[st][g0])) != 0) + (int)((d4[st] & fullmask[st][g0]) != 0);
      return c4;
    }
    uint8_t m[MAXD][4];
    memcpy(m, mx, sizeof(m));
    for (; t < t1 && cost < limit; ++t) {
      const int w = tp[t], st = w & 3, g = ((cur_edges + ((w >> 2) & 31)) >> 4) & 3;
      const uint8_t n = ++cls[st][g][w >> 7];
      if (n > m[st][g]) { m[st][g] = n; ++cost; }
    }
    if (mark) {
      memcpy(mx, m, sizeof(m));
      uint32_t touched = 0;                                       // (step, group) pairs whose counts changed
      for (int u = tp_beg[p]; u < t1; ++u) { const int w = tp[u]; touched |= 1u << ((w & 3) * 4 + (((cur_edges + ((w >> 2) & 31)) >> 4) & 3)); }
      for (int sg = 0; sg < 16; ++sg) {
        if (!(touched >> sg & 1)) continue;
        const int st = sg >> 2, g = sg & 3;
        uint16_t f = 0, nf = 0;
        for (int c = 0; c < 16; ++c) {
          f |= (uint16_t)((cls[st][g][c] >= mx[st][g] ? 1u : 0u) << c);
          nf |= (uint16_t)((cls[st][g][c] + 1 >= mx[st][g] ? 1u : 0u) << c);
        }
        fullmask[st][g] = f; nearmask[st][g] = nf;
      }
    } else
      for (int u = tp_beg[p]; u < t; ++u) { const int w = tp[u]; --cls[w & 3][((cur_edges + ((w >> 2) & 31)) >> 4) & 3][w >> 7]; }
    return cost;
  };
  auto place_diag = [&](int p) {                     // the diagonal tuples wait for the chunk to be complete (finish_diag)
    const int k = cp_off[p + 1] - cp_off[p];
    for (int a1 = 0; a1 < k; ++a1) {
      const int s = gslot[cp_pose[cp_off[p] + a1]];
      if (s >= 0) dlanes.push_back({cp_off[p] + a1, (int16_t)s, (int16_t)(((cur_edges + a1) >> 4) & 3)});
    }
  };
  // copies of the diagonal blocks: a lane may add to any of the four copies of its key frame's block, i.e. to one of four banks; per group of
  // 16 lanes a bipartite matching (augmenting paths) gives as many lanes as possible a bank of their own, the rest take their least-used one
  auto finish_diag = [&]() {
    for (int g = 0; g < 4; ++g) {
      int lanes[64], nl = 0;
      for (size_t i = 0; i < dlanes.size(); ++i) if (dlanes[i].g == g && nl < 64) lanes[nl++] = (int)i;
      if (nl == 0) continue;
      BaDiagMatch M;
      M.n = nl;
      for (int i = 0; i < nl; ++i) for (int r = 0; r < BA_SE_DCOPIES; ++r) M.bank[i][r] = (uint8_t)((BA_SE_DSTRIDE * (r * gnp + dlanes[lanes[i]].s)) & 15);
      M.run();
      for (int i = 0; i < nl; ++i) cp_rank[dlanes[lanes[i]].cp] = (uint8_t)M.choice[i];
    }
    dlanes.clear();
  };
  chunk_starts.assign(1, p_begin);
  while (placed < p_end) {
    int pick = -1, prev_pick = -1;
    if (greedy) {
      // the first candidate that costs nothing ends the scan.  "Nothing" is measured against what a full group costs anyway: sixteen
      // first-step tuples on sixteen banks without a repeat are out of reach (a group of four 4-edge points would need its last point to
      // hit exactly the four free banks), so the first step's maximum starts at two lanes per bank, the later (half as dense) steps' at one
      const int accept = 0;
      int prev = -1, scanned = 0, best_cost = 1 << 30;
      for (int p = head; p < p_end && scanned < LA; prev = p, p = nxt[p], ++scanned) {
        if (cur_edges + (cp_off[p + 1] - cp_off[p]) > 64) continue;
        const int c = cost_of(p, false, best_cost);
        if (c < best_cost) { best_cost = c; pick = p; prev_pick = prev; if (c <= accept) break; }
      }
    } else if (cur_edges + (cp_off[head + 1] - cp_off[head]) <= 64) {
      pick = head;
    }
    if (pick < 0) {
      if (cur_edges == 0) { pick = head; prev_pick = -1; }          // a point with more than 64 edges: the kernel is refused below
      else {
        chunk_starts.push_back(placed);
        finish_diag();
        cur_edges = 0;
        memset(cls, 0, sizeof(cls)); memset(mx, 1, sizeof(mx)); memset(mx[0], 2, sizeof(mx[0])); memset(fullmask, 0, sizeof(fullmask)); memset(nearmask, 0xFF, sizeof(nearmask));
        continue;
      }
    }
    if (greedy) cost_of(pick, true, 1 << 30);
    place_diag(pick);
    cur_edges += cp_off[pick + 1] - cp_off[pick];
    prank[pick] = placed; pinv[placed] = pick; ++placed;
    if (prev_pick < 0) head = nxt[pick]; else nxt[prev_pick] = nxt[pick];
  }
  finish_diag();
  };      // compose_range
  const int seg_env = ba_knobs().compose_segments;      // A/B: fixed number of segments
  const int nseg = std::max(1, seg_env > 0 ? std::min(seg_env, std::max(1, P / 64)) : std::min(8, P / 512));
  std::vector<std::vector<int>> seg_chunks(nseg);
  if (ba_plans_in_flight.load(std::memory_order_relaxed) > 1) {
    // several windows are being built side by side (a pool of host threads, bench.py): their segments are already spread over the cores --
    // one after the other here, same segments, same result
    for (int t = 0; t < nseg; ++t) compose_range((int)((long long)P * t / nseg), (int)((long long)P * (t + 1) / nseg), seg_chunks[t]);
  } else {
    std::vector<std::thread> workers;
    for (int t = 1; t < nseg; ++t)
      workers.emplace_back([&, t]() { compose_range((int)((long long)P * t / nseg), (int)((long long)P * (t + 1) / nseg), seg_chunks[t]); });
    compose_range(0, (int)((long long)P / nseg), seg_chunks[0]);
    for (std::thread& w : workers) w.join();
  }
  chunk_pt0.clear();
  for (int t = 0; t < nseg; ++t) chunk_pt0.insert(chunk_pt0.end(), seg_chunks[t].begin(), seg_chunks[t].end());
  chunk_pt0.push_back(P);
  ctick("composition");
}
// developer / test entry: the composition alone (no device needed).  pinv: P, chunk_pt0: up to P + 1 entries (n_chunks + 1 are written),
// rank: E (in the order of the caller's points, edges of a point by ascending key frame)
extern "C" int cms_ba_debug_compose(int K, const uint8_t* fixed, int P, int E, const int* e_pose, const int* e_point, int lookahead,
                                    int* pinv_out, int* chunk_pt0_out, int* n_chunks, uint8_t* rank_out) {
  if (K < 1 || P < 1 || E < 1 || !fixed || !e_pose || !e_point || !pinv_out || !chunk_pt0_out || !n_chunks)
    return cms_fail(CMS_ERR_ARG, "cms_ba_debug_compose: bad argument");
  for (int e = 0; e < E; ++e)
    if (e_pose[e] < 0 || e_pose[e] >= K || e_point[e] < 0 || e_point[e] >= P) return cms_fail(CMS_ERR_ARG, "cms_ba_debug_compose: index out of range");
  std::vector<int> prank, pinv, chunk_pt0, cp_off, cp_pose;
  std::vector<uint8_t> cp_rank;
  ba_compose_chunks(K, fixed, P, E, e_pose, e_point, lookahead, prank, pinv, chunk_pt0, cp_off, cp_pose, cp_rank);
  memcpy(pinv_out, pinv.data(), (size_t)P * sizeof(int));
  memcpy(chunk_pt0_out, chunk_pt0.data(), chunk_pt0.size() * sizeof(int));
  *n_chunks = (int)chunk_pt0.size() - 1;
  if (rank_out) memcpy(rank_out, cp_rank.data(), (size_t)E);
  return CMS_OK;
}

// Workgroups of one window for the Schur launch.  Default: all `Rtotal` workgroups run the run-major body, whose wavefronts take equal shares of the
// estimated cost of ALL chunks (run chunks, then left-over chunks: the edge-major chunk loop inside that body) -- measured on 16 tracked configs[3]
// windows (tools/experiments_r04/exp.sh emcost, rmweight): 94 us a launch.  CMS_BA_SPLIT_WORKGROUPS=1: separate workgroups for the two kinds, split
// by CMS_BA_RM_WEIGHT (cost of a run chunk in percent of an edge-major one: 100 -> 102.5 us, 60 -> 95.0, 50 -> 95.2, 40 -> 101.7: with 14 + 2
// workgroups the two edge-major ones were the launch's long pole; a left-over chunk costs about 1.8 run chunks when eight wavefronts of a
// workgroup all add to LDS, less next to run-major wavefronts).
static void ba_se_split(BaSe& se, int Rtotal, bool run_wg = false) {
  const int n_se = se.nchunks - se.n_rm;
  Rtotal = std::max(2, std::min(Rtotal, BA_SE_RANGES));
  int R_rm = 0, R_se = 0;
  if (run_wg && se.n_rm > 0) {
    // the runs have their own launches (cms_ba_schur_runwg.hip); the edge-major kernel's workgroups take the left-over chunks, about one per wavefront
    se.R_rm = 0;
    se.R = n_se > 0 ? std::max(1, std::min(Rtotal, (n_se + BA_SE_THREADS / 64 - 1) / (BA_SE_THREADS / 64))) : 0;
    se.cpw = se.R > 0 ? (n_se + se.R - 1) / se.R : 1;
    if (se.R > 0) se.R = (n_se + se.cpw - 1) / se.cpw;
    return;
  }
  if (se.n_rm > 0 && n_se > 0 && ba_knobs().unified && !ba_knobs().rm_valu) {
    // the run-major body's workgroups take the left-over chunks too (a wavefront's range is cut by cost over all chunks): se.R = 0 tells it so
    se.cpw = 1; se.R = 0; se.R_rm = std::min(Rtotal, (se.nchunks + BA_SE_THREADS / 64 - 1) / (BA_SE_THREADS / 64));
    return;
  }
  if (se.n_rm > 0 && n_se > 0) {
    const double w_rm = 0.01 * ba_knobs().rm_weight * se.n_rm, w_se = (double)n_se;
    R_rm = (int)std::lround(Rtotal * w_rm / (w_rm + w_se));
    R_rm = std::max(1, std::min(R_rm, Rtotal - 1));
    R_se = Rtotal - R_rm;
  } else if (se.n_rm > 0) R_rm = Rtotal;
  else R_se = Rtotal;
  const int per_wg = ba_knobs().rm_valu ? BA_RM_PAIRS : BA_SE_THREADS / 64;            // chunk walkers per workgroup: pairs (vector variant) or wavefronts
  if (se.n_rm > 0) R_rm = std::min(R_rm, (se.n_rm + per_wg - 1) / per_wg);              // each wants at least one chunk
  se.cpw = n_se > 0 ? std::max(1, (n_se + R_se - 1) / R_se) : 1;
  se.R = n_se > 0 ? (n_se + se.cpw - 1) / se.cpw : 0;
  se.R_rm = R_rm;
}

// sizes that follow from the window's dimensions alone (both planners)
static void ba_plan_sizes(cms_ba* b, int K, int P, int E, int np) {
  b->np = np;
  const int n = 6 * np;
  b->nblk_e = (E + 255) / 256; b->nblk_p = (P + 127) / 128;
  b->blk_lds = ((size_t)36 * (np * (np + 1) / 2) + 72 * (size_t)np + 2 * (size_t)n + 8) * sizeof(double);
  b->solve_blk = np >= 1 && np * (np + 1) / 2 <= 384 && b->blk_lds <= BA_LDS_CEILING;
  b->blk3_lds = ((size_t)BA_S3_STRIDE * (np * (np - 1) / 2) + 36 * (size_t)np + 2 * (size_t)BA_S3_STRIDE * np + 36 + 3 * (size_t)n + 8) * sizeof(double);
  b->solve_blk3 = b->solve_blk && ba_s3_threads(np) <= 1024 && 6 * np <= 60 * BA_S3_NY && b->blk3_lds <= BA_LDS_CEILING;      // (np <= 25)
}
// LDS of the edge-major / run-major Schur kernels without the per-wavefront part: the copy of the reduced system + the key frames' rotations
static size_t ba_se_fixed_lds(int K, int np) {
  const int NP2 = np * (np + 1) / 2;
  return ((size_t)(((NP2 - np) * BA_SE_SSTRIDE + 1) & ~1) + (size_t)BA_SE_DCOPIES * np * BA_SE_DSTRIDE + (size_t)K * 12) * sizeof(double);
}

// ---- everything cms_ba_create decides on the host (no device needed: cms_ba_debug_plan runs it alone) -- the internal point order (signature
// runs first, then the composed left-over chunks), the sorted edge arrays and the work lists of the edge-major / run-major Schur kernels
struct BaPlan {
  std::vector<int> pose_slot, prank, s_pose, s_point, pt_off, pose_off, pose_edges, ce0, pob, ident, lone;
  std::vector<double> s_obs, s_inv;
  std::vector<int8_t> s_face;
  std::vector<uint32_t> info;
  std::vector<int4> rm_chunk;
  std::vector<uint2> run_lane;
  std::vector<uint32_t> run_mf, run_fl, rm_cost, run_fg;
  std::vector<int> rm_cut;
  int n_rmA = 0;
  bool se_built = false;
};
template <class Tick>
static void ba_plan(cms_ba* b, BaPlan& pl, int K, const uint8_t* fixed, int P, int E, const int* e_pose, const int* e_point, const double* e_obs,
                    const double* e_invsig2, const int8_t* e_face, Tick&& tick) {
  std::vector<int>&pose_slot = pl.pose_slot, &prank = pl.prank, &s_pose = pl.s_pose, &s_point = pl.s_point, &pt_off = pl.pt_off, &pose_off = pl.pose_off,
                  &pose_edges = pl.pose_edges, &ce0 = pl.ce0, &pob = pl.pob, &ident = pl.ident, &lone = pl.lone;
  std::vector<double>&s_obs = pl.s_obs, &s_inv = pl.s_inv;
  std::vector<int8_t>& s_face = pl.s_face;
  std::vector<uint32_t>& info = pl.info;
  std::vector<int4>& rm_chunk = pl.rm_chunk;
  std::vector<uint2>& run_lane = pl.run_lane;
  std::vector<uint32_t>&run_mf = pl.run_mf, &run_fl = pl.run_fl, &rm_cost = pl.rm_cost, &run_fg = pl.run_fg;
  std::vector<int>& rm_cut = pl.rm_cut;
  bool& se_built = pl.se_built;
  struct InFlight { InFlight() { ba_plans_in_flight.fetch_add(1); } ~InFlight() { ba_plans_in_flight.fetch_sub(1); } } in_flight;
  const BaKnobs& kn = ba_knobs();
  // ---- CSR of the caller's points over their observations, a point's observations by ascending key frame (two stable counting passes)
  BA_TLV(int, cpo); BA_TLV(int, cpe); cpo.assign(P + 1, 0); cpe.assign(E, 0);
  {
    BA_TLV(int, by_pose); BA_TLV(int, cnt); by_pose.assign(E, 0); cnt.assign((size_t)std::max(K, P) + 1, 0);
    for (int e = 0; e < E; ++e) ++cnt[e_pose[e] + 1];
    for (int k = 0; k < K; ++k) cnt[k + 1] += cnt[k];
    for (int e = 0; e < E; ++e) by_pose[cnt[e_pose[e]]++] = e;
    for (int e = 0; e < E; ++e) ++cpo[e_point[e] + 1];
    for (int p = 0; p < P; ++p) cpo[p + 1] += cpo[p];
    BA_TLV(int, fill); fill.assign(cpo.begin(), cpo.end() - 1);
    for (int i = 0; i < E; ++i) { const int e = by_pose[i]; cpe[fill[e_point[e]]++] = e; }
  }
  pose_slot.assign(K, -1);
  int np = 0;
  for (int k = 0; k < K; ++k) if (!fixed[k]) pose_slot[k] = np++;
  ba_plan_sizes(b, K, P, E, np);
  const int n = 6 * np;
  // ---- can this window run the edge-major / run-major kernels at all?  (LDS copy of the reduced system, <= 31 observations per point, no
  // point seen twice by one key frame: the pair-owner kernel handles those)
  const int NP2 = np * (np + 1) / 2;
  const size_t se_fixed_lds = ba_se_fixed_lds(K, np);
  const size_t se_wave_lds = (size_t)64 * 18 * sizeof(double) + 64 * sizeof(int);
  int se_nw = BA_SE_THREADS / 64;
  while (se_nw > 2 && se_fixed_lds + se_nw * se_wave_lds > BA_LDS_CEILING) se_nw -= 2;
  bool se_ok = np >= 1 && se_fixed_lds + se_nw * se_wave_lds <= BA_LDS_CEILING && K <= 256 && np <= 62;
  for (int p = 0; p < P && se_ok; ++p) {
    if (cpo[p + 1] - cpo[p] > 31) se_ok = false;
    for (int i = cpo[p] + 1; i < cpo[p + 1] && se_ok; ++i) if (e_pose[cpe[i]] == e_pose[cpe[i - 1]]) se_ok = false;
  }
  // ---- observation signatures -> runs (cms_ba_schur_runs.hip).  Points seen by the same set of key frames are grouped (first appearance
  // orders the groups, the caller's order the points of a group: the result depends on the input alone); a group with at least
  // `run_min_chunks` full chunks becomes a run, its tail (< 3/4 of a chunk) and all smaller groups are the left-over points of the
  // edge-major kernel.  Runs need the fused path (linearisation inside the Schur kernel: three-lane solve, edge-major trial kernel) and the
  // LDS for four producer / consumer pairs.
  const size_t rm_lds = se_fixed_lds + (size_t)BA_RM_PAIRS * 2 * BA_RM_BUF * sizeof(double);
  const bool rm_ok = se_ok && kn.runs && !kn.no_fused && !kn.want_all_lists && !b->det_points && !kn.solve1 && !kn.trial_points && b->solve_blk3 && rm_lds <= BA_LDS_CEILING &&
                     BA_SE_THREADS == 128 * BA_RM_PAIRS;
  struct Run { int k, first, npts, chunks, m, kf; };          // first: a member point (its key frames are the signature)
  std::vector<Run> runs;
  BA_TLV(int, rm_points); rm_points.clear();                  // caller ids, run after run
  BA_TLV(int, left); left.clear();                            // caller ids of the left-over points, caller's order
  std::vector<int> rm_run_pt0;                                // per run: first position in rm_points
  if (rm_ok) {
    BA_TLV(int, gid); gid.assign(P, -1);
    std::vector<int> gcount, gfirst;
    std::unordered_map<uint64_t, std::vector<int>> table;     // signature hash -> groups with that hash (windows of more than 64 key frames)
    if (K <= 64) {
      // the signature IS a 64-bit set of key frames (a point is seen at most once by a key frame, checked below): open addressing on the set
      // itself, no chains, no allocation (a node-based map with a vector per signature was 0.4 of this phase's 0.67 ms)
      int cap = 1024;
      while (cap < 4 * 1024 && cap < 2 * P) cap <<= 1;            // signatures are few (hundreds); the table only has to stay sparse
      BA_TLV(uint64_t, hkey); BA_TLV(int, hval);
      auto rehash = [&](int ncap) {
        std::vector<uint64_t> ok(hkey.begin(), hkey.end()); std::vector<int> ov(hval.begin(), hval.end());
        hkey.assign(ncap, 0); hval.assign(ncap, -1);
        for (size_t i = 0; i < ok.size(); ++i)
          if (ov[i] >= 0) { size_t j = (size_t)((ok[i] * 0x9E3779B97F4A7C15ull) >> 40) & (ncap - 1); while (hval[j] >= 0) j = (j + 1) & (ncap - 1); hkey[j] = ok[i]; hval[j] = ov[i]; }
        cap = ncap;
      };
      hkey.assign(cap, 0); hval.assign(cap, -1);
      for (int p = 0; p < P; ++p) {
        uint64_t key = 0;
        for (int i = cpo[p]; i < cpo[p + 1]; ++i) key |= 1ull << e_pose[cpe[i]];
        if (__builtin_popcountll(key) != cpo[p + 1] - cpo[p]) key = ~0ull - (uint64_t)p;      // seen twice by a key frame: a group of its own (never a run)
        size_t j = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & (cap - 1);
        while (hval[j] >= 0 && hkey[j] != key) j = (j + 1) & (cap - 1);
        int g = hval[j];
        if (g < 0) {
          g = (int)gfirst.size(); gfirst.push_back(p); gcount.push_back(0); hkey[j] = key; hval[j] = g;
          if (2 * (int)gfirst.size() > cap) rehash(2 * cap);
        }
        gid[p] = g; ++gcount[g];
      }
    } else
    for (int p = 0; p < P; ++p) {
      const int k = cpo[p + 1] - cpo[p];
      uint64_t h = 1469598103934665603ull ^ (uint64_t)k;
      for (int i = cpo[p]; i < cpo[p + 1]; ++i) { h ^= (uint64_t)(e_pose[cpe[i]] + 1); h *= 1099511628211ull; }
      std::vector<int>& cand = table[h];
      int g = -1;
      for (int c : cand) {
        const int q = gfirst[c];
        if (cpo[q + 1] - cpo[q] != k) continue;
        bool same = true;
        for (int i = 0; i < k && same; ++i) same = e_pose[cpe[cpo[p] + i]] == e_pose[cpe[cpo[q] + i]];
        if (same) { g = c; break; }
      }
      if (g < 0) { g = (int)gfirst.size(); gfirst.push_back(p); gcount.push_back(0); cand.push_back(g); }
      gid[p] = g; ++gcount[g];
    }
    const int ng = (int)gfirst.size();
    std::vector<int> run_of_group(ng, -1), take(ng, 0);
    for (int g = 0; g < ng; ++g) {
      const int q = gfirst[g], k = cpo[q + 1] - cpo[q];
      int kf = 0;
      for (int i = cpo[q]; i < cpo[q + 1]; ++i) kf += pose_slot[e_pose[cpe[i]]] >= 0;
      if (k < 1 || k > 9 || kf < 1 || kf * (kf + 1) / 2 > 64 || (!kn.rm_valu && 6 * kf + 1 > 48)) continue;      // (k <= 9: the one-wavefront workgroups' own-block tasks, cms_ba_schur_runwg.hip)      // (lane tables of the vector variant / three MFMA tiles a side)
      const int m = std::min(64 / k, BA_RM_PTS);
      if (gcount[g] * 100 < kn.run_min_chunks * m * kn.run_min_pct) continue;      // at least that many FULL chunks (x run_min_pct / 100): a run pays for one set of LDS additions
      const int full = gcount[g] / m, tail = gcount[g] - full * m;
      const bool keep_tail = tail > 0 && 2 * tail >= m;          // a last chunk that is at least half full
      if (full == 0 && !keep_tail) continue;
      run_of_group[g] = (int)runs.size();
      take[g] = full * m + (keep_tail ? tail : 0);
      runs.push_back({k, q, take[g], full + (keep_tail ? 1 : 0), m, kf});
    }
    ba_rw_order_runs(runs, run_of_group, [](const Run& r) { return r.kf; });      // signatures with two tile rows first (cms_ba_schur_runwg.hip)
    rm_run_pt0.assign(runs.size() + 1, 0);
    for (size_t r = 0; r < runs.size(); ++r) rm_run_pt0[r + 1] = rm_run_pt0[r] + runs[r].npts;
    rm_points.assign(rm_run_pt0.back(), 0);
    std::vector<int> fill(rm_run_pt0.begin(), rm_run_pt0.end() - 1), seen(ng, 0);
    for (int p = 0; p < P; ++p) {
      const int g = gid[p], r = run_of_group[g];
      if (r >= 0 && seen[g] < take[g]) { rm_points[fill[r]++] = p; ++seen[g]; }
      else left.push_back(p);
    }
  } else {
    left.resize(P);
    for (int p = 0; p < P; ++p) left[p] = p;
  }
  const int P_rm = (int)rm_points.size(), PL = (int)left.size();
  tick("runs");
  // ---- internal point order: the runs' points first, then the left-over points in the chunk composition of the edge-major Schur kernel
  // (cms_ba_schur_edges.hip).  There a wavefront takes a CHUNK of whole points with at most 64 edges; in step d the lane of a point's a-th edge
  // adds its 6x6 product to the LDS block of the pose pair (pose_a, pose_(a+d) mod k), one element per instruction.  What such an instruction
  // costs is decided by where its addresses fall (tools/probe/lds_atomics.hip): the LDS takes the 64 lanes of a ds_add_f64 in four groups of
  // 16 CONSECUTIVE lanes, two clocks each when the 16 addresses fall on 16 different f64 banks (address mod 16 doubles), and two more clocks
  // for every further lane on the fullest bank (same address: six more).  Lanes 16 or more apart never compete.  Random pairs cost ~3.1 slots
  // per group (16 balls into 16 bins) -- the 2.7 lane-operations per clock measured against 7.8 on consecutive addresses.
  // The host therefore composes the chunks: points are taken from a look-ahead window over the caller's order so that, within each group
  // of 16 lanes and each step, the pairs' bank classes (pair index mod 16: the block stride is odd, and the block layout gives an element
  // and its transpose the same bank) repeat as little as possible; the diagonal tuples pick, among the four copies of their key frame's
  // diagonal block, the one whose bank is least used in their group.  Everything on the device is indexed by the internal point id;
  // cms_ba_read / cms_ba_linearize translate back.  CMS_BA_NO_PERMUTE=1 keeps the caller's order (A/B).
  BA_TLV(int, prankL); BA_TLV(int, pinvL); BA_TLV(int, chunk_pt0L); BA_TLV(int, cp_offL); BA_TLV(int, cp_poseL);
  prank.assign(P, 0);
  BA_TLV(uint8_t, cp_rankL);                          // per (left-over point, edge in pose order): copy of the diagonal blocks its (a, a) tuple goes to
  b->pinv.resize(P);
  if (PL > 0) {
    BA_TLV(int, loc); BA_TLV(int, ep); BA_TLV(int, ept); loc.assign(P, -1); ep.clear(); ept.clear();
    for (int i = 0; i < PL; ++i) loc[left[i]] = i;
    // (a deterministic window runs the pair-owner kernel on its own work lists: the look-ahead composition -- 3.6 of its 6.8 ms of planning -- would place
    // points for LDS bank classes of a kernel the window never runs; it keeps the caller's order)
    if (PL == P) ba_compose_chunks(K, fixed, P, E, e_pose, e_point, (kn.no_permute || b->det_points) ? 1 : kn.lookahead, prankL, pinvL, chunk_pt0L, cp_offL, cp_poseL, cp_rankL);
    else {
      ep.reserve(E); ept.reserve(E);
      for (int e = 0; e < E; ++e) if (loc[e_point[e]] >= 0) { ep.push_back(e_pose[e]); ept.push_back(loc[e_point[e]]); }
      // (the left-over points are the ones with rare signatures: a third of the look-ahead finds them partners almost as well, in a third of the time;
      // a sixth when they are a minority of the window -- the usual case once there are runs: 19 % of a tracked configs[3] window, most of them
      // with eight or more observations, the expensive ones to place.  Measured on 16 windows per launch: look-ahead 8 / 4 / 2 / none = 108.0 /
      // 104.9 / 114.4 / 116.6 us for the Schur kernel, and the composition is the largest single item of cms_ba_create's ~5 ms of CPU time -- a host
      // that builds 32 windows per 14 ms step inside a 16-CPU quota runs out of CPU first.  CMS_BA_LEFTOVER_LOOKAHEAD=n forces a look-ahead.)
      // Round 5: when the left-over points are a minority (<= a third of the window) they keep the caller's order -- look-ahead 1 / 4 / 8 measured
      // 88.7 / 89.6 / 86.9 us for the Schur launch (profiles/r04_experiments.txt): what the kernel needs is the matching of the diagonal copies per
      // group of 16 lanes, which every look-ahead keeps -- and the device-side planner (cms_api_ba_plan.hip) produces the very same chunks.
      const int la_left = kn.no_permute ? 1 : kn.leftover_lookahead > 0 ? kn.leftover_lookahead : (3 * PL <= P ? 1 : std::max(2, kn.lookahead / 3));
      ba_compose_chunks(K, fixed, PL, (int)ep.size(), ep.data(), ept.data(), la_left, prankL, pinvL, chunk_pt0L, cp_offL, cp_poseL, cp_rankL);
    }
  } else {
    chunk_pt0L.assign(1, 0);
  }
  for (int i = 0; i < P_rm; ++i) { prank[rm_points[i]] = i; b->pinv[i] = rm_points[i]; }
  for (int i = 0; i < PL; ++i) { prank[left[i]] = P_rm + prankL[i]; b->pinv[P_rm + prankL[i]] = left[i]; }
  // chunks: the runs' chunks (whole points of one signature), then the composed chunks of the left-over points
  b->se_chunk_pt0.clear();
  std::vector<int> rm_chunk_run;
  for (size_t r = 0; r < runs.size(); ++r)
    for (int c = 0; c < runs[r].chunks; ++c) { b->se_chunk_pt0.push_back(rm_run_pt0[r] + c * runs[r].m); rm_chunk_run.push_back((int)r); }
  const int n_rm = (int)rm_chunk_run.size();
  for (size_t c = 0; c + 1 < chunk_pt0L.size(); ++c) if (chunk_pt0L[c + 1] > chunk_pt0L[c]) b->se_chunk_pt0.push_back(P_rm + chunk_pt0L[c]);
  b->se_chunk_pt0.push_back(P);
  tick("chunks");
  // ---- edges sorted by (internal point, key frame): CSR by point; per-pose edge lists reference sorted positions
  b->perm.resize(E);
  s_pose.resize(E); s_point.resize(E); pt_off.resize(P + 1); pose_off.assign(K + 1, 0); pose_edges.resize(E);      // (all written in full below)
  s_face.resize(E);
  {
    int i = 0;
    for (int p = 0; p < P; ++p) {
      const int q = b->pinv[p];
      pt_off[p] = i;
      for (int t = cpo[q]; t < cpo[q + 1]; ++t, ++i) {
        const int e = cpe[t];
        b->perm[i] = e;
        s_pose[i] = e_pose[e]; s_point[i] = p; s_face[i] = e_face ? e_face[e] : 0;
        ++pose_off[s_pose[i] + 1];      // (measurements, informations and point positions are put in this order ON THE DEVICE: k_ba_gather)
      }
    }
    pt_off[P] = i;
  }
  for (int k = 0; k < K; ++k) pose_off[k + 1] += pose_off[k];
  {
    std::vector<int> fill(pose_off.begin(), pose_off.end() - 1);
    for (int i = 0; i < E; ++i) pose_edges[fill[s_pose[i]]++] = i;
  }
  tick("csr");
  // ---- work lists of the edge-major / run-major Schur kernels: chunks of whole points with <= 64 edges (one wavefront each), the per-edge
  // words, the runs' chunk descriptors and consumer-lane tables, the dense enumeration of the pose pairs s1 <= s2 for the solve kernel
  ce0.clear(); pob.clear(); ident.clear(); lone.clear(); info.clear(); rm_chunk.clear(); run_lane.clear(); run_mf.clear(); run_fl.clear(); rm_cost.clear(); run_fg.clear(); rm_cut.clear(); pl.n_rmA = 0;
  se_built = false;
  if (se_ok) {
    bool ok = true;
    info.resize(E);
    for (size_t c = 0; c + 1 < b->se_chunk_pt0.size() && ok; ++c) {
      const int p0 = b->se_chunk_pt0[c], p1 = b->se_chunk_pt0[c + 1];
      if (pt_off[p1] - pt_off[p0] > 64) { ok = false; break; }
      ce0.push_back(pt_off[p0]);
      for (int p = p0; p < p1; ++p) {
        const int ne = pt_off[p + 1] - pt_off[p];
        for (int a1 = 0; a1 < ne; ++a1) {
          const int e = pt_off[p] + a1;
          const int rank = p < P_rm ? ((p - p0) & (BA_SE_DCOPIES - 1)) : cp_rankL[(size_t)cp_offL[pinvL[p - P_rm]] + a1] % BA_SE_DCOPIES;      // chosen with the chunk (edges of a point are in pose order in both lists)
          info[e] = (uint32_t)a1 | ((uint32_t)ne << 5) | ((uint32_t)(pose_slot[s_pose[e]] + 1) << 10) | ((uint32_t)s_face[e] << 16) | ((uint32_t)s_pose[e] << 19) |
                    ((uint32_t)rank << 27);
        }
      }
    }
    ce0.push_back(E);
    if (ok) {
      const int nchunks = (int)ce0.size() - 1;
      pob.assign((size_t)NP2, 0); ident.resize((size_t)NP2 + 1);
      for (int I = 0; I < np; ++I)
        for (int Kc = 0; Kc <= I; ++Kc) pob[(size_t)I * (I + 1) / 2 + Kc] = Kc * np - (Kc * (Kc - 1)) / 2 + (I - Kc);   // block (I, K): pair (s1 = K, s2 = I)
      for (int i = 0; i <= NP2; ++i) ident[i] = i;
      for (int p = 0; p < P; ++p) if (pt_off[p + 1] == pt_off[p]) lone.push_back(p);      // points without observations are in no chunk
      // run chunks and the consumer lanes' tables
      const uint32_t dg_off = (uint32_t)(((NP2 - np) * BA_SE_SSTRIDE + 1) & ~1);
      rm_chunk.resize(n_rm);
      for (int c = 0; c < n_rm; ++c) {
        const Run& R = runs[rm_chunk_run[c]];
        const int p0 = b->se_chunk_pt0[c], p1 = std::min(b->se_chunk_pt0[c + 1], rm_run_pt0[rm_chunk_run[c] + 1]);
        rm_chunk[c] = make_int4(pt_off[p0], (pt_off[p1] - pt_off[p0]) | (R.k << 8) | ((p1 - p0) << 16), rm_chunk_run[c], p0);
      }
      run_lane.assign(runs.size() * 64, make_uint2(0u, 0u));
      for (size_t r = 0; r < runs.size(); ++r) {
        const int q = runs[r].first;
        int fpos[32], fslot[32], kf = 0;
        for (int i = cpo[q]; i < cpo[q + 1]; ++i) { const int s = pose_slot[e_pose[cpe[i]]]; if (s >= 0) { fpos[kf] = i - cpo[q]; fslot[kf] = s; ++kf; } }
        const int T = kf * (kf + 1) / 2, Q = 64 / T;
        int t = 0;
        for (int ia = 0; ia < kf; ++ia)
          for (int ib = ia; ib < kf; ++ib, ++t)
            for (int qq = 0; qq < Q; ++qq) {
              uint2& w = run_lane[r * 64 + (size_t)qq * T + t];
              w.x = ba_rm_lane_word(fpos[ia], fpos[ib], qq, Q, ia == ib);
              w.y = ia == ib ? dg_off + (uint32_t)(((qq % BA_SE_DCOPIES) * np + fslot[ia]) * BA_SE_DSTRIDE)
                             : (uint32_t)((fslot[ia] * np - (fslot[ia] * (fslot[ia] + 1)) / 2 + (fslot[ib] - fslot[ia] - 1)) * BA_SE_SSTRIDE);
            }
      }
      // the MFMA variant's stacked matrix of a signature: row / column i = (free key frame a = i / 6, row r = i % 6), then the rhs column
      run_mf.assign(runs.size() * 64, BA_RM_MF_NONE);
      for (size_t r = 0; r < runs.size(); ++r) {
        const int q = runs[r].first;
        int kf = 0;
        for (int i = cpo[q]; i < cpo[q + 1]; ++i) {
          const int s = pose_slot[e_pose[cpe[i]]];
          if (s < 0) continue;
          if (kf < 8) {                                            // (signatures with more free key frames exist in the vector variant only)
            for (int rr = 0; rr < 6; ++rr) run_mf[r * 64 + 6 * kf + rr] = (uint32_t)((i - cpo[q]) * 18 + 3 * rr);
            run_mf[r * 64 + 48 + kf] = (uint32_t)s;
          }
          ++kf;
        }
        if (6 * kf < 48) run_mf[r * 64 + 6 * kf] = BA_RM_MF_RHS;
        run_mf[r * 64 + 56] = (uint32_t)kf;
      }
      // ... and where a lane's accumulators go: accumulator g of tile (ti, tj) in lane l holds G[16 ti + (l >> 4) + 4 g][16 tj + (l & 15)]; the
      // upper triangle of every (key frame, key frame) block and the rhs column have a place in the LDS copy (block layout: ba_se_off; diagonal
      // blocks and right-hand sides in the diagonal copy g), everything else (lower triangle, padding) goes nowhere
      run_fl.assign(runs.size() * 64 * 12, 0xFFFFFFFFu);
      static const int tile_i[6] = {0, 0, 1, 0, 1, 2}, tile_j[6] = {0, 1, 1, 2, 2, 2};
      for (size_t r = 0; r < runs.size(); ++r) {
        const int kf = (int)run_mf[r * 64 + 56], n6 = 6 * kf;
        if (n6 + 1 > 48) continue;                                 // (vector variant only)
        for (int l = 0; l < 64; ++l)
          for (int t = 0; t < 6; ++t)
            for (int g = 0; g < 4; ++g) {
              const int I = 16 * tile_i[t] + (l >> 4) + 4 * g, N = 16 * tile_j[t] + (l & 15);
              if (!(I < n6 && N <= n6 && (N == n6 || I <= N))) continue;
              const int a1 = I / 6, r1 = I % 6, s1 = (int)run_mf[r * 64 + 48 + a1];
              uint32_t off;
              if (N == n6) off = dg_off + (uint32_t)((g * np + s1) * BA_SE_DSTRIDE + 21 + r1);
              else {
                const int a2 = N / 6, r2 = N % 6, s2 = (int)run_mf[r * 64 + 48 + a2];
                if (a1 == a2) off = dg_off + (uint32_t)((g * np + s1) * BA_SE_DSTRIDE + (r1 * 6 - (r1 * (r1 - 1)) / 2 + (r2 - r1)));
                else off = (uint32_t)((s1 * np - (s1 * (s1 + 1)) / 2 + (s2 - s1 - 1)) * BA_SE_SSTRIDE + ba_se_off(r1, r2));
              }
              uint32_t& w = run_fl[(r * 64 + l) * 12 + (4 * t + g) / 2];
              w = ((4 * t + g) & 1) ? ((w & 0x0000FFFFu) | (off << 16)) : ((w & 0xFFFF0000u) | off);
            }
      }
      // running sum of the run chunks' estimated cost: the run-major body cuts the wavefronts' ranges by it
      // (... and behind them the left-over chunks: an edge-major chunk costs a constant plus its steps of pair products -- half the edges of its
      // largest point)
      rm_cost.assign((size_t)nchunks + 1, 0u);
      for (int c = 0; c < n_rm; ++c) {
        const Run& R = runs[rm_chunk_run[c]];
        rm_cost[(size_t)c + 1] = rm_cost[c] + ba_rm_chunk_cost(R.k, (int)run_mf[(size_t)rm_chunk_run[c] * 64 + 56], rm_chunk[c].y >> 16);
      }
      for (int c = n_rm; c < nchunks; ++c) {
        int kmax = 1;
        for (int e = ce0[c]; e < ce0[c + 1]; ++e) kmax = std::max(kmax, (int)((info[e] >> 5) & 31));
        rm_cost[(size_t)c + 1] = rm_cost[c] + (uint32_t)(kn.em_cost_a + kn.em_cost_b * (kmax / 2));
      }
      // ... and for the one-wavefront workgroups (cms_ba_schur_runwg.hip): the accumulators' offsets into the global copy, the cut points of the two classes
      if (ba_want_rw_tables()) run_fg.assign(std::max<size_t>(runs.size(), 1) * 64 * 24, BA_RW_NONE);
      for (size_t r = 0; r < runs.size() && ba_want_rw_tables(); ++r) {
        const int q = runs[r].first;
        BaRunSig rs; rs.kf = 0; rs.fpos8 = 0; rs.fslot8 = 0;
        for (int i = cpo[q]; i < cpo[q + 1]; ++i) { const int sl = pose_slot[e_pose[cpe[i]]]; if (sl >= 0) rs.push(i - cpo[q], sl); }
        for (int l = 0; l < 64; ++l)
          for (int i = 0; i < 24; ++i) run_fg[(r * 64 + l) * 24 + i] = ba_run_fg_word(rs, np, l, i);
      }
      {
        int n_rmA = 0;
        while (n_rmA < n_rm && ba_rw_class(runs[rm_chunk_run[n_rmA]].kf) == 0) ++n_rmA;
        pl.n_rmA = n_rmA;
        if (ba_want_rw_tables()) {
          rm_cut.assign(2 * (BA_RW_CUTS + 1), 0);
          ba_rw_make_cuts(rm_cost, 0, n_rmA, rm_cut.data());
          ba_rw_make_cuts(rm_cost, n_rmA, n_rm, rm_cut.data() + (BA_RW_CUTS + 1));
        }
      }
      if (run_fl.empty()) run_fl.assign(12, 0xFFFFFFFFu);
      if (run_mf.empty()) run_mf.assign(64, BA_RM_MF_NONE);
      if (run_lane.empty()) run_lane.push_back(make_uint2(0u, 0u));
      if (rm_chunk.empty()) rm_chunk.push_back(make_int4(0, 0, -1, 0));
      BaSe& se = b->se;
      se.nchunks = nchunks; se.n_rm = n_rm; se.n_rmA = pl.n_rmA; se.npairs2 = NP2;
      se.cpw_t = (BA_TE_THREADS / 64) * kn.te_chunks;             // the trial kernel: te_chunks chunks per wavefront (its workgroups stage the key frames' rotations first)
      se.Rt = (nchunks + se.cpw_t - 1) / se.cpw_t;
      se.nlone = (int)lone.size();
      ba_se_split(se, BA_SE_RANGES);                              // a window on its own; a group re-splits (ba_upload_items)
      if (lone.empty()) lone.push_back(0);
      b->se_lds_fixed = se_fixed_lds; b->se_waves = se_nw; b->rm_lds = n_rm > 0 ? rm_lds : 0;
      b->n_runs = (int)runs.size(); b->rm_points = P_rm;
      se_built = true;
    }
  }
}

// developer / test entry, host only: the plan cms_ba_create makes for a window -- internal point order, chunks, signature runs, the run-major
// kernel's chunk descriptors and consumer-lane tables, the per-edge words -- so that the index arithmetic of cms_ba_schur_runs.hip can be
// replayed on the CPU (tests/test_ba_runs_cpu.py).  Sizes: pinv P, perm E, info E, chunk_pt0 P + 2, rm_chunk 4 x P, run_lane 128 x P, run_mf
// 64 x P, run_fl 768 x P (upper bounds; the last two may be NULL; counts[] = chunks, run chunks, runs, np, points inside runs, range split
// of a window on its own: R_rm, R).
extern "C" int cms_ba_debug_plan(int K, const uint8_t* fixed, int P, int E, const int* e_pose, const int* e_point, int* pinv_out, int* perm_out,
                                 uint32_t* info_out, int* chunk_pt0_out, int* rm_chunk_out, uint32_t* run_lane_out, int* counts,
                                 uint32_t* run_mf_out, uint32_t* run_fl_out) {
  if (K < 1 || P < 1 || E < 1 || !fixed || !e_pose || !e_point || !pinv_out || !perm_out || !info_out || !chunk_pt0_out || !rm_chunk_out || !run_lane_out || !counts)
    return cms_fail(CMS_ERR_ARG, "cms_ba_debug_plan: bad argument");
  for (int e = 0; e < E; ++e)
    if (e_pose[e] < 0 || e_pose[e] >= K || e_point[e] < 0 || e_point[e] >= P) return cms_fail(CMS_ERR_ARG, "cms_ba_debug_plan: index out of range");
  cms_ba* b = new cms_ba;
  b->K = K; b->P = P; b->E = E;
  BaPlan pl;
  auto t_last = std::chrono::steady_clock::now();
  const bool timing = ba_knobs().create_timing;
  ba_plan(b, pl, K, fixed, P, E, e_pose, e_point, nullptr, nullptr, nullptr, [&](const char* what) {
    const auto now = std::chrono::steady_clock::now();
    if (timing) fprintf(stderr, "[cms_ba_debug_plan] %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  });
  memcpy(pinv_out, b->pinv.data(), (size_t)P * sizeof(int));
  memcpy(perm_out, b->perm.data(), (size_t)E * sizeof(int));
  memcpy(chunk_pt0_out, b->se_chunk_pt0.data(), b->se_chunk_pt0.size() * sizeof(int));
  counts[0] = (int)b->se_chunk_pt0.size() - 1; counts[1] = pl.se_built ? b->se.n_rm : 0; counts[2] = pl.se_built ? b->n_runs : 0; counts[3] = b->np;
  counts[4] = pl.se_built ? b->rm_points : 0; counts[5] = pl.se_built ? b->se.R_rm : 0; counts[6] = pl.se_built ? b->se.R : 0; counts[7] = pl.se_built ? 1 : 0;
  if (pl.se_built) {
    memcpy(info_out, pl.info.data(), (size_t)E * sizeof(uint32_t));
    if (b->se.n_rm > 0) memcpy(rm_chunk_out, pl.rm_chunk.data(), (size_t)b->se.n_rm * sizeof(int4));
    if (b->n_runs > 0) memcpy(run_lane_out, pl.run_lane.data(), (size_t)b->n_runs * 64 * sizeof(uint2));
    if (b->n_runs > 0 && run_mf_out) memcpy(run_mf_out, pl.run_mf.data(), (size_t)b->n_runs * 64 * sizeof(uint32_t));
    if (b->n_runs > 0 && run_fl_out) memcpy(run_fl_out, pl.run_fl.data(), (size_t)b->n_runs * 64 * 12 * sizeof(uint32_t));
  }
  delete b;
  return CMS_OK;
}

#include "cms_api_ba_plan.hip"
#include "cms_api_ba_devplan.hip"

// edge i of the internal order is the caller's edge perm[i], point i the caller's point pinv[i]: measurements, informations and initial positions
// into that order (cms_ba_create uploads the caller's arrays untouched) -- and, in the same launch, what k_ba_reset does for a window that
// starts its life: current estimate = initial estimate, per-edge state and the global copy of the reduced system cleared
extern "C" __global__ void __launch_bounds__(256)
k_ba_gather(int K, int P, int E, const int* __restrict__ perm, const int* __restrict__ pinv, const double* __restrict__ raw_obs, const double* __restrict__ raw_inv,
            const double* __restrict__ raw_pts, double* __restrict__ e_obs, double* __restrict__ e_inv, double* __restrict__ pts0,
            const double* __restrict__ poses0, double* __restrict__ poses, double* __restrict__ pts, uint8_t* __restrict__ level, double* __restrict__ err,
            uint8_t* __restrict__ flags, double* __restrict__ gsum, int n_gsum, double* __restrict__ gsum_bp, int n_gsum_bp) {
  const int gs = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = t0; i < E; i += gs) {
    const int e = perm[i];
    reinterpret_cast<double2*>(e_obs)[i] = reinterpret_cast<const double2*>(raw_obs)[e];
    e_inv[i] = raw_inv[e];
    reinterpret_cast<double2*>(err)[i] = make_double2(0.0, 0.0);
    level[i] = 0; flags[i] = 0;
  }
  for (int i = t0; i < P; i += gs) {
    const int q = pinv[i];
#pragma unroll
    for (int j = 0; j < 3; ++j) { const double v = raw_pts[3 * (size_t)q + j]; pts0[3 * (size_t)i + j] = v; pts[3 * (size_t)i + j] = v; }
  }
  for (int i = t0; i < 7 * K; i += gs) poses[i] = poses0[i];
  for (int i = t0; i < n_gsum; i += gs) gsum[i] = 0.0;
  for (int i = t0; i < n_gsum_bp; i += gs) gsum_bp[i] = 0.0;
}

extern "C" int cms_ba_create(cms_ba** out, int device, int K, const double* poses, const uint8_t* fixed, int P,
                             const double* points, int E, const int* e_pose, const int* e_point, const double* e_obs,
                             const double* e_invsig2, const int8_t* e_face, double fx, double fy, double cx, double cy) {
  if (!out || K < 1 || P < 1 || E < 1 || !poses || !fixed || !points || !e_pose || !e_point || !e_obs || !e_invsig2 || !e_face)
    return cms_fail(CMS_ERR_ARG, "cms_ba_create: bad argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return cms_fail(CMS_ERR_NO_DEVICE, "no HIP device: the product path has no CPU fallback");
  HIPCHK(hipSetDevice(device));
  cms_ba* b = new cms_ba;
  b->device = device; b->K = K; b->P = P; b->E = E;
  const int det_mode_now = ba_det_mode().load();
  b->deterministic = det_mode_now != 0;
  if (b->deterministic) {
    static const int det_ranges_dflt = [] { const char* v = getenv("CMS_BA_DET_RANGES"); return v ? std::max(2, std::min((int)BA_SE_RANGES, atoi(v))) : 16; }();
    b->det_ranges = det_mode_now >= 2 ? det_mode_now : det_ranges_dflt;
    // which deterministic path: the fused chain in fixed order needs what the fused chain needs (three-lane solve: <= 25 free key frames; none of the
    // switches that take a kernel of the chain away); everything else keeps the pair-owner kernel on its own work lists
    const BaKnobs& k0 = ba_knobs();
    int np0 = 0;
    for (int k = 0; k < K; ++k) np0 += fixed[k] ? 0 : 1;
    ba_plan_sizes(b, K, P, E, np0);
    b->det_points = k0.det_points || !b->solve_blk3 || k0.no_fused || k0.solve1 || k0.trial_points || k0.rm_valu || k0.run_wg || k0.host_lm || k0.single_host_lm ||
                    k0.want_all_lists || k0.separate_reduce2 || k0.dup != 0;
  }
#define BA_TRY(x) do { int _rc = (x); if (_rc) { cms_ba_destroy(b); return _rc; } } while (0)
#define BA_HIP(x) do { hipError_t _e = (x); if (_e != hipSuccess) { cms_ba_destroy(b); return cms_fail(CMS_ERR_HIP, #x, _e); } } while (0)
  // CMS_BA_CREATE_TIMING=1: where the host side of a window's set-up goes (stderr, one line per window)
  const bool timing = ba_knobs().create_timing;
  auto t_last = std::chrono::steady_clock::now();
  std::string t_log;
  auto tick = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    char buf[64];
    snprintf(buf, sizeof(buf), " %s %.2f", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_log += buf; t_last = now;
  };
  BA_TRY(ba_lds_attrs_once(device));
  {
    // CMS_BA_STREAM_PRIORITY=low: the window's queue yields to normal-priority queues (the frame path) whenever both have workgroups ready
    int lo = 0, hi = 0;
    if (ba_knobs().stream_priority == 'l' && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
      BA_HIP(hipStreamCreateWithPriority(&b->stream, hipStreamNonBlocking, lo));
    else if (ba_tl_setup_stream) { b->stream = ba_tl_setup_stream; b->own_stream = false; }
    else {
      b->stream = ba_stream_take(device);
      if (!b->stream) BA_HIP(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
      b->pooled_stream = true;
    }
  }
  tick("stream");
  const BaKnobs& kn = ba_knobs();
  static thread_local BaPlan tl_pl;            // (keeps its vectors' capacity for the next window this host thread builds)
  BaPlan& pl = tl_pl;
  static thread_local BaFastPlan tl_fp;
  BaFastPlan& fp = tl_fp;
  // the device-side planner for the windows it takes (cms_api_ba_plan.hip: one pass over the observations on the host, the rest per point);
  // it validates the indices in that pass.  Everything else: the host plan, after the validation loop
  // ... or, inside a cms_ba_create_many call with CMS_BA_PLAN_ON_DEVICE, the plan kernel (cms_api_ba_devplan.hip): the host only checks what does not need
  // the observations, the kernel validates the indices and says so in its status word
  BaDevPlan* const dplan = ba_tl_force_host_plan ? nullptr : ba_tl_dev_plan;
  unsigned long long dp_free_mask = 0;
  const bool devp = dplan && ba_dev_plan_prepare(b, K, fixed, P, E, pl.pose_slot, dp_free_mask);
  b->dev_plan = devp;
  const int fast_rc = devp ? 1 : ba_tl_force_host_plan ? 0 : ba_plan_fast(b, fp, K, fixed, P, E, e_pose, e_point, e_face, tick);
  const bool fast = fast_rc > 0;
  bool bad_index = fast_rc < 0;
  if (!fast && !bad_index)
    for (int e = 0; e < E && !bad_index; ++e)
      bad_index = e_pose[e] < 0 || e_pose[e] >= K || e_point[e] < 0 || e_point[e] >= P || e_face[e] < 0 || e_face[e] > 4;
  if (bad_index) {
    cms_ba_destroy(b);
    return cms_fail(CMS_ERR_ARG, "cms_ba_create: edge index / face out of range (unknown-face edges must be culled by the caller)");
  }
  b->fast_plan = fast;
  if (devp) {}
  else if (fast) pl.pose_slot = fp.pose_slot;
  else ba_plan(b, pl, K, fixed, P, E, e_pose, e_point, e_obs, e_invsig2, e_face, tick);
  std::vector<int>&pose_slot = pl.pose_slot, &prank = pl.prank, &s_pose = pl.s_pose, &s_point = pl.s_point, &pt_off = pl.pt_off, &pose_off = pl.pose_off, &pose_edges = pl.pose_edges;
  std::vector<double>&s_obs = pl.s_obs, &s_inv = pl.s_inv;
  std::vector<int8_t>& s_face = pl.s_face;
  const bool se_built = fast || pl.se_built;
  const int np = b->np, n = 6 * np;
  // ---- everything the window uploads goes through ONE pinned block and ONE asynchronous copy on the window's stream (a dozen synchronous
  // hipMemcpy calls from pageable memory were 0.3 ms of a 2 ms set-up and serialised the host threads that build windows side by side)
  struct Up { const void* src; size_t bytes; void** dst; bool direct; };
  std::vector<Up> ups;
  auto up = [&](const void* src, size_t bytes, auto** dst) { ups.push_back({src, bytes, reinterpret_cast<void**>(dst), false}); };
  // one of the CALLER's arrays: with CMS_BA_INPUTS_PINNED it is not staged (a memcpy of ~3 MB per 80 k-observation window: as much host time as the
  // whole plan) but copied asynchronously from the caller's pinned memory
  const bool inputs_pinned = ba_tl_inputs_pinned;
  auto up_in = [&](const void* src, size_t bytes, auto** dst) { ups.push_back({src, bytes, reinterpret_cast<void**>(dst), inputs_pinned && bytes >= 4096}); };
  if (se_built) {
    const int NP2 = np * (np + 1) / 2;
    // (a deterministic window's first pass parks its workgroups' diagonal sums in `partial` before the first Schur launch: Rt slices of 6 np, and Rt =
    // chunks / cpw_t rounded up <= P / cpw_t + 2 -- a chunk holds at least one point)
    const size_t cpw_t_min = (size_t)(BA_TE_THREADS / 64) * (size_t)std::max(1, kn.te_chunks);
    const size_t se_partial_n = std::max((size_t)BA_SE_RANGES * NP2 * 42, b->deterministic ? ((size_t)P / cpw_t_min + 4) * 6 * (size_t)np : (size_t)0);
    BA_TRY(ba_alloc(b, &b->d_se_partial, se_partial_n)); BA_TRY(ba_alloc(b, &b->d_se_bp_partial, (size_t)BA_SE_RANGES * np * 6));
    BA_TRY(ba_alloc(b, &b->d_se_sum, (size_t)NP2 * 42));
    b->se.partial = b->d_se_partial; b->se.bp_partial = b->d_se_bp_partial;
  }
  const int dp_run_cap = devp ? std::min(BA_DP_RUN_CAP + 1, P / 7 + 1) : 0, dp_chunk_cap = devp ? E / 28 + dp_run_cap + 16 : 0;
  if (devp) {
    // what the plan kernel writes (sized by bounds: a chunk holds >= 28 observations or is the last of its run / segment) and the two tables that follow from np alone
    const int NP2 = np * (np + 1) / 2;
    fp.pob.assign((size_t)NP2, 0); fp.ident.resize((size_t)NP2 + 1);
    for (int I = 0; I < np; ++I)
      for (int Kc = 0; Kc <= I; ++Kc) fp.pob[(size_t)I * (I + 1) / 2 + Kc] = Kc * np - (Kc * (Kc - 1)) / 2 + (I - Kc);
    for (int i = 0; i <= NP2; ++i) fp.ident[i] = i;
    up(fp.pob.data(), fp.pob.size() * sizeof(int), &b->d_se_pob); up(fp.ident.data(), fp.ident.size() * sizeof(int), &b->d_se_chunk_off);
    BA_TRY(ba_alloc(b, &b->d_se_chunk_e0, (size_t)dp_chunk_cap + 1)); BA_TRY(ba_alloc(b, &b->d_se_lone, (size_t)P)); BA_TRY(ba_alloc(b, &b->d_rm_chunk, (size_t)dp_chunk_cap));
    BA_TRY(ba_alloc(b, &b->d_rm_cost, (size_t)dp_chunk_cap + 1)); BA_TRY(ba_alloc(b, &b->d_run_sig, (size_t)dp_run_cap)); BA_TRY(ba_alloc(b, &b->d_rm_cut, 1));
    BA_TRY(ba_alloc(b, &b->d_se_info, (size_t)E)); BA_TRY(ba_alloc(b, &b->d_run_mf, (size_t)dp_run_cap * 64));
    BA_TRY(ba_alloc(b, &b->d_run_fl, (size_t)dp_run_cap * 64 * 12)); BA_TRY(ba_alloc(b, &b->d_run_lane, 1));
  } else if (fast) {
    up(fp.ce0.data(), fp.ce0.size() * sizeof(int), &b->d_se_chunk_e0); up(fp.pob.data(), fp.pob.size() * sizeof(int), &b->d_se_pob);
    up(fp.ident.data(), fp.ident.size() * sizeof(int), &b->d_se_chunk_off); up(fp.lone.data(), fp.lone.size() * sizeof(int), &b->d_se_lone);
    up(fp.rm_chunk.data(), fp.rm_chunk.size() * sizeof(int4), &b->d_rm_chunk); up(fp.rm_cost.data(), fp.rm_cost.size() * sizeof(uint32_t), &b->d_rm_cost);
    up(fp.run_sig.data(), fp.run_sig.size() * sizeof(uint64_t), &b->d_run_sig);
    up(fp.rm_cut.data(), fp.rm_cut.size() * sizeof(int), &b->d_rm_cut);
    // what the expansion kernels write: the per-edge words and the runs' tables (run_lane: the vector variant's table, never read on this path)
    BA_TRY(ba_alloc(b, &b->d_se_info, (size_t)E)); BA_TRY(ba_alloc(b, &b->d_run_mf, (size_t)std::max(fp.n_runs, 1) * 64));
    BA_TRY(ba_alloc(b, &b->d_run_fl, (size_t)std::max(fp.n_runs, 1) * 64 * 12)); BA_TRY(ba_alloc(b, &b->d_run_lane, 1));
    if (ba_want_rw_tables()) BA_TRY(ba_alloc(b, &b->d_run_fg, (size_t)std::max(fp.n_runs, 1) * 64 * 24));
  } else if (se_built) {
    up(pl.ce0.data(), pl.ce0.size() * sizeof(int), &b->d_se_chunk_e0); up(pl.info.data(), pl.info.size() * sizeof(uint32_t), &b->d_se_info);
    up(pl.pob.data(), pl.pob.size() * sizeof(int), &b->d_se_pob); up(pl.ident.data(), pl.ident.size() * sizeof(int), &b->d_se_chunk_off);
    up(pl.lone.data(), pl.lone.size() * sizeof(int), &b->d_se_lone);
    up(pl.rm_chunk.data(), pl.rm_chunk.size() * sizeof(int4), &b->d_rm_chunk); up(pl.run_lane.data(), pl.run_lane.size() * sizeof(uint2), &b->d_run_lane);
    up(pl.run_mf.data(), pl.run_mf.size() * sizeof(uint32_t), &b->d_run_mf); up(pl.run_fl.data(), pl.run_fl.size() * sizeof(uint32_t), &b->d_run_fl);
    up(pl.rm_cost.data(), pl.rm_cost.size() * sizeof(uint32_t), &b->d_rm_cost);
    up(pl.run_fg.data(), pl.run_fg.size() * sizeof(uint32_t), &b->d_run_fg); up(pl.rm_cut.data(), pl.rm_cut.size() * sizeof(int), &b->d_rm_cut);
  }
  tick("se");
  // A window that has the edge-major work list runs through the grouped driver with the edge-major kernels (cms_ba_optimize_many also puts
  // it into a group of its own kind): the pair-owner and tuple-chunk kernels' work lists (co-visibility tuples: ~10 per point, 2.4 ms of
  // host time at 80 k edges) and the stored 6x3 blocks (144 B per edge) are then never touched and are not built.  The A/B switches that
  // select those kernels bring them back.
  b->se_only = fast || (se_built && b->solve_blk && !kn.want_all_lists && !b->det_points);
  if (!b->se_only) BA_TRY(ba_alloc(b, &b->d_Hpl, 18 * (size_t)E));
  // co-visibility tuples: for every point, every ordered pair of its edges whose free-pose slots satisfy s1 <= s2
  if (!b->se_only) {
    struct Tup { int pair, a1, a2; };
    std::vector<Tup> tups;
    for (int p = 0; p < P; ++p)
      for (int a1 = pt_off[p]; a1 < pt_off[p + 1]; ++a1) {
        const int s1 = pose_slot[s_pose[a1]];
        if (s1 < 0) continue;
        for (int a2 = pt_off[p]; a2 < pt_off[p + 1]; ++a2) {
          const int s2 = pose_slot[s_pose[a2]];
          if (s2 < 0 || s2 < s1) continue;
          tups.push_back({s1 * np + s2, a1, a2});
        }
      }
    {   // stable counting sort by pair (np^2 keys): the list has ~10 entries per point
      std::vector<int> cnt((size_t)np * np + 1, 0);
      for (const Tup& t : tups) ++cnt[t.pair + 1];
      for (size_t i = 1; i < cnt.size(); ++i) cnt[i] += cnt[i - 1];
      std::vector<Tup> sorted(tups.size());
      for (const Tup& t : tups) sorted[cnt[t.pair]++] = t;
      tups.swap(sorted);
    }
    std::vector<int> ps1, ps2, poff;
    std::vector<int2> tt(tups.size());
    for (size_t i = 0; i < tups.size(); ++i) {
      if (i == 0 || tups[i].pair != tups[i - 1].pair) { ps1.push_back(tups[i].pair / np); ps2.push_back(tups[i].pair % np); poff.push_back((int)i); }
      tt[i] = make_int2(tups[i].a1, tups[i].a2);
    }
    poff.push_back((int)tups.size());
    b->npairs = (int)ps1.size();
    {   // lower block (I, K) of the reduced system -> index of the pose pair (s1 = K, s2 = I) or -1 (not co-visible)
      std::vector<int> pob((size_t)std::max(np * (np + 1) / 2, 1), -1);
      for (size_t pr = 0; pr < ps1.size(); ++pr) pob[(size_t)ps2[pr] * (ps2[pr] + 1) / 2 + ps1[pr]] = (int)pr;
      BA_TRY(ba_alloc(b, &b->d_pair_of_block, pob.size()));
      BA_HIP(hipMemcpy(b->d_pair_of_block, pob.data(), pob.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    tick("tuples");
    // ---- work lists of the per-point Schur kernel (cms_ba_schur_points.hip): batches of consecutive points whose edges fit
    // LDS, ranges of consecutive batches (one workgroup each), one owner slot per co-visible pair (several helper slots for
    // the diagonal pairs, which get one tuple per edge), per (batch, slot) the tuples as 16-bit local edge ids
    {
      const int npairs = b->npairs;
      std::vector<int> pair_idx((size_t)std::max(np * np, 1), -1);
      int ndiag = 0;
      for (int pr = 0; pr < npairs; ++pr) { pair_idx[(size_t)ps1[pr] * np + ps2[pr]] = pr; ndiag += ps1[pr] == ps2[pr]; }
      bool ok = npairs > 0 && !kn.schur_chunks;
      std::vector<int> bat_e0(1, 0), bat_of_point(P, 0);
      for (int p = 0, cur = 0, curt = 0; p < P && ok; ++p) {
        const int ne = pt_off[p + 1] - pt_off[p];
        int nfree = 0;
        for (int a = pt_off[p]; a < pt_off[p + 1]; ++a) nfree += pose_slot[s_pose[a]] >= 0;
        const int nt = nfree * (nfree + 1) / 2;                    // tuples of this point
        if (ne > BA_SP_MAXE || nt > BA_SP_MAXT) { ok = false; break; }
        if (cur + ne > BA_SP_MAXE || curt + nt > BA_SP_MAXT) { bat_e0.push_back(pt_off[p]); cur = 0; curt = 0; }
        cur += ne; curt += nt;
        bat_of_point[p] = (int)bat_e0.size() - 1;
      }
      bat_e0.push_back(E);
      const int nbat = (int)bat_e0.size() - 1;
      int nhd = 1;
      if (ok) {
        nhd = ndiag > 0 ? std::min(8, std::min(BA_SP_MAXE / ndiag, (BA_SP_MAX_THREADS - (npairs - ndiag)) / ndiag)) : 1;
        if (nhd < 1 || ndiag * nhd + (npairs - ndiag) > BA_SP_MAX_THREADS) ok = false;
      }
      if (ok) {
        // slots: helpers of the diagonal pairs first, then one slot per off-diagonal pair
        std::vector<int> slot_pair;
        std::vector<int> order;
        for (int pr = 0; pr < npairs; ++pr) if (ps1[pr] == ps2[pr]) order.push_back(pr);
        const int nd_slots = ndiag * nhd;
        for (int pr = 0; pr < npairs; ++pr) if (ps1[pr] != ps2[pr]) order.push_back(pr);
        std::vector<int> slot0_of(npairs, 0);
        for (int pr : order) { slot0_of[pr] = (int)slot_pair.size(); for (int h = 0; h < (ps1[pr] == ps2[pr] ? nhd : 1); ++h) slot_pair.push_back(pr); }
        const int nslots = (int)slot_pair.size();
        // helpers of a pair are consecutive slots: [first, end) per pair id
        std::vector<int> slot_first(npairs), slot_end(npairs);
        for (int pr = 0; pr < npairs; ++pr) { slot_first[pr] = slot0_of[pr]; slot_end[pr] = slot0_of[pr] + (ps1[pr] == ps2[pr] ? nhd : 1); }
        const int bpw = std::max(1, (nbat + BA_SP_RANGES - 1) / BA_SP_RANGES);
        const int R = (nbat + bpw - 1) / bpw;
        // per batch: tuples grouped by slot (point order inside a slot), 16-bit words a1 | a2 << 8, two per u32; and the
        // nslots + 1 offsets of the slots inside the batch's tuples, 16 bit each
        const int off_stride = (nslots + 2) / 2;
        std::vector<int> rr((size_t)nbat * std::max(npairs, 1), 0);
        auto slot_of = [&](int bt, int pr) {
          if (ps1[pr] != ps2[pr]) return slot_first[pr];
          const int h = rr[(size_t)bt * npairs + pr]++ % nhd;     // round robin over the helpers, per batch
          return slot_first[pr] + h;
        };
        // batch by batch (a batch's points are consecutive): the tuples with their slot, then a stable counting sort by slot
        std::vector<int> tup_base(nbat + 1, 0);
        std::vector<uint16_t> tup16, off16((size_t)nbat * off_stride * 2, 0);
        tup16.reserve((size_t)E * 3);
        std::vector<std::pair<int, uint16_t>> v;
        std::vector<int> cnt((size_t)nslots + 2);
        for (int bt = 0, p = 0; bt < nbat; ++bt) {
          const int e0 = bat_e0[bt];
          v.clear();
          for (; p < P && bat_of_point[p] == bt; ++p) {
            for (int a1 = pt_off[p]; a1 < pt_off[p + 1]; ++a1) {
              const int s1 = pose_slot[s_pose[a1]];
              if (s1 < 0) continue;
              for (int a2 = pt_off[p]; a2 < pt_off[p + 1]; ++a2) {
                const int s2 = pose_slot[s_pose[a2]];
                if (s2 < 0 || s2 < s1) continue;
                v.push_back(std::make_pair(slot_of(bt, pair_idx[(size_t)s1 * np + s2]), (uint16_t)((a1 - e0) | ((a2 - e0) << 8))));
              }
            }
          }
          std::fill(cnt.begin(), cnt.end(), 0);
          for (const auto& x : v) ++cnt[x.first + 1];
          for (int sl = 0; sl <= nslots; ++sl) cnt[sl + 1] += cnt[sl];
          uint16_t* off = &off16[(size_t)bt * off_stride * 2];
          for (int sl = 0; sl <= nslots; ++sl) off[sl] = (uint16_t)cnt[sl];
          tup_base[bt] = (int)(tup16.size() / 2);
          const size_t base = tup16.size();
          tup16.resize(base + v.size());
          for (const auto& x : v) tup16[base + cnt[x.first]++] = x.second;
          if (tup16.size() & 1) tup16.push_back(0);
        }
        tup_base[nbat] = (int)(tup16.size() / 2);
        std::vector<int> ps0(2 * (size_t)npairs);
        for (int pr = 0; pr < npairs; ++pr) { ps0[2 * pr] = slot_first[pr]; ps0[2 * pr + 1] = slot_end[pr]; }
        BA_TRY(ba_alloc(b, &b->d_sp_bat_e0, bat_e0.size())); BA_TRY(ba_alloc(b, &b->d_sp_off, off16.size() / 2));
        BA_TRY(ba_alloc(b, &b->d_sp_list, tup16.size() / 2)); BA_TRY(ba_alloc(b, &b->d_sp_tup_base, tup_base.size())); BA_TRY(ba_alloc(b, &b->d_sp_slot_pair, slot_pair.size()));
        BA_TRY(ba_alloc(b, &b->d_sp_pair_slots, ps0.size())); BA_TRY(ba_alloc(b, &b->d_sp_partial, (size_t)R * npairs * 42));
        BA_TRY(ba_alloc(b, &b->d_sp_sum, (size_t)npairs * 42)); BA_TRY(ba_alloc(b, &b->d_sp_chunk_off, (size_t)npairs + 1));
        BA_HIP(hipMemcpy(b->d_sp_bat_e0, bat_e0.data(), bat_e0.size() * sizeof(int), hipMemcpyHostToDevice));
        BA_HIP(hipMemcpy(b->d_sp_off, off16.data(), off16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        if (!tup16.empty()) BA_HIP(hipMemcpy(b->d_sp_list, tup16.data(), tup16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        BA_HIP(hipMemcpy(b->d_sp_tup_base, tup_base.data(), tup_base.size() * sizeof(int), hipMemcpyHostToDevice));
        BA_HIP(hipMemcpy(b->d_sp_slot_pair, slot_pair.data(), slot_pair.size() * sizeof(int), hipMemcpyHostToDevice));
        BA_HIP(hipMemcpy(b->d_sp_pair_slots, ps0.data(), ps0.size() * sizeof(int), hipMemcpyHostToDevice));
        std::vector<int> ident(npairs + 1);
        for (int i = 0; i <= npairs; ++i) ident[i] = i;          // "one chunk per pair" for the solve kernel's assembly
        BA_HIP(hipMemcpy(b->d_sp_chunk_off, ident.data(), ident.size() * sizeof(int), hipMemcpyHostToDevice));
        BaSp& sp = b->sp;
        sp.nbat = nbat; sp.bpw = bpw; sp.nslots = nslots; sp.npairs = npairs; sp.R = R; sp.nd_slots = nd_slots;
        sp.bat_e0 = b->d_sp_bat_e0; sp.off32 = b->d_sp_off; sp.tup32 = b->d_sp_list; sp.tup_base = b->d_sp_tup_base; sp.off_stride = off_stride;
        sp.slot_pair = b->d_sp_slot_pair;
        sp.pair_slots = b->d_sp_pair_slots; sp.partial = b->d_sp_partial;
        b->sp_threads = (nslots + 63) / 64 * 64 + BA_SP_STAGERS;   // owner wavefronts + the staging team
        b->sp_lds = (size_t)BA_SP_MAXE * BA_SP_ROW * sizeof(double) + BA_SP_MAXT * 2 + (BA_SP_MAX_THREADS + 4) * 2;
      }
    }
    tick("sp");
    std::vector<int> pcoff(1, 0);
    std::vector<int2> crange;
    for (size_t pr = 0; pr + 1 < poff.size(); ++pr) {
      for (int t0 = poff[pr]; t0 < poff[pr + 1]; t0 += BA_TUP_CHUNK) crange.push_back(make_int2(t0, std::min(poff[pr + 1], t0 + BA_TUP_CHUNK)));
      pcoff.push_back((int)crange.size());
    }
    b->nchunks = (int)crange.size();
    BA_TRY(ba_alloc(b, &b->d_pair_chunk_off, pcoff.size())); BA_TRY(ba_alloc(b, &b->d_chunk_range, crange.size()));
    BA_TRY(ba_alloc(b, &b->d_chunk_sum, crange.size() * 42));
    BA_HIP(hipMemcpy(b->d_pair_chunk_off, pcoff.data(), pcoff.size() * sizeof(int), hipMemcpyHostToDevice));
    if (!crange.empty()) BA_HIP(hipMemcpy(b->d_chunk_range, crange.data(), crange.size() * sizeof(int2), hipMemcpyHostToDevice));
    BA_TRY(ba_alloc(b, &b->d_pair_s1, ps1.size())); BA_TRY(ba_alloc(b, &b->d_pair_s2, ps2.size()));
    BA_TRY(ba_alloc(b, &b->d_pair_off, poff.size())); BA_TRY(ba_alloc(b, &b->d_tup, tt.size()));
    if (!ps1.empty()) {
      BA_HIP(hipMemcpy(b->d_pair_s1, ps1.data(), ps1.size() * sizeof(int), hipMemcpyHostToDevice));
      BA_HIP(hipMemcpy(b->d_pair_s2, ps2.data(), ps2.size() * sizeof(int), hipMemcpyHostToDevice));
      BA_HIP(hipMemcpy(b->d_tup, tt.data(), tt.size() * sizeof(int2), hipMemcpyHostToDevice));
    }
    BA_HIP(hipMemcpy(b->d_pair_off, poff.data(), poff.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  tick("tuple-upload");
  {
    const int NP = 192;
    b->solve_lds = ((size_t)n * (n + 1) / 2 + 4 * (size_t)NP + 8) * sizeof(double);
    b->solve_in_lds = n <= 192 && b->solve_lds <= BA_LDS_CEILING;
  }
  // normalise quaternions like the SE3Quat constructor (se3quat.h:58-64, 280-285)
  std::vector<double> p0(poses, poses + 7 * (size_t)K);
  for (int k = 0; k < K; ++k) {
    double* q = &p0[7 * k + 3];
    if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
    const double nn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= nn;
  }
  // The caller's measurements, informations and point positions travel AS THEY ARE (sequential copies into the staging block) together with the
  // two permutations; k_ba_gather puts them into the internal edge / point order on the device.  On the host those three gathers were four
  // random cache lines per point over arrays no other window shares: 1.2 of cms_ba_create's 2.8 ms alone and 2.0-2.7 of ~5 ms when 32 host
  // threads build windows side by side (memory bound) -- the part a host inside a CPU quota could least afford.
  up(fixed, K, &b->d_fixed); up(pose_slot.data(), K * sizeof(int), &b->d_pose_slot);
  up_in(e_obs, 2 * (size_t)E * sizeof(double), &b->d_raw_obs); up_in(e_invsig2, E * sizeof(double), &b->d_raw_inv);
  up(p0.data(), 7 * (size_t)K * sizeof(double), &b->d_poses0); up_in(points, 3 * (size_t)P * sizeof(double), &b->d_raw_pts);
  std::vector<int> zero_off;
  if (fast) {
    // the caller's index arrays as they are; sorted edge arrays, per-edge words and the edge permutation are written by k_ba_expand_edges
    up_in(e_pose, E * sizeof(int), &b->d_raw_pose); up_in(e_point, E * sizeof(int), &b->d_raw_point); up_in(e_face, E, &b->d_raw_face);
    if (devp) {
      BA_TRY(ba_alloc(b, &b->d_pt_off, (size_t)P + 1)); BA_TRY(ba_alloc(b, &b->d_prank, (size_t)P)); BA_TRY(ba_alloc(b, &b->d_pinv, (size_t)P));
      BA_TRY(ba_alloc(b, &b->d_cpo, (size_t)P + 1)); BA_TRY(ba_alloc(b, &b->d_pcopy, (size_t)P)); BA_TRY(ba_alloc(b, &b->d_lo_copy, (size_t)E));
      BA_TRY(ba_alloc(b, &b->d_cedge, (size_t)E)); BA_TRY(ba_alloc(b, &b->d_plan_counts, BA_DP_COUNTS));
      BA_HIP(ba_pin_take(device, BA_DP_COUNTS * sizeof(int), (void**)&b->h_plan_counts, &b->h_plan_counts_bytes));
      b->h_plan_counts[0] = 0;
    } else {
    up(fp.pt_off.data(), (P + 1) * sizeof(int), &b->d_pt_off); up(fp.prank.data(), P * sizeof(int), &b->d_prank); up(fp.pinv.data(), P * sizeof(int), &b->d_pinv);
    up(fp.cpo.data(), (P + 1) * sizeof(int), &b->d_cpo); up(fp.pcopy.data(), P, &b->d_pcopy); up(fp.lo_copy.data(), fp.lo_copy.size(), &b->d_lo_copy);
    if (!fp.grouped) up(fp.cedge.data(), E * sizeof(int), &b->d_cedge);
    }
    // (the per-key-frame edge lists serve kb_ba_lin, which a group of such windows never launches: an empty CSR)
    zero_off.assign(K + 1, 0);
    up(zero_off.data(), (K + 1) * sizeof(int), &b->d_pose_off);
    BA_TRY(ba_alloc(b, &b->d_e_pose, (size_t)E)); BA_TRY(ba_alloc(b, &b->d_e_point, (size_t)E)); BA_TRY(ba_alloc(b, &b->d_e_face, (size_t)E));
    BA_TRY(ba_alloc(b, &b->d_perm, (size_t)E)); BA_TRY(ba_alloc(b, &b->d_pose_edges, 1));
    BA_TRY(ba_alloc(b, &b->d_iperm, (size_t)E));
  } else {
    up(s_pose.data(), E * sizeof(int), &b->d_e_pose); up(s_point.data(), E * sizeof(int), &b->d_e_point);
    up(s_face.data(), E, &b->d_e_face); up(pt_off.data(), (P + 1) * sizeof(int), &b->d_pt_off);
    up(pose_off.data(), (K + 1) * sizeof(int), &b->d_pose_off); up(pose_edges.data(), E * sizeof(int), &b->d_pose_edges);
    up(b->perm.data(), E * sizeof(int), &b->d_perm); up(b->pinv.data(), P * sizeof(int), &b->d_pinv);
  }
  BA_TRY(ba_alloc(b, &b->d_e_obs, 2 * (size_t)E)); BA_TRY(ba_alloc(b, &b->d_e_inv, (size_t)E)); BA_TRY(ba_alloc(b, &b->d_pts0, 3 * (size_t)P));
  // ---- buffers the device only writes / works in
  BA_TRY(ba_alloc(b, &b->d_level, E)); BA_TRY(ba_alloc(b, &b->d_err, 2 * (size_t)E)); BA_TRY(ba_alloc(b, &b->d_ow, (size_t)E));
  for (int i = 0; i < 2; ++i) { BA_TRY(ba_alloc(b, &b->d_poses[i], 7 * (size_t)K)); BA_TRY(ba_alloc(b, &b->d_pts[i], 3 * (size_t)P)); }
  BA_TRY(ba_alloc(b, &b->d_Hpp, 36 * (size_t)std::max(np, 1))); BA_TRY(ba_alloc(b, &b->d_bp, 6 * (size_t)std::max(np, 1)));
  BA_TRY(ba_alloc(b, &b->d_Hll, 9 * (size_t)P)); BA_TRY(ba_alloc(b, &b->d_bl, 3 * (size_t)P));
  BA_TRY(ba_alloc(b, &b->d_Dinv, 9 * (size_t)P)); BA_TRY(ba_alloc(b, &b->d_Hs, (size_t)std::max(n * n, 1))); BA_TRY(ba_alloc(b, &b->d_bs, std::max(n, 1)));
  BA_TRY(ba_alloc(b, &b->d_x, std::max(n, 1))); BA_TRY(ba_alloc(b, &b->d_Dg, std::max(n, 1)));
  BA_TRY(ba_alloc(b, &b->d_partial, 2 * (size_t)std::max(std::max(b->nblk_e, b->nblk_p), (E + 63) / 64 + 8) + 8)); BA_TRY(ba_alloc(b, &b->d_scal, 24));   // [8..16): developer clocks of k_ba_trial_solve
  BA_TRY(ba_alloc(b, &b->d_flags, E));
  b->d_status = reinterpret_cast<int*>(b->d_scal + 4);   // solver status travels with the scalars
  BA_HIP(ba_pin_take(device, 8 * sizeof(double), (void**)&b->h_pin, &b->h_pin_bytes));
  BA_TRY(ba_alloc(b, &b->d_pose_partial, (size_t)std::max(np, 1) * BA_POSE_CHUNKS * 27)); BA_TRY(ba_alloc(b, &b->d_db, 3 * (size_t)P));
  tick("alloc");
  // ---- the one staging block and the one copy
  {
    size_t total = 0, staged = 0;
    std::vector<size_t> offs(ups.size());
    for (size_t i = 0; i < ups.size(); ++i) if (!ups[i].direct) { offs[i] = total; total += (ups[i].bytes + 255) & ~(size_t)255; }
    staged = total;                                               // the staged entries first (one copy), the caller's pinned arrays behind them (one copy each)
    for (size_t i = 0; i < ups.size(); ++i) if (ups[i].direct) { offs[i] = total; total += (ups[i].bytes + 255) & ~(size_t)255; }
    // (the read-back of cms_ba_read reuses the pinned block: poses, points, flags)
    const size_t rd_bytes = (size_t)7 * K * 8 + 255 + (size_t)3 * P * 8 + 255 + (size_t)E + 255;
    char* dev = nullptr;
    BA_TRY(ba_alloc(b, &dev, std::max(total, (size_t)256)));
    BA_HIP(ba_stage_take(device, std::max(staged, rd_bytes), (void**)&b->h_stage, &b->h_stage_bytes));
    for (size_t i = 0; i < ups.size(); ++i) {
      if (ups[i].bytes && !ups[i].direct) memcpy(b->h_stage + offs[i], ups[i].src, ups[i].bytes);
      *ups[i].dst = dev + offs[i];
    }
    if (staged) BA_HIP(hipMemcpyAsync(dev, b->h_stage, staged, hipMemcpyHostToDevice, b->stream));
    for (size_t i = 0; i < ups.size(); ++i)
      if (ups[i].direct && ups[i].bytes) BA_HIP(hipMemcpyAsync(dev + offs[i], ups[i].src, ups[i].bytes, hipMemcpyHostToDevice, b->stream));
    b->async_pending = true;
  }
  BaDev& d = b->d;
  d.K = K; d.P = P; d.E = E; d.np = np; d.fixed = b->d_fixed; d.pose_slot = b->d_pose_slot; d.e_pose = b->d_e_pose;
  d.e_point = b->d_e_point; d.e_obs = b->d_e_obs; d.e_inv = b->d_e_inv; d.e_face = b->d_e_face; d.pt_off = b->d_pt_off;
  d.pose_off = b->d_pose_off; d.pose_edges = b->d_pose_edges; d.level = b->d_level; d.err = b->d_err; d.ow = b->d_ow;
  d.fx = fx; d.fy = fy; d.cx = cx; d.cy = cy;
  if (se_built) {
    BaSe& se = b->se;
    se.chunk_e0 = b->d_se_chunk_e0; se.e_info = b->d_se_info; se.lone = b->d_se_lone; se.rm_chunk = b->d_rm_chunk; se.run_lane = b->d_run_lane; se.run_mf = b->d_run_mf; se.run_fl = b->d_run_fl; se.rm_cost = b->d_rm_cost; se.run_fg = b->d_run_fg; se.rm_cut = b->d_rm_cut;
  }
  tick("uploads");
  b->cur = 0;
  if (fast) {
    BaExpand x;
    x.K = K; x.P = P; x.E = E; x.np = np;
    x.e_pose = b->d_raw_pose; x.e_point = b->d_raw_point; x.e_face = b->d_raw_face; x.cedge = devp ? b->d_cedge : fp.grouped ? nullptr : b->d_cedge; x.cpo = b->d_cpo;
    x.prank = b->d_prank; x.pinv = b->d_pinv; x.pt_off = b->d_pt_off; x.pcopy = b->d_pcopy; x.lo_copy = b->d_lo_copy; x.e_lo0 = devp ? 0 : fp.pt_off[fp.P_rm]; x.pose_slot = b->d_pose_slot;
    x.dcounts = devp ? b->d_plan_counts : nullptr;             // (a window planned on the device: the counts below and `grouped` are read from there)
    x.raw_obs = b->d_raw_obs; x.raw_inv = b->d_raw_inv; x.raw_pts = b->d_raw_pts; x.poses0 = b->d_poses0;
    x.perm = b->d_perm; x.iperm = b->d_iperm; x.s_pose = b->d_e_pose; x.s_point = b->d_e_point; x.s_face = b->d_e_face; x.info = b->d_se_info;
    x.e_obs = b->d_e_obs; x.e_inv = b->d_e_inv; x.pts0 = b->d_pts0; x.poses = b->d_poses[0]; x.pts = b->d_pts[0]; x.level = b->d_level; x.err = b->d_err; x.flags = b->d_flags;
    x.gsum = b->d_se_partial; x.n_gsum = b->se.npairs2 * 42; x.gsum_bp = b->d_se_bp_partial; x.n_gsum_bp = b->np * 6;
    x.ce0 = b->d_se_chunk_e0; x.n_rm = devp ? 0 : fp.n_rm; x.nchunks = devp ? 0 : fp.nchunks; x.run_sig = b->d_run_sig; x.n_runs = devp ? 0 : fp.n_runs; x.run_mf = b->d_run_mf; x.run_fl = b->d_run_fl; x.run_fg = b->d_run_fg;
    if (devp) {
      // scratch of the plan kernel + its description; cms_ba_create_many launches it in front of the expansion
      BaDevPlan dp;
      memset(&dp, 0, sizeof(dp));
      dp.K = K; dp.P = P; dp.E = E; dp.np = np; dp.chunk_cap = dp_chunk_cap; dp.em_cost_a = kn.em_cost_a; dp.em_cost_b = kn.em_cost_b; dp.free_mask = dp_free_mask;
      dp.e_pose = b->d_raw_pose; dp.e_point = b->d_raw_point; dp.e_face = b->d_raw_face; dp.pose_slot = b->d_pose_slot;
      BA_TRY(ba_alloc(b, &dp.cnt, (size_t)P)); BA_TRY(ba_alloc(b, &dp.sig, (size_t)P)); BA_TRY(ba_alloc(b, &dp.sig_i, (size_t)P)); BA_TRY(ba_alloc(b, &dp.hkey, (size_t)BA_DP_HCAP));
      BA_TRY(ba_alloc(b, &dp.gslot, (size_t)P)); BA_TRY(ba_alloc(b, &dp.ord, (size_t)P)); BA_TRY(ba_alloc(b, &dp.scan, (size_t)P + 1)); BA_TRY(ba_alloc(b, &dp.cnt_i, (size_t)P));
      BA_TRY(ba_alloc(b, &dp.seg_tmp, (size_t)P)); BA_TRY(ba_alloc(b, &dp.run_tab, (size_t)(BA_DP_RUN_CAP + 1) * 8)); BA_TRY(ba_alloc(b, &dp.chunk_pt0, (size_t)dp_chunk_cap + 1));
      BA_TRY(ba_alloc(b, &dp.chunk_run, (size_t)dp_chunk_cap + 1));
      dp.cpo = b->d_cpo; dp.cedge = b->d_cedge; dp.prank = b->d_prank; dp.pinv = b->d_pinv; dp.pt_off = b->d_pt_off; dp.pcopy = b->d_pcopy; dp.lo_copy = b->d_lo_copy;
      dp.ce0 = b->d_se_chunk_e0; dp.rm_chunk = b->d_rm_chunk; dp.rm_cost = b->d_rm_cost; dp.run_sig = reinterpret_cast<unsigned long long*>(b->d_run_sig); dp.lone = b->d_se_lone;
      dp.counts = b->d_plan_counts; dp.h_counts = b->h_plan_counts;
      static const bool dp_clk = getenv("CMS_BA_DP_CLK") != nullptr;
      if (dp_clk) { BA_TRY(ba_alloc(b, &b->d_plan_clk, 16)); dp.clk = b->d_plan_clk; }
      *dplan = dp; ba_tl_dev_plan = nullptr;
    }
    if (ba_tl_defer_expand) { *ba_tl_defer_expand = x; ba_tl_defer_expand = nullptr; }      // cms_ba_create_many launches the group's expansions together
    else hipLaunchKernelGGL(k_ba_expand_edges, dim3(std::min((std::max(E, P) + 255) / 256, 1024)), dim3(256), 0, b->stream, x);      // (+ the runs' tables)
  } else
  hipLaunchKernelGGL(k_ba_gather, dim3(std::min((std::max(E, P) + 255) / 256, 1024)), dim3(256), 0, b->stream, K, P, E, (const int*)b->d_perm, (const int*)b->d_pinv,
                     (const double*)b->d_raw_obs, (const double*)b->d_raw_inv, (const double*)b->d_raw_pts, b->d_e_obs, b->d_e_inv, b->d_pts0,
                     (const double*)b->d_poses0, b->d_poses[0], b->d_pts[0], b->d_level, b->d_err, b->d_flags,
                     b->d_se_partial, b->d_se_partial ? b->se.npairs2 * 42 : 0, b->d_se_bp_partial, b->d_se_bp_partial ? b->np * 6 : 0);
  BA_HIP(hipGetLastError());
  b->async_pending = true;
  b->gsum_clean = true;
  tick("reset");
  if (timing) fprintf(stderr, "[cms_ba_create] K %d P %d E %d ms:%s\n", K, P, E, t_log.c_str());
  *out = b;
  return CMS_OK;
}

// developer / test entry: the plan arrays AS THE DEVICE HOLDS THEM (after the window's set-up has run), whichever planner made them -- for the
// comparison of the device-side planner with cms_ba_debug_plan.  Sizes: pinv P, perm E, info E, pt_off P + 1, e_pose / e_point E, e_face E,
// chunk_e0 chunks + 1 (<= P + 2), rm_chunk 4 ints per run chunk, rm_cost chunks + 1, run_mf 64 / run_fl 768 words per run; any pointer may be NULL.
// counts[8] = chunks, run chunks, runs, free key frames, points inside runs, R_rm, R, 1 if the device-side planner made the window (2: all of it in the plan kernel).
extern "C" int cms_ba_debug_fetch_plan(cms_ba* b, int* pinv, int* perm, uint32_t* info, int* pt_off, int* e_pose, int* e_point, int8_t* e_face, int* chunk_e0,
                                       int* rm_chunk, uint32_t* rm_cost, uint32_t* run_mf, uint32_t* run_fl, int* counts) {
  if (!b || !counts) return cms_fail(CMS_ERR_ARG, "cms_ba_debug_fetch_plan: bad argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(ba_order_behind_setup(b));
  HIPCHK(hipStreamSynchronize(b->stream));
  b->async_pending = false;
  const int nch = b->se.nchunks, n_rm = b->se.n_rm, nr = b->n_runs;
  counts[0] = nch; counts[1] = n_rm; counts[2] = nr; counts[3] = b->np; counts[4] = b->rm_points; counts[5] = b->se.R_rm; counts[6] = b->se.R; counts[7] = b->dev_plan ? 2 : b->fast_plan ? 1 : 0;
  auto dl = [&](void* dst, const void* src, size_t bytes) -> int {
    if (!dst || !src || bytes == 0) return CMS_OK;
    return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? CMS_OK : cms_fail(CMS_ERR_HIP, "cms_ba_debug_fetch_plan: copy");
  };
  int rc = CMS_OK;
  if (!rc) rc = dl(pinv, b->d_pinv, (size_t)b->P * 4);
  if (!rc) rc = dl(perm, b->d_perm, (size_t)b->E * 4);
  if (!rc && nch > 0) rc = dl(info, b->d_se_info, (size_t)b->E * 4);
  if (!rc) rc = dl(pt_off, b->d_pt_off, ((size_t)b->P + 1) * 4);
  if (!rc) rc = dl(e_pose, b->d_e_pose, (size_t)b->E * 4);
  if (!rc) rc = dl(e_point, b->d_e_point, (size_t)b->E * 4);
  if (!rc) rc = dl(e_face, b->d_e_face, (size_t)b->E);
  if (!rc && nch > 0) rc = dl(chunk_e0, b->d_se_chunk_e0, ((size_t)nch + 1) * 4);
  if (!rc && n_rm > 0) rc = dl(rm_chunk, b->d_rm_chunk, (size_t)n_rm * 16);
  if (!rc && nch > 0) rc = dl(rm_cost, b->d_rm_cost, ((size_t)nch + 1) * 4);
  if (!rc && nr > 0) rc = dl(run_mf, b->d_run_mf, (size_t)nr * 64 * 4);
  if (!rc && nr > 0) rc = dl(run_fl, b->d_run_fl, (size_t)nr * 64 * 12 * 4);
  return rc;
}

// restore the initial estimate and clear the per-edge state: one launch (five copies / memsets cost more in dispatch than in work)
extern "C" __global__ void __launch_bounds__(256)
k_ba_reset(int K, int P, int E, const double* __restrict__ poses0, const double* __restrict__ pts0, double* __restrict__ poses,
           double* __restrict__ pts, uint8_t* __restrict__ level, double* __restrict__ err, uint8_t* __restrict__ flags,
           double* __restrict__ gsum, int n_gsum, double* __restrict__ gsum_bp, int n_gsum_bp) {
  const int gs = gridDim.x * blockDim.x;
  // the global copy of the reduced system the Schur kernel's workgroups add to (BaSe::gsum): zero before the first round; every round's
  // solve kernel leaves it zero again
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_gsum; i += gs) gsum[i] = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_gsum_bp; i += gs) gsum_bp[i] = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * E; i += gs) {
    err[i] = 0.0;
    if (i < E) { level[i] = 0; flags[i] = 0; }
    if (i < 7 * K) poses[i] = poses0[i];
    if (i < 3 * P) pts[i] = pts0[i];
  }
  for (int i = 2 * E + blockIdx.x * blockDim.x + threadIdx.x; i < 3 * P; i += gs) pts[i] = pts0[i];      // P large relative to E
  for (int i = 2 * E + blockIdx.x * blockDim.x + threadIdx.x; i < 7 * K; i += gs) poses[i] = poses0[i];
}
extern "C" int cms_ba_reset(cms_ba* b) {
  if (!b) return cms_fail(CMS_ERR_ARG, "null ba");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(ba_order_behind_setup(b));
  b->cur = 0;
  const int n = std::max(2 * b->E, std::max(3 * b->P, 7 * b->K));
  hipLaunchKernelGGL(k_ba_reset, dim3(std::min((n + 255) / 256, 1024)), dim3(256), 0, b->stream, b->K, b->P, b->E, (const double*)b->d_poses0,
                     (const double*)b->d_pts0, b->d_poses[0], b->d_pts[0], b->d_level, b->d_err, b->d_flags,
                     b->d_se_partial, b->d_se_partial ? b->se.npairs2 * 42 : 0, b->d_se_bp_partial, b->d_se_bp_partial ? b->np * 6 : 0);
  HIPCHK(hipGetLastError());
  b->async_pending = true;
  b->gsum_clean = true;
  return CMS_OK;       // asynchronous on the window's stream: every consumer (optimize, read) orders itself behind it
}

// chi2 of the active edges at state `which` -> d_scal[slot]; refreshes d_err
static void ba_errors(cms_ba* b, int which, int robust, double delta, int slot) {
  hipLaunchKernelGGL(k_ba_errors, dim3(b->nblk_e), dim3(256), 0, b->stream, b->d, (const double*)b->d_poses[which],
                     (const double*)b->d_pts[which], robust, delta, b->d_partial);
  hipLaunchKernelGGL(k_ba_reduce, dim3(1), dim3(256), 0, b->stream, (const double*)b->d_partial, b->nblk_e, b->d_scal + slot, 0);
}

// Levenberg-Marquardt driver (lock-step over one or many windows)
#include "cms_api_ba_lm.hip"

extern "C" int cms_ba_read(cms_ba* b, double* poses, double* points, uint8_t* outlier_flags) {
  if (!b) return cms_fail(CMS_ERR_ARG, "null ba");
  HIPCHK(hipSetDevice(b->device));
  // one pinned block (the window's upload staging, sized for this at creation), up to three copies, one synchronisation
  const size_t o_pose = 0, o_pts = ((size_t)7 * b->K * 8 + 255) & ~(size_t)255, o_flags = o_pts + (((size_t)3 * b->P * 8 + 255) & ~(size_t)255);
  if (!b->h_stage || b->h_stage_bytes < o_flags + (size_t)b->E) return cms_fail(CMS_ERR_HIP, "cms_ba_read: staging block missing");
  char* h = b->h_stage;
  // A window that runs on a stream it shares with others (cms_ba_set_stream) reads back on a stream taken from the pool: the shared one may
  // be busy with the next windows for milliseconds, and this window's results are complete (optimise returned) unless a reset is pending
  HIPCHK(ba_order_behind_setup(b));
  hipStream_t rs = b->stream;
  bool temp = false;
  if (!b->own_stream && !b->async_pending) {
    rs = ba_stream_take(b->device);
    if (!rs) HIPCHK(hipStreamCreateWithFlags(&rs, hipStreamNonBlocking));
    temp = true;
  }
  hipError_t re = hipSuccess;
  if (b->fast_plan) {      // the permutations live on the device: one kernel gathers everything in the caller's order straight into the pinned block
    hipLaunchKernelGGL(k_ba_results_to_host, dim3(std::min((std::max(b->E / 4, 3 * b->P) + 255) / 256, 512)), dim3(256), 0, rs, b->K, b->P, b->E, (const int*)b->d_prank,
                       (const int*)b->d_iperm, (const double*)b->d_poses[b->cur], (const double*)b->d_pts[b->cur], (const uint8_t*)b->d_flags,
                       poses ? reinterpret_cast<double*>(h + o_pose) : nullptr, points ? reinterpret_cast<double*>(h + o_pts) : nullptr,
                       outlier_flags ? reinterpret_cast<uint8_t*>(h + o_flags) : nullptr);
    re = hipGetLastError();
  } else {
    if (poses && re == hipSuccess) re = hipMemcpyAsync(h + o_pose, b->d_poses[b->cur], 7 * (size_t)b->K * sizeof(double), hipMemcpyDeviceToHost, rs);
    if (points && re == hipSuccess) re = hipMemcpyAsync(h + o_pts, b->d_pts[b->cur], 3 * (size_t)b->P * sizeof(double), hipMemcpyDeviceToHost, rs);
    if (outlier_flags && re == hipSuccess) re = hipMemcpyAsync(h + o_flags, b->d_flags, b->E, hipMemcpyDeviceToHost, rs);
  }
  if (re == hipSuccess) re = ba_wait_stream(rs);
  if (temp) ba_stream_give(b->device, rs);
  HIPCHK(re);
  b->async_pending = false;
  if (poses) memcpy(poses, h + o_pose, 7 * (size_t)b->K * sizeof(double));
  if (points && b->fast_plan) memcpy(points, h + o_pts, 3 * (size_t)b->P * sizeof(double));
  else if (points) {
    const double* pin = reinterpret_cast<const double*>(h + o_pts);
    for (int i = 0; i < b->P; ++i) for (int j = 0; j < 3; ++j) points[3 * (size_t)b->pinv[i] + j] = pin[3 * (size_t)i + j];
  }
  if (outlier_flags && b->fast_plan) memcpy(outlier_flags, h + o_flags, (size_t)b->E);
  else if (outlier_flags) {
    const uint8_t* f = reinterpret_cast<const uint8_t*>(h + o_flags);
    for (int i = 0; i < b->E; ++i) outlier_flags[b->perm[i]] = f[i];
  }
  return CMS_OK;
}

// ---- a window GROUP's set-up and read-back as one call each (round 6): what LocalMapping threads of many camera streams on one GPU do per step
// (Optimizer.cpp:246-357 assembles one window; :419-450 writes one back).  The host parts of the n windows run on up to `threads` threads of the
// call, the device parts of all device-planned windows are ONE launch per eight windows instead of one per window.
static_assert(sizeof(BaExpandBatch) <= 4096 && sizeof(BaReadBatch) <= 4096, "the batches travel as kernel arguments");
// standing host threads for the windows' host parts: threads that live for one call each would take the HIP runtime's per-thread state (and this
// file's thread-local plan buffers) up and down sixteen times per step -- and the runtime's tear-down of short-lived threads corrupted the heap
// under bench.py's load (glibc: double free, within a few steps)
struct BaWorkers {
  std::mutex mu; std::condition_variable cv;
  std::deque<std::function<void()>> jobs;
  int nthreads = 0;
  void ensure(int n) {
    std::lock_guard<std::mutex> lk(mu);
    for (; nthreads < std::min(n, 16); ++nthreads)
      std::thread([this]() {
        pthread_setname_np(pthread_self(), "cms-ba-plan");
        for (;;) {
          std::function<void()> job;
          { std::unique_lock<std::mutex> lk2(mu); cv.wait(lk2, [this]() { return !jobs.empty(); }); job = std::move(jobs.front()); jobs.pop_front(); }
          job();
        }
      }).detach();
  }
  void post(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu); jobs.push_back(std::move(f)); } cv.notify_one(); }
};
static BaWorkers& ba_workers() { static BaWorkers* w = new BaWorkers; return *w; }      // never destroyed: its threads wait for work until the process ends
extern "C" int cms_ba_create_many(cms_ba** out, int n, int device, const cms_ba_window* w, int threads) {
  if (!out || n < 0 || (n > 0 && !w)) return cms_fail(CMS_ERR_ARG, "cms_ba_create_many: bad argument");
  for (int i = 0; i < n; ++i) out[i] = nullptr;
  if (n == 0) return CMS_OK;
  std::vector<BaExpand> xs((size_t)n);
  std::vector<BaDevPlan> dps((size_t)n);
  static const bool dev_plan_all = getenv("CMS_BA_DEV_PLAN") != nullptr;      // A/B: every window of every call, whatever its flags say
  std::vector<char> deferred((size_t)n, 0);
  std::vector<int> rcs((size_t)n, CMS_OK);
  std::vector<std::string> errs((size_t)n);
  std::atomic<int> next(0);
  // The windows a thread of this call builds are set up on ONE stream (uploads, the plan kernel, the expansion), taken from a small pool of set-up streams:
  // every window on a pooled stream of its own spread a group's ~100 upload commands over all hardware queues, in front of whatever the frame path and the
  // Levenberg rounds had there -- measured on bench.py's step (profiles/r06_bench_runs.txt): 21.2 -> 22.4 k frames/s with four host cores, and what makes the
  // plan kernel pay (17.3 -> 20.5 k with two).  A window keeps the stream until cms_ba_set_stream moves it.  CMS_BA_SETUP_OWN_STREAMS=1: as before (A/B).
  static const bool shared_stream = getenv("CMS_BA_SETUP_OWN_STREAMS") == nullptr;
  std::mutex ss_mu; std::vector<hipStream_t> ss_taken;
  auto work = [&]() {
    if (shared_stream && hipSetDevice(device) == hipSuccess) {
      hipStream_t st = ba_setup_stream_take(device);
      ba_tl_setup_stream = st;
      if (st) { std::lock_guard<std::mutex> lk(ss_mu); ss_taken.push_back(st); }
    }
    struct Reset { ~Reset() { ba_tl_setup_stream = nullptr; } } reset_;
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) break;
      memset(&xs[(size_t)i], 0, sizeof(BaExpand));
      ba_tl_defer_expand = &xs[(size_t)i];
      ba_tl_dev_plan = (dev_plan_all || (w[i].flags & CMS_BA_PLAN_ON_DEVICE) != 0) ? &dps[(size_t)i] : nullptr;
      ba_tl_inputs_pinned = (w[i].flags & CMS_BA_INPUTS_PINNED) != 0;
      rcs[(size_t)i] = cms_ba_create(&out[i], device, w[i].K, w[i].poses, w[i].fixed, w[i].P, w[i].points, w[i].E, w[i].e_pose, w[i].e_point, w[i].e_obs, w[i].e_invsig2,
                                     w[i].e_face, w[i].fx, w[i].fy, w[i].cx, w[i].cy);
      deferred[(size_t)i] = ba_tl_defer_expand == nullptr && rcs[(size_t)i] == CMS_OK;      // (taken: a device-planned window; a host-planned one launched its own set-up kernel)
      ba_tl_defer_expand = nullptr; ba_tl_inputs_pinned = false; ba_tl_dev_plan = nullptr;
      if (rcs[(size_t)i] != CMS_OK) errs[(size_t)i] = cms_last_error();
    }
  };
  {
    const int T = std::max(1, std::min(threads > 0 ? threads : 4, n));
    std::mutex dmu; std::condition_variable dcv; int running = T - 1;
    if (T > 1) ba_workers().ensure(T - 1);
    for (int t = 1; t < T; ++t)
      ba_workers().post([&]() { work(); std::lock_guard<std::mutex> lk(dmu); if (--running == 0) dcv.notify_one(); });
    work();
    std::unique_lock<std::mutex> lk(dmu);
    dcv.wait(lk, [&]() { return running == 0; });
  }
  struct GiveBack { std::vector<hipStream_t>& v; int dev; ~GiveBack() { for (hipStream_t st : v) ba_setup_stream_give(dev, st); } } give_back_{ss_taken, device};      // (with work queued: the next call's set-ups line up behind it)
  auto fail_all = [&](int rc, const char* msg) {
    for (int i = 0; i < n; ++i) if (out[i]) { cms_ba_destroy(out[i]); out[i] = nullptr; }
    return cms_fail(rc, msg);
  };
  for (int i = 0; i < n; ++i) if (rcs[(size_t)i] != CMS_OK) return fail_all(rcs[(size_t)i], errs[(size_t)i].c_str());
  // ---- the deferred expansions: batches of eight on the first window's stream, behind every window's upload; the other windows' streams then wait for it
  std::vector<int> D;
  for (int i = 0; i < n; ++i) if (deferred[(size_t)i]) D.push_back(i);
  if (D.empty()) return CMS_OK;
  if (hipSetDevice(device) != hipSuccess) return fail_all(CMS_ERR_HIP, "cms_ba_create_many: hipSetDevice");
  static_assert(BA_DP_BATCH == BA_EXPAND_BATCH, "a batch of expansions has one batch of plans in front of it");
  std::vector<hipStream_t> plan_streams;
  for (size_t b0 = 0; b0 < D.size(); b0 += BA_EXPAND_BATCH) {
    const int nb = (int)std::min<size_t>(BA_EXPAND_BATCH, D.size() - b0);
    cms_ba* lead = out[D[b0]];
    hipEvent_t ev = ba_event_take(device);
    if (!ev) return fail_all(CMS_ERR_HIP, "cms_ba_create_many: no event");
    BaExpandBatch batch;
    int maxEP = 1;
    hipError_t re = hipSuccess;
    for (int k = 0; k < nb; ++k) {
      cms_ba* b = out[D[b0 + k]];
      batch.x[k] = xs[(size_t)D[b0 + k]];
      maxEP = std::max(maxEP, std::max(b->E, b->P));
      if (k > 0 && b->stream != lead->stream && re == hipSuccess) {      // the lead's stream waits for this window's upload
        re = hipEventRecord(ev, b->stream);
        if (re == hipSuccess) re = hipStreamWaitEvent(lead->stream, ev, 0);
      }
    }
    for (int k = nb; k < BA_EXPAND_BATCH; ++k) batch.x[k] = batch.x[0];
    {
      // the windows of the batch whose plan the device makes: one workgroup each, in front of the expansion
      BaDevPlanBatch pb;
      int nd = 0;
      for (int k = 0; k < nb; ++k) if (out[D[b0 + k]]->dev_plan) pb.x[nd++] = dps[(size_t)D[b0 + k]];
      for (int k = nd; k < BA_DP_BATCH && nd > 0; ++k) pb.x[k] = pb.x[0];
      if (nd > 0 && re == hipSuccess) {
        re = ba_dev_plan_attr_once(device);
        if (re == hipSuccess) hipLaunchKernelGGL(k_ba_plan_many, dim3(nd), dim3(BA_DP_THREADS), BA_DP_LDS, lead->stream, pb);
        if (re == hipSuccess) re = hipGetLastError();
        if (re == hipSuccess) {
          static const int m_lanes = getenv("CMS_BA_MATCH_LANES") ? atoi(getenv("CMS_BA_MATCH_LANES")) : 16;        // A/B: matchings per wavefront
          static const int m_blocks = getenv("CMS_BA_MATCH_BLOCKS") ? atoi(getenv("CMS_BA_MATCH_BLOCKS")) : BA_DP_MATCH_BLOCKS;
          for (int k = 0; k < BA_DP_BATCH; ++k) pb.x[k].pad_ = m_lanes;
          hipLaunchKernelGGL(k_ba_match_copies_many, dim3(m_blocks, nd), dim3(64), 0, lead->stream, pb);
          re = hipGetLastError();
        }
        plan_streams.push_back(lead->stream);
      }
    }
    if (re == hipSuccess) {
      hipLaunchKernelGGL(k_ba_expand_edges_many, dim3(std::min((maxEP + 255) / 256, 1024), nb), dim3(256), 0, lead->stream, batch);
      re = hipGetLastError();
    }
    if (re == hipSuccess && nb > 1) re = hipEventRecord(ev, lead->stream);
    for (int k = 1; k < nb && re == hipSuccess; ++k) {
      cms_ba* b = out[D[b0 + k]];
      if (b->stream != lead->stream) re = hipStreamWaitEvent(b->stream, ev, 0);      // ... and whatever this window's stream does next comes behind the expansion
    }
    ba_event_give(device, ev);      // (a wait that is already enqueued keeps the record it saw)
    if (re != hipSuccess) return fail_all(CMS_ERR_HIP, "cms_ba_create_many: launch failed");
  }
  // ---- windows planned on the device: the counts that size their launches are on the host once the plan kernels are through.  A window the kernel
  // gave up on (a point seen twice by a key frame, no runs, ...) is built again with the host's planners; an index out of range fails the call
  for (hipStream_t ps : plan_streams) if (ba_wait_stream(ps) != hipSuccess) return fail_all(CMS_ERR_HIP, "cms_ba_create_many: the plan kernel failed");
  for (int i : D) {
    cms_ba* b = out[i];
    if (!b->dev_plan) continue;
    const int prc = ba_dev_plan_finish(b);
    if (prc == 1) continue;
    if (prc < 0) return fail_all(CMS_ERR_ARG, "cms_ba_create: edge index / face out of range (unknown-face edges must be culled by the caller)");
    cms_ba_destroy(b); out[i] = nullptr;
    ba_tl_inputs_pinned = (w[i].flags & CMS_BA_INPUTS_PINNED) != 0;
    const int rc = cms_ba_create(&out[i], device, w[i].K, w[i].poses, w[i].fixed, w[i].P, w[i].points, w[i].E, w[i].e_pose, w[i].e_point, w[i].e_obs, w[i].e_invsig2,
                                 w[i].e_face, w[i].fx, w[i].fy, w[i].cx, w[i].cy);
    ba_tl_inputs_pinned = false;
    if (rc != CMS_OK) { const std::string msg = cms_last_error(); return fail_all(rc, msg.c_str()); }
  }
  return CMS_OK;
}

// poses / points / outlier flags of n optimised windows (any of the three arrays, or single entries of them, may be NULL): the device-planned windows'
// results are gathered in the caller's order by ONE kernel per sixteen windows straight into their pinned blocks, one stream, one wait
extern "C" int cms_ba_read_many(cms_ba** bas, int n, double** poses, double** points, uint8_t** outlier_flags) {
  if (n < 0 || (n > 0 && !bas)) return cms_fail(CMS_ERR_ARG, "cms_ba_read_many: bad argument");
  for (int i = 0; i < n; ++i) if (!bas[i]) return cms_fail(CMS_ERR_ARG, "cms_ba_read_many: null window");
  std::vector<int> F;      // windows the batched kernel takes: device-planned, results complete (nothing pending on a stream of their own), one device
  for (int i = 0; i < n; ++i) {
    cms_ba* b = bas[i];
    const size_t o_pts = ((size_t)7 * b->K * 8 + 255) & ~(size_t)255, o_flags = o_pts + (((size_t)3 * b->P * 8 + 255) & ~(size_t)255);
    const bool ok = b->fast_plan && b->device == bas[0]->device && b->h_stage && b->h_stage_bytes >= o_flags + (size_t)b->E && !b->setup_wait_pending &&
                    !(b->own_stream && b->async_pending);
    if (ok) F.push_back(i);
    else {
      const int rc = cms_ba_read(b, poses ? poses[i] : nullptr, points ? points[i] : nullptr, outlier_flags ? outlier_flags[i] : nullptr);
      if (rc) return rc;
    }
  }
  if (F.empty()) return CMS_OK;
  const int device = bas[F[0]]->device;
  HIPCHK(hipSetDevice(device));
  // windows on a shared stream with work of their group still pending there are ordered by that stream; everything else reads on a pooled stream
  hipStream_t rs = nullptr; bool temp = false;
  for (int i : F) if (bas[i]->async_pending) rs = bas[i]->stream;
  if (rs) { for (int i : F) if (bas[i]->async_pending && bas[i]->stream != rs) HIPCHK(ba_wait_stream(bas[i]->stream)); }
  else { rs = ba_stream_take(device); if (!rs) HIPCHK(hipStreamCreateWithFlags(&rs, hipStreamNonBlocking)); temp = true; }
  hipError_t re = hipSuccess;
  for (size_t b0 = 0; b0 < F.size() && re == hipSuccess; b0 += BA_READ_BATCH) {
    const int nb = (int)std::min<size_t>(BA_READ_BATCH, F.size() - b0);
    BaReadBatch batch;
    int work = 1;
    for (int k = 0; k < nb; ++k) {
      const int i = F[b0 + k];
      cms_ba* b = bas[i];
      const size_t o_pts = ((size_t)7 * b->K * 8 + 255) & ~(size_t)255, o_flags = o_pts + (((size_t)3 * b->P * 8 + 255) & ~(size_t)255);
      BaReadJob& q = batch.j[k];
      q.K = b->K; q.P = b->P; q.E = b->E; q.prank = b->d_prank; q.iperm = b->d_iperm; q.poses = b->d_poses[b->cur]; q.pts = b->d_pts[b->cur]; q.flags = b->d_flags;
      q.out_poses = (poses && poses[i]) ? reinterpret_cast<double*>(b->h_stage) : nullptr;
      q.out_pts = (points && points[i]) ? reinterpret_cast<double*>(b->h_stage + o_pts) : nullptr;
      q.out_flags = (outlier_flags && outlier_flags[i]) ? reinterpret_cast<uint8_t*>(b->h_stage + o_flags) : nullptr;
      work = std::max(work, std::max(b->E / 4, 3 * b->P));
    }
    for (int k = nb; k < BA_READ_BATCH; ++k) batch.j[k] = batch.j[0];
    hipLaunchKernelGGL(k_ba_results_to_host_many, dim3(std::min((work + 255) / 256, 256), nb), dim3(256), 0, rs, batch);
    re = hipGetLastError();
  }
  if (re == hipSuccess) re = ba_wait_stream(rs);
  if (temp) ba_stream_give(device, rs);
  HIPCHK(re);
  for (int i : F) {
    cms_ba* b = bas[i];
    const size_t o_pts = ((size_t)7 * b->K * 8 + 255) & ~(size_t)255, o_flags = o_pts + (((size_t)3 * b->P * 8 + 255) & ~(size_t)255);
    b->async_pending = false;
    if (poses && poses[i]) memcpy(poses[i], b->h_stage, 7 * (size_t)b->K * sizeof(double));
    if (points && points[i]) memcpy(points[i], b->h_stage + o_pts, 3 * (size_t)b->P * sizeof(double));
    if (outlier_flags && outlier_flags[i]) memcpy(outlier_flags[i], b->h_stage + o_flags, (size_t)b->E);
  }
  return CMS_OK;
}

extern "C" int cms_ba_run(int device, int K, double* poses, const uint8_t* fixed, int P, double* points, int E, const int* e_pose,
                          const int* e_point, const double* e_obs, const double* e_invsig2, const int8_t* e_face, double fx,
                          double fy, double cx, double cy, int its_robust, int its_final, const volatile uint8_t* stop,
                          uint8_t* outlier_flags, cms_ba_stats* stats) {
  if (outlier_flags) memset(outlier_flags, 0, E > 0 ? E : 0);
  if (stop && *stop) { if (stats) memset(stats, 0, sizeof(*stats)); return 1; }
  cms_ba* b = nullptr;
  int rc = cms_ba_create(&b, device, K, poses, fixed, P, points, E, e_pose, e_point, e_obs, e_invsig2, e_face, fx, fy, cx, cy);
  if (rc) return rc;
  rc = cms_ba_optimize(b, its_robust, its_final, stop, stats);
  if (rc == CMS_OK) rc = cms_ba_read(b, poses, points, outlier_flags);
  cms_ba_destroy(b);
  return rc;
}

extern "C" int cms_ba_linearize(int device, int K, const double* poses, const uint8_t* fixed, int P, const double* points, int E,
                                const int* e_pose, const int* e_point, const double* e_obs, const double* e_invsig2,
                                const int8_t* e_face, double fx, double fy, double cx, double cy, int robust, double huber_delta,
                                double* err, double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, double* robust_chi2_sum) {
  cms_ba* b = nullptr;
  ba_tl_force_host_plan = true;      // (this entry reports in the caller's order through the host-side permutations, and launches the per-key-frame linearisation)
  int rc = cms_ba_create(&b, device, K, poses, fixed, P, points, E, e_pose, e_point, e_obs, e_invsig2, e_face, fx, fy, cx, cy);
  ba_tl_force_host_plan = false;
  if (rc) return rc;
  if (!b->d_Hpl) {      // the window carries only the edge-major work list: the stored-block buffer this entry reports is allocated here
    rc = ba_alloc(b, &b->d_Hpl, 18 * (size_t)E);
    if (rc) { cms_ba_destroy(b); return rc; }
  }
  hipStream_t s = b->stream;
  ba_errors(b, 0, robust, huber_delta, 0);
  hipLaunchKernelGGL(k_ba_lin_points, dim3(b->nblk_p), dim3(128), 0, s, b->d, (const double*)b->d_poses[0], (const double*)b->d_pts[0],
                     robust, huber_delta, b->d_Hll, b->d_bl, b->d_Hpl);
  hipMemsetAsync(b->d_Hpp, 0, 36 * (size_t)std::max(b->np, 1) * sizeof(double), s);
  hipMemsetAsync(b->d_bp, 0, 6 * (size_t)std::max(b->np, 1) * sizeof(double), s);
  hipLaunchKernelGGL(k_ba_lin_poses, dim3(b->K, BA_POSE_CHUNKS), dim3(256), 0, s, b->d, (const double*)b->d_poses[0], (const double*)b->d_pts[0], robust,
                     huber_delta, b->d_pose_partial);
  if (b->np > 0)
    hipLaunchKernelGGL(k_ba_pose_finish, dim3(b->np), dim3(64), 0, s, b->np, (const double*)b->d_pose_partial, b->d_Hpp, b->d_bp);
  hipError_t he = hipStreamSynchronize(s);
  if (he != hipSuccess) { cms_ba_destroy(b); return cms_fail(CMS_ERR_HIP, "cms_ba_linearize", he); }
  std::vector<int> slot(K, -1);
  { int np = 0; for (int k = 0; k < K; ++k) if (!fixed[k]) slot[k] = np++; }
  std::vector<double> tmp;
  auto dl = [&](double* dptr, size_t cnt) { tmp.resize(cnt); hipMemcpy(tmp.data(), dptr, cnt * sizeof(double), hipMemcpyDeviceToHost); };
  if (err) { dl(b->d_err, 2 * (size_t)E); for (int i = 0; i < E; ++i) { err[2 * b->perm[i]] = tmp[2 * i]; err[2 * b->perm[i] + 1] = tmp[2 * i + 1]; } }
  if (Hpl) { dl(b->d_Hpl, 18 * (size_t)E); for (int i = 0; i < E; ++i) memcpy(Hpl + 18 * (size_t)b->perm[i], &tmp[18 * (size_t)i], 18 * sizeof(double)); }
  if (Hll) { dl(b->d_Hll, 9 * (size_t)P); for (int i = 0; i < P; ++i) memcpy(Hll + 9 * (size_t)b->pinv[i], &tmp[9 * (size_t)i], 9 * sizeof(double)); }
  if (bl) { dl(b->d_bl, 3 * (size_t)P); for (int i = 0; i < P; ++i) memcpy(bl + 3 * (size_t)b->pinv[i], &tmp[3 * (size_t)i], 3 * sizeof(double)); }
  if (Hpp) { dl(b->d_Hpp, 36 * (size_t)std::max(b->np, 1)); memset(Hpp, 0, 36 * (size_t)K * sizeof(double)); for (int k = 0; k < K; ++k) if (slot[k] >= 0) memcpy(Hpp + 36 * (size_t)k, &tmp[36 * (size_t)slot[k]], 36 * sizeof(double)); }
  if (bp) { dl(b->d_bp, 6 * (size_t)std::max(b->np, 1)); memset(bp, 0, 6 * (size_t)K * sizeof(double)); for (int k = 0; k < K; ++k) if (slot[k] >= 0) memcpy(bp + 6 * (size_t)k, &tmp[6 * (size_t)slot[k]], 6 * sizeof(double)); }
  if (robust_chi2_sum) hipMemcpy(robust_chi2_sum, b->d_scal, sizeof(double), hipMemcpyDeviceToHost);
  cms_ba_destroy(b);
  return CMS_OK;
}
