// batch_driver.cpp -- bench.py's step WITHOUT the interpreter: the same C-ABI calls, in the same order, from C++ threads.
//
// bench.py drives one GPU's 32 camera streams from ~25 Python threads; every library call drops and re-takes the interpreter lock, and a rank that
// gets two host cores (eight ranks under a 16-core quota) spends its step waiting for that lock and for its own glue (12.5 k
// frames/s with two cores against 21 k with eight).  The reference never uses more than two threads per stream (System.cpp:108-127: Tracking +
// LocalMapping); a host that tracks many streams per GPU is C++.  This file is that host for the bench's synthetic workload:
//
//   frame thread (the caller)   cms_pose_launch, cms_frames_upload_device, cms_frames_process, cms_area_grid, the two projection searches with their
//                               window queries (TrackWithMotionModel, TrackLocalMap), cms_kfstore_put_from_frames for the batch's key frames,
//                               cms_frames_sync, cms_pose_fetch -- Tracking.cpp:620-719 for B frames at once
//   per window group            a MAPPING thread: pending pose write-backs (cms_kfstore_update_poses), cms_kfstore_create_new_map_points,
//                               cms_kfstore_fuse_search_sets (LocalMapping.cpp:52-117, 388-466);
//                               a LOCAL-BA thread: waits for the group's windows and for the mapping thread, cms_ba_optimize_many (Optimizer.cpp:192-451);
//                               a BUILDER thread: cms_ba_create_many + cms_ba_set_stream for the windows of a coming step;
//                               a FINISHER thread: cms_ba_read_many, the write-back request, cms_ba_destroy
//
// The step loop is bench.py's: the mapping side of step s is collected at the end of step s + 1, two sets of windows are always under construction,
// the read-backs of step s finish under step s + 1.  Everything the calls need (device buffers, job lists, window descriptions) is prepared by the
// caller once; this file owns no data and makes no decision the Python loop does not make.  Built into libcubemapslam_host.so; bench.py reports the
// figure of this driver as `value` and the Python loop's next to it (config.python_step_loop).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <pthread.h>
#include <vector>

#include "cubemapslam_hip.h"

#define CBD_MAX_GROUPS 8

extern "C" {
typedef struct cbd_frame_set {       // one batch of frames with everything its tracking needs, resident on the device (bench.py: TrackSet)
  const void* d_frames;
  // TrackWithMotionModel: ORBMatcher::SearchByProjection(CurrentFrame, LastFrame, th = 15)
  int nq; const void* d_mm_frame; const void* d_mm_pose; const void* d_mm_valid; const void* d_mm_xw; const void* d_mm_oct; void* d_mm_q[5];
  void* d_cnt; void* d_off; void* d_idx; int cand_cap; void* d_tot;
  const void* d_mm_mpoff; const void* d_mm_desc; void* d_mm_pd; void* d_mm_match; const void* d_mm_ang; void* d_mm_n;
  // TrackLocalMap: Frame::isInFrustum + ORBMatcher::SearchByProjection(F, vpMapPoints, th)
  int n_mp; const void* d_lm_frame; const void* d_lm_pose; const void* d_lm_in[4]; void* d_lm_vis; void* d_lm_f[4]; void* d_lm_i[5];
  void* d_lm_off; void* d_lm_idx; int lm_cap; void* d_lm_tot; const void* d_lm_mpoff; const void* d_lm_desc; void* d_lm_pd;
  void* d_kpmp; const void* d_kpmp0; size_t kpmp_bytes;
  // ProcessNewKeyFrame: this batch's key frames per window group
  const cms_kf_from_frame* put_items[CBD_MAX_GROUPS]; int put_n[CBD_MAX_GROUPS];
} cbd_frame_set;

typedef struct cbd_group {
  cms_kfstore* store; void* ba_stream;
  // CreateNewMapPoints of the group's key frames
  int njobs; const int* cur_slot; const int* neigh_off; const int* neigh_slot; int cap; int* n_new; int* o_neigh; int* o_idx1; int* o_idx2; float* o_x3d;
  // SearchInNeighbors: both Fuse directions of every key frame, map-point sets uploaded once
  int nsets; const int* set_off; const float* pos; const float* normal; const float* min_d; const float* max_d; const uint8_t* desc;
  int nfjobs; const int* job_slot; const int* job_set; const uint8_t* skip; float th; int* best_idx; int* best_dist;
  // pose write-back of a step's windows (Optimizer.cpp:419-431): all of the group's key frames in one call
  int n_upd; const int* upd_slots; const float* upd_R; const float* upd_t; const float* upd_Ow;
  // the group's local-BA windows: two alternating sets of problems
  int nwin; const cms_ba_window* windows[2];
} cbd_group;

typedef void (*cbd_step_done_fn)(void* user, int step, const double* poses7, int nframes);

typedef struct cbd_plan {
  cms_ctx* ctx; cms_pose* po; int B, device, ngroups, create_threads, mapping_full, ahead, n_pose_edges;
  cbd_group groups[CBD_MAX_GROUPS];
  cbd_frame_set sets[2];
  cbd_step_done_fn step_done; void* user;      // optional: called by the frame thread after every step's pose fetch (the multi-GPU trajectory gather)
} cbd_plan;

typedef struct cbd_stats {
  double ba_ms_sum; long ba_jobs;               // a group's local-BA thread: wall time per step and group
  double schur_ms; long schur_launches;         // cms_ba_profile_kernel(3) of every group's first window
  double create_ms_sum; long create_windows;    // wall time of cms_ba_create_many / windows
  double tri_ms_sum, fuse_ms_sum, put_ms_sum, upd_ms_sum; long tri_calls, fuse_calls, put_calls, upd_calls;
  double wait_windows_ms, wait_tri_ms, optimize_ms;
  long new_map_points_last_step, fused_last_call;
  float stage_ms[7];                            // cms_profile_get summed over the steps
  long steps;
} cbd_stats;
}

namespace {
// one worker thread executing jobs in order
class Worker {
 public:
  explicit Worker(const std::string& name) : th_([this, name]() { pthread_setname_np(pthread_self(), name.substr(0, 15).c_str()); run(); }) {}
  ~Worker() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_one();
    th_.join();
  }
  template <class F> auto submit(F f) -> std::future<decltype(f())> {
    auto task = std::make_shared<std::packaged_task<decltype(f())()>>(std::move(f));
    auto fut = task->get_future();
    { std::lock_guard<std::mutex> lk(mu_); q_.push_back([task]() { (*task)(); }); }
    cv_.notify_one();
    return fut;
  }
 private:
  void run() {
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this]() { return stop_ || !q_.empty(); });
        if (q_.empty()) return;
        job = std::move(q_.front()); q_.pop_front();
      }
      job();
    }
  }
  std::mutex mu_; std::condition_variable cv_; std::deque<std::function<void()>> q_; bool stop_ = false;
  std::thread th_;
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Fail : std::runtime_error { using std::runtime_error::runtime_error; };
void chk(int rc, const char* what) { if (rc < 0) throw Fail(std::string(what) + ": " + cms_last_error()); }

typedef std::vector<cms_ba*> Windows;
struct BaJobResult { std::future<void> finish; double ms = 0; long n_new = 0; };
}  // namespace

struct cbd {
  cbd_plan p;
  std::vector<std::unique_ptr<Worker>> mapping, ba, builder, finisher;
  std::mutex stats_mu; cbd_stats st;
  std::mutex store_mu[CBD_MAX_GROUPS];
  std::mutex wb_mu; int wb_pending[CBD_MAX_GROUPS];           // complete groups of read-back windows whose poses still have to go to the store
  std::deque<std::pair<int, std::vector<std::future<Windows>>>> queue;      // sets of windows under construction: (problem set, per group)
  std::vector<std::future<BaJobResult>> inflight;            // the previous step's mapping side (steps overlap by one)
  std::vector<std::future<void>> reads;                      // read-backs handed to the finishers
  std::vector<double> poses7; std::vector<uint8_t> pose_out; std::vector<int> pose_inl; std::vector<cms_pose_stats> pose_st;
  std::vector<std::vector<double>> rd_poses, rd_pts; std::vector<std::vector<uint8_t>> rd_flags;      // per group: the read-back's host arrays
  std::string err;
};

namespace {
Windows build_group(cbd* d, int set, int g) {
  const cbd_group& G = d->p.groups[g];
  const double t0 = now_ms();
  Windows w((size_t)G.nwin, nullptr);
  chk(cms_ba_create_many(w.data(), G.nwin, d->p.device, G.windows[set], d->p.create_threads), "cms_ba_create_many");
  for (cms_ba* b : w) chk(cms_ba_set_stream(b, G.ba_stream), "cms_ba_set_stream");
  std::lock_guard<std::mutex> lk(d->stats_mu);
  d->st.create_ms_sum += now_ms() - t0; d->st.create_windows += G.nwin;
  return w;
}
void submit_windows(cbd* d, int set) {
  std::vector<std::future<Windows>> futs;
  for (int g = 0; g < d->p.ngroups; ++g) futs.push_back(d->builder[(size_t)g]->submit([d, set, g]() { return build_group(d, set, g); }));
  d->queue.emplace_back(set, std::move(futs));
}
void apply_write_backs(cbd* d, int g) {      // (the store's lock is held by the caller)
  const cbd_group& G = d->p.groups[g];
  for (;;) {
    { std::lock_guard<std::mutex> lk(d->wb_mu); if (d->wb_pending[g] <= 0) return; --d->wb_pending[g]; }
    const double t0 = now_ms();
    chk(cms_kfstore_update_poses(G.store, G.n_upd, G.upd_slots, G.upd_R, G.upd_t, G.upd_Ow), "cms_kfstore_update_poses");
    std::lock_guard<std::mutex> lk(d->stats_mu);
    d->st.upd_ms_sum += now_ms() - t0; ++d->st.upd_calls;
  }
}
long mapping_job(cbd* d, int g) {
  const cbd_group& G = d->p.groups[g];
  long n_new = 0;
  {
    std::lock_guard<std::mutex> lock(d->store_mu[g]);
    if (d->p.mapping_full) apply_write_backs(d, g);      // poses of the windows read back since this thread's last turn (Optimizer.cpp:419-431)
    const double t0 = now_ms();
    chk(cms_kfstore_create_new_map_points(G.store, G.njobs, G.cur_slot, G.neigh_off, G.neigh_slot, 0, G.cap, G.n_new, G.o_neigh, G.o_idx1, G.o_idx2, G.o_x3d),
        "cms_kfstore_create_new_map_points");
    for (int j = 0; j < G.njobs; ++j) n_new += G.n_new[j];
    std::lock_guard<std::mutex> lk(d->stats_mu);
    d->st.tri_ms_sum += now_ms() - t0; ++d->st.tri_calls;
  }
  if (d->p.mapping_full) {
    const double t0 = now_ms();
    {
      std::lock_guard<std::mutex> lock(d->store_mu[g]);
      chk(cms_kfstore_fuse_search_sets(G.store, G.nsets, G.set_off, G.pos, G.normal, G.min_d, G.max_d, G.desc, G.nfjobs, G.job_slot, G.job_set, G.skip, G.th,
                                       G.best_idx, G.best_dist), "cms_kfstore_fuse_search_sets");
    }
    std::lock_guard<std::mutex> lk(d->stats_mu);
    d->st.fuse_ms_sum += now_ms() - t0; ++d->st.fuse_calls;
  }
  return n_new;
}
void finish_job(cbd* d, int g, Windows w) {
  const int n = (int)w.size();
  std::vector<double*> pp((size_t)n), pq((size_t)n); std::vector<uint8_t*> pf((size_t)n);
  // (the group's windows have the sizes of its window descriptions; the arrays are reused from step to step like a mapper's own)
  const cbd_group& G = d->p.groups[g];
  size_t op = 0, oq = 0, of = 0;
  for (int i = 0; i < n; ++i) {
    const cms_ba_window& W0 = G.windows[0][i]; const cms_ba_window& W1 = G.windows[1][i];
    const size_t K = (size_t)std::max(W0.K, W1.K), P = (size_t)std::max(W0.P, W1.P), E = (size_t)std::max(W0.E, W1.E);
    pp[(size_t)i] = d->rd_poses[(size_t)g].data() + op; pq[(size_t)i] = d->rd_pts[(size_t)g].data() + oq; pf[(size_t)i] = d->rd_flags[(size_t)g].data() + of;
    op += 7 * K; oq += 3 * P; of += E;
  }
  chk(cms_ba_read_many(w.data(), n, pp.data(), pq.data(), pf.data()), "cms_ba_read_many");
  if (d->p.mapping_full) { std::lock_guard<std::mutex> lk(d->wb_mu); ++d->wb_pending[g]; }
  for (cms_ba* b : w) cms_ba_destroy(b);
}
BaJobResult ba_job(cbd* d, int g, std::shared_ptr<std::future<Windows>> wf, std::shared_ptr<std::future<long>> tri) {
  BaJobResult r;
  const double t0 = now_ms();
  Windows w = wf->get();
  const double t1 = now_ms();
  chk(cms_ba_profile_kernel(w[0], 3), "cms_ba_profile_kernel");
  r.n_new = tri->get();
  const double t2 = now_ms();
  std::vector<cms_ba_stats> stats(w.size());
  chk(cms_ba_optimize_many(w.data(), (int)w.size(), 5, 10, nullptr, stats.data()), "cms_ba_optimize_many");
  const double t3 = now_ms();
  double ms = 0; long nl = 0;
  chk(cms_ba_profile_get(w[0], &ms, &nl), "cms_ba_profile_get");
  r.finish = d->finisher[(size_t)g]->submit([d, g, w]() { finish_job(d, g, w); });
  r.ms = now_ms() - t0;
  std::lock_guard<std::mutex> lk(d->stats_mu);
  d->st.schur_ms += ms; d->st.schur_launches += nl;
  d->st.wait_windows_ms += t1 - t0; d->st.wait_tri_ms += t2 - t1; d->st.optimize_ms += t3 - t2;
  d->st.ba_ms_sum += r.ms; ++d->st.ba_jobs;
  return r;
}
void collect(cbd* d, std::vector<std::future<BaJobResult>>& ths) {
  std::vector<BaJobResult> res;
  for (auto& f : ths) res.push_back(f.get());
  if (res.empty()) return;
  for (auto& f : d->reads) f.get();            // at most two steps' read-backs are ever outstanding
  d->reads.clear();
  long n_new = 0;
  for (auto& r : res) { d->reads.push_back(std::move(r.finish)); n_new += r.n_new; }
  std::lock_guard<std::mutex> lk(d->stats_mu);
  d->st.new_map_points_last_step = n_new;
}
void tracking(cbd* d, const cbd_frame_set& S) {      // bench.py: TrackSet.enqueue_tracking
  cms_ctx* c = d->p.ctx; const int B = d->p.B;
  chk(cms_area_grid(c, B), "cms_area_grid");
  chk(cms_stream_copy_device(c, S.d_kpmp, S.d_kpmp0, S.kpmp_bytes), "cms_stream_copy_device");
  chk(cms_project_last_frame_device(c, S.nq, S.d_mm_frame, S.d_mm_pose, S.d_mm_valid, S.d_mm_xw, S.d_mm_oct, 15.0f, S.d_mm_q[0], S.d_mm_q[1], S.d_mm_q[2],
                                    S.d_mm_q[3], S.d_mm_q[4]), "cms_project_last_frame_device");
  chk(cms_features_in_area_batch_device(c, S.nq, S.d_mm_frame, S.d_mm_q[0], S.d_mm_q[1], S.d_mm_q[2], S.d_mm_q[3], S.d_mm_q[4], S.d_cnt, S.d_off, S.d_idx,
                                        S.cand_cap, S.d_tot), "cms_features_in_area_batch_device");
  chk(cms_search_local_points_device(c, B, S.d_mm_mpoff, S.d_mm_desc, S.d_off, S.d_idx, S.d_mm_pd, -1.0f, 100, S.d_kpmp, S.d_mm_match, nullptr),
      "cms_search_local_points_device");
  chk(cms_rotation_filter_device(c, B, S.d_mm_mpoff, S.d_mm_ang, S.d_kpmp, S.d_mm_match, S.d_mm_n, 1), "cms_rotation_filter_device");
  chk(cms_is_in_frustum_device(c, S.n_mp, S.d_lm_frame, S.d_lm_pose, S.d_lm_in[0], S.d_lm_in[1], S.d_lm_in[2], S.d_lm_in[3], 0.5f, 1.0f, S.d_lm_vis, S.d_lm_f[0],
                               S.d_lm_f[1], S.d_lm_i[0], S.d_lm_f[2], S.d_lm_f[3], S.d_lm_i[1], S.d_lm_i[2]), "cms_is_in_frustum_device");
  chk(cms_features_in_area_batch_device(c, S.n_mp, S.d_lm_frame, S.d_lm_f[0], S.d_lm_f[1], S.d_lm_f[3], S.d_lm_i[1], S.d_lm_i[2], S.d_lm_i[3], S.d_lm_off,
                                        S.d_lm_idx, S.lm_cap, S.d_lm_tot), "cms_features_in_area_batch_device");
  chk(cms_search_local_points_device(c, B, S.d_lm_mpoff, S.d_lm_desc, S.d_lm_off, S.d_lm_idx, S.d_lm_pd, 0.8f, 100, S.d_kpmp, S.d_lm_i[4], nullptr),
      "cms_search_local_points_device");
}
void one_step(cbd* d, int i) {
  const cbd_plan& p = d->p;
  const cbd_frame_set& S = p.sets[i & 1];
  chk(cms_pose_launch(p.po), "cms_pose_launch");                  // own stream, overlaps the frame path
  chk(cms_frames_upload_device(p.ctx, S.d_frames, p.B), "cms_frames_upload_device");
  chk(cms_frames_process(p.ctx, p.B, 1), "cms_frames_process");
  // the coming steps' windows are built under this one; this step's groups go to their threads
  auto cur = std::move(d->queue.front()); d->queue.pop_front();
  submit_windows(d, d->queue.empty() ? (cur.first ^ 1) : (d->queue.back().first ^ 1));
  std::vector<std::future<BaJobResult>> ths;
  for (int g = 0; g < p.ngroups; ++g) {
    auto tri = std::make_shared<std::future<long>>(d->mapping[(size_t)g]->submit([d, g]() { return mapping_job(d, g); }));
    auto wf = std::make_shared<std::future<Windows>>(std::move(cur.second[(size_t)g]));
    ths.push_back(d->ba[(size_t)g]->submit([d, g, wf, tri]() { return ba_job(d, g, wf, tri); }));
  }
  tracking(d, S);
  if (p.mapping_full) {                                          // ProcessNewKeyFrame: behind the tracking just enqueued, device to device
    const double t0 = now_ms();
    for (int g = 0; g < p.ngroups; ++g) chk(cms_kfstore_put_from_frames(p.groups[g].store, p.ctx, S.put_n[g], S.put_items[g]), "cms_kfstore_put_from_frames");
    std::lock_guard<std::mutex> lk(d->stats_mu);
    d->st.put_ms_sum += now_ms() - t0; ++d->st.put_calls;
  }
  chk(cms_frames_sync(p.ctx), "cms_frames_sync");                 // the step's frames are through when their poses are on the host
  chk(cms_pose_fetch(p.po, d->poses7.data(), d->pose_out.data(), d->pose_inl.data(), d->pose_st.data()), "cms_pose_fetch");
  if (p.step_done) p.step_done(p.user, i, d->poses7.data(), p.B);
  float ms7[7];
  if (cms_profile_get(p.ctx, ms7) == CMS_OK) for (int k = 0; k < 7; ++k) d->st.stage_ms[k] += ms7[k];
  // the mapping side of step s is waited for at the end of step s + 1
  std::vector<std::future<BaJobResult>> prev = std::move(d->inflight);
  d->inflight = std::move(ths);
  collect(d, prev);
  ++d->st.steps;
}
}  // namespace

extern "C" {
static thread_local std::string g_cbd_err;
const char* cbd_last_error() { return g_cbd_err.c_str(); }
#define CBD_TRY(...) try { __VA_ARGS__ } catch (const std::exception& e) { g_cbd_err = e.what(); return -1; }

cbd* cbd_create(const cbd_plan* plan, int max_pose_edges) {
  if (!plan || plan->ngroups < 1 || plan->ngroups > CBD_MAX_GROUPS || plan->B < 1) { g_cbd_err = "cbd_create: bad plan"; return nullptr; }
  cbd* d = new cbd;
  d->p = *plan;
  std::memset(&d->st, 0, sizeof(d->st));
  for (int g = 0; g < plan->ngroups; ++g) {
    const std::string gs = std::to_string(g);
    d->mapping.emplace_back(new Worker("cbd-map" + gs)); d->ba.emplace_back(new Worker("cbd-ba" + gs)); d->builder.emplace_back(new Worker("cbd-build" + gs));
    d->finisher.emplace_back(new Worker("cbd-finish" + gs));
    d->wb_pending[g] = 0;
    size_t np = 0, nq = 0, nf = 0;
    for (int i = 0; i < plan->groups[g].nwin; ++i) {
      const cms_ba_window& W0 = plan->groups[g].windows[0][i]; const cms_ba_window& W1 = plan->groups[g].windows[1][i];
      np += 7 * (size_t)std::max(W0.K, W1.K); nq += 3 * (size_t)std::max(W0.P, W1.P); nf += (size_t)std::max(W0.E, W1.E);
    }
    d->rd_poses.emplace_back(np); d->rd_pts.emplace_back(nq); d->rd_flags.emplace_back(nf);
  }
  d->poses7.assign(7 * (size_t)plan->B, 0.0); d->pose_out.assign((size_t)std::max(max_pose_edges, 1), 0); d->pose_inl.assign((size_t)plan->B, 0);
  d->pose_st.resize((size_t)plan->B);
  return d;
}
// `ahead` sets of windows go to the builders (bench.py: timed() before its warm-up steps)
int cbd_begin(cbd* d) { CBD_TRY(for (int a = 0; a < std::max(1, d->p.ahead); ++a) submit_windows(d, a & 1); return 0;) }
int cbd_steps(cbd* d, int first, int n) { CBD_TRY(for (int i = 0; i < n; ++i) one_step(d, first + i); return 0;) }
// the last step's mapping side (steps overlap by one)
int cbd_collect_inflight(cbd* d) { CBD_TRY(std::vector<std::future<BaJobResult>> prev = std::move(d->inflight); d->inflight.clear(); collect(d, prev); return 0;) }
// what the timed region ends with: the last read-backs and their pose write-backs
int cbd_finish(cbd* d) {
  CBD_TRY(
    for (auto& f : d->reads) f.get();
    d->reads.clear();
    if (d->p.mapping_full)
      for (int g = 0; g < d->p.ngroups; ++g) { std::lock_guard<std::mutex> lock(d->store_mu[g]); apply_write_backs(d, g); }
    return 0;)
}
// windows built for steps that never run are destroyed (outside the timed region, like bench.py's drain)
int cbd_drain(cbd* d) {
  CBD_TRY(
    while (!d->queue.empty()) {
      for (auto& f : d->queue.front().second) { Windows w = f.get(); for (cms_ba* b : w) cms_ba_destroy(b); }
      d->queue.pop_front();
    }
    return 0;)
}
// sizes of the plan structures as this library was compiled with them (the caller's descriptions must match: bench.py checks)
void cbd_sizes(int* out4) { out4[0] = (int)sizeof(cbd_frame_set); out4[1] = (int)sizeof(cbd_group); out4[2] = (int)sizeof(cbd_plan); out4[3] = (int)sizeof(cbd_stats); }
void cbd_stats_get(cbd* d, cbd_stats* out, int reset) {
  std::lock_guard<std::mutex> lk(d->stats_mu);
  if (out) *out = d->st;
  if (reset) std::memset(&d->st, 0, sizeof(d->st));
}
void cbd_destroy(cbd* d) {
  if (!d) return;
  try { cbd_collect_inflight(d); cbd_finish(d); cbd_drain(d); } catch (...) {}
  delete d;
}
}
