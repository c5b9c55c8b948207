// cms_api_frames.hip -- host side of the C-ABI for the frame path (context, LUT / table construction, launches).
// Included by cms_lib.hip (single translation unit together with the kernels).
#include <chrono>
#include <thread>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include "../../include/cubemapslam_hip.h"
#include "cms_types.h"
#include "orb_pattern.inc"

// CMS_CTX_CU_MASK=w0,w1,...,w7 (hex words, bit i of word j = compute unit 32 j + i of the runtime's enumeration): streams created while it is set
// (cms_ctx_create, cms_pose_create) are confined to those compute units (hipExtStreamCreateWithCUMask).  A developer experiment: the frame path and the
// local-BA rounds on disjoint parts of the chip instead of time-sharing whole CUs; bench.py sets it around the contexts it creates
// when CMS_BENCH_CU_SPLIT=n asks for the frame path on n CUs.  Read at every call, not once.
static bool cms_cu_mask_from_env(uint32_t* words, int* n_words) {
  const char* v = getenv("CMS_CTX_CU_MASK");
  if (!v || !*v) return false;
  int n = 0;
  while (*v && n < 8) {
    char* end = nullptr;
    words[n++] = (uint32_t)strtoul(v, &end, 16);
    if (end == v) return false;
    v = *end == ',' ? end + 1 : end;
    if (*end && *end != ',') return false;
  }
  *n_words = n;
  return n > 0;
}

static thread_local std::string g_cms_err;
static int cms_fail(int code, const char* what, hipError_t e = hipSuccess) {
  char buf[512];
  if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  else snprintf(buf, sizeof(buf), "%s", what);
  g_cms_err = buf;
  return code;
}
#define HIPCHK(call)                                                        \
  do {                                                                      \
    hipError_t _e = (call);                                                 \
    if (_e != hipSuccess) return cms_fail(CMS_ERR_HIP, #call, _e);          \
  } while (0)

extern "C" const char* cms_last_error(void) { return g_cms_err.c_str(); }
extern "C" int cms_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

struct cms_ctx {
  int device = 0;
  cms_camera cam;
  cms_orb_params orb;
  int max_batch = 0;
  CmsGeom g;
  float scale[CMS_MAX_LEVELS], inv_scale[CMS_MAX_LEVELS], sigma2[CMS_MAX_LEVELS], inv_sigma2[CMS_MAX_LEVELS];
  hipStream_t stream = nullptr;
  int fstride = 0;            // device fisheye row stride
  size_t fish_pitch = 0;      // bytes per fisheye frame
  int lut_stride = 0, mstride = 0;
  bool have_mask = false;
  int dirty_frames = 0;       // leading frames whose corner blocks may be non-zero (a caller-supplied canvas went through them)
  // device buffers
  uint8_t* d_fish = nullptr; uint32_t* d_lut = nullptr; uint8_t* d_pyr = nullptr; uint8_t* d_mask = nullptr;
  CmsResizeTab* d_tab = nullptr; float* d_pattern = nullptr;     // the 256 x 4 test coordinates as floats (what the kernel multiplies)
  uint32_t* d_cand = nullptr; uint16_t* d_node = nullptr; int* d_cand_cnt = nullptr; int* d_overflow = nullptr;
  uint32_t* d_qt_out = nullptr; int* d_qt_cnt = nullptr;
  uint32_t* d_cell_cand = nullptr; int* d_cell_cnt = nullptr;
  int* d_cells_all = nullptr; int* d_cells_nz = nullptr; int n_cells_all = 0, n_cells_nz = 0;   // FAST work lists
  // frame grid (Frame::AssignFeaturesToGrid), allocated on first use
  uint16_t* d_area_sorted = nullptr; int* d_area_cell_start = nullptr; int* d_area_nvalid = nullptr; int area_frames = 0;
  int* d_area_bsum = nullptr; int area_bsum_cap = 0;
  int* d_area_tmp = nullptr; size_t area_tmp_cap = 0;      // first-pass hit buffer of k_area_query (queries)
  uint8_t* h_fish_stage = nullptr; int fish_stage_frames = 0;   // pinned staging for cms_frames_upload
  // input streaming (cms_frames_upload_async): the next batch travels on its own stream while the current one is being processed
  hipStream_t copy_stream = nullptr; hipEvent_t ev_upload_done = nullptr, ev_remap_done = nullptr;
  bool upload_pending = false, remap_recorded = false;
  int dist_bounds_scaled = 0;      // cms_set_distance_bounds_mode: map points' distance bounds come from the public MapPoint getters
  hipEvent_t ev_block = nullptr; int last_batch = 0;                     // cms_frames_sync after a large batch polls this event between short sleeps instead of spinning
  hipEvent_t ev_extracted = nullptr; bool extracted_recorded = false;   // end of the last cms_frames_process (cms_stream_wait_extracted)
  uint8_t* h_stage = nullptr; size_t h_stage_bytes = 0;          // pinned staging of the one-frame host entries (one copy each way)
  CmsKeyPoint* d_kps = nullptr; uint32_t* d_aux = nullptr; uint8_t* d_desc = nullptr; int* d_kp_cnt = nullptr;
  uint32_t* d_order = nullptr; uint32_t* d_aux_sorted = nullptr; int* d_walk_cnt = nullptr; float* d_rays = nullptr;      // the batch's key-point walk: the order k_describe works in (k_cull)
  // match scratch
  void* d_match = nullptr; size_t match_bytes = 0;
  // profiling
  bool prof = false;
  bool desc_spatial = false;         // k_describe works through the batch's key points in spatial order, an eighth of the walk per XCD (CMS_DESC_SPATIAL_ORDER=1)
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t fast_lds = 0, qt_lds = 0;
};

// ---- camera model, host side (System::CreateUndistortRectifyMap -> CamModelGeneral::CubemapToFisheye, see
// CamModelGeneral.cpp:265-290, CamModelGeneral.h:359-374,388-414,458-470).  Runs once per context in double precision
// with the platform libm, exactly like the reference does at start-up; the float LUT is then quantised the way
// cv::remap quantises it (5 fractional bits) and packed into one u32 per canvas pixel.
static inline double cms_horner12(const double* c, double x) {
  double r = 0.0;
  for (int i = 11; i >= 0; --i) r = r * x + c[i];
  return r;
}
static void cms_cubemap_to_fisheye(const cms_camera& cam, double up, double vp, double* uf, double* vf) {
  const int F = cam.face;
  const double half = F / 2.0;
  float i = (float)up, j = (float)vp;
  *uf = -1; *vf = -1;
  const float fi = i / (float)F, fj = j / (float)F;
  int face = CMS_FACE_UNKNOWN;
  if (fi >= 0 && fi < 1 && fj >= 1 && fj < 2) face = CMS_FACE_LEFT;
  else if (fi >= 1 && fi < 2 && fj >= 0 && fj < 1) face = CMS_FACE_UPPER;
  else if (fi >= 1 && fi < 2 && fj >= 1 && fj < 2) face = CMS_FACE_FRONT;
  else if (fi >= 1 && fi < 2 && fj >= 2 && fj < 3) face = CMS_FACE_LOWER;
  else if (fi >= 2 && fi < 3 && fj >= 1 && fj < 2) face = CMS_FACE_RIGHT;
  if (face == CMS_FACE_UNKNOWN) return;
  i = i - (float)((int)(i / (float)F) * F);
  j = j - (float)((int)(j / (float)F) * F);
  const double lx = ((double)i - half) * 1.0 / half, ly = ((double)j - half) * 1.0 / half, lz = 1.0;
  double x, y, z;
  switch (face) {
    case CMS_FACE_FRONT: x = lx; y = ly; z = lz; break;
    case CMS_FACE_LEFT: x = -lz; y = ly; z = lx; break;
    case CMS_FACE_RIGHT: x = lz; y = ly; z = -lx; break;
    case CMS_FACE_LOWER: x = lx; y = lz; z = -ly; break;
    default: x = lx; y = -lz; z = ly; break;  // UPPER
  }
  double norm = sqrt(x * x + y * y);
  if (norm == 0.0) norm = 1e-14;
  const double theta = atan(-z / norm);
  const double rho = cms_horner12(cam.invpol, theta);
  const double uu = x / norm * rho, vv = y / norm * rho;
  *uf = uu * cam.c + vv * cam.d + cam.u0;
  *vf = uu * cam.e + vv + cam.v0;
  if (*uf < 0 || *uf >= cam.Iw || *vf < 0 || *vf >= cam.Ih) { *uf = -1; *vf = -1; }
}
static void cms_build_lut(const cms_camera& cam, int lut_stride, std::vector<uint32_t>& lut) {
  const int W = 3 * cam.face;
  lut.assign((size_t)W * lut_stride, 0u);
  for (int y = 0; y < W; ++y)
    for (int x = 0; x < W; ++x) {
      double u, v;
      cms_cubemap_to_fisheye(cam, (double)x, (double)y, &u, &v);
      if (u < 0 || v < 0 || u >= cam.Iw || v >= cam.Ih) continue;   // map entry stays (0,0) (System.cpp:316-317)
      const float mu = (float)u, mv = (float)v;
      const int sx = (int)lrint((double)(mu * 32.f)), sy = (int)lrint((double)(mv * 32.f));
      const uint32_t X = (uint32_t)(sx >> 5), Y = (uint32_t)(sy >> 5), ax = (uint32_t)(sx & 31), ay = (uint32_t)(sy & 31);
      lut[(size_t)y * lut_stride + x] = X | (Y << 11) | (ax << 22) | (ay << 27);
    }
}

static inline short cms_sat_short(float v) {
  const long iv = lrint((double)v);
  return (short)(iv < -32768 ? -32768 : iv > 32767 ? 32767 : iv);
}
// cv::resize INTER_LINEAR coefficient tables (imgproc: fx = (dx+0.5)*scale - 0.5, 11-bit weights)
static void cms_resize_table(int sn, int dn, bool clamp_frac, CmsResizeTab* out) {
  const double inv_scale = (double)dn / sn, scale = 1. / inv_scale;
  for (int d = 0; d < dn; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    if (clamp_frac) {
      if (s < 0) { f = 0; s = 0; }
      if (s >= sn - 1) { f = 0; s = sn - 1; }
    }
    out[d].s = (short)s;
    out[d].a0 = cms_sat_short((1.f - f) * 2048);
    out[d].a1 = cms_sat_short(f * 2048);
    out[d].pad = 0;
  }
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static void cms_ctx_free(cms_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  void* ptrs[] = {c->d_fish, c->d_lut, c->d_pyr, c->d_mask, c->d_tab, c->d_pattern, c->d_cand, c->d_node, c->d_cand_cnt,
                  c->d_overflow, c->d_qt_out, c->d_qt_cnt, c->d_kps, c->d_aux, c->d_order, c->d_aux_sorted, c->d_desc, c->d_kp_cnt, c->d_match, c->d_cell_cand,
                  c->d_cell_cnt, c->d_cells_all, c->d_cells_nz, c->d_area_sorted, c->d_area_cell_start, c->d_area_nvalid, c->d_area_bsum, c->d_area_tmp, c->d_walk_cnt, c->d_rays};
  for (void* p : ptrs) if (p) hipFree(p);
  if (c->h_fish_stage) (void)hipHostFree(c->h_fish_stage);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  for (int i = 0; i < 8; ++i) if (c->ev[i]) hipEventDestroy(c->ev[i]);
  if (c->ev_extracted) hipEventDestroy(c->ev_extracted);
  if (c->ev_block) hipEventDestroy(c->ev_block);
  if (c->ev_upload_done) hipEventDestroy(c->ev_upload_done);
  if (c->ev_remap_done) hipEventDestroy(c->ev_remap_done);
  if (c->copy_stream) hipStreamDestroy(c->copy_stream);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int cms_ctx_create(cms_ctx** out, int device, const cms_camera* cam, const cms_orb_params* orb, int max_batch) {
  if (!out || !cam || !orb || max_batch < 1) return cms_fail(CMS_ERR_ARG, "cms_ctx_create: bad argument");
  *out = nullptr;
  if (orb->nlevels < 1 || orb->nlevels > CMS_MAX_LEVELS) return cms_fail(CMS_ERR_UNSUPPORTED, "nlevels must be 1..12");
  if (cam->face < 32 || 3 * cam->face > 4095) return cms_fail(CMS_ERR_UNSUPPORTED, "face size must satisfy 32 <= F and 3F <= 4095");
  if (cam->Iw < 2 || cam->Ih < 2 || cam->Iw > 2047 || cam->Ih > 2047) return cms_fail(CMS_ERR_UNSUPPORTED, "fisheye size must be <= 2047");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return cms_fail(CMS_ERR_NO_DEVICE, "no HIP device: the product path has no CPU fallback");
  if (device < 0 || device >= ndev) return cms_fail(CMS_ERR_ARG, "device index out of range");
  HIPCHK(hipSetDevice(device));
  cms_ctx* c = new cms_ctx;
  c->device = device; c->cam = *cam; c->orb = *orb; c->max_batch = max_batch;
  CmsGeom& g = c->g;
  memset(&g, 0, sizeof(g));
  const int L = orb->nlevels, F = cam->face, W = 3 * F;
  g.nlevels = L; g.W = W; g.F = F; g.ini_th = orb->ini_th_fast; g.min_th = orb->min_th_fast;
  // ---- ORBextractor::ORBextractor tables (ORBExtractor.cpp:386-418)
  const double scaleFactor = orb->scale_factor;
  c->scale[0] = 1.0f; c->sigma2[0] = 1.0f;
  for (int i = 1; i < L; ++i) { c->scale[i] = (float)(c->scale[i - 1] * scaleFactor); c->sigma2[i] = c->scale[i] * c->scale[i]; }
  for (int i = 0; i < L; ++i) { c->inv_scale[i] = 1.0f / c->scale[i]; c->inv_sigma2[i] = 1.0f / c->sigma2[i]; }
  {
    const float factor = (float)(1.0f / scaleFactor);
    float nDesired = orb->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; ++l) { g.lv[l].quota = (int)lrint((double)nDesired); sum += g.lv[l].quota; nDesired *= factor; }
    g.lv[L - 1].quota = orb->nfeatures - sum > 0 ? orb->nfeatures - sum : 0;
  }
  // ---- level geometry (ORBExtractor.cpp:928-934, 745-762)
  size_t off = 0, cand_off = 0, tab_off = 0;
  int cell0 = 0, kp_off = 0, maxq = 0, wCellMax = 0, hCellMax = 0;
  for (int l = 0; l < L; ++l) {
    CmsLevel& lv = g.lv[l];
    lv.w = (int)lrint((double)((float)W * c->inv_scale[l]));
    lv.h = lv.w;
    if (lv.w < 2 * CMS_MINB + 30) { delete c; return cms_fail(CMS_ERR_UNSUPPORTED, "pyramid level too small for the FAST cell grid"); }
    lv.stride = (int)align_up((size_t)lv.w + 8, 128);
    lv.off = off;
    off += align_up((size_t)lv.stride * (lv.h + 1), 256);
    const float width = (float)(lv.w - 2 * CMS_MINB), height = (float)(lv.h - 2 * CMS_MINB);
    lv.nCols = (int)(width / 30.f); lv.nRows = (int)(height / 30.f);
    lv.wCell = (int)ceilf(width / lv.nCols); lv.hCell = (int)ceilf(height / lv.nRows);
    lv.cell0 = cell0; cell0 += lv.nCols * lv.nRows;
    lv.cand_cap = (lv.w * lv.h) / 4 + 4096;
    lv.cand_off = cand_off; cand_off += align_up((size_t)lv.cand_cap, 64);
    lv.kp_off = kp_off; kp_off += lv.quota + 3;
    lv.scale = c->scale[l];
    lv.patch_size = (float)(int)(31 * c->scale[l]);
    lv.tab_off = tab_off; tab_off += (size_t)lv.w + lv.h;
    if (lv.quota > maxq) maxq = lv.quota;
    if (lv.wCell > wCellMax) wCellMax = lv.wCell;
    if (lv.hCell > hCellMax) hCellMax = lv.hCell;
    if (lv.nCols > 255 || lv.nRows > 255) { delete c; return cms_fail(CMS_ERR_UNSUPPORTED, "more than 255 FAST cells per row"); }
  }
  g.total_cells = cell0; g.kp_cap = kp_off; g.pyr_bytes = off; g.cand_total = cand_off;
  g.qt_maxn = 8;
  while (g.qt_maxn < maxq + 3) g.qt_maxn <<= 1;
  if (g.qt_maxn > 2048) { delete c; return cms_fail(CMS_ERR_UNSUPPORTED, "per-level feature quota above 2045 is not supported"); }
  g.tile_h = hCellMax + 6;
  g.tile_stride = (int)align_up((size_t)wCellMax + 6 + 3, 4) + 4;
  g.sc_h = hCellMax + 2;
  g.sc_stride = (int)align_up((size_t)wCellMax + 2, 4);
  g.list_cap = wCellMax * hCellMax;
  g.cell_cap = (int)align_up((size_t)((wCellMax + 1) / 2) * ((hCellMax + 1) / 2), 8);   // strict 3x3 maxima cannot be 8-adjacent
  g.dbg_stop = getenv("CMS_DBG_FAST_STOP") ? atoi(getenv("CMS_DBG_FAST_STOP")) : 0;
  // CMS_DESC_SPATIAL_ORDER=1: k_describe works through the batch's key points band by band of their levels (k_cull builds the walk), an eighth of
  // the walk per XCD: a third of the HBM fetches of the octree's list order (profiles/r02_describe_order.txt: 2.39 GB -> 0.86 GB per 256 frames).
  // It is NOT the default: the kernel is bound by its vector-ALU issue, alone (0.600 against 0.576 ms per 256 frames) and inside bench.py's
  // step next to the local BA's traffic (round 4, four profiled runs each: 0.84-1.02 ms against 0.65-0.87 in list order) -- the bytes it
  // saves are not what the kernel waits for.
  c->desc_spatial = getenv("CMS_DESC_SPATIAL_ORDER") != nullptr && g.kp_cap <= CMS_ORDER_MAX && max_batch <= 65535;
  g.gauss_column_mode = 0;
  if (wCellMax > 60 || hCellMax > 60) { delete c; return cms_fail(CMS_ERR_UNSUPPORTED, "FAST cell larger than 60 pixels"); }
  if (orb->scale_factor < 1.01f || orb->scale_factor > 1.9f) { delete c; return cms_fail(CMS_ERR_UNSUPPORTED, "scaleFactor must be in [1.01, 1.9]"); }
  g.fast_cell_lds = (int)align_up((size_t)g.tile_h * g.tile_stride + (size_t)g.sc_h * g.sc_stride + 2 * (size_t)g.list_cap + 16, 16);
  c->fast_lds = CMS_FAST_WPB * (size_t)g.fast_cell_lds;
  c->qt_lds = 64 * (size_t)g.qt_maxn + 4 * 512 + 64;

  c->fstride = (int)align_up((size_t)cam->Iw, 64);
  c->fish_pitch = (size_t)c->fstride * cam->Ih;
  c->lut_stride = (int)align_up((size_t)W, 4);
  c->mstride = (int)align_up((size_t)W, 64);
  const size_t B = (size_t)max_batch;
#define ALLOC(ptr, bytes) do { hipError_t _e = hipMalloc((void**)&(ptr), (bytes)); if (_e != hipSuccess) { cms_ctx_free(c); return cms_fail(CMS_ERR_HIP, "hipMalloc " #ptr, _e); } } while (0)
  // CMS_FRAME_STREAM_PRIORITY=high|low (developer knob): dispatch priority of the frame path's queue against the mapping side's
  hipError_t e;
  {
    const char* pr = getenv("CMS_FRAME_STREAM_PRIORITY");
    int lo = 0, hi = 0;
    uint32_t cu_mask[8]; int cu_words = 0;
    if (cms_cu_mask_from_env(cu_mask, &cu_words))      // developer A/B (CMS_CTX_CU_MASK, read at every call): the context's stream confined to a set of compute units
      e = hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)cu_words, cu_mask);
    else
    if (pr && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
      e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, pr[0] == 'h' ? hi : lo);
    else
      e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  }
  if (e != hipSuccess) { cms_ctx_free(c); return cms_fail(CMS_ERR_HIP, "hipStreamCreate", e); }
  for (int i = 0; i < 8; ++i) hipEventCreate(&c->ev[i]);
  ALLOC(c->d_fish, B * c->fish_pitch + 256);
  ALLOC(c->d_lut, (size_t)W * c->lut_stride * 4);
  ALLOC(c->d_pyr, B * g.pyr_bytes + 256);
  ALLOC(c->d_mask, (size_t)W * c->mstride);
  ALLOC(c->d_tab, tab_off * sizeof(CmsResizeTab));
  ALLOC(c->d_pattern, 1024 * sizeof(float));
  ALLOC(c->d_cand, B * g.cand_total * 4);
  ALLOC(c->d_node, B * g.cand_total * 2);
  ALLOC(c->d_cand_cnt, B * L * sizeof(int));
  ALLOC(c->d_overflow, sizeof(int));
  ALLOC(c->d_cell_cand, B * g.total_cells * g.cell_cap * 4);
  ALLOC(c->d_cell_cnt, B * g.total_cells * sizeof(int));
  ALLOC(c->d_qt_out, B * g.kp_cap * 4);
  ALLOC(c->d_qt_cnt, B * L * sizeof(int));
  ALLOC(c->d_kps, B * g.kp_cap * sizeof(CmsKeyPoint));
  ALLOC(c->d_aux, B * g.kp_cap * 4);
  ALLOC(c->d_order, B * g.kp_cap * 4);
  ALLOC(c->d_walk_cnt, 16);
  ALLOC(c->d_rays, B * g.kp_cap * 12);      // mvKeyRays of every frame (k_cull)
  ALLOC(c->d_aux_sorted, B * g.kp_cap * 4);
  ALLOC(c->d_desc, B * g.kp_cap * 32);
  ALLOC(c->d_kp_cnt, B * sizeof(int));
#undef ALLOC
  // zero-init: canvas corner blocks stay 0 forever (cubemap_lafida.cpp:110-111), mask defaults to "all valid"
  hipMemset(c->d_pyr, 0, B * g.pyr_bytes + 256);
  hipMemset(c->d_fish, 0, B * c->fish_pitch + 256);
  hipMemset(c->d_mask, 255, (size_t)W * c->mstride);
  hipMemset(c->d_overflow, 0, sizeof(int));
  hipMemset(c->d_kp_cnt, 0, B * sizeof(int));
  {
    std::vector<uint32_t> lut;
    cms_build_lut(*cam, c->lut_stride, lut);
    hipMemcpy(c->d_lut, lut.data(), lut.size() * 4, hipMemcpyHostToDevice);
    std::vector<CmsResizeTab> tab(tab_off);
    for (int l = 1; l < L; ++l) {
      cms_resize_table(g.lv[l - 1].w, g.lv[l].w, true, &tab[g.lv[l].tab_off]);
      cms_resize_table(g.lv[l - 1].h, g.lv[l].h, false, &tab[g.lv[l].tab_off + g.lv[l].w]);
    }
    // extent of the exactly-zero corner regions per level (exact, from the very tables k_resize uses): a destination
    // pixel is zero when both source taps sx, sx+1 lie inside the previous level's zero region (w == h, x and y tables agree
    // on the index part)
    g.lv[0].zlo = F; g.lv[0].zhi = F;
    for (int l = 1; l < L; ++l) {
      const CmsResizeTab* tx = &tab[g.lv[l].tab_off];
      const int sw = g.lv[l - 1].w, dw = g.lv[l].w;
      int lo = 0, hi = 0;
      while (lo < dw && std::min((int)tx[lo].s + 1, sw - 1) < g.lv[l - 1].zlo) ++lo;
      while (hi < dw && (int)tx[dw - 1 - hi].s >= sw - g.lv[l - 1].zhi) ++hi;
      g.lv[l].zlo = lo; g.lv[l].zhi = hi;
    }
    hipMemcpy(c->d_tab, tab.data(), tab.size() * sizeof(CmsResizeTab), hipMemcpyHostToDevice);
    // FAST work lists: every cell the reference loop visits (ORBExtractor.cpp:764-781), and the subset that does not lie
    // wholly inside the zero corner regions of a remapped cross (same tests as in k_fast_cells)
    std::vector<int> all, nz;
    for (int l = 0; l < L; ++l) {
      if (g.lv[l].nRows > 255 || g.lv[l].nCols > 255) { cms_ctx_free(c); return cms_fail(CMS_ERR_UNSUPPORTED, "cms_ctx_create: more than 255 FAST cells per row"); }
      const CmsLevel& lv = g.lv[l];
      const int maxBX = lv.w - CMS_MINB, maxBY = lv.h - CMS_MINB;
      for (int ci = 0; ci < lv.nRows; ++ci)
        for (int cj = 0; cj < lv.nCols; ++cj) {
          const int iniY = CMS_MINB + ci * lv.hCell, iniX = CMS_MINB + cj * lv.wCell;
          if (iniY >= maxBY - 3 || iniX >= maxBX - 6) continue;
          const int maxY = std::min(iniY + lv.hCell + 6, maxBY), maxX = std::min(iniX + lv.wCell + 6, maxBX);
          if (maxX - iniX - 6 <= 0 || maxY - iniY - 6 <= 0) continue;
          // list entry: level (4 bits) | row (8) | column (8): the kernel needs no division to find its cell
          const int id = l | (ci << 4) | (cj << 12);
          all.push_back(id);
          const bool zero = (maxX <= lv.zlo || iniX >= lv.w - lv.zhi) && (maxY <= lv.zlo || iniY >= lv.h - lv.zhi);
          if (!zero) nz.push_back(id);
        }
    }
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Listing the cells so that every XCD walks ONE
    // contiguous eighth of the row-major cell order keeps the 128-byte lines that horizontally adjacent cells share inside one
    // L2 (PMC: FETCH_SIZE of k_fast_cells was 2.8x the bytes of the non-zero cells with the plain order).
    auto xcd_stripe = [](std::vector<int>& v) {
      if (getenv("CMS_FAST_NO_XCD_STRIPE")) return;
      const size_t n = v.size(), per = (n + 7) / 8;
      std::vector<int> out;
      out.reserve(n);
      for (size_t j = 0; j < per; ++j)
        for (size_t x = 0; x < 8; ++x) if (x * per + j < n) out.push_back(v[x * per + j]);
      v.swap(out);
    };
    xcd_stripe(all); xcd_stripe(nz);
    c->n_cells_all = (int)all.size(); c->n_cells_nz = (int)nz.size();
    if (hipMalloc((void**)&c->d_cells_all, std::max<size_t>(all.size(), 1) * 4) != hipSuccess ||
        hipMalloc((void**)&c->d_cells_nz, std::max<size_t>(nz.size(), 1) * 4) != hipSuccess) {
      cms_ctx_free(c); return cms_fail(CMS_ERR_HIP, "hipMalloc cell lists");
    }
    hipMemcpy(c->d_cells_all, all.data(), all.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(c->d_cells_nz, nz.data(), nz.size() * 4, hipMemcpyHostToDevice);
    float patf[1024];
    for (int i = 0; i < 1024; ++i) patf[i] = (float)kOrbPattern[i];
    hipMemcpy(c->d_pattern, patf, sizeof(patf), hipMemcpyHostToDevice);
  }
  e = hipFuncSetAttribute((const void*)k_quadtree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->qt_lds);
  if (e != hipSuccess) { cms_ctx_free(c); return cms_fail(CMS_ERR_HIP, "hipFuncSetAttribute(k_quadtree)", e); }
  e = hipDeviceSynchronize();
  if (e != hipSuccess) { cms_ctx_free(c); return cms_fail(CMS_ERR_HIP, "ctx init", e); }
  *out = c;
  return CMS_OK;
}

extern "C" void cms_ctx_destroy(cms_ctx* ctx) { cms_ctx_free(ctx); }
extern "C" void* cms_ctx_stream(cms_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int cms_ctx_geometry(const cms_ctx* c, cms_geometry* o) {
  if (!c || !o) return cms_fail(CMS_ERR_ARG, "cms_ctx_geometry: null");
  memset(o, 0, sizeof(*o));
  o->W = c->g.W; o->F = c->g.F; o->nlevels = c->g.nlevels; o->kp_cap = c->g.kp_cap; o->max_batch = c->max_batch;
  for (int l = 0; l < c->g.nlevels; ++l) {
    o->level_w[l] = c->g.lv[l].w; o->level_h[l] = c->g.lv[l].h; o->level_quota[l] = c->g.lv[l].quota;
    o->level_cells[l] = c->g.lv[l].nCols * c->g.lv[l].nRows;
    o->scale[l] = c->scale[l]; o->inv_scale[l] = c->inv_scale[l]; o->sigma2[l] = c->sigma2[l]; o->inv_sigma2[l] = c->inv_sigma2[l];
  }
  o->pyramid_bytes_per_frame = c->g.pyr_bytes; o->candidate_entries_per_frame = c->g.cand_total;
  o->fisheye_stride = c->fstride;
  return CMS_OK;
}

extern "C" void* cms_frames_input(cms_ctx* c) { return c ? (void*)c->d_fish : nullptr; }

// wait = false (cms_remap_extract): the copy reads the library's own pinned staging block, the caller's buffer is free once the rows have been
// repacked; the entry synchronises the stream before it returns anyway
static int cms_frames_upload_impl(cms_ctx* c, const uint8_t* fisheye, int fstride, size_t frame_pitch, int B, bool wait) {
  if (!c || !fisheye || B < 1 || B > c->max_batch || fstride < c->cam.Iw) return cms_fail(CMS_ERR_ARG, "cms_frames_upload: bad argument");
  HIPCHK(hipSetDevice(c->device));
  // A pitched 2D copy from pageable memory goes row by row through the runtime (3.5 ms for one 754 x 480 frame); the rows are
  // repacked to the device pitch in a pinned staging buffer instead and leave as one linear copy (0.05 ms)
  const size_t per_frame = c->fish_pitch;
  if (!c->h_fish_stage || c->fish_stage_frames < B) {
    if (c->h_fish_stage) (void)hipHostFree(c->h_fish_stage);
    c->h_fish_stage = nullptr; c->fish_stage_frames = 0;
    if (hipHostMalloc((void**)&c->h_fish_stage, per_frame * (size_t)B) == hipSuccess) c->fish_stage_frames = B;
  }
  if (c->h_fish_stage) {
    for (int b = 0; b < B; ++b)
      for (int r = 0; r < c->cam.Ih; ++r)
        memcpy(c->h_fish_stage + (size_t)b * per_frame + (size_t)r * c->fstride, fisheye + (size_t)b * frame_pitch + (size_t)r * fstride, (size_t)c->cam.Iw);
    HIPCHK(hipMemcpyAsync(c->d_fish, c->h_fish_stage, per_frame * (size_t)B, hipMemcpyHostToDevice, c->stream));
  } else {
    for (int b = 0; b < B; ++b)
      HIPCHK(hipMemcpy2DAsync(c->d_fish + (size_t)b * c->fish_pitch, c->fstride, fisheye + (size_t)b * frame_pitch, fstride,
                              c->cam.Iw, c->cam.Ih, hipMemcpyHostToDevice, c->stream));
  }
  if (wait || !c->h_fish_stage) HIPCHK(hipStreamSynchronize(c->stream));   // the caller may free / reuse its host buffer as soon as we return
  return CMS_OK;
}
extern "C" int cms_frames_upload(cms_ctx* c, const uint8_t* fisheye, int fstride, size_t frame_pitch, int B) {
  return cms_frames_upload_impl(c, fisheye, fstride, frame_pitch, B, true);
}

// Input streaming.  The fisheye staging buffer is read by exactly one kernel of a batch, k_remap (the first one), so one buffer is
// enough for double buffering: the copy of batch s + 1 starts on the copy stream as soon as k_remap of batch s has finished and runs
// under the pyramid / FAST / octree / descriptor kernels of batch s; cms_frames_process of batch s + 1 waits for it on the device.
// The source must be pinned (cms_host_alloc, hipHostMalloc or hipHostRegister) and stay untouched until the next cms_frames_process
// has been enqueued and cms_frames_sync returned, or until cms_frames_upload_wait.
extern "C" int cms_frames_upload_async(cms_ctx* c, const uint8_t* fisheye, int fstride, size_t frame_pitch, int B) {
  if (!c || !fisheye || B < 1 || B > c->max_batch || fstride < c->cam.Iw) return cms_fail(CMS_ERR_ARG, "cms_frames_upload_async: bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (!c->copy_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->ev_upload_done, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_remap_done, hipEventDisableTiming));
    // first upload of this context: a remap of an earlier, not yet synchronised cms_frames_process may still read the staging buffer, and it
    // was enqueued before this event existed -- everything the frame stream holds so far goes in front of the copy
    HIPCHK(hipEventRecord(c->ev_remap_done, c->stream));
    c->remap_recorded = true;
  }
  if (c->remap_recorded) HIPCHK(hipStreamWaitEvent(c->copy_stream, c->ev_remap_done, 0));   // write-after-read on the staging buffer
  if (fstride == c->fstride && (frame_pitch == c->fish_pitch || B == 1)) {
    HIPCHK(hipMemcpyAsync(c->d_fish, fisheye, c->fish_pitch * (size_t)B, hipMemcpyHostToDevice, c->copy_stream));   // caller uses the device layout
  } else {
    for (int b = 0; b < B; ++b)
      HIPCHK(hipMemcpy2DAsync(c->d_fish + (size_t)b * c->fish_pitch, c->fstride, fisheye + (size_t)b * frame_pitch, fstride,
                              c->cam.Iw, c->cam.Ih, hipMemcpyHostToDevice, c->copy_stream));
  }
  HIPCHK(hipEventRecord(c->ev_upload_done, c->copy_stream));
  c->upload_pending = true;
  return CMS_OK;
}
// device-to-device variant for inputs that are already resident in HBM (several batches kept on the device, one staging buffer):
// d_src has the staging layout [B][Ih][fisheye_stride]; asynchronous on the ctx stream, in order with cms_frames_process
extern "C" int cms_frames_upload_device(cms_ctx* c, const void* d_src, int B) {
  if (!c || !d_src || B < 1 || B > c->max_batch) return cms_fail(CMS_ERR_ARG, "cms_frames_upload_device: bad argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(c->d_fish, d_src, c->fish_pitch * (size_t)B, hipMemcpyDeviceToDevice, c->stream));
  return CMS_OK;
}
// a device-to-device copy on the frame path's stream: what a host keeps between its own device-resident arrays and the calls above and below (e.g. the
// per-key-point map-point slots of a batch, reset before every tracking pass: Frame::Frame starts with mvpMapPoints all NULL, Frame.cpp:99-102)
extern "C" int cms_stream_copy_device(cms_ctx* c, void* d_dst, const void* d_src, size_t bytes) {
  if (!c || (bytes > 0 && (!d_dst || !d_src))) return cms_fail(CMS_ERR_ARG, "cms_stream_copy_device: bad argument");
  if (bytes == 0) return CMS_OK;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, c->stream));
  return CMS_OK;
}
extern "C" int cms_frames_upload_wait(cms_ctx* c) {
  if (!c) return cms_fail(CMS_ERR_ARG, "null ctx");
  HIPCHK(hipSetDevice(c->device));
  if (c->copy_stream) HIPCHK(hipStreamSynchronize(c->copy_stream));
  return CMS_OK;
}
extern "C" int cms_host_alloc(void** out, size_t bytes) {
  if (!out || bytes == 0) return cms_fail(CMS_ERR_ARG, "cms_host_alloc: bad argument");
  HIPCHK(hipHostMalloc(out, bytes));
  return CMS_OK;
}
extern "C" void cms_host_free(void* p) { if (p) (void)hipHostFree(p); }

// Which definition of cv::GaussianBlur's 8-bit column pass the descriptors are computed from: 0 (default) the integer formula
// (sum + 32768) >> 16, 1 the float evaluation of an x86 (SSE2) build of OpenCV <= 3.2, which rounds ties to even.
// A maintainer with a real OpenCV picks the one that matches it; the two differ on rare pixels by one grey level.
extern "C" int cms_set_gaussian_mode(cms_ctx* c, int column_mode) {
  if (!c || column_mode < 0 || column_mode > 1) return cms_fail(CMS_ERR_ARG, "cms_set_gaussian_mode: mode must be 0 or 1");
  c->g.gauss_column_mode = column_mode;
  return CMS_OK;
}

extern "C" int cms_set_mask(cms_ctx* c, const uint8_t* mask, int mstride) {
  if (!c || !mask || mstride < c->g.W) return cms_fail(CMS_ERR_ARG, "cms_set_mask: bad argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpy2DAsync(c->d_mask, c->mstride, mask, mstride, c->g.W, c->g.W, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->have_mask = true;
  return CMS_OK;
}

static int cms_launch_frames(cms_ctx* c, int B, int from_fisheye) {
  c->last_batch = B;
  c->g.skip_zero_cells = from_fisheye ? 1 : 0;   // a caller-supplied canvas (cms_extract) may hold anything in its corner blocks
  // the corner blocks of every level are 0 from cms_ctx_create on and k_remap never writes there; only a caller-supplied canvas
  // can dirty them, in which case the next remapped launch rewrites the zeros in full
  const int clean = (from_fisheye && c->dirty_frames == 0) ? 1 : 0;
  if (!from_fisheye) c->dirty_frames = std::max(c->dirty_frames, B);
  else if (B >= c->dirty_frames) c->dirty_frames = 0;
  const CmsGeom& g = c->g;
  const int L = g.nlevels;
  hipStream_t s = c->stream;
  if (c->prof) hipEventRecord(c->ev[0], s);
  if (from_fisheye) {
    if (c->upload_pending) { HIPCHK(hipStreamWaitEvent(s, c->ev_upload_done, 0)); c->upload_pending = false; }   // streamed input (cms_frames_upload_async)
    dim3 grid((g.W / 4 + 255) / 256 + 1, g.W, std::min((B + CMS_REMAP_FPT - 1) / CMS_REMAP_FPT, CMS_REMAP_ZSPLIT));
    hipLaunchKernelGGL(k_remap, grid, dim3(256), 0, s, (const uint8_t*)c->d_fish, c->fish_pitch, c->fstride, c->cam.Iw,
                       c->cam.Ih, (const uint32_t*)c->d_lut, c->lut_stride, c->d_pyr, g.pyr_bytes, g.W, g.lv[0].stride, g.F, clean ? 0 : 1, B);
    if (c->copy_stream) { HIPCHK(hipEventRecord(c->ev_remap_done, s)); c->remap_recorded = true; }   // the staging buffer is free from here on
  }
  if (c->prof) hipEventRecord(c->ev[1], s);
  // One launch per level (k_resize).  CMS_RESIZE_FUSED=1: levels (1, 2), (3, 4), (5, 6) two per launch (k_resize2: the intermediate level is computed in LDS
  // and never read back) -- bit-identical, and measured 2.7x SLOWER as written (2.42 against 0.89 ms per 256 frames for the pyramid): the resize is bound by its
  // per-pixel integer arithmetic and LDS byte reads, not by HBM, so the saved read buys nothing and the second stage's irregular rectangle costs.  Kept as an
  // experiment behind the switch.
  static const bool rz_single = getenv("CMS_RESIZE_FUSED") == nullptr;
  for (int l = 1; l < L; ++l) {
    dim3 block(64, 4);
    const double ratio = (double)g.lv[l - 1].w / g.lv[l].w;
    if (!rz_single && l + 1 < L) {
      const double ratio2 = (double)g.lv[l].w / g.lv[l + 1].w;
      const int la = (int)align_up((size_t)ceil(256 * ratio2) + 34, 16);                 // mid rectangle: k_resize's staged width for the tile of level l + 1
      const int ls = (int)align_up((size_t)ceil((la + 2) * ratio) + 34, 16);             // ... and the src rectangle behind it
      // rows of mid a tile can need: floor(yb r) - 1 .. ceil((yl + 1) r) + 1 -> CMS_RZ_ROWS r + 4; rows of src behind them likewise (24 and 33 at the
      // pyramid's 1.2; the kernel's buffers hold CMS_RZ2_AROWS / CMS_RZ2_SROWS)
      const int arows = (int)ceil(CMS_RZ_ROWS * ratio2) + 4;
      const int srows = (int)ceil(arows * ratio) + 4;
      if (la <= 512 && ls <= 512 && arows <= CMS_RZ2_AROWS && srows <= CMS_RZ2_SROWS) {
        const CmsLevel& d = g.lv[l + 1];
        CmsResize2 a;
        a.src = g.lv[l - 1]; a.mid = g.lv[l]; a.dst = d;
        a.tabx1 = (const CmsResizeTab*)(c->d_tab + g.lv[l].tab_off); a.taby1 = a.tabx1 + g.lv[l].w;
        a.tabx2 = (const CmsResizeTab*)(c->d_tab + d.tab_off); a.taby2 = a.tabx2 + d.w;
        a.ls = ls; a.la = la; a.arows = CMS_RZ2_AROWS;
        a.lo1 = (int)floor(ratio * 65536.0); a.hi1 = (int)ceil(ratio * 65536.0); a.lo2 = (int)floor(ratio2 * 65536.0); a.hi2 = (int)ceil(ratio2 * 65536.0);
        a.skip_zero = clean;
        const size_t lds = (size_t)ls * CMS_RZ2_SROWS + (size_t)la * CMS_RZ2_AROWS + (size_t)(la + 16) * sizeof(CmsResizeTab);
        dim3 grid((d.w + 255) / 256, (d.h + CMS_RZ_ROWS - 1) / CMS_RZ_ROWS, B);
        hipLaunchKernelGGL(k_resize2, grid, block, lds, s, c->d_pyr, g.pyr_bytes, a);
        ++l;
        continue;
      }
    }
    const CmsLevel& d = g.lv[l];
    dim3 grid((d.w + 255) / 256, (d.h + CMS_RZ_ROWS - 1) / CMS_RZ_ROWS, B);
    const int ls = (int)align_up((size_t)ceil(256 * ratio) + 34, 16);    // LDS row stride of the staged source rectangle (16-byte columns)
    const int lrows = (int)ceil(CMS_RZ_ROWS * ratio) + 7;               // the kernel's integer bounds can be one row wider on either side
    hipLaunchKernelGGL(k_resize, grid, block, (size_t)ls * lrows, s, c->d_pyr, g.pyr_bytes, g.lv[l - 1], d,
                       (const CmsResizeTab*)(c->d_tab + d.tab_off), (const CmsResizeTab*)(c->d_tab + d.tab_off + d.w), ls, clean,
                       (int)floor(ratio * 65536.0), (int)ceil(ratio * 65536.0));
  }
  if (c->prof) hipEventRecord(c->ev[2], s);
  HIPCHK(hipMemsetAsync(c->d_cell_cnt, 0, (size_t)B * g.total_cells * sizeof(int), s));
  {
#if CMS_FAST_LIST
    const int* list = from_fisheye ? c->d_cells_nz : c->d_cells_all;
    const int nlist = from_fisheye ? c->n_cells_nz : c->n_cells_all;
#else
    const int* list = nullptr;
    const int nlist = g.total_cells;
#endif
    if (nlist > 0)
      hipLaunchKernelGGL(k_fast_cells, dim3((nlist + CMS_FAST_WPB - 1) / CMS_FAST_WPB, B), dim3(64 * CMS_FAST_WPB), c->fast_lds, s, (const uint8_t*)c->d_pyr, g.pyr_bytes, g,
                         list, nlist, c->d_cell_cand, c->d_cell_cnt, c->d_overflow);
  }
  if (c->prof) hipEventRecord(c->ev[3], s);
  hipLaunchKernelGGL(k_quadtree, dim3(L, B), dim3(512), c->qt_lds, s, g, (const uint32_t*)c->d_cell_cand, (const int*)c->d_cell_cnt,
                     c->d_cand, c->d_cand_cnt, c->d_overflow,
                     c->d_node, c->d_qt_out, c->d_qt_cnt, c->desc_spatial ? c->d_walk_cnt : nullptr);
  if (c->prof) hipEventRecord(c->ev[4], s);
  hipLaunchKernelGGL(k_cull, dim3(B), dim3(256), 0, s, g, (const uint32_t*)c->d_qt_out, (const int*)c->d_qt_cnt,
                     (const uint8_t*)c->d_mask, c->mstride, c->d_kps, c->d_aux, c->d_kp_cnt, c->desc_spatial ? c->d_order : nullptr, c->d_aux_sorted, c->d_walk_cnt, c->d_rays);
  if (c->prof) hipEventRecord(c->ev[5], s);
  {
    auto kdesc = g.gauss_column_mode == 1 ? k_describe_sse2 : k_describe;
    if (c->desc_spatial)      // the batch's walk in eight segments, one per XCD (see k_cull); the grid covers the longest walk possible (+ 8: the segments' rounding)
      hipLaunchKernelGGL(kdesc, dim3((unsigned)(((size_t)B * g.kp_cap + 8 + CMS_DESC_WPB - 1) / CMS_DESC_WPB)), dim3(64 * CMS_DESC_WPB), 0, s,
                         (const uint8_t*)c->d_pyr, g.pyr_bytes, g, c->d_kps, (const uint32_t*)c->d_aux, (const int*)c->d_kp_cnt, (const float*)c->d_pattern, c->d_desc,
                         (const uint32_t*)c->d_order, (const uint32_t*)c->d_aux_sorted, (const int*)c->d_walk_cnt);
    else
      hipLaunchKernelGGL(kdesc, dim3((g.kp_cap + CMS_DESC_WPB - 1) / CMS_DESC_WPB, B), dim3(64 * CMS_DESC_WPB), 0, s, (const uint8_t*)c->d_pyr, g.pyr_bytes, g, c->d_kps,
                         (const uint32_t*)c->d_aux, (const int*)c->d_kp_cnt, (const float*)c->d_pattern, c->d_desc, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const int*)nullptr);
  }
  if (c->prof) hipEventRecord(c->ev[6], s);
  if (c->ev_extracted) { HIPCHK(hipEventRecord(c->ev_extracted, s)); c->extracted_recorded = true; }
  HIPCHK(hipGetLastError());
  return CMS_OK;
}

// Scheduling aid: make another HIP stream (a mapping-side context's, a local-BA group's) wait on the device until the extraction
// launched by the last cms_frames_process of `c` has finished.  The extraction kernels and the local-BA chain each fill the chip on their
// own; run side by side they only slow each other down (the extractor drops from 43 % to 33 % of its byte roofline inside bench.py's
// step, the step is no shorter), so the mapping side of a step is queued behind the extraction and overlaps the tracking kernels instead.
extern "C" int cms_stream_wait_extracted(cms_ctx* c, void* hip_stream) {
  if (!c || !hip_stream) return cms_fail(CMS_ERR_ARG, "cms_stream_wait_extracted: bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (!c->ev_extracted) { HIPCHK(hipEventCreateWithFlags(&c->ev_extracted, hipEventDisableTiming)); return CMS_OK; }   // armed from the next process call on
  if (c->extracted_recorded) HIPCHK(hipStreamWaitEvent((hipStream_t)hip_stream, c->ev_extracted, 0));
  return CMS_OK;
}

extern "C" int cms_frames_process(cms_ctx* c, int B, int from_fisheye) {
  if (!c || B < 1 || B > c->max_batch) return cms_fail(CMS_ERR_ARG, "cms_frames_process: bad batch");
  HIPCHK(hipSetDevice(c->device));
  return cms_launch_frames(c, B, from_fisheye);
}
extern "C" int cms_frames_sync(cms_ctx* c) {
  if (!c) return cms_fail(CMS_ERR_ARG, "null ctx");
  HIPCHK(hipSetDevice(c->device));
  static const bool relaxed = getenv("CMS_BA_RELAXED_WAIT") != nullptr;      // (one switch for every host wait that can sleep: see ba_wait_stream)
  if (relaxed && c->last_batch >= 32) {
    // a large batch keeps the stream busy for milliseconds: the calling thread sleeps until the stream is through instead of spinning on it (a
    // host core per context otherwise; the wake-up costs some tens of microseconds, nothing against such a batch).  Small batches: spin as before
    // (polled with short sleeps: hipEventSynchronize on a hipEventBlockingSync event was measured to keep the core just as busy on this runtime)
    if (!c->ev_block) HIPCHK(hipEventCreateWithFlags(&c->ev_block, hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->ev_block, c->stream));
    for (;;) {
      const hipError_t q = hipEventQuery(c->ev_block);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) return cms_fail(CMS_ERR_HIP, "cms_frames_sync", q);
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  int ov = 0;
  HIPCHK(hipMemcpy(&ov, c->d_overflow, sizeof(int), hipMemcpyDeviceToHost));
  if (ov) return cms_fail(CMS_ERR_OVERFLOW, "candidate list overflow");
  return CMS_OK;
}
extern "C" int cms_frames_results(cms_ctx* c, void** d_kps, void** d_desc, void** d_counts) {
  if (!c) return cms_fail(CMS_ERR_ARG, "null ctx");
  if (d_kps) *d_kps = c->d_kps;
  if (d_desc) *d_desc = c->d_desc;
  if (d_counts) *d_counts = c->d_kp_cnt;
  return CMS_OK;
}
extern "C" int cms_frames_rays(cms_ctx* c, void** d_rays) {
  if (!c || !d_rays) return cms_fail(CMS_ERR_ARG, "cms_frames_rays: bad argument");
  *d_rays = c->d_rays;
  return CMS_OK;
}
extern "C" int cms_frames_fetch_rays(cms_ctx* c, int b, float* rays, int cap, int* n) {
  if (!c || b < 0 || b >= c->max_batch || !n || !rays) return cms_fail(CMS_ERR_ARG, "cms_frames_fetch_rays: bad argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  int cnt = 0;
  HIPCHK(hipMemcpy(&cnt, c->d_kp_cnt + b, sizeof(int), hipMemcpyDeviceToHost));
  *n = cnt;
  if (cnt > cap) return cms_fail(CMS_ERR_OVERFLOW, "cms_frames_fetch_rays: caller capacity too small");
  if (cnt > 0) HIPCHK(hipMemcpy(rays, c->d_rays + (size_t)b * c->g.kp_cap * 3, (size_t)cnt * 12, hipMemcpyDeviceToHost));
  return CMS_OK;
}
extern "C" int cms_frames_fetch(cms_ctx* c, int b, cms_keypoint* kps, uint8_t* desc, int cap, int* n) {
  if (!c || b < 0 || b >= c->max_batch || !n) return cms_fail(CMS_ERR_ARG, "cms_frames_fetch: bad argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  int cnt = 0;
  HIPCHK(hipMemcpy(&cnt, c->d_kp_cnt + b, sizeof(int), hipMemcpyDeviceToHost));
  *n = cnt;
  if (cnt > cap) return cms_fail(CMS_ERR_OVERFLOW, "cms_frames_fetch: caller capacity too small");
  if (cnt > 0) {
    if (kps) HIPCHK(hipMemcpy(kps, c->d_kps + (size_t)b * c->g.kp_cap, (size_t)cnt * sizeof(cms_keypoint), hipMemcpyDeviceToHost));
    if (desc) HIPCHK(hipMemcpy(desc, c->d_desc + (size_t)b * c->g.kp_cap * 32, (size_t)cnt * 32, hipMemcpyDeviceToHost));
  }
  return CMS_OK;
}

extern "C" int cms_remap(cms_ctx* c, const uint8_t* fisheye, int fstride, uint8_t* cubemap, int cstride) {
  if (!c || !fisheye || !cubemap || cstride < c->g.W) return cms_fail(CMS_ERR_ARG, "cms_remap: bad argument");
  int rc = cms_frames_upload(c, fisheye, fstride, 0, 1);
  if (rc) return rc;
  const CmsGeom& g = c->g;
  dim3 grid((g.W / 4 + 255) / 256 + 1, g.W, 1);
  hipLaunchKernelGGL(k_remap, grid, dim3(256), 0, c->stream, (const uint8_t*)c->d_fish, c->fish_pitch, c->fstride, c->cam.Iw,
                     c->cam.Ih, (const uint32_t*)c->d_lut, c->lut_stride, c->d_pyr, g.pyr_bytes, g.W, g.lv[0].stride, g.F, 1, 1);
  const int F = g.F;
  const int fx0[5] = {F, 0, 2 * F, F, F}, fy0[5] = {F, F, F, 0, 2 * F};
  for (int f = 0; f < 5; ++f)
    HIPCHK(hipMemcpy2DAsync(cubemap + (size_t)fy0[f] * cstride + fx0[f], cstride, c->d_pyr + (size_t)fy0[f] * g.lv[0].stride + fx0[f],
                            g.lv[0].stride, F, F, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return CMS_OK;
}

extern "C" int cms_extract(cms_ctx* c, const uint8_t* cubemap, int cstride, cms_keypoint* kps, uint8_t* desc, int cap, int* n) {
  if (!c || !cubemap || !n || cstride < c->g.W) return cms_fail(CMS_ERR_ARG, "cms_extract: bad argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpy2DAsync(c->d_pyr, c->g.lv[0].stride, cubemap, cstride, c->g.W, c->g.W, hipMemcpyHostToDevice, c->stream));
  int rc = cms_launch_frames(c, 1, 0);
  if (rc) return rc;
  rc = cms_frames_sync(c);
  if (rc) return rc;
  return cms_frames_fetch(c, 0, kps, desc, cap, n);
}
static int cms_hstage(cms_ctx* c, size_t bytes);
// One frame, host buffers in and out: ONE synchronisation.  The image leaves from the pinned upload staging without a wait, the kernels follow,
// and overflow flag | key-point count | key points | descriptors come back as four asynchronous copies into the pinned result staging behind
// them (the upload, cms_frames_sync and cms_frames_fetch one after the other were five synchronous round trips of 15-20 us each).
static int cms_remap_extract_impl(cms_ctx* c, const uint8_t* fisheye, int fstride, cms_keypoint* kps, uint8_t* desc, float* rays, int cap, int* n);
extern "C" int cms_remap_extract(cms_ctx* c, const uint8_t* fisheye, int fstride, cms_keypoint* kps, uint8_t* desc, int cap, int* n) {
  return cms_remap_extract_impl(c, fisheye, fstride, kps, desc, nullptr, cap, n);
}
extern "C" int cms_remap_extract_rays(cms_ctx* c, const uint8_t* fisheye, int fstride, cms_keypoint* kps, uint8_t* desc, float* rays, int cap, int* n) {
  return cms_remap_extract_impl(c, fisheye, fstride, kps, desc, rays, cap, n);
}
static int cms_remap_extract_impl(cms_ctx* c, const uint8_t* fisheye, int fstride, cms_keypoint* kps, uint8_t* desc, float* rays, int cap, int* n) {
  if (!c || !fisheye || !n || cap < 0) return cms_fail(CMS_ERR_ARG, "cms_remap_extract: bad argument");
  int rc = cms_frames_upload_impl(c, fisheye, fstride, 0, 1, false);
  if (rc) return rc;
  rc = cms_launch_frames(c, 1, 1);
  if (rc) return rc;
  const int m = std::min(cap, c->g.kp_cap);
  const size_t o_kp = 256, o_desc = o_kp + (((size_t)m * sizeof(cms_keypoint) + 255) & ~(size_t)255), o_rays = o_desc + (((size_t)m * 32 + 255) & ~(size_t)255),
               total = o_rays + (rays ? (size_t)m * 12 : 0);
  rc = cms_hstage(c, total);
  if (rc) return rc;
  uint8_t* h = c->h_stage;
  hipStream_t s = c->stream;
  HIPCHK(hipMemcpyAsync(h, c->d_overflow, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(h + 64, c->d_kp_cnt, sizeof(int), hipMemcpyDeviceToHost, s));
  if (m > 0 && kps) HIPCHK(hipMemcpyAsync(h + o_kp, c->d_kps, (size_t)m * sizeof(cms_keypoint), hipMemcpyDeviceToHost, s));
  if (m > 0 && desc) HIPCHK(hipMemcpyAsync(h + o_desc, c->d_desc, (size_t)m * 32, hipMemcpyDeviceToHost, s));
  if (m > 0 && rays) HIPCHK(hipMemcpyAsync(h + o_rays, c->d_rays, (size_t)m * 12, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  int ov = 0, cnt = 0;
  memcpy(&ov, h, sizeof(int)); memcpy(&cnt, h + 64, sizeof(int));
  if (ov) return cms_fail(CMS_ERR_OVERFLOW, "candidate list overflow");
  *n = cnt;
  if (cnt > cap) return cms_fail(CMS_ERR_OVERFLOW, "cms_frames_fetch: caller capacity too small");
  if (cnt > 0) {
    if (kps) memcpy(kps, h + o_kp, (size_t)cnt * sizeof(cms_keypoint));
    if (desc) memcpy(desc, h + o_desc, (size_t)cnt * 32);
    if (rays) memcpy(rays, h + o_rays, (size_t)cnt * 12);
  }
  return CMS_OK;
}

// ---- debug read-back
extern "C" int cms_debug_lut(cms_ctx* c, uint32_t* out, int* stride) {
  if (!c || !out) return cms_fail(CMS_ERR_ARG, "null");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpy(out, c->d_lut, (size_t)c->g.W * c->lut_stride * 4, hipMemcpyDeviceToHost));
  if (stride) *stride = c->lut_stride;
  return CMS_OK;
}
extern "C" int cms_debug_level(cms_ctx* c, int b, int level, uint8_t* dst, int dstride) {
  if (!c || !dst || b < 0 || b >= c->max_batch || level < 0 || level >= c->g.nlevels) return cms_fail(CMS_ERR_ARG, "cms_debug_level");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  const CmsLevel& lv = c->g.lv[level];
  HIPCHK(hipMemcpy2D(dst, dstride, c->d_pyr + (size_t)b * c->g.pyr_bytes + lv.off, lv.stride, lv.w, lv.h, hipMemcpyDeviceToHost));
  return CMS_OK;
}
extern "C" int cms_debug_cubemap(cms_ctx* c, int b, uint8_t* dst, int dstride) { return cms_debug_level(c, b, 0, dst, dstride); }
extern "C" int cms_debug_candidates(cms_ctx* c, int b, int level, int* xys, int cap, int* n) {
  if (!c || !n || b < 0 || b >= c->max_batch || level < 0 || level >= c->g.nlevels) return cms_fail(CMS_ERR_ARG, "cms_debug_candidates");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  int cnt = 0;
  HIPCHK(hipMemcpy(&cnt, c->d_cand_cnt + b * c->g.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
  const CmsLevel& lv = c->g.lv[level];
  if (cnt > lv.cand_cap) cnt = lv.cand_cap;
  *n = cnt;
  std::vector<uint32_t> tmp(cnt > 0 ? cnt : 1);
  if (cnt > 0) HIPCHK(hipMemcpy(tmp.data(), c->d_cand + (size_t)b * c->g.cand_total + lv.cand_off, (size_t)cnt * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < cnt && i < cap; ++i) {
    xys[3 * i] = (int)(tmp[i] & 0xFFF) - CMS_MINB; xys[3 * i + 1] = (int)((tmp[i] >> 12) & 0xFFF) - CMS_MINB; xys[3 * i + 2] = (int)(tmp[i] >> 24);
  }
  return CMS_OK;
}
extern "C" int cms_debug_distributed(cms_ctx* c, int b, int level, int* xys, int cap, int* n) {
  if (!c || !n || b < 0 || b >= c->max_batch || level < 0 || level >= c->g.nlevels) return cms_fail(CMS_ERR_ARG, "cms_debug_distributed");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  int cnt = 0;
  HIPCHK(hipMemcpy(&cnt, c->d_qt_cnt + b * c->g.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
  *n = cnt;
  const CmsLevel& lv = c->g.lv[level];
  std::vector<uint32_t> tmp(cnt > 0 ? cnt : 1);
  if (cnt > 0) HIPCHK(hipMemcpy(tmp.data(), c->d_qt_out + (size_t)b * c->g.kp_cap + lv.kp_off, (size_t)cnt * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < cnt && i < cap; ++i) {
    xys[3 * i] = (int)(tmp[i] & 0xFFF); xys[3 * i + 1] = (int)((tmp[i] >> 12) & 0xFFF); xys[3 * i + 2] = (int)(tmp[i] >> 24);
  }
  return CMS_OK;
}

extern "C" int cms_profile_enable(cms_ctx* c, int on) { if (!c) return CMS_ERR_ARG; c->prof = on != 0; return CMS_OK; }
extern "C" int cms_profile_get(cms_ctx* c, float* ms7) {
  if (!c || !ms7) return cms_fail(CMS_ERR_ARG, "null");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int i = 0; i < 6; ++i) HIPCHK(hipEventElapsedTime(&ms7[i], c->ev[i], c->ev[i + 1]));
  HIPCHK(hipEventElapsedTime(&ms7[6], c->ev[0], c->ev[6]));
  return CMS_OK;
}

// ---- matching
extern "C" int cms_hamming_best2_device(cms_ctx* c, const void* qdesc, const void* q_row, int nq, const void* tdesc, const void* cand_off,
                                        const void* cand_idx, const void* t_level, const void* t_excluded, void* best_idx,
                                        void* best_dist, void* best_level, void* second_dist, void* second_level) {
  if (!c || nq < 0) return cms_fail(CMS_ERR_ARG, "cms_hamming_best2_device: bad argument");
  if (nq == 0) return CMS_OK;
  HIPCHK(hipSetDevice(c->device));
  hipLaunchKernelGGL(k_hamming_best2, dim3((nq + 3) / 4), dim3(256), 0, c->stream, (const uint4*)qdesc, (const int*)q_row, nq, (const uint4*)tdesc,
                     (const int*)cand_off, (const int*)cand_idx, (const int*)t_level, (const uint8_t*)t_excluded, (int*)best_idx,
                     (int*)best_dist, (int*)best_level, (int*)second_dist, (int*)second_level);
  HIPCHK(hipGetLastError());
  return CMS_OK;
}
static int cms_scratch(cms_ctx* c, size_t bytes) {
  if (bytes <= c->match_bytes) return CMS_OK;
  if (c->d_match) hipFree(c->d_match);
  c->d_match = nullptr; c->match_bytes = 0;
  HIPCHK(hipMalloc(&c->d_match, bytes));
  c->match_bytes = bytes;
  return CMS_OK;
}
static int cms_hstage(cms_ctx* c, size_t bytes) {
  if (bytes <= c->h_stage_bytes) return CMS_OK;
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  c->h_stage = nullptr; c->h_stage_bytes = 0;
  HIPCHK(hipHostMalloc((void**)&c->h_stage, bytes + bytes / 2));
  c->h_stage_bytes = bytes + bytes / 2;
  return CMS_OK;
}
extern "C" int cms_hamming_best2(cms_ctx* c, const uint8_t* qdesc, int nq, const uint8_t* tdesc, int nt, const int* cand_off,
                                 const int* cand_idx, const int* t_level, const uint8_t* t_excluded, int* best_idx, int* best_dist,
                                 int* best_level, int* second_dist, int* second_level) {
  if (!c || nq < 0 || nt < 0 || (nq > 0 && (!qdesc || !cand_off || !best_idx || !best_dist || !second_dist)))
    return cms_fail(CMS_ERR_ARG, "cms_hamming_best2: bad argument");
  if (nq == 0) return CMS_OK;
  HIPCHK(hipSetDevice(c->device));
  const int ncand = cand_off[nq];
  for (int q = 0; q < nq; ++q)
    if (cand_off[q + 1] - cand_off[q] >= (1 << 22)) return cms_fail(CMS_ERR_UNSUPPORTED, "candidate list longer than 2^22");
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes, 256); return r; };
  const size_t oq = take((size_t)nq * 32), ot = take((size_t)nt * 32 + 32), ooff = take((size_t)(nq + 1) * 4), oidx = take((size_t)ncand * 4 + 4),
               olv = take((size_t)nt * 4 + 4), oex = take((size_t)nt + 4), oout = take((size_t)nq * 4 * 5);
  int rc = cms_scratch(c, o);
  if (rc) return rc;
  uint8_t* base = (uint8_t*)c->d_match;
  hipStream_t s = c->stream;
  HIPCHK(hipMemcpyAsync(base + oq, qdesc, (size_t)nq * 32, hipMemcpyHostToDevice, s));
  if (nt > 0) HIPCHK(hipMemcpyAsync(base + ot, tdesc, (size_t)nt * 32, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(base + ooff, cand_off, (size_t)(nq + 1) * 4, hipMemcpyHostToDevice, s));
  if (ncand > 0) HIPCHK(hipMemcpyAsync(base + oidx, cand_idx, (size_t)ncand * 4, hipMemcpyHostToDevice, s));
  if (t_level && nt > 0) HIPCHK(hipMemcpyAsync(base + olv, t_level, (size_t)nt * 4, hipMemcpyHostToDevice, s));
  if (t_excluded && nt > 0) HIPCHK(hipMemcpyAsync(base + oex, t_excluded, (size_t)nt, hipMemcpyHostToDevice, s));
  int* dout = (int*)(base + oout);
  rc = cms_hamming_best2_device(c, base + oq, nullptr, nq, base + ot, base + ooff, base + oidx, t_level ? base + olv : nullptr,
                                t_excluded ? base + oex : nullptr, dout, dout + nq, dout + 2 * nq, dout + 3 * nq, dout + 4 * nq);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(best_idx, dout, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(best_dist, dout + nq, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
  if (best_level) HIPCHK(hipMemcpyAsync(best_level, dout + 2 * nq, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(second_dist, dout + 3 * nq, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
  if (second_level) HIPCHK(hipMemcpyAsync(second_level, dout + 4 * nq, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return CMS_OK;
}
extern "C" int cms_hamming_matrix(cms_ctx* c, const uint8_t* a, int na, const uint8_t* b, int nb, uint16_t* out) {
  if (!c || na < 0 || nb < 0) return cms_fail(CMS_ERR_ARG, "cms_hamming_matrix: bad argument");
  if (na == 0 || nb == 0) return CMS_OK;
  HIPCHK(hipSetDevice(c->device));
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes, 256); return r; };
  const size_t oa = take((size_t)na * 32), ob = take((size_t)nb * 32), oo = take((size_t)na * nb * 2);
  int rc = cms_scratch(c, o);
  if (rc) return rc;
  uint8_t* base = (uint8_t*)c->d_match;
  hipStream_t s = c->stream;
  HIPCHK(hipMemcpyAsync(base + oa, a, (size_t)na * 32, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(base + ob, b, (size_t)nb * 32, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_hamming_matrix, dim3((nb + 15) / 16, (na + 15) / 16), dim3(256), 0, s, (const uint4*)(base + oa), na,
                     (const uint4*)(base + ob), nb, (uint16_t*)(base + oo));
  HIPCHK(hipMemcpyAsync(out, base + oo, (size_t)na * nb * 2, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return CMS_OK;
}
