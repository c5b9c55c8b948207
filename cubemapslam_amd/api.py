"""ctypes binding of libcubemapslam_hip.so (the C-ABI in include/cubemapslam_hip.h).

Plumbing only: every call goes straight to the hand-written HIP library.  There is NO CPU fallback: if the shared
library is missing or no MI355X is visible the calls raise.
"""
import time
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CMS_HIP_LIB") or os.path.join(_HERE, "lib", "libcubemapslam_hip.so")   # override: developer A/B builds (tools/ab_build.sh)


class CmsError(RuntimeError):
    pass


class Camera(C.Structure):
    _fields_ = [("c", C.c_double), ("d", C.c_double), ("e", C.c_double), ("u0", C.c_double), ("v0", C.c_double),
                ("invpol", C.c_double * 12), ("pol", C.c_double * 5), ("Iw", C.c_int), ("Ih", C.c_int),
                ("face", C.c_int), ("fov_deg", C.c_double)]


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int)]


class Geometry(C.Structure):
    _fields_ = [("W", C.c_int), ("F", C.c_int), ("nlevels", C.c_int), ("kp_cap", C.c_int), ("max_batch", C.c_int),
                ("level_w", C.c_int * 12), ("level_h", C.c_int * 12), ("level_quota", C.c_int * 12),
                ("level_cells", C.c_int * 12), ("scale", C.c_float * 12), ("inv_scale", C.c_float * 12),
                ("sigma2", C.c_float * 12), ("inv_sigma2", C.c_float * 12), ("pyramid_bytes_per_frame", C.c_size_t),
                ("candidate_entries_per_frame", C.c_size_t), ("fisheye_stride", C.c_int)]


class BaStats(C.Structure):
    _fields_ = [("iterations_done", C.c_int * 2), ("chi2_initial", C.c_double * 2), ("chi2_final", C.c_double * 2),
                ("lambda_final", C.c_double * 2), ("n_outliers_mid", C.c_int), ("n_outliers_final", C.c_int)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4")])

_lib = None


def lib():
    """Load the HIP library; raises if it has not been built (python -m cubemapslam_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CmsError("libcubemapslam_hip.so is missing -- build it with `python -m cubemapslam_amd.build` "
                           "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.cms_last_error.restype = C.c_char_p
        for n in ("cms_ctx_stream", "cms_frames_input", "cms_ba_stream"):
            getattr(L, n).restype = C.c_void_p
            getattr(L, n).argtypes = [C.c_void_p]
        L.cms_ctx_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.cms_ctx_destroy.argtypes = [C.c_void_p]
        L.cms_ctx_destroy.restype = None
        L.cms_ctx_geometry.argtypes = [C.c_void_p, C.c_void_p]
        L.cms_remap.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.cms_set_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.cms_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.cms_remap_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.cms_frames_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int]
        L.cms_frames_upload_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int]
        L.cms_frames_upload_wait.argtypes = [C.c_void_p]
        L.cms_frames_upload_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.cms_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
        L.cms_host_free.argtypes = [C.c_void_p]
        L.cms_host_free.restype = None
        L.cms_ba_profile_kernel.argtypes = [C.c_void_p, C.c_int]
        L.cms_ba_profile_get.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cms_ba_debug_compose.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cms_frames_process.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.cms_frames_sync.argtypes = [C.c_void_p]
        L.cms_frames_results.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cms_frames_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.cms_debug_lut.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cms_debug_cubemap.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.cms_debug_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.cms_debug_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.cms_debug_distributed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.cms_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.cms_profile_get.argtypes = [C.c_void_p, C.c_void_p]
        L.cms_hamming_best2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 9
        L.cms_hamming_best2_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 10
        L.cms_hamming_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.cms_ba_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                    C.c_double, C.c_double]
        L.cms_ba_reset.argtypes = [C.c_void_p]
        L.cms_ba_optimize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.cms_ba_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cms_ba_destroy.argtypes = [C.c_void_p]
        L.cms_ba_destroy.restype = None
        L.cms_ba_run.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                 C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cms_ba_linearize.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                       C.c_double, C.c_int, C.c_double] + [C.c_void_p] * 7
        L.cms_area_set_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.cms_area_grid.argtypes = [C.c_void_p, C.c_int]
        L.cms_features_in_area.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p]
        L.cms_features_in_area_device.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_void_p]
        L.cms_features_in_area_batch_device.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 9 + [C.c_int, C.c_void_p]
        L.cms_area_set_descriptors.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.cms_search_local_points.argtypes = ([C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_float] * 3 + [C.c_int, C.c_int] +
                                              [C.c_void_p] * 9)
        L.cms_search_by_projection.argtypes = ([C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_int, C.c_int, C.c_int] +
                                               [C.c_void_p] * 3)
        L.cms_project_last_frame_device.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_float] + [C.c_void_p] * 5
        L.cms_rotation_filter_device.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
        L.cms_is_in_frustum_device.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_float] + [C.c_void_p] * 8
        L.cms_search_local_points_device.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_int] + [C.c_void_p] * 3
        L.cms_create_new_map_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.cms_kfstore_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.cms_kfstore_destroy.argtypes = [C.c_void_p]
        L.cms_kfstore_destroy.restype = None
        L.cms_kfstore_put.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.cms_kfstore_update.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.cms_kfstore_fuse_search.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_float, C.c_void_p, C.c_void_p]
        L.cms_kfstore_fuse_search_sets.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 3 + [C.c_float, C.c_void_p, C.c_void_p]
        L.cms_kfstore_put_from_frame.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]
        L.cms_kfstore_update_poses.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
        L.cms_kfstore_debug_fetch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 12
        L.cms_ba_debug_fetch_plan.argtypes = [C.c_void_p] * 14
        L.cms_kfstore_create_new_map_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.cms_distinctive_descriptors.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cms_update_normal_and_depth.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8
        L.cms_fuse_search.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_void_p, C.c_void_p]
        L.cms_pose_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.cms_pose_destroy.argtypes = [C.c_void_p]
        L.cms_pose_destroy.restype = None
        L.cms_pose_stream.argtypes = [C.c_void_p]
        L.cms_pose_stream.restype = C.c_void_p
        _edge = [C.c_void_p] * 5 + [C.c_double] * 4
        L.cms_pose_upload.argtypes = [C.c_void_p, C.c_int] + _edge + [C.c_void_p]
        L.cms_pose_launch.argtypes = [C.c_void_p]
        L.cms_pose_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cms_pose_optimize_batch.argtypes = [C.c_void_p, C.c_int] + _edge + [C.c_void_p] * 4
        L.cms_pose_optimize.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_double] * 4 + [C.c_void_p] * 4
        _lib = L
    return _lib


def _p(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def _chk(rc, what):
    if rc < 0:
        raise CmsError("%s failed (%d): %s" % (what, rc, lib().cms_last_error().decode()))
    return rc


def make_camera(d):
    cam = Camera()
    for k in ("c", "d", "e", "u0", "v0", "Iw", "Ih", "face", "fov_deg"):
        setattr(cam, k, d[k])
    for i in range(12):
        cam.invpol[i] = d["invpol"][i] if i < len(d["invpol"]) else 0.0
    for i in range(5):
        cam.pol[i] = d["pol"][i] if i < len(d["pol"]) else 0.0
    return cam


class Context:
    """cms_ctx: remap LUT + extractor tables + device buffers for up to max_batch frames."""

    def __init__(self, cam, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, max_batch=1, device=0):
        self.cam = make_camera(cam) if isinstance(cam, dict) else cam
        self.orb = OrbParams(nfeatures, scale_factor, nlevels, ini_th, min_th)
        self.h = C.c_void_p()
        _chk(lib().cms_ctx_create(C.byref(self.h), device, C.byref(self.cam), C.byref(self.orb), max_batch), "cms_ctx_create")
        self.geom = Geometry()
        _chk(lib().cms_ctx_geometry(self.h, C.byref(self.geom)), "cms_ctx_geometry")
        self.W = self.geom.W
        self.max_batch = max_batch

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            lib().cms_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return lib().cms_ctx_stream(self.h)

    def set_mask(self, mask):
        mask = np.ascontiguousarray(mask, np.uint8)
        _chk(lib().cms_set_mask(self.h, _p(mask), mask.strides[0]), "cms_set_mask")

    def set_distance_bounds_mode(self, scaled):
        f = lib().cms_set_distance_bounds_mode
        f.argtypes = [C.c_void_p, C.c_int]
        _chk(f(self.h, int(scaled)), "cms_set_distance_bounds_mode")

    def set_gaussian_mode(self, column_mode):
        f = lib().cms_set_gaussian_mode
        f.argtypes = [C.c_void_p, C.c_int]
        _chk(f(self.h, int(column_mode)), "cms_set_gaussian_mode")

    def remap(self, fisheye, cubemap=None):
        fisheye = np.ascontiguousarray(fisheye, np.uint8)
        if cubemap is None:
            cubemap = np.zeros((self.W, self.W), np.uint8)
        _chk(lib().cms_remap(self.h, _p(fisheye), fisheye.strides[0], _p(cubemap), cubemap.strides[0]), "cms_remap")
        return cubemap

    def extract(self, cubemap, cap=None):
        cubemap = np.ascontiguousarray(cubemap, np.uint8)
        cap = cap or self.geom.kp_cap
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        _chk(lib().cms_extract(self.h, _p(cubemap), cubemap.strides[0], _p(kps), _p(desc), cap, C.byref(n)), "cms_extract")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def remap_extract(self, fisheye, cap=None):
        fisheye = np.ascontiguousarray(fisheye, np.uint8)
        cap = cap or self.geom.kp_cap
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        _chk(lib().cms_remap_extract(self.h, _p(fisheye), fisheye.strides[0], _p(kps), _p(desc), cap, C.byref(n)), "cms_remap_extract")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def remap_extract_rays(self, fisheye, cap=None):
        """cms_remap_extract_rays: key points, descriptors and Frame::mvKeyRays (n x 3 float) of one frame, one synchronisation"""
        fisheye = np.ascontiguousarray(fisheye, np.uint8)
        cap = cap or self.geom.kp_cap
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8); rays = np.zeros((cap, 3), np.float32)
        n = C.c_int()
        _chk(lib().cms_remap_extract_rays(self.h, _p(fisheye), fisheye.strides[0], _p(kps), _p(desc), _p(rays), cap, C.byref(n)), "cms_remap_extract_rays")
        return kps[:n.value].copy(), desc[:n.value].copy(), rays[:n.value].copy()

    def fetch_rays(self, b):
        """cms_frames_fetch_rays: mvKeyRays of frame b of the last process() call"""
        cap = self.geom.kp_cap
        rays = np.zeros((cap, 3), np.float32)
        n = C.c_int()
        _chk(lib().cms_frames_fetch_rays(self.h, b, _p(rays), cap, C.byref(n)), "cms_frames_fetch_rays")
        return rays[:n.value].copy()

    def rays_ptr(self):
        a = C.c_void_p()
        _chk(lib().cms_frames_rays(self.h, C.byref(a)), "cms_frames_rays")
        return a.value

    # ---- batched device path
    def upload(self, frames):
        frames = np.ascontiguousarray(frames, np.uint8)
        assert frames.ndim == 3
        _chk(lib().cms_frames_upload(self.h, _p(frames), frames.strides[1], frames.strides[0], frames.shape[0]), "cms_frames_upload")

    def upload_async(self, frames):
        """input streaming: `frames` must live in pinned memory (host_alloc); the copy runs on the context's copy stream and the next
        process() waits for it on the device"""
        assert frames.ndim == 3 and frames.dtype == np.uint8 and frames.flags.c_contiguous
        _chk(lib().cms_frames_upload_async(self.h, _p(frames), frames.strides[1], frames.strides[0], frames.shape[0]), "cms_frames_upload_async")

    def upload_device(self, d_ptr, B):
        """staging <- device buffer [B][Ih][fisheye_stride] (inputs resident in HBM), asynchronous on the ctx stream"""
        _chk(lib().cms_frames_upload_device(self.h, C.c_void_p(int(d_ptr)), B), "cms_frames_upload_device")

    def stream_wait_extracted(self, hip_stream):
        """the given HIP stream waits (on the device) for the extraction of this context's last process() call"""
        f = lib().cms_stream_wait_extracted
        f.argtypes = [C.c_void_p, C.c_void_p]
        _chk(f(self.h, C.c_void_p(int(hip_stream))), "cms_stream_wait_extracted")

    def upload_wait(self):
        _chk(lib().cms_frames_upload_wait(self.h), "cms_frames_upload_wait")

    def process(self, B, from_fisheye=True):
        _chk(lib().cms_frames_process(self.h, B, 1 if from_fisheye else 0), "cms_frames_process")

    def sync(self):
        _chk(lib().cms_frames_sync(self.h), "cms_frames_sync")

    def fetch(self, b):
        cap = self.geom.kp_cap
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        _chk(lib().cms_frames_fetch(self.h, b, _p(kps), _p(desc), cap, C.byref(n)), "cms_frames_fetch")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def results_ptrs(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _chk(lib().cms_frames_results(self.h, C.byref(a), C.byref(b), C.byref(c)), "cms_frames_results")
        return a.value, b.value, c.value

    def input_ptr(self):
        return lib().cms_frames_input(self.h)

    def profile(self, on=True):
        lib().cms_profile_enable(self.h, 1 if on else 0)

    def profile_ms(self):
        ms = np.zeros(7, np.float32)
        _chk(lib().cms_profile_get(self.h, _p(ms)), "cms_profile_get")
        return dict(zip(("remap", "pyramid", "fast", "octree", "cull", "describe", "total"), ms.tolist()))

    # ---- debug read-back
    def debug_lut(self):
        W = self.W
        out = np.zeros((W, (W + 3) // 4 * 4), np.uint32)
        s = C.c_int()
        _chk(lib().cms_debug_lut(self.h, _p(out), C.byref(s)), "cms_debug_lut")
        return out[:, :W]

    def debug_level(self, b, l):
        w, h = self.geom.level_w[l], self.geom.level_h[l]
        out = np.zeros((h, w), np.uint8)
        _chk(lib().cms_debug_level(self.h, b, l, _p(out), w), "cms_debug_level")
        return out

    def debug_candidates(self, b, l):
        cap = self.geom.level_w[l] * self.geom.level_h[l] // 4 + 4096
        out = np.zeros((cap, 3), np.int32)
        n = C.c_int()
        _chk(lib().cms_debug_candidates(self.h, b, l, _p(out), cap, C.byref(n)), "cms_debug_candidates")
        return out[:n.value].copy()

    def debug_distributed(self, b, l):
        cap = self.geom.level_quota[l] + 8
        out = np.zeros((cap, 3), np.int32)
        n = C.c_int()
        _chk(lib().cms_debug_distributed(self.h, b, l, _p(out), cap, C.byref(n)), "cms_debug_distributed")
        return out[:n.value].copy()

    # ---- matching
    def hamming_best2(self, qdesc, tdesc, cand_off, cand_idx, tlevel=None, texcl=None):
        qdesc = np.ascontiguousarray(qdesc, np.uint8); tdesc = np.ascontiguousarray(tdesc, np.uint8)
        cand_off = np.ascontiguousarray(cand_off, np.int32); cand_idx = np.ascontiguousarray(cand_idx, np.int32)
        tl = np.ascontiguousarray(tlevel, np.int32) if tlevel is not None else None
        tx = np.ascontiguousarray(texcl, np.uint8) if texcl is not None else None
        nq = len(qdesc)
        outs = [np.zeros(max(nq, 1), np.int32) for _ in range(5)]
        _chk(lib().cms_hamming_best2(self.h, _p(qdesc), nq, _p(tdesc), len(tdesc), _p(cand_off), _p(cand_idx), _p(tl), _p(tx),
                                     *[_p(o) for o in outs]), "cms_hamming_best2")
        keys = ("best_idx", "best_dist", "best_level", "second_dist", "second_level")
        return {k: o[:nq] for k, o in zip(keys, outs)}

    def area_set_keypoints(self, b, kps):
        """put caller key points into frame slot b (tests / callers that extracted elsewhere)"""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        _chk(lib().cms_area_set_keypoints(self.h, b, len(kps), _p(kps)), "cms_area_set_keypoints")

    def area_grid(self, B):
        """Frame::AssignFeaturesToGrid for frames 0..B-1 (device-resident key points)"""
        _chk(lib().cms_area_grid(self.h, B), "cms_area_grid")

    def features_in_area(self, b, qx, qy, qr, qmin, qmax, cap=None):
        """Frame::GetFeaturesInArea for a batch of queries against frame b -> (off[nq+1], idx[total]) in the reference's order"""
        qx = np.ascontiguousarray(qx, np.float32); qy = np.ascontiguousarray(qy, np.float32); qr = np.ascontiguousarray(qr, np.float32)
        qmin = np.ascontiguousarray(qmin, np.int32); qmax = np.ascontiguousarray(qmax, np.int32)
        nq = len(qx)
        cap = cap or max(1, 64 * nq + 1024)
        off = np.zeros(nq + 1, np.int32); idx = np.zeros(cap, np.int32); tot = C.c_int(0)
        _chk(lib().cms_features_in_area(self.h, b, nq, _p(qx), _p(qy), _p(qr), _p(qmin), _p(qmax), _p(off), _p(idx), cap, C.byref(tot)),
             "cms_features_in_area")
        return off, idx[:tot.value]

    def features_in_area_device(self, b, nq, d_q5, d_cnt, d_off, d_idx, cap, idx_base, d_total):
        """device-pointer variant: d_q5 = (qx, qy, qr, qmin, qmax) device addresses"""
        _chk(lib().cms_features_in_area_device(self.h, b, nq, *[C.c_void_p(int(a)) for a in d_q5], C.c_void_p(int(d_cnt)), C.c_void_p(int(d_off)),
                                               C.c_void_p(int(d_idx)), cap, idx_base, C.c_void_p(int(d_total))), "cms_features_in_area_device")

    def features_in_area_batch_device(self, nq, d_qframe, d_q5, d_cnt, d_off, d_idx, cap, d_total):
        """all frames of the batch in one launch sequence; d_qframe[q] = frame searched by query q; indices are batch rows"""
        _chk(lib().cms_features_in_area_batch_device(self.h, nq, C.c_void_p(int(d_qframe)), *[C.c_void_p(int(a)) for a in d_q5], C.c_void_p(int(d_cnt)),
                                                     C.c_void_p(int(d_off)), C.c_void_p(int(d_idx)), cap, C.c_void_p(int(d_total))),
             "cms_features_in_area_batch_device")

    def area_set_descriptors(self, b, desc):
        desc = np.ascontiguousarray(desc, np.uint8)
        _chk(lib().cms_area_set_descriptors(self.h, b, len(desc), _p(desc)), "cms_area_set_descriptors")

    def search_local_points(self, b, pose15, pos, normal, min_dist, max_dist, mp_desc, kp_mp, viewing_cos_limit=0.5, th=1.0, nnratio=0.8,
                            th_high=100):
        """Frame::isInFrustum + ORBMatcher::SearchByProjection(F, vpMapPoints, th) for frame b (area_grid first).  kp_mp: int32 per key
        point (>= 0 = taken), updated in place.  Returns dict(in_view, proj_x, proj_y, level, view_cos, match, n_matches, rounds)."""
        pose15 = np.ascontiguousarray(pose15, np.float32).reshape(15)
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3); normal = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
        n = len(pos)
        min_dist = np.ascontiguousarray(min_dist, np.float32); max_dist = np.ascontiguousarray(max_dist, np.float32)
        mp_desc = np.ascontiguousarray(mp_desc, np.uint8).reshape(n, 32)
        assert kp_mp.dtype == np.int32 and kp_mp.flags.c_contiguous
        vis = np.zeros(n, np.uint8); px = np.zeros(n, np.float32); py = np.zeros(n, np.float32); lvl = np.zeros(n, np.int32)
        vc = np.zeros(n, np.float32); match = np.full(n, -1, np.int32)
        nm = C.c_int(0); rounds = C.c_int(0)
        _chk(lib().cms_search_local_points(self.h, b, _p(pose15), n, _p(pos), _p(normal), _p(min_dist), _p(max_dist), _p(mp_desc),
                                           viewing_cos_limit, th, nnratio, th_high, len(kp_mp), _p(kp_mp), _p(vis), _p(px), _p(py), _p(lvl),
                                           _p(vc), _p(match), C.addressof(nm), C.addressof(rounds)), "cms_search_local_points")
        return dict(in_view=vis, proj_x=px, proj_y=py, level=lvl, view_cos=vc, match=match, n_matches=nm.value, rounds=rounds.value)

    def search_by_projection(self, b, pose12, valid, Xw, octave, angle, mp_desc, kp_mp, th=15.0, check_ori=True, th_high=100):
        """ORBMatcher::SearchByProjection(CurrentFrame, LastFrame, th, mono) against frame slot b -> (match, n_matches); kp_mp in/out"""
        a = [np.ascontiguousarray(pose12, np.float32), np.ascontiguousarray(valid, np.uint8), np.ascontiguousarray(Xw, np.float32),
             np.ascontiguousarray(octave, np.int32), np.ascontiguousarray(angle, np.float32), np.ascontiguousarray(mp_desc, np.uint8)]
        n = len(a[1])
        match = np.full(n, -1, np.int32); nm = C.c_int(0)
        assert kp_mp.dtype == np.int32
        _chk(lib().cms_search_by_projection(self.h, b, _p(a[0]), n, *[_p(v) for v in a[1:]], th, int(check_ori), th_high, len(kp_mp), _p(kp_mp), _p(match),
                                            C.addressof(nm)), "cms_search_by_projection")
        return match, nm.value

    def search_for_initialization(self, b2, k1, d1, prev_matched, window=100, nnratio=0.9, check_ori=True):
        """ORBMatcher::SearchForInitialization(F1, F2 = frame slot b2, vbPrevMatched, vnMatches12, windowSize) -> (matches12, nmatches); prev_matched
        (n1, 2) float32 is updated in place"""
        k1 = np.ascontiguousarray(k1, KP_DTYPE); d1 = np.ascontiguousarray(d1, np.uint8)
        assert prev_matched.dtype == np.float32 and prev_matched.flags.c_contiguous and prev_matched.shape == (len(k1), 2)
        m12 = np.full(max(len(k1), 1), -1, np.int32); nm = C.c_int(0)
        f = lib().cms_search_for_initialization
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        _chk(f(self.h, b2, len(k1), _p(k1), _p(d1), _p(prev_matched), int(window), float(nnratio), int(check_ori), _p(m12), C.addressof(nm)),
             "cms_search_for_initialization")
        return m12[:len(k1)], nm.value

    def project_last_frame_device(self, n, d_qframe, d_pose12, d_valid, d_Xw, d_oct, th, d_q5):
        """d_q5 = (qx, qy, qr, qmin, qmax) raw device pointers"""
        v = lambda a: C.c_void_p(int(a)) if a else None
        _chk(lib().cms_project_last_frame_device(self.h, n, v(d_qframe), v(d_pose12), v(d_valid), v(d_Xw), v(d_oct), th, *[v(a) for a in d_q5]),
             "cms_project_last_frame_device")

    def rotation_filter_device(self, B, d_mp_off, d_last_angle, d_kp_mp, d_mp_match, d_n_matches, check_orientation=True):
        v = lambda a: C.c_void_p(int(a)) if a else None
        _chk(lib().cms_rotation_filter_device(self.h, B, v(d_mp_off), v(d_last_angle), v(d_kp_mp), v(d_mp_match), v(d_n_matches), int(check_orientation)),
             "cms_rotation_filter_device")

    def is_in_frustum_device(self, nmp, d_mp_frame, d_pose15, d_pos, d_normal, d_min, d_max, viewing_cos_limit, th, d_outs5, d_q3):
        """d_outs5 = (in_view u8, proj_x, proj_y, level, view_cos); d_q3 = (qr, qmin, qmax) or (0, 0, 0); raw device pointers"""
        v = lambda a: C.c_void_p(int(a)) if a else None
        _chk(lib().cms_is_in_frustum_device(self.h, nmp, v(d_mp_frame), v(d_pose15), v(d_pos), v(d_normal), v(d_min), v(d_max),
                                            viewing_cos_limit, th, *[v(a) for a in d_outs5], *[v(a) for a in d_q3]), "cms_is_in_frustum_device")

    def search_local_points_device(self, B, d_mp_off, d_mp_desc, d_cand_off, d_cand_idx, d_pair_dist, nnratio, th_high, d_kp_mp, d_mp_match,
                                   d_rounds=0):
        v = lambda a: C.c_void_p(int(a)) if a else None
        _chk(lib().cms_search_local_points_device(self.h, B, v(d_mp_off), v(d_mp_desc), v(d_cand_off), v(d_cand_idx), v(d_pair_dist), nnratio,
                                                  th_high, v(d_kp_mp), v(d_mp_match), v(d_rounds)), "cms_search_local_points_device")

    def hamming_best2_device(self, qdesc, q_row, nq, tdesc, cand_off, cand_idx, t_level, t_excl, outs):
        """all arguments are raw device pointers (ints); asynchronous on the ctx stream"""
        _chk(lib().cms_hamming_best2_device(self.h, _p(qdesc), _p(q_row), nq, _p(tdesc), _p(cand_off), _p(cand_idx), _p(t_level),
                                            _p(t_excl), *[_p(o) for o in outs]), "cms_hamming_best2_device")

    def hamming_matrix(self, a, b):
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
        out = np.zeros((len(a), len(b)), np.uint16)
        _chk(lib().cms_hamming_matrix(self.h, _p(a), len(a), _p(b), len(b), _p(out)), "cms_hamming_matrix")
        return out


class PinnedArray:
    """uint8 numpy array over pinned host memory from cms_host_alloc (freed with the object)"""

    def __init__(self, shape):
        n = int(np.prod(shape))
        self.p = C.c_void_p()
        _chk(lib().cms_host_alloc(C.byref(self.p), n), "cms_host_alloc")
        self.array = np.ctypeslib.as_array((C.c_uint8 * n).from_address(self.p.value)).reshape(shape)

    def close(self):
        if self.p:
            self.array = None
            lib().cms_host_free(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Keyframe(C.Structure):
    """cms_keyframe (include/cubemapslam_hip.h)"""
    _fields_ = [("n", C.c_int), ("kps", C.c_void_p), ("desc", C.c_void_p), ("rays", C.c_void_p), ("mp", C.c_void_p),
                ("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3),
                ("nnodes", C.c_int), ("node_id", C.c_void_p), ("node_off", C.c_void_p), ("node_feat", C.c_void_p), ("median_depth", C.c_float)]


def make_keyframe(kf):
    """kf: dict with x, y, octave, angle, desc, rays, mp, R, t, Ow, node_id, node_off, node_feat, median_depth (synth.keyframe_set entries
    plus 'rays').  Returns (Keyframe, keep-alive list)."""
    n = len(kf["x"])
    kps = np.zeros(n, KP_DTYPE); kps["x"] = kf["x"]; kps["y"] = kf["y"]; kps["octave"] = kf["octave"]; kps["angle"] = kf["angle"]
    keep = [kps, np.ascontiguousarray(kf["desc"], np.uint8), np.ascontiguousarray(kf["rays"], np.float32), np.ascontiguousarray(kf["mp"], np.int32),
            np.ascontiguousarray(kf["node_id"], np.int32), np.ascontiguousarray(kf["node_off"], np.int32), np.ascontiguousarray(kf["node_feat"], np.int32)]
    K = Keyframe()
    K.n = n; K.kps = keep[0].ctypes.data; K.desc = keep[1].ctypes.data; K.rays = keep[2].ctypes.data; K.mp = keep[3].ctypes.data
    K.Rcw[:] = [float(v) for v in np.asarray(kf["R"], np.float32).reshape(9)]
    K.tcw[:] = [float(v) for v in kf["t"]]; K.Ow[:] = [float(v) for v in kf["Ow"]]
    K.nnodes = len(keep[4]); K.node_id = keep[4].ctypes.data; K.node_off = keep[5].ctypes.data; K.node_feat = keep[6].ctypes.data
    K.median_depth = float(kf["median_depth"])
    return K, keep


def search_for_triangulation(ctx, K1, K2, E12=None, check_orientation=False):
    """cms_search_for_triangulation: ORBMatcher::SearchForTriangulation(pKF1, pKF2, E12, ...) alone.  K1, K2: Keyframe structs; E12 (9 floats,
    row major) or None = ComputeE12 of the two poses.  Returns (matches12 int32[n1], nmatches)."""
    m = np.full(max(K1.n, 1), -1, np.int32)
    n = C.c_int()
    e = None if E12 is None else np.ascontiguousarray(E12, np.float32)
    lib().cms_search_for_triangulation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    _chk(lib().cms_search_for_triangulation(ctx.h, C.byref(K1), C.byref(K2), _p(e) if e is not None else None, int(check_orientation), _p(m), C.byref(n)),
         "cms_search_for_triangulation")
    return m[:K1.n].copy(), n.value


def create_new_map_points(ctx, jobs, check_orientation=False, cap=None):
    """jobs: list of (current Keyframe, [neighbour Keyframes in covisibility order]).  Returns per job (neigh, idx1, idx2, x3d)."""
    nj = len(jobs)
    cur = (Keyframe * max(nj, 1))(*[j[0] for j in jobs])
    flat = [k for j in jobs for k in j[1]]
    neigh = (Keyframe * max(len(flat), 1))(*flat)
    off = np.concatenate([[0], np.cumsum([len(j[1]) for j in jobs])]).astype(np.int32)
    cap = cap if cap is not None else max([j[0].n for j in jobs] + [1])
    n_new = np.zeros(max(nj, 1), np.int32)
    on = np.zeros((max(nj, 1), cap), np.int32); o1 = np.zeros_like(on); o2 = np.zeros_like(on); ox = np.zeros((max(nj, 1), cap, 3), np.float32)
    _chk(lib().cms_create_new_map_points(ctx.h, nj, C.addressof(cur), _p(off), C.addressof(neigh), int(check_orientation), cap, _p(n_new), _p(on), _p(o1),
                                         _p(o2), _p(ox)), "cms_create_new_map_points")
    return [(on[j, :n_new[j]].copy(), o1[j, :n_new[j]].copy(), o2[j, :n_new[j]].copy(), ox[j, :n_new[j]].copy()) for j in range(nj)]


class KeyframeStore:
    """cms_kfstore: key frames resident on the device in slots"""

    def __init__(self, ctx, max_keyframes, max_features=2048, max_nodes=2048):
        self.ctx = ctx
        self.h = C.c_void_p()
        _chk(lib().cms_kfstore_create(C.byref(self.h), ctx.h, max_keyframes, max_features, max_nodes), "cms_kfstore_create")

    def close(self):
        if self.h:
            lib().cms_kfstore_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def put(self, slot, K):
        _chk(lib().cms_kfstore_put(self.h, slot, C.byref(K)), "cms_kfstore_put")

    def update(self, slot, R=None, t=None, Ow=None, median_depth=None, mp=None):
        f = lambda a, dt: None if a is None else np.ascontiguousarray(a, dt)
        a = [f(R, np.float32), f(t, np.float32), f(Ow, np.float32), None if median_depth is None else np.array([median_depth], np.float32), f(mp, np.int32)]
        _chk(lib().cms_kfstore_update(self.h, slot, *[_p(v) for v in a]), "cms_kfstore_update")

    def put_from_frame(self, slot, src_ctx, b, n, kf):
        """cms_kfstore_put_from_frame: frame b of src_ctx's last batch (key points, descriptors, rays, grid: device to device) becomes the key frame in
        `slot`; kf supplies what the host holds: R, t, Ow, median_depth, mp (or None), node_id / node_off / node_feat.  Asynchronous."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        i32 = lambda a: None if a is None else np.ascontiguousarray(a, np.int32)
        R, t, Ow = f32(np.asarray(kf["R"]).reshape(9)), f32(kf["t"]), f32(kf["Ow"])
        mp, nid, noff, nfeat = i32(kf.get("mp")), i32(kf["node_id"]), i32(kf["node_off"]), i32(kf["node_feat"])
        L = lib()
        _chk(L.cms_kfstore_put_from_frame(self.h, slot, src_ctx.h, b, n, _p(R), _p(t), _p(Ow), float(kf["median_depth"]), _p(mp), len(nid), _p(nid), _p(noff), _p(nfeat)),
             "cms_kfstore_put_from_frame")

    def put_from_frames(self, src_ctx, items):
        """cms_kfstore_put_from_frames: several key frames of src_ctx's last batch in ONE call (one kernel); items = [(slot, b, n, kf), ...] with kf as in
        put_from_frame.  All items are checked before anything is touched: on an error the store is unchanged."""
        class KfFromFrame(C.Structure):      # cms_kf_from_frame (include/cubemapslam_hip.h)
            _fields_ = [("slot", C.c_int), ("b", C.c_int), ("n", C.c_int), ("Rcw", C.c_void_p), ("tcw", C.c_void_p), ("Ow", C.c_void_p), ("median_depth", C.c_float),
                        ("mp", C.c_void_p), ("nnodes", C.c_int), ("node_id", C.c_void_p), ("node_off", C.c_void_p), ("node_feat", C.c_void_p)]
        arr = (KfFromFrame * len(items))()
        keep = []
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        i32 = lambda a: None if a is None else np.ascontiguousarray(a, np.int32)
        ptr = lambda a: None if a is None else a.ctypes.data
        for q, (slot, b, n, kf) in zip(arr, items):
            R, t, Ow = f32(np.asarray(kf["R"]).reshape(9)), f32(kf["t"]), f32(kf["Ow"])
            mp, nid, noff, nfeat = i32(kf.get("mp")), i32(kf["node_id"]), i32(kf["node_off"]), i32(kf["node_feat"])
            keep.append((R, t, Ow, mp, nid, noff, nfeat))
            q.slot = slot; q.b = b; q.n = n; q.Rcw = ptr(R); q.tcw = ptr(t); q.Ow = ptr(Ow); q.median_depth = float(kf["median_depth"]); q.mp = ptr(mp)
            q.nnodes = len(nid); q.node_id = ptr(nid); q.node_off = ptr(noff); q.node_feat = ptr(nfeat)
        L = lib()
        L.cms_kfstore_put_from_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _chk(L.cms_kfstore_put_from_frames(self.h, src_ctx.h, len(items), arr), "cms_kfstore_put_from_frames")

    def debug_fetch(self, slot, max_features=16384, max_nodes=16384):
        """cms_kfstore_debug_fetch: the slot's device contents as a dict"""
        hdr = np.zeros(21, np.uint32); misc = np.zeros(2, np.int32)
        L = lib()
        _chk(L.cms_kfstore_debug_fetch(self.h, slot, *([None] * 10), _p(hdr), _p(misc)), "cms_kfstore_debug_fetch")
        n, nn = int(hdr[1]), int(hdr[3])
        o = dict(kps=np.zeros(n, KP_DTYPE), desc=np.zeros((n, 32), np.uint8), rays=np.zeros((n, 3), np.float32), mp=np.zeros(n, np.int32), feat_node=np.zeros(n, np.int32),
                 sorted=np.zeros(n, np.uint16), node_id=np.zeros(nn, np.int32), node_off=np.zeros(nn + 1, np.int32), node_feat=np.zeros(max(n, 1), np.int32),
                 cell_start=np.zeros(5 * 50 * 50 + 1, np.int32))
        _chk(L.cms_kfstore_debug_fetch(self.h, slot, _p(o["kps"]), _p(o["desc"]), _p(o["rays"]), _p(o["mp"]), _p(o["feat_node"]), _p(o["sorted"]), _p(o["node_id"]),
                                       _p(o["node_off"]), _p(o["node_feat"]), _p(o["cell_start"]), _p(hdr), _p(misc)), "cms_kfstore_debug_fetch")
        o["node_feat"] = o["node_feat"][:int(o["node_off"][-1]) if nn else 0]
        o["header"] = hdr; o["nvalid"] = int(misc[0]); o["kp_cnt"] = int(misc[1])
        return o

    def update_poses(self, slots, R, t, Ow):
        """cms_kfstore_update_poses: poses of resident key frames after a local BA, one asynchronous call"""
        slots = np.ascontiguousarray(slots, np.int32)
        a = [np.ascontiguousarray(R, np.float32).reshape(-1), np.ascontiguousarray(t, np.float32).reshape(-1), np.ascontiguousarray(Ow, np.float32).reshape(-1)]
        assert len(a[0]) == 9 * len(slots) and len(a[1]) == 3 * len(slots) and len(a[2]) == 3 * len(slots)
        _chk(lib().cms_kfstore_update_poses(self.h, len(slots), _p(slots), _p(a[0]), _p(a[1]), _p(a[2])), "cms_kfstore_update_poses")

    def fuse_search(self, jobs, th=3.0):
        """jobs: list of (slot, dict(skip, pos, normal, min_dist, max_dist, desc)) -> per job (best_idx, best_dist)"""
        nj = len(jobs)
        slots = np.array([j[0] for j in jobs], np.int32)
        off = np.concatenate([[0], np.cumsum([len(j[1]["pos"]) for j in jobs])]).astype(np.int32)
        cat = lambda k, dt: np.ascontiguousarray(np.concatenate([np.asarray(j[1][k]) for j in jobs]), dt)
        a = [cat("skip", np.uint8), cat("pos", np.float32), cat("normal", np.float32), cat("min_dist", np.float32), cat("max_dist", np.float32), cat("desc", np.uint8)]
        n = int(off[-1])
        bi = np.zeros(n, np.int32); bd = np.zeros(n, np.int32)
        _chk(lib().cms_kfstore_fuse_search(self.h, nj, _p(slots), _p(off), *[_p(v) for v in a], th, _p(bi), _p(bd)), "cms_kfstore_fuse_search")
        return [(bi[off[j]:off[j + 1]].copy(), bd[off[j]:off[j + 1]].copy()) for j in range(nj)]

    def fuse_search_sets(self, sets, jobs, th=3.0):
        """cms_kfstore_fuse_search_sets.  sets: list of dict(pos, normal, min_dist, max_dist, desc); jobs: list of (slot, set index, skip array or None)
        -> per job (best_idx, best_dist)"""
        soff = np.concatenate([[0], np.cumsum([len(q["pos"]) for q in sets])]).astype(np.int32)
        cat = lambda k, dt: np.ascontiguousarray(np.concatenate([np.asarray(q[k]) for q in sets]), dt)
        a = [cat("pos", np.float32), cat("normal", np.float32), cat("min_dist", np.float32), cat("max_dist", np.float32), cat("desc", np.uint8)]
        slots = np.array([j[0] for j in jobs], np.int32); jset = np.array([j[1] for j in jobs], np.int32)
        sizes = [int(soff[j[1] + 1] - soff[j[1]]) for j in jobs]
        eoff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        skip = np.ascontiguousarray(np.concatenate([np.zeros(sz, np.uint8) if j[2] is None else np.asarray(j[2], np.uint8) for j, sz in zip(jobs, sizes)]), np.uint8)
        bi = np.zeros(int(eoff[-1]), np.int32); bd = np.zeros(int(eoff[-1]), np.int32)
        _chk(lib().cms_kfstore_fuse_search_sets(self.h, len(sets), _p(soff), *[_p(v) for v in a], len(jobs), _p(slots), _p(jset), _p(skip), th, _p(bi), _p(bd)),
             "cms_kfstore_fuse_search_sets")
        return [(bi[eoff[j]:eoff[j + 1]].copy(), bd[eoff[j]:eoff[j + 1]].copy()) for j in range(len(jobs))]

    def create_new_map_points(self, jobs, check_orientation=False, cap=2048, copy=True):
        """jobs: list of (current slot, [neighbour slots in covisibility order]) -> per job (neigh, idx1, idx2, x3d).
        The job arrays and the output buffers are kept between calls with the same job list (a mapping thread calls this once per key frame,
        next to threads that hold the interpreter lock: fresh 0.8 MB of zeroed arrays per call were ~2 ms of a 3 ms call whose library
        part takes 0.3 ms).  copy=False returns views into those buffers, valid until the next call."""
        nj = len(jobs)
        key = (cap, tuple((int(j[0]), tuple(int(s) for s in j[1])) for j in jobs))      # (the content, not the list's identity: a caller may edit its list in place)
        st = getattr(self, "_cnmp", None)
        if st is None or st[0] != key:
            cur = np.array([j[0] for j in jobs], np.int32)
            neigh = np.array([s for j in jobs for s in j[1]] + [0], np.int32)
            off = np.concatenate([[0], np.cumsum([len(j[1]) for j in jobs])]).astype(np.int32)
            n_new = np.zeros(max(nj, 1), np.int32)
            on = np.zeros((max(nj, 1), cap), np.int32); o1 = np.zeros_like(on); o2 = np.zeros_like(on); ox = np.zeros((max(nj, 1), cap, 3), np.float32)
            arrs = (cur, off, neigh, n_new, on, o1, o2, ox)
            st = self._cnmp = (key, arrs, [_p(a) for a in arrs])
        cur, off, neigh, n_new, on, o1, o2, ox = st[1]
        pc, po, pn, pnn, pon, po1, po2, pox = st[2]
        t0 = time.perf_counter()
        _chk(lib().cms_kfstore_create_new_map_points(self.h, nj, pc, po, pn, int(check_orientation), cap, pnn, pon, po1, po2, pox), "cms_kfstore_create_new_map_points")
        self.last_call_ms = 1e3 * (time.perf_counter() - t0)      # the library call alone (the wrapper's array handling is Python's)
        if not copy:
            return [(on[j, :n_new[j]], o1[j, :n_new[j]], o2[j, :n_new[j]], ox[j, :n_new[j]]) for j in range(nj)]
        return [(on[j, :n_new[j]].copy(), o1[j, :n_new[j]].copy(), o2[j, :n_new[j]].copy(), ox[j, :n_new[j]].copy()) for j in range(nj)]


def distinctive_descriptors(ctx, obs_off, desc):
    obs_off = np.ascontiguousarray(obs_off, np.int32); desc = np.ascontiguousarray(desc, np.uint8)
    out = np.zeros(len(obs_off) - 1, np.int32)
    _chk(lib().cms_distinctive_descriptors(ctx.h, len(out), _p(obs_off), _p(desc), _p(out)), "cms_distinctive_descriptors")
    return out


def update_normal_and_depth(ctx, obs_off, pos, obs_Ow, ref_Ow, ref_level, normal, min_dist, max_dist):
    """normal / min_dist / max_dist: float32 arrays updated in place"""
    a = [np.ascontiguousarray(obs_off, np.int32), np.ascontiguousarray(pos, np.float32), np.ascontiguousarray(obs_Ow, np.float32),
         np.ascontiguousarray(ref_Ow, np.float32), np.ascontiguousarray(ref_level, np.int32)]
    _chk(lib().cms_update_normal_and_depth(ctx.h, len(a[0]) - 1, *[_p(v) for v in a], _p(normal), _p(min_dist), _p(max_dist)), "cms_update_normal_and_depth")


def fuse_search(ctx, b, pose15, skip, pos, normal, min_dist, max_dist, desc, th):
    """search half of ORBMatcher::Fuse against key-frame slot b -> (best_idx, best_dist)"""
    n = len(pos)
    a = [np.ascontiguousarray(pose15, np.float32), np.ascontiguousarray(skip, np.uint8), np.ascontiguousarray(pos, np.float32),
         np.ascontiguousarray(normal, np.float32), np.ascontiguousarray(min_dist, np.float32), np.ascontiguousarray(max_dist, np.float32),
         np.ascontiguousarray(desc, np.uint8)]
    bi = np.zeros(n, np.int32); bd = np.zeros(n, np.int32)
    _chk(lib().cms_fuse_search(ctx.h, b, _p(a[0]), n, *[_p(v) for v in a[1:]], th, _p(bi), _p(bd)), "cms_fuse_search")
    return bi, bd


class BundleAdjuster:
    """cms_ba: one local-BA window resident on the device."""

    def __init__(self, prob, device=0):
        self.prob = prob
        self.K, self.P, self.E = len(prob["poses"]), len(prob["points"]), len(prob["e_pose"])
        self.h = C.c_void_p()
        poses = np.ascontiguousarray(prob["poses"], np.float64); pts = np.ascontiguousarray(prob["points"], np.float64)
        _chk(lib().cms_ba_create(C.byref(self.h), device, self.K, _p(poses), _p(prob["fixed"]), self.P, _p(pts), self.E,
                                 _p(prob["e_pose"]), _p(prob["e_point"]), _p(np.ascontiguousarray(prob["e_obs"], np.float64)),
                                 _p(prob["e_invsig2"]), _p(prob["e_face"]), prob["fx"], prob["fy"], prob["cx"], prob["cy"]),
             "cms_ba_create")

    @classmethod
    def from_handle(cls, prob, handle):
        self = cls.__new__(cls)
        self.prob = prob
        self.K, self.P, self.E = len(prob["poses"]), len(prob["points"]), len(prob["e_pose"])
        self.h = C.c_void_p(handle)
        return self

    def reset(self):
        _chk(lib().cms_ba_reset(self.h), "cms_ba_reset")

    def optimize(self, its=(5, 10), stop=None):
        st = BaStats()
        stop_arr = np.array([1 if stop else 0], np.uint8)
        rc = _chk(lib().cms_ba_optimize(self.h, its[0], its[1], _p(stop_arr), C.byref(st)), "cms_ba_optimize")
        return rc, st

    def read(self):
        poses = np.zeros((self.K, 7)); pts = np.zeros((self.P, 3)); flags = np.zeros(self.E, np.uint8)
        _chk(lib().cms_ba_read(self.h, _p(poses), _p(pts), _p(flags)), "cms_ba_read")
        return poses, pts, flags

    def fetch_plan(self):
        """cms_ba_debug_fetch_plan: the plan arrays as the device holds them (whichever planner made the window)"""
        P, E = self.P, self.E
        out = dict(pinv=np.zeros(P, np.int32), perm=np.zeros(E, np.int32), info=np.zeros(E, np.uint32), pt_off=np.zeros(P + 1, np.int32),
                   e_pose=np.zeros(E, np.int32), e_point=np.zeros(E, np.int32), e_face=np.zeros(E, np.int8), chunk_e0=np.zeros(P + 2, np.int32),
                   rm_chunk=np.zeros((P, 4), np.int32), rm_cost=np.zeros(P + 2, np.uint32), run_mf=np.zeros((P, 64), np.uint32), run_fl=np.zeros((P, 64, 12), np.uint32))
        cnt = np.zeros(8, np.int32)
        L = lib()
        _chk(L.cms_ba_debug_fetch_plan(self.h, _p(out["pinv"]), _p(out["perm"]), _p(out["info"]), _p(out["pt_off"]), _p(out["e_pose"]), _p(out["e_point"]),
                                       _p(out["e_face"]), _p(out["chunk_e0"]), _p(out["rm_chunk"]), _p(out["rm_cost"]), _p(out["run_mf"]), _p(out["run_fl"]), _p(cnt)),
             "cms_ba_debug_fetch_plan")
        nch, n_rm, nr = int(cnt[0]), int(cnt[1]), int(cnt[2])
        out.update(n_chunks=nch, n_rm=n_rm, n_runs=nr, np=int(cnt[3]), rm_points=int(cnt[4]), R_rm=int(cnt[5]), R=int(cnt[6]), device_planned=bool(cnt[7]), plan_kernel=int(cnt[7]) == 2)
        out["chunk_e0"] = out["chunk_e0"][:nch + 1].copy(); out["rm_chunk"] = out["rm_chunk"][:n_rm].copy(); out["rm_cost"] = out["rm_cost"][:nch + 1].copy()
        out["run_mf"] = out["run_mf"][:nr].copy(); out["run_fl"] = out["run_fl"][:nr].copy()
        return out

    @property
    def stream(self):
        return lib().cms_ba_stream(self.h)

    def set_stream(self, hip_stream):
        f = lib().cms_ba_set_stream
        f.argtypes = [C.c_void_p, C.c_void_p]
        _chk(f(self.h, C.c_void_p(int(hip_stream))), "cms_ba_set_stream")

    def profile_kernel(self, kernel_id):
        """HIP events around one kernel of the grouped driver's rounds (group owned by this handle); 3 = kb_ba_schur_points"""
        _chk(lib().cms_ba_profile_kernel(self.h, kernel_id), "cms_ba_profile_kernel")

    def profile_get(self):
        ms = C.c_double(0); n = C.c_long(0)
        _chk(lib().cms_ba_profile_get(self.h, C.byref(ms), C.byref(n)), "cms_ba_profile_get")
        return ms.value, n.value

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            lib().cms_ba_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ba_compose_chunks(fixed, n_points, e_pose, e_point, lookahead=48):
    """Host-only: the chunk composition of the edge-major Schur kernel (cms_ba_debug_compose).  Returns (pinv, chunk_pt0, rank)."""
    fixed = np.ascontiguousarray(fixed, np.uint8); e_pose = np.ascontiguousarray(e_pose, np.int32); e_point = np.ascontiguousarray(e_point, np.int32)
    P, E = int(n_points), len(e_pose)
    pinv = np.zeros(P, np.int32); pt0 = np.zeros(P + 1, np.int32); rank = np.zeros(E, np.uint8)
    n = C.c_int(0)
    _chk(lib().cms_ba_debug_compose(len(fixed), _p(fixed), P, E, _p(e_pose), _p(e_point), int(lookahead), _p(pinv), _p(pt0), C.byref(n), _p(rank)),
         "cms_ba_debug_compose")
    return pinv, pt0[:n.value + 1].copy(), rank


def ba_plan(fixed, n_points, e_pose, e_point, tables=False):
    """Host-only: the plan cms_ba_create makes (cms_ba_debug_plan): internal order, chunks, signature runs and the run-major kernel's tables."""
    fixed = np.ascontiguousarray(fixed, np.uint8); e_pose = np.ascontiguousarray(e_pose, np.int32); e_point = np.ascontiguousarray(e_point, np.int32)
    P, E = int(n_points), len(e_pose)
    pinv = np.zeros(P, np.int32); perm = np.zeros(E, np.int32); info = np.zeros(E, np.uint32); pt0 = np.zeros(P + 2, np.int32)
    rmc = np.zeros((P, 4), np.int32); rl = np.zeros((P, 64, 2), np.uint32); cnt = np.zeros(8, np.int32)
    mf = np.zeros((P, 64), np.uint32) if tables else None; fl = np.zeros((P, 64, 12), np.uint32) if tables else None
    _chk(lib().cms_ba_debug_plan(len(fixed), _p(fixed), P, E, _p(e_pose), _p(e_point), _p(pinv), _p(perm), _p(info), _p(pt0), _p(rmc), _p(rl), _p(cnt),
                                 _p(mf), _p(fl)), "cms_ba_debug_plan")
    nch, n_rm, nruns = int(cnt[0]), int(cnt[1]), int(cnt[2])
    return dict(pinv=pinv, perm=perm, info=info, chunk_pt0=pt0[:nch + 1].copy(), rm_chunk=rmc[:n_rm].copy(), run_lane=rl[:nruns].copy(), n_chunks=nch,
                run_mf=None if mf is None else mf[:nruns].copy(), run_fl=None if fl is None else fl[:nruns].copy(),
                n_rm=n_rm, n_runs=nruns, np=int(cnt[3]), rm_points=int(cnt[4]), R_rm=int(cnt[5]), R=int(cnt[6]), usable=bool(cnt[7]))


def ba_plan_fast(fixed, n_points, e_pose, e_point):
    """Host-only: the device-side planner's result for a window (cms_ba_debug_plan_fast); `usable` False when it does not take the window."""
    fixed = np.ascontiguousarray(fixed, np.uint8); e_pose = np.ascontiguousarray(e_pose, np.int32); e_point = np.ascontiguousarray(e_point, np.int32)
    P, E = int(n_points), len(e_pose)
    pinv = np.zeros(P, np.int32); perm = np.zeros(E, np.int32); info = np.zeros(E, np.uint32); pt0 = np.zeros(P + 2, np.int32)
    rmc = np.zeros((P, 4), np.int32); cnt = np.zeros(8, np.int32); mf = np.zeros((P, 64), np.uint32); fl = np.zeros((P, 64, 12), np.uint32)
    _chk(lib().cms_ba_debug_plan_fast(len(fixed), _p(fixed), P, E, _p(e_pose), _p(e_point), _p(pinv), _p(perm), _p(info), _p(pt0), _p(rmc), _p(cnt), _p(mf), _p(fl)),
         "cms_ba_debug_plan_fast")
    nch, n_rm, nruns = int(cnt[0]), int(cnt[1]), int(cnt[2])
    return dict(pinv=pinv, perm=perm, info=info, chunk_pt0=pt0[:nch + 1].copy(), rm_chunk=rmc[:n_rm].copy(), n_chunks=nch, run_mf=mf[:nruns].copy(), run_fl=fl[:nruns].copy(),
                n_rm=n_rm, n_runs=nruns, np=int(cnt[3]), rm_points=int(cnt[4]), R_rm=int(cnt[5]), R=int(cnt[6]), usable=bool(cnt[7]))


def ba_run_fg(fixed, n_points, e_pose, e_point, fast=False):
    """Host-only: the tables of the opt-in one-wavefront run workgroups (cms_ba_debug_run_fg); None when the planner does not take the window."""
    fixed = np.ascontiguousarray(fixed, np.uint8); e_pose = np.ascontiguousarray(e_pose, np.int32); e_point = np.ascontiguousarray(e_point, np.int32)
    P, E = int(n_points), len(e_pose)
    fg = np.zeros((P, 64, 24), np.uint32); cut = np.zeros((2, 1025), np.int32); cnt = np.zeros(4, np.int32)
    _chk(lib().cms_ba_debug_run_fg(len(fixed), _p(fixed), P, E, _p(e_pose), _p(e_point), 1 if fast else 0, _p(fg), _p(cut), _p(cnt)), "cms_ba_debug_run_fg")
    if cnt[0] < 0:
        return None
    return dict(run_fg=fg[:int(cnt[0])].copy(), rm_cut=cut, n_runs=int(cnt[0]), n_rm=int(cnt[1]), n_rmA=int(cnt[2]), np=int(cnt[3]))


class BaWindow(C.Structure):      # cms_ba_window (include/cubemapslam_hip.h)
    _fields_ = [("K", C.c_int), ("poses", C.c_void_p), ("fixed", C.c_void_p), ("P", C.c_int), ("points", C.c_void_p), ("E", C.c_int), ("e_pose", C.c_void_p),
                ("e_point", C.c_void_p), ("e_obs", C.c_void_p), ("e_invsig2", C.c_void_p), ("e_face", C.c_void_p), ("fx", C.c_double), ("fy", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("flags", C.c_int)]


def pin_problem(prob):
    """a copy of a BA problem whose observation-sized arrays lie in pinned host memory (cms_host_alloc), for cms_ba_window.flags = CMS_BA_INPUTS_PINNED;
    the PinnedArray objects travel with the dict and keep the memory alive"""
    q = dict(prob); keep = []
    for key, dt in (("points", np.float64), ("e_pose", np.int32), ("e_point", np.int32), ("e_obs", np.float64), ("e_invsig2", np.float64), ("e_face", np.int8)):
        a = np.ascontiguousarray(prob[key], dt)
        pa = PinnedArray((max(a.nbytes, 16),))
        v = pa.array[:a.nbytes].view(dt).reshape(a.shape); v[...] = a
        q[key] = v; keep.append(pa)
    q["_pinned"] = keep
    return q


def ba_window_array(probs):
    """(array of cms_ba_window, the numpy arrays it points into) for cms_ba_create_many; the problems' arrays must stay alive while the call runs"""
    arr = (BaWindow * len(probs))()
    keep = []
    for q, p in zip(arr, probs):
        a = [np.ascontiguousarray(p["poses"], np.float64), np.ascontiguousarray(p["fixed"], np.uint8), np.ascontiguousarray(p["points"], np.float64),
             np.ascontiguousarray(p["e_pose"], np.int32), np.ascontiguousarray(p["e_point"], np.int32), np.ascontiguousarray(p["e_obs"], np.float64),
             np.ascontiguousarray(p["e_invsig2"], np.float64), np.ascontiguousarray(p["e_face"], np.int8)]
        keep.append(a)
        q.K = len(a[0]); q.poses = a[0].ctypes.data; q.fixed = a[1].ctypes.data; q.P = len(a[2]); q.points = a[2].ctypes.data; q.E = len(a[3])
        q.e_pose = a[3].ctypes.data; q.e_point = a[4].ctypes.data; q.e_obs = a[5].ctypes.data; q.e_invsig2 = a[6].ctypes.data; q.e_face = a[7].ctypes.data
        q.fx = p["fx"]; q.fy = p["fy"]; q.cx = p["cx"]; q.cy = p["cy"]
        q.flags = (1 if p.get("_pinned") else 0) | (2 if p.get("_plan_on_device") else 0)      # CMS_BA_INPUTS_PINNED: pin_problem() made the arrays; CMS_BA_PLAN_ON_DEVICE
    return arr, keep


def ba_create_many(probs, device=0, threads=4, windows=None):
    """cms_ba_create_many: the windows of a group in one call (host parts on `threads` threads, one set-up launch per eight device-planned windows).
    windows: a prepared ba_window_array(probs) (bench.py builds it once per problem set)."""
    n = len(probs)
    arr, keep = windows if windows is not None else ba_window_array(probs)
    handles = (C.c_void_p * n)()
    L = lib()
    L.cms_ba_create_many.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    _chk(L.cms_ba_create_many(handles, n, device, arr, threads), "cms_ba_create_many")
    return [BundleAdjuster.from_handle(p, handles[i]) for i, p in enumerate(probs)]


def ba_read_many(bas):
    """cms_ba_read_many: [(poses, points, outlier flags)] of optimised windows, one gather launch per sixteen device-planned windows"""
    n = len(bas)
    outs = [(np.zeros((b.K, 7)), np.zeros((b.P, 3)), np.zeros(b.E, np.uint8)) for b in bas]
    handles = (C.c_void_p * n)(*[b.h for b in bas])
    pp = (C.c_void_p * n)(*[o[0].ctypes.data for o in outs]); pq = (C.c_void_p * n)(*[o[1].ctypes.data for o in outs]); pf = (C.c_void_p * n)(*[o[2].ctypes.data for o in outs])
    L = lib()
    L.cms_ba_read_many.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    _chk(L.cms_ba_read_many(handles, n, pp, pq, pf), "cms_ba_read_many")
    return outs


def ba_set_deterministic(on):
    """cms_ba_set_deterministic: windows created afterwards run the fixed-order (bit-repeatable) kernels, like the reference's single-threaded g2o."""
    _chk(lib().cms_ba_set_deterministic(int(on) if not isinstance(on, bool) else (1 if on else 0)), "cms_ba_set_deterministic")      # (an integer >= 2: workgroups per window)


def ba_get_deterministic():
    return bool(lib().cms_ba_get_deterministic())


def ba_optimize_many(bas, its=(5, 10), stop=None, stop_array=None):
    """cms_ba_optimize_many: advance several BundleAdjuster windows in lock-step from one host thread.  stop_array: a uint8[1] array
    another thread may set while the call runs (Optimizer::LocalBundleAdjustment's pbStopFlag); None = no flag."""
    n = len(bas)
    handles = (C.c_void_p * n)(*[b.h for b in bas])
    stats = (BaStats * n)()
    stop_arr = stop_array if stop_array is not None else (np.array([1], np.uint8) if stop else None)
    lib().cms_ba_optimize_many.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rc = _chk(lib().cms_ba_optimize_many(handles, n, its[0], its[1], _p(stop_arr), stats), "cms_ba_optimize_many")
    return rc, list(stats)


def ba_run(prob, its=(5, 10), stop=None, device=0):
    poses = np.array(prob["poses"], np.float64, copy=True); pts = np.array(prob["points"], np.float64, copy=True)
    E = len(prob["e_pose"])
    flags = np.zeros(E, np.uint8)
    st = BaStats()
    stop_arr = np.array([1 if stop else 0], np.uint8)
    rc = _chk(lib().cms_ba_run(device, len(poses), _p(poses), _p(prob["fixed"]), len(pts), _p(pts), E, _p(prob["e_pose"]),
                               _p(prob["e_point"]), _p(np.ascontiguousarray(prob["e_obs"], np.float64)), _p(prob["e_invsig2"]),
                               _p(prob["e_face"]), prob["fx"], prob["fy"], prob["cx"], prob["cy"], its[0], its[1], _p(stop_arr),
                               _p(flags), C.byref(st)), "cms_ba_run")
    return dict(rc=rc, poses=poses, points=pts, outliers=flags, stats=st)


def ba_linearize(prob, robust=True, delta=float(np.sqrt(5.991)), device=0):
    K, P, E = len(prob["poses"]), len(prob["points"]), len(prob["e_pose"])
    poses = np.ascontiguousarray(prob["poses"], np.float64); pts = np.ascontiguousarray(prob["points"], np.float64)
    o = dict(err=np.zeros((E, 2)), Hpp=np.zeros((K, 6, 6)), bp=np.zeros((K, 6)), Hll=np.zeros((P, 3, 3)), bl=np.zeros((P, 3)),
             Hpl=np.zeros((E, 6, 3)), chi=np.zeros(1))
    _chk(lib().cms_ba_linearize(device, K, _p(poses), _p(prob["fixed"]), P, _p(pts), E, _p(prob["e_pose"]), _p(prob["e_point"]),
                                _p(np.ascontiguousarray(prob["e_obs"], np.float64)), _p(prob["e_invsig2"]), _p(prob["e_face"]),
                                prob["fx"], prob["fy"], prob["cx"], prob["cy"], 1 if robust else 0, float(delta), _p(o["err"]),
                                _p(o["Hpp"]), _p(o["bp"]), _p(o["Hll"]), _p(o["bl"]), _p(o["Hpl"]), _p(o["chi"])), "cms_ba_linearize")
    return o


class PoseStats(C.Structure):
    _fields_ = [("rounds", C.c_int), ("n_bad", C.c_int), ("iterations_done", C.c_int * 4)]


class PoseOptimizer:
    """Optimizer::PoseOptimization for a batch of frames (cms_pose_*): problems are dicts like synth.pose_problem()."""

    def __init__(self, max_frames, max_edges, device=0):
        self.h = C.c_void_p()
        _chk(lib().cms_pose_create(C.byref(self.h), device, max_frames, max_edges), "cms_pose_create")
        self.nf = 0

    def close(self):
        if self.h:
            lib().cms_pose_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream(self):
        return lib().cms_pose_stream(self.h)

    def upload(self, probs):
        self.nf = len(probs)
        cnt = [len(p["Xw"]) for p in probs]
        self.off = np.zeros(self.nf + 1, np.int32); self.off[1:] = np.cumsum(cnt)
        cat = lambda k, dt, w: (np.ascontiguousarray(np.concatenate([np.asarray(p[k], dt).reshape(-1, w) for p in probs]), dt)
                                if self.off[-1] else np.zeros((0, w), dt))
        Xw, obs, inv, face = cat("Xw", np.float64, 3), cat("obs", np.float64, 2), cat("invsig2", np.float64, 1), cat("face", np.int8, 1)
        poses = np.ascontiguousarray(np.stack([p["pose0"] for p in probs]), np.float64)
        p0 = probs[0]
        _chk(lib().cms_pose_upload(self.h, self.nf, _p(self.off), _p(Xw), _p(obs), _p(inv), _p(face), p0["fx"], p0["fy"], p0["cx"],
                                   p0["cy"], _p(poses)), "cms_pose_upload")

    def launch(self):
        _chk(lib().cms_pose_launch(self.h), "cms_pose_launch")

    def fetch(self):
        poses = np.zeros((self.nf, 7)); out = np.zeros(max(int(self.off[-1]), 1), np.uint8)
        ninl = np.zeros(self.nf, np.int32); st = (PoseStats * self.nf)()
        _chk(lib().cms_pose_fetch(self.h, _p(poses), _p(out), _p(ninl), C.byref(st)), "cms_pose_fetch")
        outs = [out[self.off[f]:self.off[f + 1]].copy() for f in range(self.nf)]
        return ninl, poses, outs, list(st)

    def optimize(self, probs):
        self.upload(probs); self.launch()
        return self.fetch()

    def optimize_batch(self, probs):
        """cms_pose_optimize_batch on this handle (a few frames: the direct path through the handle's pinned block); the resident batch of
        upload / launch is left alone.  Returns (n_inliers, poses, outlier flags per frame, stats)."""
        nf = len(probs)
        cnt = [len(q["Xw"]) for q in probs]
        off = np.zeros(nf + 1, np.int32); off[1:] = np.cumsum(cnt)
        cat = lambda k, dt, w: (np.ascontiguousarray(np.concatenate([np.asarray(q[k], dt).reshape(-1, w) for q in probs]), dt)
                                if off[-1] else np.zeros((0, w), dt))
        Xw, obs, inv, face = cat("Xw", np.float64, 3), cat("obs", np.float64, 2), cat("invsig2", np.float64, 1), cat("face", np.int8, 1)
        poses = np.ascontiguousarray(np.stack([q["pose0"] for q in probs]), np.float64)
        out = np.zeros(max(int(off[-1]), 1), np.uint8); ninl = np.zeros(nf, np.int32); st = (PoseStats * nf)()
        p0 = probs[0]
        _chk(lib().cms_pose_optimize_batch(self.h, nf, _p(off), _p(Xw), _p(obs), _p(inv), _p(face), C.c_double(p0["fx"]), C.c_double(p0["fy"]),
                                           C.c_double(p0["cx"]), C.c_double(p0["cy"]), _p(poses), _p(out), _p(ninl), C.byref(st)), "cms_pose_optimize_batch")
        return ninl, poses, [out[off[f]:off[f + 1]].copy() for f in range(nf)], list(st)


def pose_optimize(prob, n=None, device=0):
    """One frame through the one-shot entry point cms_pose_optimize; returns (n_inliers, pose7, outlier flags, stats)."""
    n = len(prob["Xw"]) if n is None else n
    pose = np.array(prob["pose0"], np.float64, copy=True)
    out = np.zeros(max(n, 1), np.uint8); ninl = np.zeros(1, np.int32); st = PoseStats()
    Xw = np.ascontiguousarray(prob["Xw"][:n], np.float64); obs = np.ascontiguousarray(prob["obs"][:n], np.float64)
    inv = np.ascontiguousarray(prob["invsig2"][:n], np.float64); face = np.ascontiguousarray(prob["face"][:n], np.int8)
    _chk(lib().cms_pose_optimize(device, n, _p(Xw), _p(obs), _p(inv), _p(face), prob["fx"], prob["fy"], prob["cx"], prob["cy"],
                                 _p(pose), _p(out), _p(ninl), C.byref(st)), "cms_pose_optimize")
    return int(ninl[0]), pose, out[:n], st
