"""The CPU oracle behind the closed-loop harness' backend interface (cubemapslam_amd/harness.py): test infrastructure only."""
import numpy as np
import orc


class OracleBackend:
    name = "oracle"

    def __init__(self, camd, mask, gaussian_mode=0):
        self.camd = camd
        self.cam = orc.make_camera(camd)
        self.m1, self.m2 = orc.build_lut(self.cam)
        self.mask = mask
        nf = camd["nfeatures"]
        self.orb_ini = orc.Orb(nfeatures=3 * nf, gaussian_column_mode=gaussian_mode)
        self.orb_trk = orc.Orb(nfeatures=nf, gaussian_column_mode=gaussian_mode)
        self.cur = None
        # ORBextractor's tables (ORBExtractor.cpp:386-403): sf[i] = float(sf[i-1] * (double)1.2f), sigma2 = sf^2, inverse in float
        sf = [np.float32(1.0)]
        for _ in range(7):
            sf.append(np.float32(np.float64(sf[-1]) * np.float64(np.float32(1.2))))
        self.sf = np.array(sf, np.float32)
        self.inv_sigma2 = (np.float32(1.0) / (self.sf * self.sf)).astype(np.float32)

    def scale_factors(self):
        return self.sf, self.inv_sigma2

    def extract(self, fisheye, init):
        cube = orc.fisheye_to_cubemap(self.cam, self.m1, self.m2, np.ascontiguousarray(fisheye))
        k, d = (self.orb_ini if init else self.orb_trk).extract(self.cam, cube, self.mask)
        self.cur = (k, d)
        self.last_rays = orc.keyframe_rays(self.cam, k["x"], k["y"])      # Frame::ComputeKeyPointRays
        return k, d

    def search_for_initialization(self, k1, d1, k2, d2, prev_matched):
        return orc.search_for_initialization(self.cam, k1, d1, k2, d2, prev_matched, 100, 0.9, True)

    def search_by_projection(self, k, d, pose12, valid, Xw, octave, angle, mp_desc, kp_mp, th):
        return orc.search_by_projection_frames(self.cam, pose12[:9], pose12[9:], k["x"], k["y"], k["octave"], k["angle"], d, self.sf, valid, Xw, octave, angle,
                                               mp_desc, kp_mp, th=th, check_ori=True)

    def search_local_points(self, k, d, pose15, pos, normal, min_dist, max_dist, mp_desc, kp_mp, th):
        fr = orc.is_in_frustum(self.cam, pose15, pos, normal, min_dist, max_dist)
        match, nm = orc.search_local_points(self.cam, k["x"], k["y"], k["octave"], d, self.sf, fr, mp_desc, kp_mp, th=th)
        return match, nm, fr["in_view"]

    def pose_optimize(self, prob):
        n, pose, out, st = orc.pose_optimize(prob)
        return n, pose, out

    def local_ba(self, prob):
        r = orc.ba_run(prob)
        return r["poses"], r["points"], r["outliers"], list(r["stats"].iterations_done)

    def close(self):
        pass
