import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from cubemapslam_amd import api, synth
import bench
B, F = 256, 550
camd = synth.camera("lafida", F)
frames = np.concatenate([bench.make_stream_frames(camd, 8, 100 + s) for s in range(B // 8)])
for ini, mn in ((20, 7), (20, 20), (7, 7)):
    ctx = api.Context(camd, nfeatures=camd["nfeatures"], max_batch=B, ini_th=ini, min_th=mn)
    ctx.set_mask(synth.cubemap_valid_mask(camd))
    ctx.upload(frames); ctx.profile(True)
    acc = {}
    for i in range(8):
        ctx.process(B, True); ctx.sync()
        if i >= 3:
            for k, v in ctx.profile_ms().items(): acc[k] = acc.get(k, 0.0) + v / 5
    nkp = np.mean([len(ctx.fetch(b)[0]) for b in range(16)])
    print("ini %d min %d: fast %.3f ms  octree %.3f  describe %.3f  kp/frame %.0f" % (ini, mn, acc["fast"], acc["octree"], acc["describe"], nkp))
    ctx.close()
