"""Frame path only (remap + extract), B frames per launch (32 streams x 8 frames like bench.py), for rocprofv3 runs and A/B timing."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from cubemapslam_amd import api, synth
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
F = int(sys.argv[2]) if len(sys.argv) > 2 else 550
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
camd = synth.camera("lafida", F)
ctx = api.Context(camd, nfeatures=camd["nfeatures"], max_batch=B)
ctx.set_mask(synth.cubemap_valid_mask(camd))
fps = 8 if B % 8 == 0 else 1
frames = np.concatenate([bench.make_stream_frames(camd, fps, 100 + s) for s in range(B // fps)])
ctx.upload(frames)
ctx.profile(True)
acc = {}
for i in range(iters + 2):
    ctx.process(B, True); ctx.sync()
    if i >= 2:
        for k, v in ctx.profile_ms().items():
            acc[k] = acc.get(k, 0.0) + v / iters
g = ctx.geom
P = [g.level_w[l] * g.level_h[l] for l in range(g.nlevels)]
nkp = np.mean([len(ctx.fetch(b)[0]) for b in range(min(B, 16))])
b_extract = (2 * P[0] + sum(P[l - 1] + P[l] for l in range(1, g.nlevels))) + 3 * sum(P) + nkp * (709 + 961 + 32 + 28)
ext = sum(acc[k] for k in ("pyramid", "fast", "octree", "cull", "describe"))
print({k: round(v, 4) for k, v in acc.items()}, "extractor %.2f us/frame = %.3f of 8 TB/s (SURVEY 8d bytes)" % (1e3 * ext / B, b_extract * B / (ext * 1e-3) / 8e12))
