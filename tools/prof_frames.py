"""Frame path only (remap + extract), B frames per launch, for rocprofv3 runs."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from cubemapslam_amd import api, synth
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
F = int(sys.argv[2]) if len(sys.argv) > 2 else 550
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
camd = synth.camera("lafida", F)
ctx = api.Context(camd, nfeatures=camd["nfeatures"], max_batch=B)
ctx.set_mask(synth.cubemap_valid_mask(camd))
ctx.upload(bench.make_frames(camd, B, 100))
ctx.profile(True)
for i in range(iters):
    ctx.process(B, True); ctx.sync()
print({k: round(v, 4) for k, v in ctx.profile_ms().items()})
