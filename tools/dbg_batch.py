import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from cubemapslam_amd import api, synth
import bench
B = int(sys.argv[1]); mode = sys.argv[2]
camd = synth.camera("lafida", 550)
ctx = api.Context(camd, nfeatures=2000, max_batch=B)
print("ctx ok", flush=True)
ctx.set_mask(synth.cubemap_valid_mask(camd)); print("mask ok", flush=True)
fr = bench.make_frames(camd, B, 100)
if mode != "noupload":
    ctx.upload(fr); ctx.sync(); print("upload ok", flush=True)
ctx.process(B, mode == "remap"); ctx.sync(); print("process ok", B, mode, flush=True)
