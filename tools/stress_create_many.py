"""Developer aid: cms_ba_create_many / cms_ba_read_many from two host threads, over and over (heap checks: MALLOC_CHECK_=3)."""
import sys, os, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from cubemapslam_amd import api, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "many"
probs = [synth.ba_problem(K=20, P=22150, obs_per_point=4, F=550, seed=42 + i, views="track") for i in range(8)]
camd = synth.camera("lafida", 550)
ctxs = [api.Context(camd, nfeatures=500, max_batch=1) for _ in range(2)]
def loop(t):
    mine = probs[4 * t:4 * t + 4] * 4
    arr = api.ba_window_array(mine)
    for it in range(30):
        if mode == "many":
            grp = api.ba_create_many(mine, threads=4, windows=arr)
        else:
            grp = [api.BundleAdjuster(p) for p in mine]
        if mode != "nostream":
            for b in grp:
                b.set_stream(ctxs[t].stream)
        api.ba_optimize_many(grp, (2, 2))
        outs = api.ba_read_many(grp) if mode != "single" else [b.read() for b in grp]
        for b in grp:
            b.close()
    print("thread", t, "done", flush=True)
ths = [threading.Thread(target=loop, args=(t,)) for t in range(2)]
for t in ths: t.start()
for t in ths: t.join()
print("ok")
