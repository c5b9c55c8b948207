#!/bin/bash
# developer helper (run on the GPU box through gpurun): bench.py with the given extra args, one line of key numbers
# usage: tools/gb.sh <tag> [bench args...]       env is inherited (A/B knobs)
tag=$1; shift
python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 --confined-steps 0 "$@" > gpurun_out/gb_$tag.log 2>&1
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
l = [x for x in open("gpurun_out/gb_%s.log" % tag) if x.startswith("{")]
if not l:
    print(tag, "FAILED"); print(open("gpurun_out/gb_%s.log" % tag).read()[-1500:])
else:
    j = json.loads(l[-1]); c = j["config"]
    st = c.get("stage_ms_per_step", {})
    ro = [r for r in (j.get("roofline"), j.get("roofline_other")) if r]
    print("%-14s value %8.1f  step %6.3f ms  ba %6.3f ms  frames %5.2f ms  ext_frac %s  create %s ms  %s  %s" % (
        tag, j.get("value", 0), j.get("ms_per_step", 0), c.get("ba_ms_per_step", 0), st.get("total", 0),
        (c.get("extractor_vs_survey_bytes") or {}).get("frac_of_8TBps"), (c.get("ba_window_setup") or {}).get("ms_per_window_inside_the_step"),
        " ".join("%s=%.1fus" % (r["kernel"], 1e3 * r["ms_per_launch"]) for r in ro), c.get("ba_worker_ms")))
PY
