#!/bin/bash
# the default bench line (what the driver runs) with its key numbers printed: bash tools/experiments_r04/bench_full.sh [tag] [bench args...]
TAG=${1:-default}; shift
mkdir -p gpurun_out/r04
( time python bench.py "$@" ) > gpurun_out/r04/bench_$TAG.json 2> gpurun_out/r04/bench_$TAG.err; tail -4 gpurun_out/r04/bench_$TAG.err
python - gpurun_out/r04/bench_$TAG.json <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
j = json.loads(l[-1]); c = j["config"]
print("value", j["value"], "ms_per_step", j["ms_per_step"], "ba", c["ba_ms_per_step"], "worker", c["ba_worker_ms"])
print("extractor", c["extractor_vs_survey_bytes"]); print("extract_only", c.get("extract_only")); print("host", c.get("host"))
print("random", c.get("ba_views_random")); print("optimise_only", c["optimise_only"])
print("roofline", {k: j["roofline"][k] for k in ("kernel", "frac", "ms_per_launch", "achieved")}); print("ba_check", c["ba_check"])
print("one_call", c["one_local_ba_call"], "setup", c["ba_window_setup"]["ms_per_window_one_after_the_other"], c["ba_window_setup"]["ms_per_window_inside_the_step"])
print("closed", c["single_stream_closed_loop"]); print("cpu", j["cpu_baseline"]["value"] if j["cpu_baseline"] else None, j["cpu_baseline"]["two_threads_like_the_reference"] if j["cpu_baseline"] else None)
print("streaming", c["with_input_streaming"])
PY
