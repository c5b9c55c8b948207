#!/bin/bash
# developer helper (GPU box): the step with the process confined to the given cores, for a few environments
# usage: tools/confined.sh <cores e.g. 0,1> <tag> [ENV=val ...]
R=$GRAFT_REPO_ROOT; cores=$1; tag=$2; shift; shift
env "$@" CMS_BENCH_NO_PY_LOOP=1 taskset -c $cores python $R/bench.py --steps ${STEPS:-12} --warmup 3 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 --confined-steps 0 --extract-only-steps 0 --random-views-steps 0 --mapping-only-steps 0 --unpipelined-steps 0 --deterministic-steps 0 $BENCH_ARGS > $R/gpurun_out/conf_$tag.json 2> $R/gpurun_out/conf_$tag.err
python - $R/gpurun_out/conf_$tag.json $tag $cores <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "FAILED")
else:
    j = json.loads(l[-1]); h = j["config"]["host"]
    print("%-14s cores %-8s value %8.1f  step %6.2f ms  cores used %s  waits %s  driver %s" % (sys.argv[2], sys.argv[3], j["value"], j["ms_per_step"], h.get("host_cores_used"), h.get("host_waits"), h.get("step_driver", "")[:6]))
PY
