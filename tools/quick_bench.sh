#!/bin/bash
# Developer aid: bench.py's headline pass only (every extra pass off), extra arguments / environment passed through.
#   tools/quick_bench.sh TAG [ENV=VALUE ...] -- [bench.py arguments]
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
ENVS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
[ "${1:-}" = "--" ] && shift
env "${ENVS[@]}" CMS_BENCH_NO_PY_LOOP=1 python $R/bench.py --extract-only-steps 0 --random-views-steps 0 --optimise-only-steps 0 --unpipelined-steps 0 --mapping-only-steps 0 --deterministic-steps 0 \
  --closed-loop-frames 0 --confined-steps 0 --cpu-frames 0 --no-streaming-pass --verify-windows 0 "$@" > $R/gpurun_out/$TAG.json 2> $R/gpurun_out/$TAG.err
python - <<PY
import json
try:
    d = json.loads(open("$R/gpurun_out/$TAG.json").read().strip().splitlines()[-1])
    print("$TAG", "value", d["value"], "ms_per_step", d["ms_per_step"], "schur ms/launch", d["roofline"]["ms_per_launch"], "frac", d["roofline"]["frac"])
except Exception as e:
    print("$TAG failed:", e); print(open("$R/gpurun_out/$TAG.err").read()[-800:])
PY
