#!/bin/bash
# Developer aid: the deterministic pass of bench.py next to the Python loop's default step (the pass it is compared with), other extra passes off.
#   tools/det_leg.sh [tag] [ENV=VALUE ...]
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-det}; shift
env "$@" python $R/bench.py --extract-only-steps 0 --random-views-steps 0 --optimise-only-steps 0 --unpipelined-steps 0 --mapping-only-steps 0 --deterministic-steps 10 \
  > $R/gpurun_out/$TAG.json 2> $R/gpurun_out/$TAG.err
python - <<PY
import json
d = json.loads(open("$R/gpurun_out/$TAG.json").read().strip().splitlines()[-1])
c = d["config"]
print("$TAG", "value", d["value"], "python loop", (c.get("python_step_loop") or {}).get("value"), "deterministic", {k: c["deterministic"][k] for k in ("value", "ms_per_step", "ba_ms_per_step", "of_python_step_loop")})
PY
