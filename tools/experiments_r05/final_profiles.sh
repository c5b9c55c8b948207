#!/bin/bash
# round 5: the committed evidence (profiles/r05_*): run on the GPU box, then `python tools/summarise_profiles.py r05` in the build container
O=gpurun_out/r05; mkdir -p $O
bash tools/run_profiles.sh r05 > $O/run_profiles.log 2>&1
bash tools/experiments_r05/exp.sh create > /dev/null 2>&1
bash tools/experiments_r05/exp.sh create_in_step > /dev/null 2>&1
bash tools/experiments_r05/exp.sh frames > /dev/null 2>&1
timeout 900 python bench.py > $O/bench_default_4.json 2> $O/bench_default_4.err
timeout 900 python bench.py > $O/bench_default_5.json 2> $O/bench_default_5.err
ls gpurun_out/prof | head -50; tail -3 $O/run_profiles.log
