#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the same command bench.py is judged on          -> gpurun_out/prof/<tag>_bench_*
#   2. FETCH_SIZE and WRITE_SIZE of the frame-path kernels, one --pmc pass each   -> gpurun_out/prof/<tag>_pmc_{fetch,write}_*
#   3. --kernel-trace --stats of CreateNewMapPoints (8 key frames x 20 neighbours) and of 8 local-BA windows in lock-step
# Summaries are then copied into profiles/ by tools/summarise_profiles.py.
set -u
TAG=${1:-r01}
PB=${PROF_B:-256}          # frames per dispatch of the PMC passes = bench.py's default --batch
export PROF_B=$PB
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_bench -- python $R/bench.py --steps 10 --warmup 2 --cpu-frames 0 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o ${TAG}_pmc_fetch -- python $R/tools/prof_frames.py $PB 550 3 > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o ${TAG}_pmc_write -- python $R/tools/prof_frames.py $PB 550 3 > $OUT/${TAG}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_mapping -- python $R/tools/prof_tri.py 8 20 5 > $OUT/${TAG}_mapping.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_ba8 -- python $R/tools/prof_ba_many.py 8 > $OUT/${TAG}_ba8.log 2>&1
# 4. SQ instruction mix of the frame-path kernels (three --pmc passes, kernel trace only)                 -> gpurun_out/pmc/
bash $R/tools/pmc_mix.sh > $OUT/${TAG}_pmc_mix.log 2>&1
ls $OUT
