#!/bin/bash
# round 3, GPU call 1: the new local-BA path -- parity first, then A/B timings (run through gpurun from the repo root)
set -u
O=gpurun_out/r1; mkdir -p $O
echo "== new BA tests" ; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tracked_windows or signature_runs or group_of_sixteen or repeatable" > $O/t_new.log 2>&1; tail -15 $O/t_new.log
echo "== full suite"; timeout 1500 python -m pytest tests -m gpu -q > $O/t_all.log 2>&1; tail -12 $O/t_all.log
echo "== BA alone, 16 windows"
for v in "track" "random"; do
  python tools/prof_ba_many.py 16 $v diff > $O/ba16_$v.log 2>&1; tail -3 $O/ba16_$v.log
done
CMS_BA_NO_RUNS=1 python tools/prof_ba_many.py 16 track diff > $O/ba16_track_noruns.log 2>&1; tail -2 $O/ba16_track_noruns.log
CMS_BA_SEPARATE_REDUCE=1 python tools/prof_ba_many.py 16 track diff > $O/ba16_track_sepred.log 2>&1; tail -2 $O/ba16_track_sepred.log
CMS_BA_CREATE_TIMING=1 python tools/prof_ba_many.py 2 track > $O/ba_create_timing.log 2>&1; grep cms_ba_create $O/ba_create_timing.log | tail -3
echo "== bench A/B"
bash tools/gb.sh runs
CMS_BA_NO_RUNS=1 bash tools/gb.sh noruns
CMS_BA_SEPARATE_REDUCE=1 bash tools/gb.sh sepred
bash tools/gb.sh random --ba-views random
bash tools/gb.sh runs2
bash tools/gb.sh grp1 --ba-groups 1
bash tools/gb.sh grp4 --ba-groups 4
CMS_BENCH_PART=ba bash tools/gb.sh baonly
echo "== probe"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe/lds_atomics.hip -o /tmp/lds_atomics 2>/dev/null && /tmp/lds_atomics > $O/probe_lds_atomics.txt 2>&1; tail -5 $O/probe_lds_atomics.txt
echo "== full bench line"
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -c 1500 $O/bench_full.json; tail -5 $O/bench_full.err
