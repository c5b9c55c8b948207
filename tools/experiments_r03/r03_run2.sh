#!/bin/bash
set -u
O=gpurun_out/r2; mkdir -p $O
L=$PWD/cubemapslam_amd/lib/ab_r02.so
for args in "312 track" "142 track" "204 track 20 12000 2 0.0" "313 random"; do
  echo "== $args"
  python tools/diag_ba_window.py $args 2>&1 | tail -14
  CMS_BA_NO_RUNS=1 python tools/diag_ba_window.py $args 2>&1 | tail -8
  CMS_BA_RUNS_AS_EDGES=1 python tools/diag_ba_window.py $args 2>&1 | head -1
  CMS_BA_SEPARATE_REDUCE=1 python tools/diag_ba_window.py $args 2>&1 | head -1
  CMS_BA_DETERMINISTIC=1 python tools/diag_ba_window.py $args 2>&1 | head -1
  CMS_HIP_LIB=$L python tools/diag_ba_window.py $args 2>&1 | head -1
done > $O/diag.log 2>&1
cat $O/diag.log
