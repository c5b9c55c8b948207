#!/bin/bash
set -u
bash tools/gb.sh base
CMS_BENCH_BA_FIRST=0 bash tools/gb.sh bafirst0
CMS_BENCH_BA_FIRST=300 bash tools/gb.sh bafirst300
CMS_BENCH_BA_FIRST=800 bash tools/gb.sh bafirst800
CMS_BA_RM_VALU=1 CMS_BA_RM_WEIGHT=60 bash tools/gb.sh valu
CMS_BA_NO_RUNS=1 bash tools/gb.sh noruns
bash tools/gb.sh base2
