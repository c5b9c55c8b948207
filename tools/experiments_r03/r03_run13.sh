#!/bin/bash
set -u
bash tools/gb.sh base
CMS_BENCH_TRI_LAST=1 bash tools/gb.sh trilast
bash tools/gb.sh base2
CMS_BENCH_TRI_LAST=1 bash tools/gb.sh trilast2
