#!/bin/bash
# round 5, GPU call 1: the whole GPU suite after the device-side planner / ADVICE fixes / DPP Hamming, the host cost of a window's set-up, a default bench run
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gputests1.txt
CMS_BA_CREATE_TIMING=1 timeout 120 python tools/prof_ba_create.py 8 > $O/create_fast.txt 2>&1
CMS_BA_HOST_PLAN=1 CMS_BA_CREATE_TIMING=1 timeout 120 python tools/prof_ba_create.py 8 > $O/create_host.txt 2>&1
timeout 300 python tools/prof_ba_many.py 16 track diff 3 2>&1 | grep -v "^$" | tail -6 > $O/ba16_fast.txt
CMS_BA_HOST_PLAN=1 timeout 300 python tools/prof_ba_many.py 16 track diff 3 2>&1 | tail -6 > $O/ba16_host.txt
timeout 600 python bench.py > $O/bench1.json 2> $O/bench1.err
CMS_BA_HOST_PLAN=1 bash tools/gb.sh r05_hostplan > $O/gb_hostplan.txt 2>&1
bash tools/gb.sh r05_fastplan > $O/gb_fastplan.txt 2>&1
tail -3 $O/gputests1.txt; tail -3 $O/create_fast.txt; tail -3 $O/create_host.txt; cat $O/ba16_fast.txt | tail -3; cat $O/gb_hostplan.txt $O/gb_fastplan.txt | cut -c1-400; tail -c 600 $O/bench1.json
