#!/bin/bash
set -u
bash tools/gb.sh sdma
CMS_BA_STAGE_COHERENT=1 bash tools/gb.sh coherent
bash tools/gb.sh sdma2
CMS_BA_STAGE_COHERENT=1 bash tools/gb.sh coherent2
python tools/prof_ba_many.py 16 track diff 2>&1 | grep -E "create|read"
