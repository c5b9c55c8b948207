#!/bin/bash
# host cost of a window against the Schur kernel's time: composition of the left-over chunks on / off / with shorter look-ahead
set -u
for v in "" "CMS_BA_NO_PERMUTE=1" "CMS_BA_LOOKAHEAD=12" "CMS_BA_LOOKAHEAD=6"; do
  echo "== ${v:-default}"
  env $v python tools/prof_ba_many.py 16 track diff 2>&1 | grep "lock-step\|cms_ba_create" | tail -2
done
for i in 1 2; do
bash tools/gb.sh base_$i
CMS_BA_NO_PERMUTE=1 bash tools/gb.sh noperm_$i
CMS_BA_LOOKAHEAD=6 bash tools/gb.sh la6_$i
done
