#!/bin/bash
# the windows' own (creation) streams at low priority like the frame path's, the group streams that carry the Levenberg rounds at normal priority
set -u
for i in 1 2 3; do
bash tools/gb.sh prepnormal_$i
CMS_BA_STREAM_PRIORITY=low bash tools/gb.sh preplow_$i
done
