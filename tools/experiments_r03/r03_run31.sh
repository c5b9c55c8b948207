#!/bin/bash
# window groups per step, again, with the frame path at low priority and the global copy of the reduced system
set -u
for i in 1 2; do
bash tools/gb.sh g2_$i
bash tools/gb.sh g3_$i --ba-groups 3
bash tools/gb.sh g1_$i --ba-groups 1
done
