#!/bin/bash
# measurements / informations / point positions gathered into the internal order on the device (k_ba_gather) against on the host (before)
set -u
L=$PWD/cubemapslam_amd/lib
for i in 1 2 3; do
bash tools/gb.sh gather_$i
CMS_HIP_LIB=$L/ab_pre.so bash tools/gb.sh hostgather_$i
done
