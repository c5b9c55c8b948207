#!/bin/bash
# Developer aid: build a variant of the HIP library with extra -D flags for A/B timing on the GPU box.
#   tools/ab_build.sh NAME -DCMS_FAST_WPB=2 ...   ->  cubemapslam_amd/lib/ab_NAME.so   (use with CMS_HIP_LIB=<path>)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value "$@" \
  $R/cubemapslam_amd/csrc/cms_lib.hip -o $R/cubemapslam_amd/lib/ab_$NAME.so
echo built $R/cubemapslam_amd/lib/ab_$NAME.so
