#!/bin/bash
# Developer aid: build the HIP library as of git revision REV into cubemapslam_amd/lib/ab_NAME.so (A/B partner of the working tree).
#   tools/ab_build_rev.sh REV NAME [-D...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
REV=$1; NAME=$2; shift 2
T=$(mktemp -d)
(cd $R && git archive $REV cubemapslam_amd/csrc include) | tar -x -C $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value "$@" $T/cubemapslam_amd/csrc/cms_lib.hip -o $R/cubemapslam_amd/lib/ab_$NAME.so
rm -rf $T
echo built $R/cubemapslam_amd/lib/ab_$NAME.so from $REV
