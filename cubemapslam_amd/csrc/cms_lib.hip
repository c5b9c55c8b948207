// cms_lib.hip -- single translation unit of libcubemapslam_hip.so (kernels + C-ABI host side), gfx950 only.
// Build: python -m cubemapslam_amd.build   (hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC)
#include "cms_extract_kernels.hip"
#include "cms_match_kernels.hip"
#include "cms_area_kernels.hip"
#include "cms_ba_kernels.hip"
#include "cms_ba_fused.hip"
#include "cms_ba_schur_points.hip"
#include "cms_ba_wrappers.hip"
#include "cms_pose_opt.hip"
#include "cms_api_frames.hip"
#include "cms_api_area.hip"
#include "cms_api_ba.hip"
#include "cms_api_pose.hip"
