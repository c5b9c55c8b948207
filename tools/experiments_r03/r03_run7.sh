#!/bin/bash
CMS_HIP_LIB=$PWD/cubemapslam_amd/lib/ab_rmclk.so python tools/prof_rm_clk.py 16 2>&1 | tail -12
CMS_HIP_LIB=$PWD/cubemapslam_amd/lib/ab_rmclk.so python tools/prof_rm_clk.py 1 2>&1 | tail -12
