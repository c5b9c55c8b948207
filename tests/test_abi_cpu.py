"""CPU-side checks of the product library: it builds, loads, exports every symbol include/cubemapslam_hip.h declares,
and refuses to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re
import numpy as np
import pytest
from cubemapslam_amd import api, build, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_the_declared_abi():
    build.build(verbose=False)
    hdr = open(os.path.join(ROOT, "include", "cubemapslam_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(cms_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    L = api.lib()
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing


def test_struct_layouts_match_the_header():
    assert C.sizeof(api.Camera) == 5 * 8 + 12 * 8 + 5 * 8 + 3 * 4 + 4 + 8
    assert C.sizeof(api.OrbParams) == 20 and api.KP_DTYPE.itemsize == 24
    assert C.sizeof(api.BaStats) == 8 + 6 * 8 + 8
    assert C.sizeof(api.PoseStats) == 6 * 4


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.CmsError) as ei:
        api.Context(synth.camera("lafida", 150), nfeatures=500)
    assert "no HIP device" in str(ei.value) or "failed" in str(ei.value)
    with pytest.raises(api.CmsError):
        api.ba_run(synth.ba_problem(K=3, P=10, obs_per_point=2, seed=1))
    with pytest.raises(api.CmsError):
        api.PoseOptimizer(1, 16)
    with pytest.raises(api.CmsError):
        api.pose_optimize(synth.pose_problem(N=20, seed=1))


def test_product_sources_do_not_touch_the_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(ROOT, "cubemapslam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", ".inc")):
                s = open(os.path.join(root, f), errors="ignore").read()
                if re.search(r"liborc|orc_api\.h|oracle/orc_|import orc\b", s):
                    bad.append(f)
    assert not bad, bad
