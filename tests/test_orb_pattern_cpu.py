"""The rBRIEF test pattern (a10: steered BRIEF) is reference-held DATA: 256 x 4 integers in src/ORBExtractor.cpp:121-379 (`bit_pattern_31_`), committed here as
generated tables for the product (cubemapslam_amd/csrc/orb_pattern.inc) and the oracle (oracle/orc_pattern.inc) by tools/gen_orb_pattern.py.  Both tables must be
the same 1024 numbers, their digest must be the one recorded when they were generated, and -- in the build container, where /root/reference exists -- they must
be the reference's numbers as its source holds them today."""
import hashlib
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIGEST = "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"


def _table(path):
    txt = open(path).read()
    body = txt[txt.index("{") + 1:txt.rindex("}")]
    v = np.array([int(x) for x in re.findall(r"-?\d+", body)], np.int64)
    assert v.size == 1024, (path, v.size)
    return v


def test_product_and_oracle_hold_the_same_pattern_and_its_digest_is_the_recorded_one():
    a = _table(os.path.join(ROOT, "cubemapslam_amd", "csrc", "orb_pattern.inc"))
    b = _table(os.path.join(ROOT, "oracle", "orc_pattern.inc"))
    assert np.array_equal(a, b)
    assert np.abs(a).max() <= 15                       # every test point lies inside the 31 x 31 patch
    assert hashlib.sha256(a.astype(np.int8).tobytes()).hexdigest() == DIGEST


def test_pattern_equals_the_reference_source_where_it_is_present():
    ref = "/root/reference/src/ORBExtractor.cpp"
    if not os.path.exists(ref):
        pytest.skip("no reference checkout on this machine (the GPU box): the digest test pins the table")
    src = open(ref).read()
    i = src.index("bit_pattern_31_")
    body = re.sub(r"/\*.*?\*/", "", src[src.index("=", i):src.index("};", i)], flags=re.S)
    nums = np.array([int(x) for x in re.findall(r"-?\d+", body)], np.int64)
    assert nums.size == 1024
    assert np.array_equal(nums, _table(os.path.join(ROOT, "cubemapslam_amd", "csrc", "orb_pattern.inc")))


def test_the_oracles_named_constants_are_the_references_where_it_is_present():
    """TH_HIGH / TH_LOW / HISTO_LENGTH (ORBMatcher.cpp:42-44: 12 bins here, not ORB-SLAM2's 30), PATCH_SIZE / HALF_PATCH_SIZE / EDGE_THRESHOLD
    (ORBExtractor.cpp:43-45), the chi2 gate 5.991 and the Huber widths sqrt(5.991) of PoseOptimization / LocalBundleAdjustment (Optimizer.cpp:77,138,300,384):
    read out of the reference's text and out of the oracle's sources, compared as numbers."""
    ref = "/root/reference/src"
    if not os.path.isdir(ref):
        pytest.skip("no reference checkout on this machine (the GPU box)")
    def const(path, name):
        m = re.search(r"\b%s\s*=\s*(-?[0-9.]+)" % re.escape(name), open(path).read())
        assert m, (path, name)
        return float(m.group(1))
    orc = os.path.join(ROOT, "oracle")
    for name, ref_file, orc_file in (("TH_HIGH", "ORBMatcher.cpp", None), ("TH_LOW", "ORBMatcher.cpp", "orc_track.cpp"), ("HISTO_LENGTH", "ORBMatcher.cpp", "orc_track.cpp"),
                                     ("PATCH_SIZE", "ORBExtractor.cpp", "orc_orb.cpp"), ("HALF_PATCH_SIZE", "ORBExtractor.cpp", "orc_orb.cpp"),
                                     ("EDGE_THRESHOLD", "ORBExtractor.cpp", "orc_orb.cpp")):
        r = const(os.path.join(ref, ref_file), name)
        if orc_file:
            assert const(os.path.join(orc, orc_file), name) == r, name
        else:
            assert r == 100.0                      # (TH_HIGH is an argument of the oracle's matchers: the tests pass 100)
    opt = open(os.path.join(ref, "Optimizer.cpp")).read()
    ba = open(os.path.join(orc, "orc_ba.cpp")).read()
    assert "thHuberMono = sqrt(5.991)" in opt and "deltaMono = sqrt(5.991)" in opt and opt.count("e->chi2()>5.991") >= 2
    assert ba.count("std::sqrt(5.991)") >= 2 and "c2 > 5.991" in ba and "chi2 > 5.991f" in ba
