"""The oracle against its own committed regression vectors (tests/golden/oracle_v1.npz, made by tests/golden/make_golden.py): integer /
index / byte outputs and float outputs of the bit-exact stages identical, solver outputs to 1e-9."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden


def test_oracle_reproduces_golden_vectors():
    want = np.load(os.path.join(HERE, "golden", "oracle_v1.npz"))
    got = make_golden.build()
    assert set(got) == set(want.files)
    loose = {"ba_chi2_final", "ba_points_head", "pose_pose"}
    for k in want.files:
        if k in loose:
            assert np.allclose(got[k], want[k], rtol=1e-9, atol=1e-12), k
        else:
            assert np.array_equal(np.asarray(got[k]), want[k]), k
