"""Reference-held input data committed as fixtures (tests/golden/make_reference_fixtures.py made them in the build container):
the cubemap masks of /root/reference/Masks and the parsed values of /root/reference/Config/*.yaml.  Data only."""
import os
import numpy as np

_G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MASK_KEYS = {("lafida", 450): "lafida_450", ("lafida", 550): "lafida_550", ("lafida", 650): "lafida_650", ("front", 650): "front_650", ("left", 650): "left_650"}


def reference_mask(name, F):
    """the reference's own cubemap mask for camera `name` at face size F as a uint8 image (0 / 255), 3F x 3F"""
    z = np.load(os.path.join(_G, "reference_masks.npz"))
    key = MASK_KEYS[(name, F)]
    h, w = (int(v) for v in z[key + "_shape"])
    bits = np.unpackbits(z[key + "_bits"], axis=1)[:, :w]
    assert bits.shape == (h, w) and h == 3 * F
    return np.ascontiguousarray(bits * np.uint8(255))


def reference_config(key):
    """parsed values of Config/<key>_params.yaml: (camera struct bytes, orb params bytes, [fps, withFisheyeMask, RGB])"""
    z = np.load(os.path.join(_G, "reference_configs.npz"))
    return bytes(z[key + "_camera"]), bytes(z[key + "_orb"]), z[key + "_misc"]
