import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def clouds():
    z = np.load(os.path.join(GOLDEN, "fixture_clouds.npz"))
    return {0: z["cloud_bin_0"], 1: z["cloud_bin_1"]}


@pytest.fixture(scope="session")
def images():
    z = np.load(os.path.join(GOLDEN, "fixture_images.npz"))
    # [1,3,120,160] float32, as scripts/generate_desc.py:96-97 builds it
    return {i: np.transpose(z[f"image_{i}"], (2, 0, 1))[None].copy() for i in (0, 1)}


@pytest.fixture(scope="session")
def golden():
    return dict(np.load(os.path.join(GOLDEN, "golden_descriptors.npz")))


@pytest.fixture(scope="session")
def head_map():
    return np.load(os.path.join(GOLDEN, "head_map_xyz.npz"))["xyz"]


@pytest.fixture(scope="session")
def seeded_sd():
    import imf_oracle as O
    return O.seeded_state_dict(seed=0, with_unused_image_layers=True)


@pytest.fixture
def fast_mode():
    """The split-f16 FAST mode (imf_conv_args.variant 6: two f16 parts per operand, operand images between layers, the fused
    head, IMF_FLAG_RANGE + fp32 recompute) for the duration of one test.  The process default since round 5 is variant 3
    (bf16x3: exact fp32 operands, fp32 buffers, no range guard); the tests of variant 6's own machinery pin it with this."""
    from imfnet_amd import ops
    prev = ops.CONV_VARIANT
    ops.CONV_VARIANT = 6
    yield
    ops.CONV_VARIANT = prev
