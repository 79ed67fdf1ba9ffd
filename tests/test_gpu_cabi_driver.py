"""The C ABI used from C (no Python, no torch in the client): tests/cabi/driver.cpp includes include/imfnet_hip.h, links
libimfnet_hip.so and checks voxelisation + rulebook + one fused convolution against exactly known results."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "build", "cabi_driver")


@pytest.mark.gpu
def test_c_driver_runs_against_the_shared_library():
    if not os.path.exists(DRIVER):
        subprocess.run(["make", "-C", os.path.join(ROOT, "imfnet_amd", "csrc"), "driver"], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "imfnet_amd") + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([DRIVER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "C ABI driver OK" in r.stdout, r.stdout
