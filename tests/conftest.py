import os
import sys

import pytest

# The GPU tests check results, not speed: let MIOpen pick its convolution kernels heuristically instead of benchmarking
# every new shape on a fresh box (the exhaustive find costs minutes for the fp32 ResNeXt / U-Net shapes of Boost).
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stable-diffusion-webui-depthmap-script_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch
