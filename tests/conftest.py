import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: exhaustive CPU check (about 10 s)")


@pytest.fixture(scope="session")
def orc():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def model():
    import oracle_lib
    return oracle_lib.KernelMathModel()


@pytest.fixture(scope="session")
def lena_bgr():
    import numpy as np
    from PIL import Image
    path = os.path.join(ROOT, "tests", "golden", "lena_color_1024.png")
    rgb = np.array(Image.open(path).convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])  # cv::imread order: B, G, R


@pytest.fixture(scope="session")
def gpu():
    import torch
    import jpeg_amd
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (there is no CPU fallback)")
    jpeg_amd.init(0)
    return torch.device("cuda:0")
