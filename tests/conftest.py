import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "grad: the test records an autograd graph (training path); every other test runs "
                                       "under torch.no_grad(), i.e. on the inference operators")


@pytest.fixture(autouse=True)
def _inference_unless_marked(request):
    """Inference is `torch.no_grad()` (ScenePipeline, inference_utils): with gradients enabled the model tree switches to its
    torch-operator gradient path (bev_blocks.grad_path), and a parity test that forgot no_grad() would silently test torch
    instead of the HIP kernels.  Tests of the training path opt in with @pytest.mark.grad."""
    import torch
    if request.node.get_closest_marker("grad"):
        yield
        return
    with torch.no_grad():
        yield


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return load
