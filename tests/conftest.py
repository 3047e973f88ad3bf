import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

REFDATA = os.path.join(REPO, "tests", "golden", "refdata")
GOLDEN = os.path.join(REPO, "tests", "golden")
EMUL_LIB = os.path.join(REPO, "tests", "emul", "libdtrl_emul.so")   # lane-loop build of the kernel source: TESTS ONLY, lives outside the product package
HIP_LIB = os.path.join(REPO, "deepterrainrl_amd", "lib", "libdtrl.so")
REFERENCE = "/root/reference"
# the test process's own choice, before any HIP runtime starts (the package leaves the environment alone: deepterrainrl_amd.configure_hw_queues)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """Build the oracle + both libraries on demand (CPU box: hipcc cross-compiles gfx950 without a GPU)."""
    need = [os.path.join(REPO, "oracle", "_ref", "libdtrl_oracle.so"), EMUL_LIB, HIP_LIB]
    if all(os.path.exists(p) for p in need):
        return
    import __graft_entry__
    __graft_entry__.build()


@pytest.fixture(scope="session", autouse=True)
def built():
    _ensure_built()


def _emul_scenario_cls():
    import deepterrainrl_amd

    class EmulScenario(deepterrainrl_amd.BatchScenario):
        """BatchScenario bound to the lane-loop CPU build of the kernel source (tests/emul/): checks host logic and kernel math without a GPU."""
        def _library(self):
            return deepterrainrl_amd._bind(EMUL_LIB)
    return EmulScenario


class _Lazy:
    def __call__(self, *a, **k):
        return _emul_scenario_cls()(*a, **k)


EmulScenario = _Lazy()   # callable like the class; resolved lazily so that importing conftest does not import the package


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests need a HIP device: skip them (instead of failing in dtrl_create) when none is visible."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = os.path.exists("/dev/kfd")
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="needs a HIP device (run with -m gpu on an MI355X box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def om():
    from oracle import model
    return model


@pytest.fixture(scope="session")
def da():
    import deepterrainrl_amd
    return deepterrainrl_amd


def dog_policy(om, scale="data/policies/dog/models/dog_mace3_slopes_mixed_model_scale.txt", seed=1234):
    desc = om.parse_deploy_prototxt(os.path.join(REFDATA, "data/policies/dog/nets/dog_mace3_deploy.prototxt"))
    w = om.xavier_weights(desc, seed)
    io, isc, oo, osc = om.load_scale_file(os.path.join(REFDATA, scale))
    return desc, w, io, isc, oo, osc
