import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A tree that was never built (fresh clone): build once, like `__graft_entry__.build()` does --
    # the HIP libraries (hipcc cross-compiles without a GPU), the generated solver models and the
    # CPU oracles the tests check them against.  Nothing is built when the artefacts are there.
    need = [os.path.join(ROOT, "tpl_amd", "libtplhip.so"),
            os.path.join(ROOT, "tpl_amd", "models", "libilqr_trajectory_tracking_mpc.so"),
            os.path.join(ROOT, "oracle", "libdp_lon_oracle.so"),
            os.path.join(ROOT, "oracle", "libilqr_oracle_trajectory_tracking_mpc.so")]
    if not all(os.path.exists(f) for f in need):
        import __graft_entry__
        __graft_entry__.build()


def _have_gpu():
    try:
        import torch
        return torch.cuda.device_count() > 0 and os.path.exists("/dev/kfd")
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
