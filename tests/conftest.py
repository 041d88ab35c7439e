import os
import sys

import pytest

# PM_TEST_HW_QUEUES=<n>: run the suite with that many HIP hardware queues (GPU_MAX_HW_QUEUES; the runtime reads it once,
# when it starts — i.e. at the torch.cuda call below).  Unset = the runtime's default of 4, which is what the suite
# has been verified under; the multi-rank and multi-pool tests are worth a run under 16.
if os.environ.get("PM_TEST_HW_QUEUES"):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ["PM_TEST_HW_QUEUES"])

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# PM_EXP_DEFINES=A=1,B: the whole suite against a VARIANT build of the engine library (extra -D flags; the mechanism of
# tools/variant_bench.py) — how an experiment behind an #ifdef is put through the parity tests before it becomes the
# default.  Unset = the product library.
# PM_EXP_LIB=<path>: the same with a variant library built beforehand (tools/build_variants.py builds them where there is
# no GPU; they travel to the GPU box with the tree, and the box's minutes go into the tests instead of hipcc).
if os.environ.get("PM_EXP_LIB"):
    from protocol_amd import build as _B
    assert os.path.exists(os.environ["PM_EXP_LIB"]), os.environ["PM_EXP_LIB"]
    _B.LIB_PATH = os.path.abspath(os.environ["PM_EXP_LIB"])
    _B.needs_build = lambda: False
elif os.environ.get("PM_EXP_DEFINES"):
    from protocol_amd import build as _B
    _alt = os.path.join(os.path.dirname(_B.LIB_PATH), "libpm_engine_exp.so")
    _B.build(force=True, defines=[d for d in os.environ["PM_EXP_DEFINES"].split(",") if d], out=_alt)
    _B.LIB_PATH = _alt
    _B.needs_build = lambda: False


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
