import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _reset_error_checking():
    """Some code paths switch the device-status read-back off for latency; every test starts from the default ("deferred": a launch's status
    word is read at the next synchronisation point, ops.check_deferred) with nothing pending."""
    yield
    try:
        from gabotorch_amd import ops
        ops.set_error_checking("deferred")
        for ring in ops._status_rings.values():
            ring.reset()
        ops._deferred.clear()
    except Exception:   # noqa: BLE001
        pass


@pytest.fixture(params=["sync", "deferred"])
def raising(request):
    """`with raising("not positive definite"): <launches>` under both ways a device-side data error reaches the caller: at the call
    (ops.set_error_checking(True): a stream synchronisation per call, the reference's torch.cholesky behaviour, spd_utils_torch.py:87) and at the
    next ops.check_deferred() (the default: no synchronisation in the call)."""
    import contextlib

    from gabotorch_amd import ops
    mode = request.param
    ops.set_error_checking(True if mode == "sync" else "deferred")

    def factory(match):
        @contextlib.contextmanager
        def cm():
            with pytest.raises(RuntimeError, match=match):
                yield
                if mode == "deferred":
                    ops.check_deferred()
        return cm()
    factory.mode = mode
    return factory
