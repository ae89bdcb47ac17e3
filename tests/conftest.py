import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; oracle/bfir_oracle.c)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def bfir():
    """The product package; its HIP library must already be built."""
    import foo_dsp_bfir_amd as b
    b.load()
    return b


class env_override:
    """Set environment switches the library reads at engine creation / per launch and put back what was there before --
    a switch given to the whole run (scripts/gpu_env_matrix.sh) must survive a test that flips it for one engine."""
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def rel_err(y, ref):
    """max |y - ref| / max |ref|: the norm the 1e-5 / 1e-12 tolerances are stated in."""
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-300))


# north_star: "within 1e-5 relative fp32 (1e-12 fp64)"
TOL = {4: 1e-5, 8: 1e-12}
