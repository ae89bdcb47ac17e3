"""The HIP engine against the committed golden vectors (tests/golden/, see make_golden.py:
seeded inputs + independent direct-form convolution + oracle output)."""
import glob
import os

import numpy as np
import pytest

from conftest import TOL, rel_err

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_engine_reproduces_golden(bfir, path):
    g = np.load(path)
    s, L, B, C, taps, nb = (int(v) for v in g["params"])
    eng = bfir.Brutefir(L, B, s, C)
    eng.set_chunk(16)
    assert eng.set_coeff(list(g["h"])) == 0
    rc, y = eng.run(g["x"])
    assert rc == 0
    assert rel_err(y, g["y_oracle"]) <= TOL[s]
    assert rel_err(y, g["y_direct"]) <= TOL[s]
