"""The offline drivers on the GPU engine against the same call sequence
run on the CPU oracle."""
import numpy as np
import pytest

from conftest import TOL, rel_err

pytestmark = pytest.mark.gpu


class _OracleFilter:
    """The oracle behind the Brutefir python interface, for the shared driver code."""
    def __init__(self, orc, L, B, s, C):
        self.e = orc.Engine(L, B, s, C)

    def set_coeff(self, coeffs, n_coeffs, length, coeff_blocks, scale):
        return self.e.set_coeff(coeffs[:n_coeffs], coeff_blocks, scale, length)

    def run(self, x):
        return self.e.run(x)

    def overflow(self, c):
        return self.e.overflow(c)


@pytest.mark.parametrize("s", [4, 8])
def test_convolve_impulses(orc, bfir, s):
    from foo_dsp_bfir_amd import preprocessor as pp
    rng = np.random.default_rng(31)
    dt = orc.real_dtype(s)
    L, C = 256, 2
    imps = [(np.stack(orc.synth_ir(rng, C, n, dt), axis=1), sc) for n, sc in ((900, 1.0), (700, 0.5), (1000, 2.0))]
    got = pp.convolve_impulses(imps, L, s)
    want = pp.convolve_impulses(imps, L, s, make_engine=lambda *a: _OracleFilter(orc, *a))
    assert got.shape == want.shape == (1000, C)
    assert rel_err(got, want) <= TOL[s]


@pytest.mark.parametrize("s", [4, 8])
def test_calculate_attenuation(orc, bfir, s):
    from foo_dsp_bfir_amd import preprocessor as pp
    rng = np.random.default_rng(32)
    dt = orc.real_dtype(s)
    L, C, n = 512, 2, 1800
    ir = np.stack(orc.synth_ir(rng, C, n, dt), axis=1) * dt(40.0)       # loud enough to need attenuation
    noise = orc.synth_audio(rng, pp.get_next_multiple(n, L), C, dt)
    got = pp.calculate_attenuation(ir, L, s, noise)
    want = pp.calculate_attenuation(ir, L, s, noise, make_engine=lambda *a: _OracleFilter(orc, *a))
    assert want < 0.0
    assert abs(got - want) <= 1e-4 if s == 4 else abs(got - want) <= 1e-10
    quiet = pp.calculate_attenuation(ir * dt(1e-3), L, s, noise)
    assert quiet == 0.0
