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


@pytest.mark.parametrize("s", [4, 8])
def test_calculate_attenuation_skips_failed_blocks_and_goes_on(orc, bfir, s):
    """A run() that returns -1 only drops ITS block from the scan (preprocessor.cpp:329-356), and -- because
    brutefir::run returns before it advances (brutefir.cpp:316-321, 337-340) -- from the history too: the blocks
    after it are scanned as if it had never been fed.  The expectation is the reference's own loop, block by block
    on ONE oracle engine (which restates that early return); the product replays it on the GPU."""
    from foo_dsp_bfir_amd import preprocessor as pp
    rng = np.random.default_rng(33)
    dt = orc.real_dtype(s)
    L, C, n = 256, 2, 1500
    B = pp.get_next_multiple(n, L) // L
    ir = np.stack(orc.synth_ir(rng, C, n, dt), axis=1) * dt(25.0)
    noise = orc.synth_audio(rng, B * L, C, dt)
    noise[2 * L + 17, 1] = np.inf                                  # block 2 fails; its neighbours must not
    noise[4 * L + 3, 0] = np.nan                                   # and block 4
    # the reference's loop verbatim on the oracle
    taps = [np.ascontiguousarray(np.concatenate([ir[:, c], np.zeros(B * L - n, dt)])) for c in range(C)]
    ref = orc.Engine(L, B, s, C)
    assert ref.set_coeff(taps, B, 1.0, L) == 0
    want_max, failed = 0.0, []
    for b in range(B):
        rc, out = ref.run(noise[b * L:(b + 1) * L])
        if rc == 0:
            want_max = max(want_max, float(np.abs(out).max()))
        else:
            failed.append(b)
    assert failed == [2, 4] and want_max > 1.0
    want = -20.0 * np.log10(want_max)
    got = pp.calculate_attenuation(ir, L, s, noise)
    assert abs(got - want) <= (1e-4 if s == 4 else 1e-10)
    # the shared driver code on oracle-backed filters takes the same replay path
    again = pp.calculate_attenuation(ir, L, s, noise, make_engine=lambda *a: _OracleFilter(orc, *a))
    assert abs(again - want) <= 1e-12
