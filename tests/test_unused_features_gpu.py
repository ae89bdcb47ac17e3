"""Methods the reference declares on fftw_convolver but never calls:
N-input mixnscale, dirac_convolve, convolve_eval, crossfade_inplace, runtime_coeffs2cbuf,
verify_cbuf -- against the oracle's restatement of the same reference lines."""
import numpy as np
import pytest

from conftest import TOL, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("s", [4, 8])
@pytest.mark.parametrize("n_bufs", [2, 3, 4, 7])
def test_mixnscale_many_inputs_bit_exact(orc, bfir, s, n_bufs):
    L = 256
    dt = orc.real_dtype(s)
    rng = np.random.default_rng(n_bufs)
    ins = [rng.standard_normal(2 * L).astype(dt) for _ in range(n_bufs)]
    scales = list(rng.uniform(-2, 2, n_bufs))
    cv = bfir.FftwConvolver(L, s)
    for mode, omode in ((bfir.MIXMODE_INPUT, orc.MIXMODE_INPUT), (bfir.MIXMODE_OUTPUT, orc.MIXMODE_OUTPUT)):
        out = cv.new_cbuf()
        cv.convolver_mixnscale(ins, out, scales, n_bufs, mode)
        assert np.array_equal(out, orc.mixnscale_n(ins, scales, omode))
    with pytest.raises(bfir.BfirError):
        cv.convolver_mixnscale(ins, out, scales, n_bufs, 2)       # INPUT_ADD has no case in the reference either


@pytest.mark.parametrize("s", [4, 8])
def test_dirac_convolve_bit_exact(orc, bfir, s):
    L = 512
    x = np.random.default_rng(1).standard_normal(2 * L).astype(orc.real_dtype(s))
    cv = bfir.FftwConvolver(L, s)
    out = cv.new_cbuf(); cv.convolver_dirac_convolve(x, out)
    assert np.array_equal(out, orc.dirac_convolve(x))
    y = x.copy(); cv.convolver_dirac_convolve_inplace(y)
    assert np.array_equal(y, out)


@pytest.mark.parametrize("s", [4, 8])
def test_convolve_eval(orc, bfir, s):
    L = 256
    dt = orc.real_dtype(s)
    rng = np.random.default_rng(2)
    cv = bfir.FftwConvolver(L, s)
    buf, rbuf = np.zeros(3 * L, dt), np.zeros(3 * L, dt)
    for _ in range(3):                                    # the buffer carries the overlap between calls
        hc = orc.r2hc(rng.standard_normal(2 * L).astype(dt))
        out = cv.new_cbuf(); cv.convolver_convolve_eval(hc, buf, out)
        want = orc.convolve_eval(hc, rbuf)
        assert rel_err(out, want) <= TOL[s] and rel_err(buf, rbuf) <= TOL[s]


@pytest.mark.parametrize("s", [4, 8])
def test_crossfade_inplace(orc, bfir, s):
    L = 256
    dt = orc.real_dtype(s)
    rng = np.random.default_rng(3)
    mk = lambda: orc.mixnscale(orc.r2hc(rng.standard_normal(2 * L).astype(dt)), 1.0 / (2 * L), orc.MIXMODE_INPUT)
    a, b = mk(), mk()
    tail = rng.standard_normal(3 * L).astype(dt)          # the double path reads buffer[n_fft ..]
    cv = bfir.FftwConvolver(L, s)
    ia, ib, ibuf = a.copy(), b.copy(), tail.copy()
    cv.convolver_crossfade_inplace(ia, ib, ibuf)
    ra, rb, rbuf = a.copy(), b.copy(), tail.copy()
    orc.crossfade_inplace(ra, rb, rbuf)
    assert rel_err(ia, ra) <= TOL[s]
    assert rel_err(ib, rb) <= TOL[s]                      # time-domain crossfade spectrum left behind
    assert rel_err(ibuf[:2 * L], rbuf[:2 * L]) <= TOL[s] and np.array_equal(ibuf[2 * L:], tail[2 * L:])


@pytest.mark.parametrize("s", [4, 8])
def test_runtime_coeffs2cbuf_and_verify(orc, bfir, s):
    L = 512
    dt = orc.real_dtype(s)
    taps = np.random.default_rng(4).standard_normal(L).astype(dt)
    cv = bfir.FftwConvolver(L, s)
    dest = cv.new_cbuf(); cv.convolver_runtime_coeffs2cbuf(taps, dest)
    assert np.array_equal(dest, cv.convolver_coeffs2cbuf(taps, L, 1.0))
    assert rel_err(dest, orc.coeffs2cbuf(taps, L, 1.0)) <= TOL[s]
    assert cv.convolver_verify_cbuf([dest, dest.copy()], 2)
    bad = dest.copy(); bad[17] = np.nan
    assert not cv.convolver_verify_cbuf([dest, bad], 2)
