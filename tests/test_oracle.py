"""CPU tests of the oracle itself (no GPU): against the committed golden vectors, an
independent long-double direct convolution, scipy's pocketfft and algebraic properties.

The oracle is PARITY UNPINNED with respect to the reference (see oracle/bfir_oracle.h):
the reference holds no fixtures and cannot run here; these tests pin it to the published
definition of the arithmetic instead."""
import glob
import os

import numpy as np
import pytest
import scipy.fft

from conftest import TOL, rel_err

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def _hc_from_rfft(x):
    n = x.size
    X = scipy.fft.rfft(x.astype(np.float64))
    hc = np.empty(n)
    hc[:n // 2 + 1] = X.real
    hc[n // 2 + 1:] = X.imag[1:n // 2][::-1]
    return hc


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("n", [4, 8, 32, 256, 2048, 8192, 32768])
def test_r2hc_hc2r_match_fftw_definition(orc, dt, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(dt)
    hc = orc.r2hc(x)
    tol = 2e-6 if dt == np.float32 else 1e-14
    assert rel_err(hc, _hc_from_rfft(x)) < tol
    # FFTW_HC2R is the unnormalised inverse: HC2R(R2HC(x)) = n x
    assert rel_err(orc.hc2r(hc) / n, x) < 5 * tol


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_grouped_layout(orc, dt):
    """mixnscale INPUT: out[8g+j] = Re X_{4g+j}, out[8g+4+j] = Im X_{4g+j}, out[4] = Re X_{n/2}."""
    n = 64
    hc = np.arange(1, n + 1).astype(dt)
    g = orc.mixnscale(hc, 2.0, orc.MIXMODE_INPUT)
    for k in range(n // 2):
        assert g[8 * (k // 4) + k % 4] == 2 * hc[k]
        if k:
            assert g[8 * (k // 4) + 4 + k % 4] == 2 * hc[n - k]
    assert g[4] == 2 * hc[n // 2]
    back = orc.mixnscale(g, 0.5, orc.MIXMODE_OUTPUT)
    assert np.array_equal(back, hc)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_convolve_is_complex_product_with_real_dc_and_nyquist(orc, dt):
    n = 128
    rng = np.random.default_rng(1)
    b, c, d0 = (rng.standard_normal(n).astype(dt) for _ in range(3))

    def cplx(v):
        z = np.zeros(n // 2, dtype=np.complex128)
        for k in range(n // 2):
            z[k] = v[8 * (k // 4) + k % 4] + 1j * v[8 * (k // 4) + 4 + k % 4]
        return z
    d = orc.convolve(b, c)
    zb, zc, zd = cplx(b), cplx(c), cplx(d)
    tol = 1e-6 if dt == np.float32 else 1e-14
    assert np.abs(zd[1:] - zb[1:] * zc[1:]).max() < tol * 10
    assert abs(d[0] - b[0] * c[0]) < tol and abs(d[4] - b[4] * c[4]) < tol
    da = orc.convolve_add(b, c, d0)
    assert rel_err(da, d0.astype(np.float64) + d) < tol
    assert np.array_equal(orc.convolve_inplace(b, c), d)


def test_coeffs2cbuf_rejects_nonfinite(orc):
    h = np.ones(10, np.float32)
    h[3] = np.nan
    assert orc.coeffs2cbuf(h, 16) is None
    assert orc.coeffs2cbuf(np.ones(10, np.float32), 16) is not None


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(orc, path):
    g = np.load(path)
    s, L, B, C, taps, nb = (int(v) for v in g["params"])
    eng = orc.Engine(L, B, s, C)
    assert eng.set_coeff(list(g["h"])) == 0
    rc, y = eng.run(g["x"])
    assert rc == 0
    assert np.array_equal(y, g["y_oracle"])             # regression: bit-exact with the stored run
    assert rel_err(y, g["y_direct"]) <= TOL[s] / 10      # definition: direct-form convolution


def test_zero_added_latency_and_tail_padding(orc):
    """Taps sit in the upper half of the padded block (fftw_convolver.cpp:491), so block t of
    the output is the convolution up to and including block t of the input."""
    L, B, C = 64, 3, 1
    h = np.zeros(150, np.float64); h[0] = 1.0; h[149] = -0.5
    x = np.zeros((6 * L, 1)); x[10, 0] = 1.0
    eng = orc.Engine(L, B, 8, C); eng.set_coeff([h])
    _, y = eng.run(x)
    assert abs(y[10, 0] - 1.0) < 1e-14 and abs(y[159, 0] + 0.5) < 1e-14
    y[10, 0] = 0; y[159, 0] = 0
    assert np.abs(y).max() < 1e-14


def test_overflow_bookkeeping(orc):
    real = np.array([0.5, -1.5, 1.0, 2.0, np.nan, -1.0], np.float32)
    raw = np.zeros((6, 2), np.float32)
    of = orc.Overflow(); of.max = 1.0
    orc.real2raw(real, raw, 1, of)
    assert of.n_overflows == 2 and of.largest == 2.0        # strict compares; NaN never counts
    assert np.array_equal(raw[:4, 1], real[:4]) and np.all(raw[:, 0] == 0)


def test_reset_does_not_clear_time_history(orc):
    L, B, C = 32, 2, 1
    rng = np.random.default_rng(4)
    h = [rng.standard_normal(40)]
    x = rng.standard_normal((3 * L, 1))
    a = orc.Engine(L, B, 8, C); a.set_coeff(h)
    a.run(x[:2 * L]); a.reset()         # two calls: input_timecbuf[n][0] now holds block 1
    _, y1 = a.run(x[2 * L:])
    fresh = orc.Engine(L, B, 8, C); fresh.set_coeff(h)
    _, y2 = fresh.run(x[2 * L:])
    assert not np.allclose(y1, y2)      # the stale half block still feeds the first FFT
    b = orc.Engine(L, B, 8, C); b.set_coeff(h)
    b.run(x[:L]); b.reset()             # one call: buffer 0 was never written
    assert np.array_equal(b.run(x[2 * L:])[1], y2)


def test_engine_argument_checks(orc):
    for args in [(100, 2, 4, 2), (64, 2, 5, 2), (64, 2, 4, 9), (64, 0, 4, 2)]:
        with pytest.raises(ValueError):
            orc.Engine(*args)


def test_sampled_reference_equals_the_long_run(orc):
    """oracle.sampled_reference (used for launches too long to run whole on the CPU): a fresh engine
    fed the B+1 blocks ending at block g reproduces block g of the long run bit for bit."""
    L, B, C, nb = 64, 5, 3, 23
    rng = np.random.default_rng(12)
    for s in (4, 8):
        dt = orc.real_dtype(s)
        h = orc.synth_ir(rng, C, B * L - 9, dt)
        x = orc.synth_audio(rng, nb * L, C, dt)
        eng = orc.Engine(L, B, s, C); eng.set_coeff(h)
        _, y = eng.run(x)
        pts = [0, 1, B - 1, B, B + 1, 17, nb - 1]
        ref = orc.sampled_reference(h, lambda g: x[g * L:(g + 1) * L], pts, L, B, s, C)
        for g in pts:
            assert np.array_equal(ref[g], y[g * L:(g + 1) * L])
