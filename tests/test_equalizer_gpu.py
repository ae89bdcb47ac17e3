"""FFT plans beyond one workgroup's LDS (four-step), the equalizer render
and its WAV cache file, against the oracle / scipy."""
import os

import numpy as np
import pytest
import scipy.fft

from conftest import TOL, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("s", [4, 8])
@pytest.mark.parametrize("order", [1, 2, 3, 4, 5, 6, 12, 13, 14, 16, 18])   # below 5: direct sums
def test_fft_plans_any_size(orc, bfir, s, order):
    n = 1 << order
    dt = orc.real_dtype(s)
    x = np.random.default_rng(order).standard_normal(n).astype(dt)
    fwd, inv = bfir.FftPlan(order, False, s), bfir.FftPlan(order, True, s)
    hc = fwd.execute(x)
    X = scipy.fft.rfft(x.astype(np.float64))
    want = np.empty(n); want[:n // 2 + 1] = X.real; want[n // 2 + 1:] = X.imag[1:n // 2][::-1]
    assert rel_err(hc, want) <= TOL[s]
    assert rel_err(inv.execute(hc) / n, x) <= TOL[s]        # FFTW_HC2R is the unnormalised inverse
    if 2 <= order <= 16:
        assert rel_err(hc, orc.r2hc(x)) <= TOL[s]


@pytest.mark.parametrize("s", [4, 8])
def test_equalizer_render_matches_oracle(orc, bfir, s, tmp_path):
    srate, L, blocks, C = 44100, 1024, 64, 2            # the plug-in's EQ: 65536-point HC2R (common.h:18-19)
    freq = [31.5, 63, 125, 250, 500, 1000, 2000, 4000, 8000, 16000]
    mag = [3.0, -2.0, 0.0, 1.5, -6.0, 4.0, 0.5, -1.0, 2.0, -3.0]
    phase = [0.0, 10.0, -20.0, 0.0, 5.0, 0.0, -15.0, 30.0, 0.0, 0.0]
    eq = bfir.Equalizer(L, blocks, s, C, srate)
    ir = eq.generate(freq, mag, phase, cache_dir=str(tmp_path))
    of, om, op = orc.equalizer_bands(srate, freq, mag, phase)
    assert np.allclose(of, eq.freq, rtol=0, atol=0) and np.allclose(om, eq.mag, rtol=1e-15) and np.allclose(op, eq.phase, rtol=1e-15)
    want = orc.equalizer_render(L * blocks, of, om, op, s)
    assert ir.shape == (L * blocks // 2, C) and np.array_equal(ir[:, 0], ir[:, 1])
    # Not the hot path's 1e-5 / 1e-12: those are stated for L <= 16384-sample blocks of audio.  Here the
    # spectrum is cos/sin(rad) with |rad| up to taps*pi/2 ~ 1e5: the reference forms `rad` in float
    # (equalizer.cpp:239-259), whose spacing at 1e5 is 2^-7, and the GPU's and the host's single-precision
    # cos/sin of such an argument agree to ~1e-5 absolute at best (argument reduction), which is the
    # spectrum's relative error before any transform.  fp64: a 65536-point transform of O(1) data, ~log2(N) eps.
    assert rel_err(ir[:, 0], want) <= (1e-4 if s == 4 else 1e-10)
    # the cache file: name scheme of make_filename, readable back, re-used on the next call
    files = os.listdir(tmp_path)
    assert len(files) == 1 and files[0].startswith("eq-") and files[0].endswith("-%d-%d-%d-%d.wav" % (L * blocks // 2, s, C, srate))
    from foo_dsp_bfir_amd import wavio
    back, rate = wavio.read_wav(os.path.join(tmp_path, files[0]))
    assert rate == srate and np.array_equal(back, ir)
    again = bfir.Equalizer(L, blocks, s, C, srate).generate(freq, mag, phase, cache_dir=str(tmp_path))
    assert np.array_equal(again, ir)


def test_flat_equalizer_is_a_delayed_impulse(bfir):
    """All bands at 0 dB, zero phase: |H| = 1 with linear phase -pi*n -> an impulse in the middle
    of the taps-sample HC2R output, i.e. at sample 0 of the upper half."""
    eq = bfir.Equalizer(1024, 8, 8, 1, 48000)
    ir = eq.generate([1000.0], [0.0], [0.0])[:, 0]
    assert abs(ir[0] - 1.0) < 1e-9 and np.abs(ir[1:]).max() < 1e-9
