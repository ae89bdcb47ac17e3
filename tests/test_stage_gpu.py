"""Stage-level entry points (bfir_convolver_*, one per fftw_convolver method) against the
oracle's restatement of the same reference loop, and a block processed through the
reference's own per-stage call sequence (brutefir.cpp:252-334) against the fused engine."""
import numpy as np
import pytest

from conftest import TOL, rel_err

pytestmark = pytest.mark.gpu


def _conv(bfir, L, s):
    return bfir.FftwConvolver(L, s)


@pytest.mark.parametrize("s", [4, 8])
@pytest.mark.parametrize("L", [16, 1024, 4096])
def test_time2freq_freq2time(orc, bfir, s, L):
    dt = orc.real_dtype(s)
    rng = np.random.default_rng(L)
    cv = _conv(bfir, L, s)
    assert cv.convolver_cbufsize() == 2 * L * s
    x = rng.standard_normal(2 * L).astype(dt)
    hc = cv.new_cbuf(); cv.convolver_time2freq(x, hc)
    assert rel_err(hc, orc.r2hc(x)) <= TOL[s]
    back = cv.new_cbuf(); cv.convolver_freq2time(hc, back)
    assert rel_err(back / (2 * L), x) <= TOL[s]
    assert rel_err(back, orc.hc2r(hc)) <= TOL[s]


@pytest.mark.parametrize("s", [4, 8])
def test_mixnscale_is_bit_exact(orc, bfir, s):
    L = 512
    dt = orc.real_dtype(s)
    cv = _conv(bfir, L, s)
    hc = np.random.default_rng(0).standard_normal(2 * L).astype(dt)
    g = cv.new_cbuf(); cv.convolver_mixnscale([hc], g, [0.37], 1, bfir.MIXMODE_INPUT)
    assert np.array_equal(g, orc.mixnscale(hc, 0.37, orc.MIXMODE_INPUT))
    o = cv.new_cbuf(); cv.convolver_mixnscale([g], o, [1.7], 1, bfir.MIXMODE_OUTPUT)
    assert np.array_equal(o, orc.mixnscale(g, 1.7, orc.MIXMODE_OUTPUT))


@pytest.mark.parametrize("s", [4, 8])
def test_convolve_family_is_bit_exact(orc, bfir, s):
    """The stage kernels keep the reference's operation order (separate multiply and add)."""
    L = 2048
    dt = orc.real_dtype(s)
    rng = np.random.default_rng(1)
    b, c, d0 = (rng.standard_normal(2 * L).astype(dt) for _ in range(3))
    cv = _conv(bfir, L, s)
    d = cv.new_cbuf(); cv.convolver_convolve(b, c, d)
    assert np.array_equal(d, orc.convolve(b, c))
    da = d0.copy(); cv.convolver_convolve_add(b, c, da)
    assert np.array_equal(da, orc.convolve_add(b, c, d0))
    bi = b.copy(); cv.convolver_convolve_inplace(bi, c)
    assert np.array_equal(bi, orc.convolve_inplace(b, c))


@pytest.mark.parametrize("s,fmt", [(4, 8), (8, 8), (4, 10), (8, 10)])
def test_raw2cbuf_cbuf2raw_bit_exact(orc, bfir, s, fmt):
    from foo_dsp_bfir_amd.convolver import make_buffer_format
    L, C, ch = 256, 3, 1
    dt = orc.real_dtype(s)
    rng = np.random.default_rng(2)
    raw = (rng.uniform(-2, 2, (L, C))).astype(orc.fmt_dtype(fmt))
    cv = _conv(bfir, L, s)
    bf = make_buffer_format(fmt, ch, C)
    cbuf, nxt = cv.new_cbuf(), cv.new_cbuf()
    cbuf[:L] = 7.0
    cv.convolver_raw2cbuf(raw, cbuf, nxt, bf)
    want = raw[:, ch].astype(dt)
    assert np.array_equal(nxt[:L], want) and np.array_equal(cbuf[L:], want) and np.all(cbuf[:L] == 7.0)
    out = np.full((L, C), 9.0, dtype=orc.fmt_dtype(fmt))
    of = bfir.Overflow(); of.max = 1.0
    cv.convolver_cbuf2raw(np.r_[cbuf[L:], cbuf[L:]], out, bf, of)   # reads the first L samples
    ref_out = np.full((L, C), 9.0, dtype=orc.fmt_dtype(fmt))
    rof = orc.Overflow(); rof.max = 1.0
    orc.real2raw(want, ref_out, ch, rof)
    assert np.array_equal(out, ref_out)
    assert of.n_overflows == rof.n_overflows > 0 and of.largest == rof.largest


@pytest.mark.parametrize("s", [4, 8])
def test_coeffs2cbuf(orc, bfir, s):
    L = 1024
    dt = orc.real_dtype(s)
    taps = np.random.default_rng(3).standard_normal(700).astype(dt)
    cv = _conv(bfir, L, s)
    got = cv.convolver_coeffs2cbuf(taps, taps.size, 0.25)
    assert rel_err(got, orc.coeffs2cbuf(taps, L, 0.25)) <= TOL[s]
    taps[5] = np.inf
    assert cv.convolver_coeffs2cbuf(taps, taps.size, 1.0) is None


@pytest.mark.parametrize("s", [4, 8])
def test_reference_call_sequence_equals_fused_engine(orc, bfir, s):
    """Drive the facade exactly as brutefir::run does and compare with bfir_engine_run."""
    from foo_dsp_bfir_amd.convolver import make_buffer_format
    L, B, C, nb = 256, 3, 2, 7
    fmt = 8 if s == 4 else 10
    dt = orc.real_dtype(s)
    rng = np.random.default_rng(5)
    h = orc.synth_ir(rng, C, B * L - 30, dt)
    x = orc.synth_audio(rng, nb * L, C, dt)
    cv = _conv(bfir, L, s)
    # coeff::preprocess_coeff (coeff.cpp:292-354)
    coeffs = [[cv.convolver_coeffs2cbuf(h[n][b * L:(b + 1) * L], min(L, max(0, h[n].size - b * L)), 1.0)
               for b in range(B)] for n in range(C)]
    fdl = [[cv.new_cbuf() for _ in range(B)] for _ in range(C)]
    ocbuf = [cv.new_cbuf() for _ in range(C)]
    tbuf = [[cv.new_cbuf(), cv.new_cbuf()] for _ in range(C)]
    ifreq, ofreq, tout = cv.new_cbuf(), cv.new_cbuf(), cv.new_cbuf()
    ofl = [bfir.Overflow() for _ in range(C)]
    for o in ofl:
        o.max = 1.0
    y = np.zeros_like(x)
    cur = 0
    for t in range(nb):
        inb, outb = x[t * L:(t + 1) * L], y[t * L:(t + 1) * L]
        for n in range(C):
            cv.convolver_raw2cbuf(inb, tbuf[n][cur], tbuf[n][1 - cur], make_buffer_format(fmt, n, C))
            cv.convolver_time2freq(tbuf[n][cur], ifreq)
            slot = t % B
            cv.convolver_mixnscale([ifreq], fdl[n][slot], [1.0], 1, bfir.MIXMODE_INPUT)
            cv.convolver_convolve(fdl[n][slot], coeffs[n][0], ocbuf[n])
            for i in range(1, min(B, t + 1)):
                cv.convolver_convolve_add(fdl[n][(t - i) % B], coeffs[n][i], ocbuf[n])
            cv.convolver_mixnscale([ocbuf[n]], ofreq, [1.0], 1, bfir.MIXMODE_OUTPUT)
            cv.convolver_freq2time(ofreq, tout)
            cv.convolver_cbuf2raw(tout, outb, make_buffer_format(fmt, n, C), ofl[n])
        cur = 1 - cur
    eng = bfir.Brutefir(L, B, s, C); eng.set_coeff(h)
    rc, y_eng = eng.run(x)
    ref = orc.Engine(L, B, s, C); ref.set_coeff(h); _, y_ref = ref.run(x)
    assert rc == 0
    assert rel_err(y, y_ref) <= TOL[s] and rel_err(y_eng, y_ref) <= TOL[s]
    assert rel_err(y, y_eng) <= TOL[s]
