"""Methods the reference declares on fftw_convolver but never calls:
N-input mixnscale, dirac_convolve, convolve_eval, crossfade_inplace, runtime_coeffs2cbuf,
verify_cbuf, debug_dump_cbuf and the td convolver of the dead delay class -- against the oracle's
restatement of the same reference lines."""
import ctypes as C
import re

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


@pytest.mark.parametrize("s", [4, 8])
@pytest.mark.parametrize("n_taps", [2, 3, 8, 9, 31, 100, 1000, 5000, 40000])
def test_td_new_and_convolve_match_oracle(orc, bfir, s, n_taps):
    """convolver_td_new / _td_convolve (fftw_convolver.cpp:708-777): block lengths 2 ... 65536, i.e. the direct sums
    (up to 16 reals), the workgroup FFT and the four-step FFT behind the same three calls."""
    dt = orc.real_dtype(s)
    rng = np.random.default_rng(n_taps + s)
    h = rng.standard_normal(n_taps).astype(dt)
    cv = bfir.FftwConvolver(256, s)
    assert cv.convolver_td_block_length(n_taps) == orc.td_block_length(n_taps)
    tdc = cv.convolver_td_new(h, n_taps)
    bl, spec = orc.td_new(h)
    assert tdc.blocklen == bl and tdc.coeffs.dtype == dt and tdc.coeffs.size == 2 * bl
    assert rel_err(tdc.coeffs, spec) <= TOL[s]
    for _ in range(2):
        x = rng.standard_normal(2 * bl).astype(dt)
        y = x.copy()
        cv.convolver_td_convolve(tdc, y)
        assert rel_err(y, orc.td_convolve(spec, x)) <= TOL[s]
    tdc.close()


def test_td_refusals(bfir):
    cv = bfir.FftwConvolver(256, 4)
    for n in (0, -5, 1):          # 0 / negative: the reference's -1 and NULL; 1: undefined there, refused here
        assert cv.convolver_td_block_length(n) == -1
        assert cv.convolver_td_new(np.ones(4, np.float32), n) is None


@pytest.mark.parametrize("s", [4, 8])
def test_td_convolver_as_the_delay_class_uses_it(orc, bfir, s):
    """delay.cpp:150-180: [rest | block] in, first half out, block after block == linear convolution with the taps."""
    dt = orc.real_dtype(s)
    rng = np.random.default_rng(11)
    h = rng.standard_normal(63).astype(dt)
    cv = bfir.FftwConvolver(512, s)
    tdc = cv.convolver_td_new(h, h.size)
    bl = tdc.blocklen
    sig = rng.standard_normal(12 * bl).astype(dt)
    rest, out = np.zeros(bl, dt), np.empty_like(sig)
    for i in range(0, sig.size, bl):
        blk = np.concatenate((rest, sig[i:i + bl]))
        rest = blk[bl:].copy()
        cv.convolver_td_convolve(tdc, blk)
        out[i:i + bl] = blk[:bl]
    want = np.convolve(sig.astype(np.float64), h.astype(np.float64))[:sig.size]
    assert rel_err(out, want) <= TOL[s]


@pytest.mark.parametrize("s", [4, 8])
def test_debug_dump_cbuf_matches_oracle(orc, bfir, s, tmp_path):
    """convolver_debug_dump_cbuf (fftw_convolver.cpp:604-651): same line format and count as the oracle's file, the
    values within the path's tolerance (an inverse transform lies in between), the taps themselves come back."""
    dt = orc.real_dtype(s)
    L = 1024
    rng = np.random.default_rng(5)
    taps = [rng.standard_normal(L).astype(dt), rng.standard_normal(300).astype(dt)]
    cv = bfir.FftwConvolver(L, s)
    cbufs = [cv.convolver_coeffs2cbuf(taps[0], L, 0.25), cv.convolver_coeffs2cbuf(taps[1], 300, 1.0)]
    got_path, want_path = tmp_path / "hip.txt", tmp_path / "orc.txt"
    cv.convolver_debug_dump_cbuf(got_path, cbufs, 2)
    assert orc.debug_dump_cbuf(want_path, cbufs) == 0
    got, want = got_path.read_text().split("\n"), want_path.read_text().split("\n")
    assert len(got) == len(want) == 2 * L + 1 and got[-1] == ""
    assert all(re.fullmatch(r"-?\d\.\d{16}e[+-]\d{2,3}", ln) for ln in got[:-1])
    g, w = np.array([float(v) for v in got[:-1]]), np.array([float(v) for v in want[:-1]])
    assert rel_err(g, w) <= TOL[s]
    assert rel_err(g[:L], 0.25 * taps[0].astype(np.float64)) <= TOL[s]
    assert rel_err(g[L:L + 300], taps[1]) <= TOL[s] and np.abs(g[L + 300:]).max() <= TOL[s] * np.abs(taps[1]).max()
    cv.convolver_debug_dump_cbuf(tmp_path / "no-such-dir" / "x.txt", cbufs, 2)      # logged and skipped, as there
    assert bfir.load().bfir_convolver_debug_dump_cbuf(cv._h, str(tmp_path / "no-such-dir" / "x.txt").encode(),
                                                      (C.c_void_p * 1)(cbufs[0].ctypes.data), 1) == bfir.ERR_IO
