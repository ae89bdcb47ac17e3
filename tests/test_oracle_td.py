"""Oracle restatement of the class's last public methods (fftw_convolver.hpp:139-166): convolve_inplace_ordered,
convolver_td_block_length / _td_new / _td_convolve and convolver_debug_dump_cbuf, pinned on the CPU by numpy's FFT,
by the defining sums and by the way delay.cpp:150-180 uses the td convolver.  (No reference vectors exist for
these either: parity unpinned, as for the rest of the path.)"""
import re

import numpy as np
import pytest

from conftest import TOL, rel_err


def test_td_block_length_is_log2_roof(orc):
    # brutefir/log2.h:33-51: the next power of two; one tap has no defined answer there (-1 here)
    for n, want in ((0, -1), (-3, -1), (1, -1), (2, 2), (3, 4), (4, 4), (5, 8), (31, 32), (32, 32), (33, 64),
                    (1000, 1024), (65537, 131072)):
        assert orc.td_block_length(n) == want


@pytest.mark.parametrize("s", [4, 8])
def test_convolve_inplace_ordered_is_the_complex_product(orc, s):
    dt = orc.real_dtype(s)
    rng = np.random.default_rng(s)
    for n in (4, 8, 64, 1024):
        b, c = rng.standard_normal(n).astype(dt), rng.standard_normal(n).astype(dt)
        got = orc.convolve_inplace_ordered(b, c)
        # half-complex order: r0 .. r_{n/2}, i_{n/2-1} .. i_1
        cplx = lambda v: np.concatenate(([v[0]], v[1:n // 2] + 1j * v[:n // 2:-1], [v[n // 2]])).astype(np.complex128)
        want = cplx(b.astype(np.float64)) * cplx(c.astype(np.float64))
        want[0], want[-1] = float(b[0]) * float(c[0]), float(b[n // 2]) * float(c[n // 2])
        assert rel_err(cplx(got.astype(np.float64)).real, want.real) <= 4 * np.finfo(dt).eps
        assert rel_err(cplx(got.astype(np.float64)).imag[1:-1], want.imag[1:-1]) <= 4 * np.finfo(dt).eps
        assert got[0] == b[0] * c[0] and got[n // 2] == b[n // 2] * c[n // 2]


@pytest.mark.parametrize("s", [4, 8])
@pytest.mark.parametrize("n_taps", [2, 3, 5, 8, 9, 31, 100, 1000])
def test_td_new_and_convolve_against_numpy(orc, s, n_taps):
    dt = orc.real_dtype(s)
    rng = np.random.default_rng(n_taps)
    h = rng.standard_normal(n_taps).astype(dt)
    bl, spec = orc.td_new(h)
    assert bl == orc.td_block_length(n_taps) and spec.size == 2 * bl
    padded = np.zeros(2 * bl)
    padded[bl:bl + n_taps] = h
    ref = np.fft.rfft(padded) / (2 * bl)
    assert rel_err(spec[:bl + 1], ref.real) <= TOL[s]
    x = rng.standard_normal(2 * bl).astype(dt)
    y = orc.td_convolve(spec, x)
    want = np.fft.irfft(np.fft.rfft(x.astype(np.float64)) * np.fft.rfft(padded), 2 * bl)
    assert rel_err(y, want) <= TOL[s]


def test_td_convolver_as_the_delay_class_uses_it(orc):
    """delay.cpp:150-180: [rest | block] in, first half out.  Tap k lies at lag blocklen + k of the circular product,
    so a filter of n taps applied that way IS the linear convolution of the signal with the taps."""
    rng = np.random.default_rng(9)
    h = rng.standard_normal(31)
    bl, spec = orc.td_new(h)
    sig = rng.standard_normal(16 * bl)
    rest, out = np.zeros(bl), np.empty_like(sig)
    for i in range(0, sig.size, bl):
        blk = np.concatenate((rest, sig[i:i + bl]))
        rest = blk[bl:].copy()
        out[i:i + bl] = orc.td_convolve(spec, blk)[:bl]
    assert rel_err(out, np.convolve(sig, h)[:sig.size]) <= 1e-13


@pytest.mark.parametrize("s", [4, 8])
def test_debug_dump_gives_back_the_taps(orc, s, tmp_path):
    """coeffs2cbuf folds 1/n_fft into the spectrum, the dump's unnormalised HC2R takes it out again: the file
    lists scale * taps, one "%.16e" line each (fftw_convolver.cpp:604-651)."""
    dt = orc.real_dtype(s)
    L = 256
    rng = np.random.default_rng(3)
    taps = [rng.standard_normal(L).astype(dt), rng.standard_normal(100).astype(dt)]
    cbufs = [orc.coeffs2cbuf(taps[0], L, 0.5), orc.coeffs2cbuf(taps[1], L, 1.0)]
    vals = orc.debug_dump_values(cbufs[0])
    assert rel_err(vals, 0.5 * taps[0].astype(np.float64)) <= TOL[s]
    path = tmp_path / "dump.txt"
    assert orc.debug_dump_cbuf(path, cbufs) == 0
    lines = path.read_text().split("\n")
    assert lines[-1] == "" and len(lines) == 2 * L + 1
    assert all(re.fullmatch(r"-?\d\.\d{16}e[+-]\d{2,3}", ln) for ln in lines[:-1])
    got = np.array([float(ln) for ln in lines[:-1]])
    assert np.array_equal(got[:L].astype(dt), vals)                     # 17 digits round-trip a float / double
    want1 = np.concatenate((taps[1], np.zeros(L - 100))).astype(np.float64)
    assert rel_err(got[L:], want1) <= TOL[s]
    assert orc.debug_dump_cbuf(tmp_path / "no-such-dir" / "x.txt", cbufs) == -1
