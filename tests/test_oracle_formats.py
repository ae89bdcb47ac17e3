"""CPU tests of the oracle's sample-format restatement (raw2real / real2raw for all eleven
formats, brutefir/raw2real.cpp, real2raw.cpp, dither.cpp:196-262) against numpy."""
import numpy as np
import pytest


@pytest.mark.parametrize("fmt", range(1, 12))
def test_raw2real_matches_numpy_decoding(orc, fmt):
    rng = np.random.default_rng(fmt)
    frames, C = 50, 3
    if fmt >= 8:
        vals = rng.uniform(-1, 1, (frames, C))
        raw = vals.astype(orc.FMT_DTYPES[fmt])
        want = raw.astype(np.float64)
    else:
        bits = 8 * orc.FMT_BYTES[fmt]
        want = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), (frames, C), dtype=np.int64)
        raw = orc.encode_ints(want, fmt)
        assert np.array_equal(orc.decode_ints(raw, fmt), want)
    for ch in range(C):
        got = orc.raw2real_fmt(raw, ch, fmt, 8)
        assert np.array_equal(got, want[:, ch].astype(np.float64))


@pytest.mark.parametrize("fmt", [1, 2, 3, 4, 5])
def test_real2raw_integer_requantiser(orc, fmt):
    """dither{f,d}_real2int_no_dither: floor(x + 0.5), except exact negative integers of x + 0.5
    come out one lower (dither.cpp:220-223), clipping counted in n_overflows / largest."""
    bits = 8 * orc.FMT_BYTES[fmt]
    imin, imax = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    real = np.array([0.0, 0.49, 0.5, 1.5, -0.49, -0.5, -0.51, -1.5, -2.5, imax + 0.4, imax + 0.6, imin - 0.6,
                     imin - 0.4, imin + 0.3], np.float64)
    raw = orc.raw_frames(fmt, real.size, 1)
    of = orc.Overflow(); of.max = float(imax)
    orc.real2raw_fmt(real, raw, 0, fmt, of)
    got = orc.decode_ints(raw, fmt)[:, 0]
    want = []
    for v in real + 0.5:
        if v < 0:
            want.append(imin if v <= imin else int(v) - 1)     # int() truncates toward zero
        else:
            want.append(imax if v > imax else int(v))
    assert list(got) == want
    # -0.5 -> 0; -1.5 and -2.5 hit exact negative integers after the +0.5 and land one lower
    assert got[5] == 0 and got[7] == -2 and got[8] == -3
    assert of.n_overflows == sum(1 for v in real + 0.5 if (v < 0 and v <= imin) or (v >= 0 and v > imax))


def test_full_scale_is_negative_for_32_bit_like_the_reference(orc):
    """get_full_scale does (double)(1 << 31) in int arithmetic (brutefir.cpp:395-398)."""
    L = orc.lib()
    assert L.orc_fmt_out_scale(6) == -2147483648.0 and L.orc_fmt_in_scale(6) == -1.0 / 2147483648.0
    assert L.orc_fmt_out_scale(2) == 32768.0 and L.orc_fmt_max(2) == 32767.0 and L.orc_fmt_max(8) == 1.0
