"""All eleven sample formats (brutefir/global.h:24-34) through the staging
kernels.  Stage level: integer and byte work is bit-exact with the oracle.  Engine level:
integer outputs may differ from the oracle's by one LSB where the fp32/fp64 sample lands
within the arithmetic tolerance of a rounding boundary."""
import numpy as np
import pytest

from conftest import TOL, rel_err

pytestmark = pytest.mark.gpu

INT_FMTS = [1, 2, 3, 4, 5, 6, 7]
ALL_FMTS = list(range(1, 12))


def _random_raw(orc, rng, fmt, frames, C):
    if fmt >= 8:
        v = rng.uniform(-1, 1, (frames, C))
        return v.astype(orc.FMT_DTYPES[fmt])
    bits = 8 * orc.FMT_BYTES[fmt]
    v = rng.integers(-(1 << (bits - 1)), (1 << (bits - 1)), (frames, C), dtype=np.int64)
    return orc.encode_ints(v, fmt)


@pytest.mark.parametrize("fmt", ALL_FMTS)
@pytest.mark.parametrize("s", [4, 8])
def test_raw2cbuf_every_format_bit_exact(orc, bfir, fmt, s):
    from foo_dsp_bfir_amd.convolver import make_buffer_format
    L, C, ch = 128, 3, 2
    raw = _random_raw(orc, np.random.default_rng(fmt), fmt, L, C)
    cv = bfir.FftwConvolver(L, s)
    cbuf, nxt = cv.new_cbuf(), cv.new_cbuf()
    cv.convolver_raw2cbuf(raw, cbuf, nxt, make_buffer_format(fmt, ch, C))
    want = orc.raw2real_fmt(raw, ch, fmt, s)
    assert np.array_equal(nxt[:L], want) and np.array_equal(cbuf[L:], want)


@pytest.mark.parametrize("fmt", ALL_FMTS)
@pytest.mark.parametrize("s", [4, 8])
def test_cbuf2raw_every_format_bit_exact(orc, bfir, fmt, s):
    from foo_dsp_bfir_amd.convolver import make_buffer_format
    L, C, ch = 256, 2, 1
    dt = orc.real_dtype(s)
    rng = np.random.default_rng(100 + fmt)
    if fmt >= 8:
        real = rng.uniform(-1.5, 1.5, L).astype(dt)
        fmax = 1.0
    else:
        full = float(1 << (8 * orc.FMT_BYTES[fmt] - 1))
        real = (rng.uniform(-1.2, 1.2, L) * min(full, 2.0 ** 30)).astype(dt)   # some samples clip
        real[:8] = np.array([-3.0, -2.5, -0.5, -0.49, 0.49, 0.5, 2.5, 3.0], dt)   # the rounding quirks
        fmax = orc.lib().orc_fmt_max(fmt)
    cv = bfir.FftwConvolver(L, s)
    out = np.ascontiguousarray(_random_raw(orc, rng, fmt, L, C))   # other channel must survive
    ref = out.copy()
    of = bfir.Overflow(); of.max = fmax
    rof = orc.Overflow(); rof.max = fmax
    cv.convolver_cbuf2raw(np.r_[real, real], out, make_buffer_format(fmt, ch, C), of)
    orc.real2raw_fmt(real, ref, ch, fmt, rof)
    assert np.array_equal(out, ref)
    assert (of.n_overflows, of.intlargest, of.largest) == (rof.n_overflows, rof.intlargest, rof.largest)


@pytest.mark.parametrize("in_fmt,out_fmt,s", [(2, 2, 4), (3, 5, 8), (4, 6, 4), (6, 2, 8), (1, 4, 4),
                                              (9, 11, 8), (7, 9, 4), (2, 8, 8), (8, 3, 4)])
def test_engine_with_integer_and_big_endian_formats(orc, bfir, in_fmt, out_fmt, s):
    L, B, C, nb = 256, 3, 2, 9
    rng = np.random.default_rng(in_fmt * 16 + out_fmt)
    h = orc.synth_ir(rng, C, B * L - 11, orc.real_dtype(s))
    x = _random_raw(orc, rng, in_fmt, nb * L, C)
    ref = orc.Engine(L, B, s, C, in_fmt, out_fmt); ref.set_coeff(h)
    rc_ref, y_ref = ref.run(x)
    eng = bfir.Brutefir(L, B, s, C, in_fmt, out_fmt); eng.set_chunk(4); assert eng.set_coeff(h) == 0
    rc, y = eng.run(x)
    assert rc == rc_ref == 0
    if out_fmt >= 8:
        assert rel_err(y.astype(np.float64), y_ref.astype(np.float64)) <= TOL[s]
    else:
        a, b = orc.decode_ints(y, out_fmt), orc.decode_ints(y_ref, out_fmt)
        full = float(1 << (8 * orc.FMT_BYTES[out_fmt] - 1))
        lsb_tol = max(1, int(np.ceil(TOL[s] * full)))    # 1e-5 of full scale in LSBs, at least one
        assert np.abs(a - b).max() <= lsb_tol
        if lsb_tol == 1:
            assert (a != b).mean() < 0.01
    for c in range(C):
        o, r = eng.overflow(c), ref.overflow(c)
        assert o.max == r.max and o.n_overflows == r.n_overflows == 0
        if out_fmt < 8:
            assert abs(o.intlargest - r.intlargest) <= max(1, int(np.ceil(TOL[s] * abs(r.intlargest))))


def test_dither_flag_is_a_no_op_on_float_output(orc, bfir):
    """apply_dither only acts on integer formats (fftw_convolver.cpp:421, 444); see tests/test_dither_gpu.py."""
    L, B, C, nb = 256, 2, 2, 5
    rng = np.random.default_rng(1)
    h = orc.synth_ir(rng, C, 400, np.float32); x = orc.synth_audio(rng, nb * L, C, np.float32)
    a = bfir.Brutefir(L, B, 4, C, 8, 8, apply_dither=True); a.set_coeff(h)
    b = bfir.Brutefir(L, B, 4, C, 8, 8, apply_dither=False); b.set_coeff(h)
    assert np.array_equal(a.run(x)[1], b.run(x)[1])
