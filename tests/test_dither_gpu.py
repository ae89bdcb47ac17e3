"""HP-TPDF dither on integer outputs (brutefir/dither.cpp, real2raw.cpp:38-317).

Integer work, so the bar is BIT-EXACT with the oracle:
  * the random table of the product equals the oracle's (and a python restatement of the generator);
  * stage level: convolver_cbuf2raw(apply_dither) on the same working-precision block gives the same
    integers, the same dither_state and the same overflow bookkeeping, for S8/S16/S24/S32 LE/BE,
    fp32 and fp64, over many consecutive blocks (the error feedback carries over) and table wraps;
  * engine level: the dithering engine equals the oracle's dither applied to the output of the SAME
    engine run with float output (power-of-two output scales are exact in floating point, so the
    samples entering the quantiser are identical; BFIR_PAIR=0 keeps both on the same kernels), for
    one call, many calls, chunked launches and a batch of engines."""
import os

import numpy as np
import pytest

from conftest import env_override

from test_oracle_dither import _taus_table

pytestmark = pytest.mark.gpu

INT_FMTS = [1, 2, 3, 4, 5, 6, 7]


def test_random_table_matches_oracle_and_generator(orc, bfir):
    d = bfir.Dither(3, 500, 4, 0, 256)
    t = d.table()
    assert np.array_equal(t, orc.Dither(3, 500, 4, 0, 256).table())
    assert np.array_equal(t[:2048], _taus_table(2048))
    assert [d.states[c].randtab_ptr for c in range(3)] == [1, 5001, 10001]
    with pytest.raises(bfir.BfirError):
        bfir.Dither(2, 500, 4, 900, 256)                     # budget below n_channels * max(1 s, one loop)


@pytest.mark.parametrize("fmt", INT_FMTS)
@pytest.mark.parametrize("s", [4, 8])
def test_cbuf2raw_with_dither_bit_exact(orc, bfir, fmt, s):
    from foo_dsp_bfir_amd.convolver import make_buffer_format
    L, C, ch, nblk, srate = 128, 3, 1, 24, 40               # spacing 400 -> table of 1201 bytes: wraps often
    dt = orc.real_dtype(s)
    rng = np.random.default_rng(7 * fmt + s)
    full = float(1 << (8 * orc.FMT_BYTES[fmt] - 1))
    dth = bfir.Dither(C, srate, s, 0, L)
    ref = orc.Dither(C, srate, s, 0, L)
    cv = bfir.FftwConvolver(L, s, dth)
    of = bfir.Overflow(); of.max = orc.lib().orc_fmt_max(fmt)
    rof = orc.Overflow(); rof.max = of.max
    for b in range(nblk):
        amp = 1.3 if b % 5 == 4 else 0.7                     # every fifth block clips
        real = (rng.uniform(-amp, amp, L) * min(full, 2.0 ** 30)).astype(dt)
        if b == 0:
            real[:8] = np.array([-3.0, -2.5, -0.5, -0.49, 0.49, 0.5, 2.5, 3.0], dt)
        out = orc.encode_ints(rng.integers(-100, 100, (L, C)), fmt)
        out = np.ascontiguousarray(out); want = out.copy()
        cv.convolver_cbuf2raw(np.r_[real, real], out, make_buffer_format(fmt, ch, C), of, True, dth.states[ch])
        ref.real2raw(real, want, ch, fmt, rof)
        assert np.array_equal(out, want), (b, np.flatnonzero(orc.decode_ints(out, fmt)[:, ch] != orc.decode_ints(want, fmt)[:, ch])[:5])
        assert dth.states[ch].randtab_ptr == ref.randtab_ptr(ch)
        assert (of.n_overflows, of.intlargest, of.largest) == (rof.n_overflows, rof.intlargest, rof.largest)
    assert of.n_overflows > 0 or orc.FMT_BYTES[fmt] == 4    # the 32-bit test amplitude stays below full scale


def test_cbuf2raw_dither_needs_the_instance(bfir):
    from foo_dsp_bfir_amd.convolver import make_buffer_format
    cv = bfir.FftwConvolver(64, 4)
    of = bfir.Overflow(); of.max = 32767.0
    st = bfir.Dither(1, 100, 4, 0, 64).states[0]
    with pytest.raises(bfir.BfirError):                      # "Dither instance not set." (fftw_convolver.cpp:412-416)
        cv.convolver_cbuf2raw(np.zeros(128, np.float32), np.zeros((64, 1), "<i2"), make_buffer_format(2, 0, 1), of, True, st)
    # a dither built for the other precision would read the convolver's samples as the wrong type: refused
    # (the reference builds both with one realsize, brutefir.cpp:709-719)
    d8 = bfir.Dither(1, 100, 8, 0, 64)
    cv._dither = d8
    with pytest.raises(bfir.BfirError):
        cv.convolver_cbuf2raw(np.zeros(128, np.float32), np.zeros((64, 1), "<i2"), make_buffer_format(2, 0, 1), of, True,
                              d8.states[0])


def _planar(bfir, *a, **kw):
    with env_override(BFIR_PAIR="0"):
        return bfir.Brutefir(*a, **kw)


class _OracleDither:
    """The oracle's dither applied block by block, channel by channel (brutefir.cpp:252-334 order) to the
    float output of the same engine; keeps its state between calls like the engine's m_dither."""

    def __init__(self, orc, fmt, s, L, C, srate):
        self.orc, self.fmt, self.s, self.L, self.C = orc, fmt, s, L, C
        self.d = orc.Dither(C, srate, s, 0, L)
        self.of = [orc.Overflow() for _ in range(C)]
        for o in self.of:
            o.max = orc.lib().orc_fmt_max(fmt)

    def reset_counters(self):                                # brutefir::reset, brutefir.cpp:346-367
        for o in self.of:
            o.n_overflows, o.largest, o.intlargest = 0, 0.0, 0

    def apply(self, yf):
        orc, L = self.orc, self.L
        dt = orc.real_dtype(self.s)
        scale = dt(orc.lib().orc_fmt_out_scale(self.fmt))
        nb = yf.shape[0] // L
        exp = orc.raw_frames(self.fmt, nb * L, self.C)
        for t in range(nb):
            for c in range(self.C):
                self.d.real2raw((yf[t * L:(t + 1) * L, c] * scale).astype(dt), exp[t * L:(t + 1) * L], c, self.fmt, self.of[c])
        return exp


@pytest.mark.parametrize("s,fmt,C,chunk", [(4, 2, 2, 3), (4, 4, 3, 64), (8, 5, 2, 5), (8, 7, 4, 1), (4, 3, 8, 7), (4, 6, 1, 2)])
def test_engine_with_dither_bit_exact(orc, bfir, s, fmt, C, chunk):
    L, B, nb, srate = 256, 3, 31, 150                       # spacing 1500: every channel wraps a few times
    rng = np.random.default_rng(fmt + C)
    dt = orc.real_dtype(s)
    h = orc.synth_ir(rng, C, B * L - 9, dt)
    x = orc.synth_audio(rng, nb * L, C, dt)
    in_fmt = 8 if s == 4 else 10
    probe = _planar(bfir, L, B, s, C, in_fmt, in_fmt); probe.set_coeff(h)
    gain = 1.25 / float(np.abs(probe.run(x)[1]).max())      # loud enough to clip now and then
    ef = _planar(bfir, L, B, s, C, in_fmt, in_fmt, sampling_rate=srate); ef.set_chunk(chunk); assert ef.set_coeff(h, scale=gain) == 0
    rc, yf = ef.run(x); assert rc == 0
    ed = bfir.Brutefir(L, B, s, C, in_fmt, fmt, sampling_rate=srate, apply_dither=True); ed.set_chunk(chunk)
    assert ed.set_coeff(h, scale=gain) == 0
    # three calls of different sizes: the dither state carries over like the engine's history
    parts = [ed.run(x[a * L:b * L])[1] for a, b in ((0, 11), (11, 12), (12, nb))]
    yd = np.concatenate(parts)
    od = _OracleDither(orc, fmt, s, L, C, srate)
    assert np.array_equal(yd, od.apply(yf))

    def same_counters():
        for c in range(C):
            o = ed.overflow(c)
            assert (o.n_overflows, o.intlargest, o.largest) == (od.of[c].n_overflows, od.of[c].intlargest, od.of[c].largest)
    same_counters()
    assert sum(ed.overflow(c).n_overflows for c in range(C)) > 0
    # dither is not plain requantisation
    en = bfir.Brutefir(L, B, s, C, in_fmt, fmt, sampling_rate=srate); en.set_chunk(chunk); en.set_coeff(h, scale=gain)
    if not (s == 4 and orc.FMT_BYTES[fmt] == 4):             # a +-1 LSB dither vanishes in fp32 next to 2^30 (also in the reference)
        assert (orc.decode_ints(en.run(x)[1], fmt) != orc.decode_ints(yd, fmt)).mean() > 0.2
    # reset() touches counters only (brutefir.cpp:346-367): table position and error feedback carry on
    ed.reset(); ef.reset(); od.reset_counters()
    _, yd2 = ed.run(x[:4 * L]); _, yf2 = ef.run(x[:4 * L])
    assert np.array_equal(yd2, od.apply(yf2))
    same_counters()


def test_batch_of_engines_dithers_each_like_a_single_instance(orc, bfir):
    s, fmt, L, B, C, E, nb, srate = 4, 2, 128, 2, 2, 5, 20, 60
    rng = np.random.default_rng(3)
    hs = [orc.synth_ir(rng, C, B * L, np.float32) for _ in range(E)]
    xs = np.stack([orc.synth_audio(rng, nb * L, C, np.float32) for _ in range(E)])
    ef = _planar(bfir, L, B, s, C, 8, 8, sampling_rate=srate, n_engines=E)
    ed = bfir.Brutefir(L, B, s, C, 8, fmt, sampling_rate=srate, apply_dither=True, n_engines=E)
    for e in range(E):
        assert ef.set_coeff(hs[e], engine_index=e) == 0 and ed.set_coeff(hs[e], engine_index=e) == 0
    _, yf = ef.run(xs); rc, yd = ed.run(xs)
    assert rc == 0
    for e in range(E):                                       # every instance has its own m_dither
        assert np.array_equal(yd[e], _OracleDither(orc, fmt, s, L, C, srate).apply(yf[e]))
