"""The oracle's restatement of class dither (brutefir/dither.cpp) -- CPU checks.

PARITY UNPINNED: the reference holds no vectors for its dither.  What is checked here is what the
source states: the generator's recurrence and seeding (dither.cpp:419-449), the table layout
(:62-74, :105-109), the preloop's wrap (:127-139), the map (:73-104) and the quantiser's algebra
(:141-194), plus the statistical properties that make it HP-TPDF dither."""
import numpy as np

from conftest import rel_err  # noqa: F401  (shared fixtures)


def _taus_table(n):
    """Independent python restatement of tausinit(state, 0) + tausrand (32-bit arithmetic by masking)."""
    M = 0xFFFFFFFF

    def step(s, a, b, c, d):
        return (((s & c) << d) & M) ^ ((((s << a) & M) ^ s) >> b)
    s0 = (69069 * 1) & M; s1 = (69069 * s0) & M; s2 = (69069 * s1) & M
    out = []
    for i in range(6 + n):
        s0 = step(s0, 13, 19, 4294967294, 12); s1 = step(s1, 2, 25, 4294967288, 4); s2 = step(s2, 3, 11, 4294967280, 17)
        if i >= 6:
            out.append((s0 ^ s1 ^ s2) & 0xFF)
    return np.array(out, dtype=np.uint8).view(np.int8)


def test_table_is_the_tausworthe_stream_and_channels_are_spaced(orc):
    d = orc.Dither(3, 500, 4, 0, 256)           # spacing = 10 * 500
    t = d.table()
    assert t.size == 3 * 5000 + 1
    assert np.array_equal(t[:4096], _taus_table(4096))
    assert [d.randtab_ptr(c) for c in range(3)] == [1, 5001, 10001]
    # an explicit byte budget shrinks the spacing down to max(1 s, one loop) and no further (:44-58)
    assert orc.Dither(2, 500, 4, 1400, 256).table().size == 2 * 700 + 1
    import pytest
    with pytest.raises(ValueError):
        orc.Dither(2, 500, 4, 900, 256)


def test_quantiser_algebra_and_noise_shaping(orc):
    """Error feedback {1, -1}: y - x = e[n-1] - e[n-2] - e[n] with |e| <= 1.5 LSB, so without clipping
    every output is within 4.5 LSB of its input, and the error spectrum is high-pass shaped."""
    n, C = 1 << 15, 1
    rng = np.random.default_rng(5)
    for s in (4, 8):
        dt = orc.real_dtype(s)
        x = (rng.uniform(-20000, 20000, n)).astype(dt)
        d = orc.Dither(C, 44100, s, 0, 1024)
        raw = np.zeros((n, 1), "<i2")
        of = orc.Overflow(); of.max = 32767.0
        for b in range(n // 1024):               # block by block like run(): one preloop per block
            d.real2raw(x[b * 1024:(b + 1) * 1024], raw[b * 1024:(b + 1) * 1024], 0, 2, of)
        err = raw[:, 0].astype(np.float64) - x.astype(np.float64)
        assert np.abs(err).max() <= 4.6 and of.n_overflows == 0   # y - x = e[n-1] - e[n-2] - e[n], |e| <= 1.5
        assert of.intlargest == np.abs(raw).max()
        # high-pass shaped (difference of two uniform bytes, noise transfer 1 - z^-1 + z^-2): little
        # error power at the bottom of the band, most at the top
        E = np.abs(np.fft.rfft(err - err.mean())) ** 2
        assert E[1: E.size // 16].mean() < 0.1 * E[-E.size // 16:].mean()
        # and it is dither: different from plain requantisation, reproducible from a fresh instance
        raw0 = np.zeros((n, 1), "<i2"); of0 = orc.Overflow(); of0.max = 32767.0
        orc.real2raw_fmt(x, raw0, 0, 2, of0)
        assert (raw0 != raw).mean() > 0.3
        d2 = orc.Dither(C, 44100, s, 0, 1024); raw2 = np.zeros((n, 1), "<i2"); of2 = orc.Overflow(); of2.max = 32767.0
        for b in range(n // 1024):
            d2.real2raw(x[b * 1024:(b + 1) * 1024], raw2[b * 1024:(b + 1) * 1024], 0, 2, of2)
        assert np.array_equal(raw2, raw)


def test_preloop_wraps_and_carries_the_last_byte(orc):
    L, srate = 64, 100                           # spacing = max(1000, max(100, 64)) = 1000: wraps every ~15 blocks
    d = orc.Dither(2, srate, 4, 0, L)
    size = d.table().size
    ptrs = []
    x = np.zeros(L, np.float32); raw = np.zeros((L, 2), "<i2"); of = orc.Overflow(); of.max = 32767.0
    for _ in range(70):
        d.real2raw(x, raw, 1, 2, of)             # channel 1 starts at 1001
        ptrs.append(d.randtab_ptr(1))
    assert max(ptrs) < size and min(ptrs) == 1 + L          # wrapped: restarted at 1 and advanced one block
    assert all(b - a == L or b == 1 + L for a, b in zip(ptrs, ptrs[1:]))


def test_clipping_bookkeeping_keeps_the_reference_quirk(orc):
    """On a clip the peak test looks at the undithered sample and stores the dithered one (dither.cpp:163-166)."""
    d = orc.Dither(1, 44100, 8, 0, 16)
    x = np.array([40000.0, -50000.0, 100.0, 32767.4] + [0.0] * 12)
    raw = np.zeros((16, 1), "<i2"); of = orc.Overflow(); of.max = 32767.0
    d.real2raw(x, raw, 0, 2, of)
    assert raw[0, 0] == 32767 and raw[1, 0] == -32768
    assert of.n_overflows >= 2 and 39998.0 < of.largest < 50003.0


def test_engine_with_dither_matches_stage_calls(orc):
    """orc_engine with apply_dither = the float engine's output pushed through Dither.real2raw block by
    block, channel by channel (brutefir.cpp:326-331 order) -- power-of-two output scales are exact."""
    L, B, C, nb, srate = 128, 3, 3, 40, 60      # spacing = max(600, 128): table wraps several times
    rng = np.random.default_rng(2)
    for s, fmt in ((4, 2), (8, 4), (4, 7)):
        dt = orc.real_dtype(s)
        h = orc.synth_ir(rng, C, B * L - 5, dt)
        x = orc.synth_audio(rng, nb * L, C, dt)
        ef = orc.Engine(L, B, s, C, None, None, srate, False); ef.set_coeff(h); _, yf = ef.run(x)
        ed = orc.Engine(L, B, s, C, None, fmt, srate, True); ed.set_coeff(h); rc, yd = ed.run(x)
        assert rc == 0
        d = orc.Dither(C, srate, s, 0, L)
        exp = orc.raw_frames(fmt, nb * L, C)
        ofs = [orc.Overflow() for _ in range(C)]
        for o in ofs:
            o.max = orc.lib().orc_fmt_max(fmt)
        scale = orc.lib().orc_fmt_out_scale(fmt)
        for t in range(nb):
            for c in range(C):
                d.real2raw((yf[t * L:(t + 1) * L, c] * dt(scale)).astype(dt), exp[t * L:(t + 1) * L], c, fmt, ofs[c])
        assert np.array_equal(exp, yd)
        for c in range(C):
            o = ed.overflow(c)
            assert (o.n_overflows, o.intlargest, o.largest) == (ofs[c].n_overflows, ofs[c].intlargest, ofs[c].largest)
