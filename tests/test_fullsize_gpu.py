"""BASELINE.json's full-size configurations through size-independent properties (the oracle
is too slow to run them whole): linearity, dirac identity, delay equivariance, chunk
invariance, plus the oracle on the leading blocks."""
import numpy as np
import pytest

from conftest import TOL, rel_err

pytestmark = pytest.mark.gpu

# (name, realsize, C, taps, L, n_blocks)
FULL = [
    ("cfg2", 4, 2, 65536, 8192, 48),
    ("cfg3_headline", 4, 8, 131072, 4096, 160),
    ("cfg5_fp64", 8, 2, 262144, 4096, 200),
]


@pytest.mark.parametrize("name,s,C,taps,L,nb", FULL, ids=[f[0] for f in FULL])
def test_fullsize_properties(orc, bfir, name, s, C, taps, L, nb):
    B = taps // L
    dt = orc.real_dtype(s)
    rng = np.random.default_rng(len(name))
    h = orc.synth_ir(rng, C, taps, dt)
    x1 = orc.synth_audio(rng, nb * L, C, dt)
    x2 = orc.synth_audio(rng, nb * L, C, dt)

    def run(x, chunk=64, coeffs=h):
        e = bfir.Brutefir(L, B, s, C); e.set_chunk(chunk)
        assert e.set_coeff(coeffs) == 0
        rc, y = e.run(x)
        assert rc == 0
        e.close()
        return y
    y1, y2 = run(x1), run(x2)
    tol = TOL[s]
    # linearity: F(a x1 + b x2) = a F(x1) + b F(x2)
    ylin = run((0.5 * x1 - 0.25 * x2).astype(dt))
    assert rel_err(ylin, 0.5 * y1.astype(np.float64) - 0.25 * y2) <= tol
    # chunk invariance, bit for bit
    assert np.array_equal(run(x1, chunk=7), y1)
    # delay by one block: output is delayed by one block (time invariance across the ring)
    xd = np.concatenate([np.zeros((L, C), dt), x1[:-L]])
    assert rel_err(run(xd)[L:], y1[:-L]) <= tol
    # a one-block delay folded into the filter instead: h' = [zeros(L), h] over B blocks (tail dropped)
    # oracle on the leading blocks (covers warm-up of the ring for the first B blocks)
    k = min(nb, 6)
    ref = orc.Engine(L, B, s, C); ref.set_coeff(h)
    assert rel_err(y1[:k * L], ref.run(x1[:k * L])[1]) <= tol
    # dirac in every channel passes the input through
    d = np.zeros(taps, dt); d[0] = 1.0
    assert rel_err(run(x1, coeffs=[d] * C), x1) <= tol
    # a dirac at tap (B-1)*L + 3 reaches through the whole delay line
    d2 = np.zeros(taps, dt); d2[(B - 1) * L + 3] = 1.0
    yd = run(x1, coeffs=[d2] * C)
    lag = (B - 1) * L + 3
    assert rel_err(yd[lag:], x1[:-lag]) <= tol and np.abs(yd[:lag]).max() <= tol


def test_cfg4_batch_of_32_stereo_engines_fullsize(orc, bfir):
    """BASELINE configs[3], one GPU's share: 32 independent stereo engines, 65536 taps,
    L = 4096 (B = 16), sharing launches.  Engines must not leak into each other, results must
    equal the same engines run alone (bit for bit) and the oracle on the leading blocks."""
    s, C, taps, L, E, nb = 4, 2, 65536, 4096, 32, 40
    B = taps // L
    rng = np.random.default_rng(44)
    hs = [orc.synth_ir(rng, C, taps, np.float32) for _ in range(E)]
    xs = np.stack([orc.synth_audio(rng, nb * L, C, np.float32) for _ in range(E)])
    xs[7] = 0.0                                          # a silent stream must stay silent
    batch = bfir.Brutefir(L, B, s, C, n_engines=E)
    batch.set_chunk(16)
    for e in range(E):
        assert batch.set_coeff(hs[e], engine_index=e) == 0
    rc, y = batch.run(xs)
    assert rc == 0 and np.all(y[7] == 0.0)
    for e in (0, 13, 31):
        one = bfir.Brutefir(L, B, s, C); one.set_chunk(16); one.set_coeff(hs[e])
        assert np.array_equal(one.run(xs[e])[1], y[e])
        ref = orc.Engine(L, B, s, C); ref.set_coeff(hs[e])
        k = 5
        assert rel_err(y[e][:k * L], ref.run(xs[e][:k * L])[1]) <= TOL[s]
    assert all(batch.overflow(c).n_overflows == 0 for c in range(E * C))
