"""The float fast path (csrc/pair.hip): two channels per complex transform on raw frames.

It is selected for FLOAT_LE in and out, fp32 arithmetic, an even channel count and
512 <= L <= 8192; BFIR_PAIR=0 at engine creation keeps the planar staging kernels.  Both must
agree with the oracle (1e-5) and keep the reference's bookkeeping: time history across calls,
chunks and reset (brutefir.cpp:255-260, 346-367), overflow statistics and the NaN verdict
(real2raw.cpp:321-336, brutefir.cpp:316-321)."""
import os

import numpy as np
import pytest

from conftest import TOL, env_override, rel_err

pytestmark = pytest.mark.gpu


def _planar_engine(bfir, *args, **kw):
    with env_override(BFIR_PAIR="0"):
        return bfir.Brutefir(*args, **kw)


def _data(orc, C, taps, frames, seed):
    rng = np.random.default_rng(seed)
    return orc.synth_ir(rng, C, taps, np.float32), orc.synth_audio(rng, frames, C, np.float32)


@pytest.mark.parametrize("L,B,C,nb,chunk", [(512, 3, 2, 9, 4), (1024, 5, 6, 13, 1), (4096, 2, 8, 5, 3), (8192, 2, 2, 4, 2)])
def test_pair_and_planar_paths_agree(orc, bfir, L, B, C, nb, chunk):
    h, x = _data(orc, C, B * L - 11, nb * L, seed=L + C)
    ref = orc.Engine(L, B, 4, C); ref.set_coeff(h); _, y_ref = ref.run(x)
    outs = []
    for make in (bfir.Brutefir, lambda *a: _planar_engine(bfir, *a)):
        eng = make(L, B, 4, C); eng.set_chunk(chunk); assert eng.set_coeff(h) == 0
        rc, y = eng.run(x)
        assert rc == 0 and rel_err(y, y_ref) <= TOL[4]
        outs.append((y, [eng.overflow(c) for c in range(C)]))
        eng.close()
    assert rel_err(outs[0][0], outs[1][0]) <= 2e-6          # two FFT factorizations of the same sums
    for a, b in zip(outs[0][1], outs[1][1]):
        assert a.n_overflows == b.n_overflows and abs(a.largest - b.largest) <= 1e-5 * max(b.largest, 1e-30)


def test_block_by_block_equals_batched_on_the_pair_path(orc, bfir):
    """The plug-in's pattern (foo_dsp_bfir.cpp:311-349): one run() per block; every chunk is one block,
    so the history hand-over of one-block chunks carries the whole run."""
    L, B, C, nb = 512, 4, 4, 11
    h, x = _data(orc, C, 1900, nb * L, seed=8)
    a = bfir.Brutefir(L, B, 4, C); a.set_coeff(h)
    _, y_all = a.run(x)
    b = bfir.Brutefir(L, B, 4, C); b.set_coeff(h)
    parts = [b.run(x[t * L:(t + 1) * L])[1] for t in range(nb)]
    assert np.array_equal(np.concatenate(parts), y_all)
    ref = orc.Engine(L, B, 4, C); ref.set_coeff(h)
    assert rel_err(y_all, ref.run(x)[1]) <= TOL[4]


def test_reset_keeps_time_history_on_the_pair_path(orc, bfir):
    """brutefir::reset clears counters only (brutefir.cpp:346-367): the first block after it still
    sees whatever block input_timecbuf[n][0] holds -- after an odd number of blocks the one before last."""
    L, B, C = 512, 3, 2
    h, x = _data(orc, C, 1300, 10 * L, seed=21)
    ref = orc.Engine(L, B, 4, C); ref.set_coeff(h)
    eng = bfir.Brutefir(L, B, 4, C); eng.set_coeff(h); eng.set_chunk(2)
    for nblk in (3, 2, 1, 1):
        seg = x[:nblk * L]
        _, yr = ref.run(seg); rc, y = eng.run(seg)
        assert rc == 0 and rel_err(y, yr) <= TOL[4]
        ref.reset(); eng.reset()
        x = x[nblk * L:]
    _, yr = ref.run(x); rc, y = eng.run(x)
    assert rc == 0 and rel_err(y, yr) <= TOL[4]


def test_overflow_statistics_and_nan_verdict_on_the_pair_path(orc, bfir):
    L, B, C, nb = 512, 2, 4, 10
    rng = np.random.default_rng(10)   # no |sample| within 1e-5 of the clip threshold
    h = [np.r_[np.float32(g), np.zeros(700, np.float32)] for g in (0.5, 1.5, 3.0, 0.25)]   # pure gains
    x = orc.synth_audio(rng, nb * L, C, np.float32)
    ref = orc.Engine(L, B, 4, C); ref.set_coeff(h); _, y_ref = ref.run(x)
    eng = bfir.Brutefir(L, B, 4, C); eng.set_coeff(h); eng.set_chunk(4); rc, y = eng.run(x)
    assert rc == 0 and rel_err(y, y_ref) <= TOL[4]
    assert np.abs(np.abs(y_ref.astype(np.float64)) - 1.0).min() > 1e-5   # nothing sits on the threshold
    for c in range(C):
        o, r = eng.overflow(c), ref.overflow(c)
        assert o.n_overflows == r.n_overflows
        assert abs(o.largest - r.largest) <= 1e-5 * r.largest
    assert eng.overflow(0).n_overflows == 0 and eng.overflow(2).n_overflows > 0
    # a NaN in block 1: the reference stops there with -1, the batched call reports -1 at the end
    bad = x.copy(); bad[L + 7, 3] = np.nan
    eng2 = bfir.Brutefir(L, B, 4, C); eng2.set_coeff(h)
    ref2 = orc.Engine(L, B, 4, C); ref2.set_coeff(h)
    assert ref2.run(bad)[0] == -1 and eng2.run(bad)[0] == -1


def test_pair_path_needs_8_byte_aligned_frames(bfir):
    """The float fast path moves a stereo frame as one 8-byte access and refuses frames it cannot; the general path
    (BFIR_PAIR=0, one sample per access) takes the same buffer and gives the bits of the aligned run."""
    import torch
    L, B, C = 512, 2, 2
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal(2 * L * C).astype(np.float32))
    buf = torch.zeros(2 * L * C + 2, device="cuda", dtype=torch.float32)
    out = torch.zeros_like(buf)
    got = {}
    for pair in ("1", "0"):
        with env_override(BFIR_PAIR=pair):
            for off in (2, 1):                                  # samples: 8-byte aligned, then 4
                eng = bfir.Brutefir(L, B, 4, C)
                eng.set_coeff([np.ones(10, np.float32)] * C)
                buf.zero_(); buf[off:off + 2 * L * C] = x.cuda(); out.zero_()
                if pair == "1" and off == 1:
                    with pytest.raises(bfir.BfirError):
                        eng.run_device(buf.data_ptr() + 4 * off, out.data_ptr() + 4 * off, 2)
                    continue
                eng.run_device(buf.data_ptr() + 4 * off, out.data_ptr() + 4 * off, 2)
                assert eng.sync() == 0
                got[pair, off] = out[off:off + 2 * L * C].cpu().numpy().copy()
    assert np.array_equal(got["0", 1], got["0", 2])
    assert rel_err(got["0", 2], got["1", 2]) <= TOL[4]


@pytest.mark.parametrize("L,B,C,nb,chunk", [(1024, 40, 2, 50, 16), (512, 70, 3, 80, 32), (1024, 33, 4, 37, 5)])
def test_more_partitions_than_one_register_batch(orc, bfir, L, B, C, nb, chunk):
    """B > 32: the streaming MAC runs in batches of 32 partitions that continue the sums left in Y
    (the plug-in's own shape: FILTER_LEN 1024 and as many partitions as the impulse needs)."""
    h, x = _data(orc, C, B * L - 5, nb * L, seed=B + C)
    ref = orc.Engine(L, B, 4, C); ref.set_coeff(h); _, y_ref = ref.run(x)
    eng = bfir.Brutefir(L, B, 4, C); eng.set_chunk(chunk); assert eng.set_coeff(h) == 0
    rc, y = eng.run(x)
    assert rc == 0 and rel_err(y, y_ref) <= TOL[4]
    # chunking stays bit-exact across the batches
    eng2 = bfir.Brutefir(L, B, 4, C); eng2.set_chunk(nb); eng2.set_coeff(h)
    assert np.array_equal(eng2.run(x)[1], y)
    # and the batched sums are the very sums of the grouped-layout MAC kernels (general path for both)
    with env_override(BFIR_PAIR="0"):
        a = bfir.Brutefir(L, B, 4, C)
        with env_override(BFIR_MAC_VARIANT="8"):
            b = bfir.Brutefir(L, B, 4, C)
    for e in (a, b):
        e.set_chunk(chunk); e.set_coeff(h)
    assert np.array_equal(a.run(x)[1], b.run(x)[1])


def test_automatic_chunk_equals_explicit_chunks(orc, bfir):
    """set_chunk(0) (the default) picks the launch size itself; results do not depend on it."""
    L, B, C, nb = 512, 3, 2, 70
    h, x = _data(orc, C, 1300, nb * L, seed=3)
    outs = []
    for chunk in (0, 7, 64):
        eng = bfir.Brutefir(L, B, 4, C)
        assert eng.set_chunk(chunk) in (0, None)
        eng.set_coeff(h)
        rc, y = eng.run(x)
        assert rc == 0
        outs.append(y)
        eng.close()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    ref = orc.Engine(L, B, 4, C); ref.set_coeff(h)
    assert rel_err(outs[0], ref.run(x)[1]) <= TOL[4]


# ---- pairs in TIME: odd channel counts and single channels on the float fast path (k_fwd_tp_ps / k_inv_tp_ps) -----------
# Blocks t and t + 1 of ONE channel are the real and the imaginary part of a transform, so the fast path no longer
# needs an even channel count.  Which blocks share a transform depends on where a launch starts,
# so results of different launch cuts agree to rounding (the 2e-6 the two FFT factorisations differ by), not bit for bit.
@pytest.mark.parametrize("L,B,C,nb,chunk", [(512, 3, 1, 9, 4), (1024, 5, 3, 13, 1), (4096, 2, 7, 6, 3), (4096, 32, 1, 70, 16),
                                            (8192, 2, 1, 5, 2), (2048, 4, 5, 11, 64), (1024, 6, 3, 12, 5)])
def test_time_paired_path_matches_oracle_and_planar_path(orc, bfir, L, B, C, nb, chunk):
    h, x = _data(orc, C, B * L - 11, nb * L, seed=L + C)
    ref = orc.Engine(L, B, 4, C); ref.set_coeff(h); _, y_ref = ref.run(x)
    outs = []
    for make in (bfir.Brutefir, lambda *a: _planar_engine(bfir, *a)):
        eng = make(L, B, 4, C); eng.set_chunk(chunk); assert eng.set_coeff(h) == 0
        rc, y = eng.run(x)
        assert rc == 0 and rel_err(y, y_ref) <= TOL[4]
        outs.append((y, [eng.overflow(c) for c in range(C)]))
        eng.close()
    assert rel_err(outs[0][0], outs[1][0]) <= 2e-6
    for a, b in zip(outs[0][1], outs[1][1]):
        assert a.n_overflows == b.n_overflows and abs(a.largest - b.largest) <= 1e-5 * max(b.largest, 1e-30)


def test_time_paired_path_is_the_default_for_odd_channel_counts(orc, bfir):
    """BFIR_PAIR_TIME=0 keeps such engines on the general path; the two agree to rounding, and differ in the last
    bits (which is how this test knows the fast path ran)."""
    if any(os.environ.get(k) for k in ("BFIR_PAIR", "BFIR_PAIR_PERSIST", "BFIR_PAIR_TIME")):
        pytest.skip("a path-selecting switch is set for the whole run (scripts/gpu_env_matrix.sh)")
    L, B, C, nb = 1024, 4, 3, 16
    h, x = _data(orc, C, B * L - 5, nb * L, seed=77)
    a = bfir.Brutefir(L, B, 4, C); a.set_coeff(h); _, ya = a.run(x)
    with env_override(BFIR_PAIR_TIME="0"):
        b = bfir.Brutefir(L, B, 4, C)
    b.set_coeff(h); _, yb = b.run(x)
    assert rel_err(ya, yb) <= 2e-6 and not np.array_equal(ya, yb)


def test_time_paired_history_across_calls_chunks_and_reset(orc, bfir):
    """One-block calls (every chunk a single block paired with nothing), odd and even call lengths, reset()."""
    L, B, C = 512, 3, 3
    h, x = _data(orc, C, 1300, 23 * L, seed=22)
    ref = orc.Engine(L, B, 4, C); ref.set_coeff(h)
    eng = bfir.Brutefir(L, B, 4, C); eng.set_coeff(h); eng.set_chunk(4)
    pos = 0
    for k, nblk in enumerate((3, 1, 1, 2, 5, 1, 4, 6)):
        seg = x[pos * L:(pos + nblk) * L]; pos += nblk
        _, yr = ref.run(seg); rc, y = eng.run(seg)
        assert rc == 0 and rel_err(y, yr) <= TOL[4], (k, nblk)
        if k in (2, 5):
            ref.reset(); eng.reset()


def test_time_paired_overflow_statistics_and_nan_verdict(orc, bfir):
    L, B, C, nb = 512, 2, 3, 9
    rng = np.random.default_rng(11)
    h = [np.r_[np.float32(g), np.zeros(700, np.float32)] for g in (0.5, 3.0, 0.25)]   # pure gains
    x = orc.synth_audio(rng, nb * L, C, np.float32)
    ref = orc.Engine(L, B, 4, C); ref.set_coeff(h); _, y_ref = ref.run(x)
    eng = bfir.Brutefir(L, B, 4, C); eng.set_coeff(h); eng.set_chunk(4); rc, y = eng.run(x)
    assert rc == 0 and rel_err(y, y_ref) <= TOL[4]
    assert np.abs(np.abs(y_ref.astype(np.float64)) - 1.0).min() > 1e-5   # nothing sits on the threshold
    for c in range(C):
        o, r = eng.overflow(c), ref.overflow(c)
        assert o.n_overflows == r.n_overflows and abs(o.largest - r.largest) <= 1e-5 * r.largest
    assert eng.overflow(0).n_overflows == 0 and eng.overflow(1).n_overflows > 0
    bad = x.copy(); bad[4 * L + 7, 2] = np.nan
    eng2 = bfir.Brutefir(L, B, 4, C); eng2.set_coeff(h)
    ref2 = orc.Engine(L, B, 4, C); ref2.set_coeff(h)
    assert ref2.run(bad)[0] == -1 and eng2.run(bad)[0] == -1
