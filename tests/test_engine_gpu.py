"""Parity of the fused HIP engine (bfir_engine_* through the C ABI) with the CPU oracle.

Tolerance: max|y - y_oracle| / max|y_oracle| <= 1e-5 (realsize 4) / 1e-12 (realsize 8),
the figures BASELINE.json's north_star states."""
import numpy as np
import pytest

from conftest import TOL, env_override, rel_err

pytestmark = pytest.mark.gpu

# (realsize, L, B, C, taps, n_blocks, chunk)
CONFIGS = [
    (4, 4096, 2, 2, 8192, 7, 64),       # BASELINE configs[0] shape
    (4, 8192, 8, 2, 65536, 19, 8),      # configs[1] shape, first 2B+3 blocks
    (4, 4096, 32, 8, 131072, 67, 16),   # configs[2] headline shape, 2B+3 blocks
    (8, 4096, 64, 2, 262144, 131, 32),  # configs[4] fp64, 2B+3 blocks
    (4, 1024, 1, 2, 1000, 5, 2),        # single partition (convolve_inplace branch), ragged tail
    (8, 1024, 4, 3, 4000, 11, 3),       # plug-in default L, odd channel count, ragged tail
    (4, 256, 5, 8, 1100, 23, 7),        # chunk not dividing the run, B not a power of two
    (4, 16, 3, 1, 40, 9, 4),            # smallest supported partition
    (8, 64, 2, 8, 128, 6, 1),           # block-at-a-time
    (4, 16384, 2, 1, 20000, 3, 2),      # largest fp32 partition
    (8, 8192, 2, 1, 9000, 3, 2),        # largest fp64 partition
]


def _make(orc, s, C, taps, frames, seed):
    rng = np.random.default_rng(seed)
    dt = orc.real_dtype(s)
    return orc.synth_ir(rng, C, taps, dt), orc.synth_audio(rng, frames, C, dt)


@pytest.mark.parametrize("s,L,B,C,taps,nb,chunk", CONFIGS)
def test_run_matches_oracle(orc, bfir, s, L, B, C, taps, nb, chunk):
    h, x = _make(orc, s, C, taps, nb * L, seed=L + B + C)
    ref = orc.Engine(L, B, s, C)
    assert ref.set_coeff(h) == 0
    rc_ref, y_ref = ref.run(x)
    eng = bfir.Brutefir(L, B, s, C)
    eng.set_chunk(chunk)
    assert not eng.is_initialized()
    assert eng.set_coeff(h) == 0
    assert eng.is_initialized()
    rc, y = eng.run(x)
    assert rc == rc_ref == 0
    assert rel_err(y, y_ref) <= TOL[s]
    for c in range(C):
        o, r = eng.overflow(c), ref.overflow(c)
        assert o.max == r.max == 1.0
        assert abs(o.largest - r.largest) <= TOL[s] * max(r.largest, 1e-30)
        assert o.n_overflows == r.n_overflows == 0
    eng.close()


@pytest.mark.parametrize("s,L", [(4, 512), (8, 512), (8, 1024), (8, 4096), (4, 4096)])
def test_partition_spectra_match_oracle(orc, bfir, s, L):
    """The partition spectra as bfir_engine_read_coeff hands them out -- in the reference's grouped layout whatever the
    engine keeps internally ((re, im) pairs: fp32 from 256 points, fp64 on the run kernels from 1024)."""
    B, C = 5, 2
    taps = 4 * L + 52                 # ragged tail: block 4 holds 52 taps
    h, _ = _make(orc, s, C, taps, L, seed=11)
    ref = orc.Engine(L, B, s, C)
    ref.set_coeff(h, scale=0.5)
    eng = bfir.Brutefir(L, B, s, C)
    assert eng.set_coeff(h, scale=0.5) == 0
    for c in range(C):
        for b in range(B):
            assert rel_err(eng.coeff_block(c, b), ref.coeff_block(c, b)) <= TOL[s]


def test_chunking_is_bitwise_invariant(orc, bfir):
    s, L, B, C, nb = 4, 1024, 6, 4, 29
    h, x = _make(orc, s, C, B * L - 17, nb * L, seed=5)
    outs = []
    for chunk in (1, 2, 5, 8, 64):
        eng = bfir.Brutefir(L, B, s, C)
        eng.set_chunk(chunk)
        assert eng.set_coeff(h) == 0
        rc, y = eng.run(x)
        assert rc == 0
        outs.append(y)
        eng.close()
    for y in outs[1:]:
        assert np.array_equal(y, outs[0])


def test_streaming_calls_equal_one_call(orc, bfir):
    """run() block by block (the plug-in's pattern, foo_dsp_bfir.cpp:311-349) == one batched call."""
    s, L, B, C, nb = 8, 256, 4, 2, 13
    h, x = _make(orc, s, C, 1000, nb * L, seed=6)
    a = bfir.Brutefir(L, B, s, C); a.set_coeff(h)
    _, y_all = a.run(x)
    b = bfir.Brutefir(L, B, s, C); b.set_coeff(h)
    parts = [b.run(x[t * L:(t + 1) * L])[1] for t in range(nb)]
    assert np.array_equal(np.concatenate(parts), y_all)


@pytest.mark.parametrize("in_fmt,out_fmt,s", [(8, 8, 8), (10, 10, 4), (8, 10, 4), (10, 8, 8)])
def test_mixed_sample_widths(orc, bfir, in_fmt, out_fmt, s):
    """float32 I/O around fp64 arithmetic is the plug-in's own configuration
    (foo_dsp_bfir/common.h:17, foo_dsp_bfir.cpp:283-284)."""
    L, B, C, nb = 512, 3, 2, 8
    rng = np.random.default_rng(3)
    h = orc.synth_ir(rng, C, 1400, orc.real_dtype(s))
    x = orc.synth_audio(rng, nb * L, C, orc.fmt_dtype(in_fmt))
    ref = orc.Engine(L, B, s, C, in_fmt, out_fmt); ref.set_coeff(h)
    _, y_ref = ref.run(x)
    eng = bfir.Brutefir(L, B, s, C, in_fmt, out_fmt); eng.set_coeff(h)
    rc, y = eng.run(x)
    assert rc == 0 and y.dtype == y_ref.dtype
    tol = 1e-5 if (s == 4 or out_fmt == 8) else 1e-12
    assert rel_err(y, y_ref) <= tol


def test_overflow_counters(orc, bfir):
    s, L, B, C, nb = 4, 512, 2, 3, 10
    rng = np.random.default_rng(9)
    h = [np.r_[np.float32(g), np.zeros(700, np.float32)] for g in (0.5, 1.5, 3.0)]  # pure gains
    x = orc.synth_audio(rng, nb * L, C, np.float32)
    ref = orc.Engine(L, B, s, C); ref.set_coeff(h); _, y_ref = ref.run(x)
    eng = bfir.Brutefir(L, B, s, C); eng.set_coeff(h); rc, y = eng.run(x)
    assert rc == 0 and rel_err(y, y_ref) <= TOL[s]
    # exact counts are only comparable when no sample sits on the threshold
    margin = np.abs(np.abs(y_ref.astype(np.float64)) - 1.0).min()
    assert margin > 1e-5
    for c in range(C):
        o, r = eng.overflow(c), ref.overflow(c)
        assert o.n_overflows == r.n_overflows
        assert abs(o.largest - r.largest) <= 1e-5 * r.largest
    assert eng.overflow(0).n_overflows == 0 and eng.overflow(2).n_overflows > 0
    report = eng.check_overflows()
    assert [n for n, _, _ in report] == [0, 1, 2] and report[2][2] > 0.0
    eng.reset()
    assert eng.overflow(2).n_overflows == 0 and eng.overflow(2).largest == 0.0


def test_reset_keeps_time_history_like_reference(orc, bfir):
    """brutefir::reset clears counters only (brutefir.cpp:346-367): the first block after it
    still sees the stale half of input_timecbuf[n][0]."""
    s, L, B, C = 8, 128, 3, 2
    h, x = _make(orc, s, C, 300, 9 * L, seed=21)
    ref = orc.Engine(L, B, s, C); ref.set_coeff(h)
    eng = bfir.Brutefir(L, B, s, C); eng.set_coeff(h)
    for nblk in (3, 2, 1, 1):   # odd and even call counts before each reset
        seg = x[:nblk * L]
        _, yr = ref.run(seg); rc, y = eng.run(seg)
        assert rc == 0 and rel_err(y, yr) <= TOL[s]
        ref.reset(); eng.reset()
        x = x[nblk * L:]
    _, yr = ref.run(x); rc, y = eng.run(x)
    assert rc == 0 and rel_err(y, yr) <= TOL[s]


def test_nonfinite_input_returns_minus_one(orc, bfir):
    s, L, B, C = 4, 256, 2, 2
    h, x = _make(orc, s, C, 400, 4 * L, seed=2)
    x[L + 5, 1] = np.nan
    ref = orc.Engine(L, B, s, C); ref.set_coeff(h)
    eng = bfir.Brutefir(L, B, s, C); eng.set_coeff(h)
    assert ref.run(x)[0] == -1
    assert eng.run(x)[0] == -1
    # the verdict is consumed; a clean engine state needs B clean blocks, as in the reference


def test_nonfinite_coefficient_returns_minus_two(orc, bfir):
    eng = bfir.Brutefir(256, 2, 4, 2)
    h = [np.ones(300, np.float32), np.ones(300, np.float32)]
    h[1][7] = np.inf
    assert eng.set_coeff(h) == -2
    assert not eng.is_initialized()
    assert eng.run(np.zeros((256, 2), np.float32))[0] == bfir.ERR_STATE


def test_fewer_coefficient_blocks_than_filter_blocks(orc, bfir):
    """coeffs[n].n_blocks < n_blocks (brutefir.cpp:292)."""
    s, L, B, C, nb = 4, 256, 6, 2, 15
    h, x = _make(orc, s, C, 3 * L, nb * L, seed=8)
    ref = orc.Engine(L, B, s, C); ref.set_coeff(h, coeff_blocks=3)
    eng = bfir.Brutefir(L, B, s, C); assert eng.set_coeff(h, coeff_blocks=3) == 0
    _, yr = ref.run(x); rc, y = eng.run(x)
    assert rc == 0 and rel_err(y, yr) <= TOL[s]


def test_dirac_is_identity(bfir):
    """coeff::load_dirac_coeff's impulse (brutefir/coeff.cpp) passes audio through unchanged."""
    L, B, C, nb = 1024, 4, 2, 9
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (nb * L, C)).astype(np.float32)
    d = np.zeros(B * L, np.float32); d[0] = 1.0
    eng = bfir.Brutefir(L, B, 4, C); eng.set_coeff([d, d])
    rc, y = eng.run(x)
    assert rc == 0 and np.abs(y - x).max() <= 1e-5


def test_batch_of_engines_matches_single_engines(orc, bfir):
    """BASELINE configs[3] shape at reduced size: independent stereo streams sharing launches."""
    s, L, B, C, nb, E = 4, 512, 4, 2, 11, 5
    rng = np.random.default_rng(17)
    hs = [orc.synth_ir(rng, C, B * L, np.float32) for _ in range(E)]
    xs = np.stack([orc.synth_audio(rng, nb * L, C, np.float32) for _ in range(E)])
    batch = bfir.Brutefir(L, B, s, C, n_engines=E)
    batch.set_chunk(4)
    for e in range(E):
        assert batch.set_coeff(hs[e], engine_index=e) == 0
    rc, y = batch.run(xs)
    assert rc == 0
    for e in range(E):
        ref = orc.Engine(L, B, s, C); ref.set_coeff(hs[e])
        assert rel_err(y[e], ref.run(xs[e])[1]) <= TOL[s]
        one = bfir.Brutefir(L, B, s, C); one.set_chunk(4); one.set_coeff(hs[e])
        assert np.array_equal(one.run(xs[e])[1], y[e])


def test_run_device_on_torch_stream(orc, bfir):
    import torch
    s, L, B, C, nb = 4, 2048, 8, 8, 40
    h, x = _make(orc, s, C, B * L, nb * L, seed=4)
    ref = orc.Engine(L, B, s, C); ref.set_coeff(h); _, y_ref = ref.run(x)
    eng = bfir.Brutefir(L, B, s, C); eng.set_chunk(16); eng.set_coeff(h)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty_like(d_in)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()          # a real (non-null) hipStream_t owned by torch
    assert side.cuda_stream != 0
    with torch.cuda.stream(side):
        eng.run_device(d_in.data_ptr(), d_out.data_ptr(), nb, stream=side.cuda_stream)
        done = torch.cuda.Event(); done.record(side)
    done.synchronize()                  # torch's event sees the work: same runtime, same stream
    assert rel_err(d_out.cpu().numpy(), y_ref) <= TOL[s]
    assert eng.sync() == 0


def test_engine_strides_beyond_4_gib(orc, bfir):
    """The engines of a batch may lie more than 2^32 bytes apart (the bench's own 8-channel stream is a 32 GiB
    stride): the C ABI carries the strides as int64_t (include/bfir_hip.h), the kernels as 64-bit offsets."""
    import torch
    s, L, B, C, nb, E = 4, 1024, 3, 2, 6, 2
    stride = (4 << 30) + 65536                                   # bytes between engine 0 and engine 1
    hs = [_make(orc, s, C, B * L, nb * L, seed=70 + e)[0] for e in range(E)]
    xs = [_make(orc, s, C, B * L, nb * L, seed=80 + e)[1] for e in range(E)]
    d_in = torch.zeros(stride + nb * L * C * 4, dtype=torch.uint8, device="cuda")
    d_out = torch.zeros_like(d_in)
    for e in range(E):
        raw = torch.from_numpy(xs[e].view(np.uint8).reshape(-1)).cuda()
        d_in[e * stride:e * stride + raw.numel()] = raw
    eng = bfir.Brutefir(L, B, s, C, n_engines=E)
    for e in range(E):
        assert eng.set_coeff(hs[e], engine_index=e) == 0
    eng.run_device(d_in.data_ptr(), d_out.data_ptr(), nb, in_stride_bytes=stride, out_stride_bytes=stride)
    assert eng.sync() == 0
    for e in range(E):
        y = d_out[e * stride:e * stride + nb * L * C * 4].cpu().numpy().view(np.float32).reshape(nb * L, C)
        ref = orc.Engine(L, B, s, C); ref.set_coeff(hs[e])
        assert rel_err(y, ref.run(xs[e])[1]) <= TOL[s]
    eng.close()


def test_create_rejects_bad_arguments(bfir):
    for args in [(1000, 2, 4, 2), (1024, 2, 6, 2), (1024, 2, 4, 9), (1024, 0, 4, 2), (8, 2, 4, 2),
                 (16384, 2, 8, 1)]:
        with pytest.raises(bfir.BfirError):
            bfir.Brutefir(*args)
    with pytest.raises(bfir.BfirError):
        bfir.Brutefir(1024, 2, 4, 2, in_format=12)  # not a BF_SAMPLE_FORMAT_* code


def _run_kernel_twiddles(s, L):
    """fp64 engines with 1024 <= L <= 8192 in direct mode transform runs of blocks in k_fwd_run (and, one channel per
    workgroup, k_inv_run; kernels.hip), whose twiddles are products of a few base roots instead of table entries: spectra
    equal to the one-transform kernels' to a few 1e-16, not to the bit (chunking still never changes a bit: every launch of
    such an engine goes through the same kernel)."""
    return s == 8 and 1024 <= L <= 8192


def _same_outputs(a, b, s, L, out_is_f32=False):
    if not _run_kernel_twiddles(s, L):
        return np.array_equal(a, b)
    return rel_err(a, b) <= (2e-7 if out_is_f32 else 1e-14)


def _same_stats(a, b, s, L):
    if not _run_kernel_twiddles(s, L):
        return a == b
    return all(na == nb_ and abs(la - lb) <= 1e-6 * max(abs(lb), 1e-300) for (na, la), (nb_, lb) in zip(a, b))


@pytest.mark.parametrize("s,L,B,C,in_fmt,out_fmt", [(8, 1024, 5, 2, 8, 8),     # the plug-in: REALSIZE 8, float32 frames
                                                   (8, 2048, 3, 2, 8, 8),     # ... both channels in one workgroup
                                                   (8, 256, 3, 3, 10, 10), (4, 256, 4, 3, 8, 8), (4, 16384, 2, 1, 8, 10),
                                                   (8, 4096, 2, 5, 10, 8), (4, 64, 3, 2, 8, 8)])
def test_direct_path_equals_staging_kernels_bit_for_bit(orc, bfir, s, L, B, C, in_fmt, out_fmt):
    """Engines outside the float fast path whose frames are FLOAT_LE / FLOAT64_LE read and write the raw frames
    from inside k_fwd / k_inv (direct mode) instead of through k_stage_in / k_stage_out and planar buffers:
    the same arithmetic, so the same bits (_run_kernel_twiddles: the one exception) -- across chunked launches,
    one-block calls, reset and the overflow bookkeeping -- and within tolerance of the oracle."""
    import os
    nb = 13
    rng = np.random.default_rng(L + C)
    h = orc.synth_ir(rng, C, B * L - 7, orc.real_dtype(s))
    x = (orc.synth_audio(rng, nb * L, C, np.float64) * 1.7).astype(orc.fmt_dtype(in_fmt))   # some samples overflow 1.0
    outs = []
    for direct in (True, False):
        with env_override(BFIR_DIRECT="1" if direct else "0"):   # 1 forces it where the engine would not pick it itself
            eng = bfir.Brutefir(L, B, s, C, in_fmt, out_fmt)
        eng.set_chunk(4)
        assert eng.set_coeff(h, scale=12.0) == 0
        parts = [eng.run(x[a * L:b * L])[1] for a, b in ((0, 6), (6, 7), (7, nb))]
        stats = [(eng.overflow(c).n_overflows, eng.overflow(c).largest) for c in range(C)]
        eng.reset()
        parts.append(eng.run(x[:3 * L])[1])
        stats += [(eng.overflow(c).n_overflows, eng.overflow(c).largest) for c in range(C)]
        outs.append((np.concatenate(parts), stats))
        eng.close()
    assert _same_outputs(outs[0][0], outs[1][0], s, L, out_fmt == 8) and _same_stats(outs[0][1], outs[1][1], s, L)
    assert all(big > 0.0 for _, big in outs[0][1][:C])          # the peak bookkeeping ran (counts depend on the data)
    ref = orc.Engine(L, B, s, C, in_fmt, out_fmt); ref.set_coeff(h, scale=12.0)
    y_ref = ref.run(x)[1]
    assert rel_err(outs[0][0][:nb * L], y_ref) <= (1e-5 if (s == 4 or out_fmt == 8) else 1e-12)


@pytest.mark.parametrize("misalign", [0, 8])
def test_stereo_float_frames_direct_batch_and_alignment(orc, bfir, misalign):
    """The plug-in's shape (REALSIZE 8, float32 stereo frames) runs k_inv (and, with BFIR_RUN64=0, k_fwd) with both
    channels of a block in one workgroup and 16-byte frame accesses; buffers that are only 8-byte aligned fall back to one
    channel per workgroup.  Either way: the staging kernels' result (to the bit with the table twiddles, see
    _run_kernel_twiddles), for a batch of engines on device buffers."""
    import os
    import torch
    L, B, C, nb, ne = 1024, 4, 2, 9, 3
    rng = np.random.default_rng(5 + misalign)
    hs = [orc.synth_ir(rng, C, B * L - 3, np.float64) for _ in range(ne)]
    x = (orc.synth_audio(rng, ne * nb * L, C, np.float64) * 1.3).astype(np.float32).reshape(ne, nb * L, C)
    pad = misalign // 4
    d_x = torch.zeros(x.size + 4, dtype=torch.float32, device="cuda")
    d_x[pad:pad + x.size] = torch.from_numpy(x.reshape(-1)).cuda()
    outs = []
    for direct in (None, "0"):
        with env_override(**({} if direct is None else {"BFIR_DIRECT": direct})):
            eng = bfir.Brutefir(L, B, 8, C, 8, 8, n_engines=ne)
        eng.set_chunk(4)
        for g in range(ne):
            assert eng.set_coeff(hs[g], engine_index=g) == 0
        d_y = torch.zeros(x.size + 4, dtype=torch.float32, device="cuda")
        stride = nb * L * C * 4
        for a, b in ((0, 5), (5, 6), (6, nb)):
            eng.run_device(d_x.data_ptr() + misalign + a * L * C * 4, d_y.data_ptr() + misalign + a * L * C * 4, b - a,
                           in_stride_bytes=stride, out_stride_bytes=stride)
        assert eng.sync() == 0
        outs.append(d_y[pad:pad + x.size].cpu().numpy().reshape(ne, nb * L, C))
        eng.close()
    assert _same_outputs(outs[0], outs[1], 8, L, True)
    for g in range(ne):
        ref = orc.Engine(L, B, 8, C, 8, 8); ref.set_coeff(hs[g])
        assert rel_err(outs[0][g], ref.run(x[g])[1]) <= 1e-5      # float32 output frames


@pytest.mark.parametrize("C,misalign,L", [(4, 0, 1024), (6, 8, 512), (8, 0, 2048), (8, 4, 1024)])
def test_wide_float_frames_direct_channel_pairs(orc, bfir, C, misalign, L):
    """fp64 arithmetic on float32 frames with an even channel count above two (5.1 / 7.1 audio through the plug-in's
    shipped REALSIZE 8): k_fwd / k_inv take a channel PAIR per workgroup, 8 bytes of every frame at the frame stride
    (round 3; such engines ran the staging kernels before).  Same bits as the staging path -- one-block and multi-block
    chunks, call-to-call continuation, a batch of engines, device buffers that are 8-byte aligned (the pair path) or
    only 4-byte aligned (one channel per workgroup)."""
    import torch
    B, nb, ne = 3, 9, 2
    rng = np.random.default_rng(50 + C + misalign)
    hs = [orc.synth_ir(rng, C, B * L - 3, np.float64) for _ in range(ne)]
    x = (orc.synth_audio(rng, ne * nb * L, C, np.float64) * 1.3).astype(np.float32).reshape(ne, nb * L, C)
    pad = misalign // 4
    d_x = torch.zeros(x.size + 4, dtype=torch.float32, device="cuda")
    d_x[pad:pad + x.size] = torch.from_numpy(x.reshape(-1)).cuda()
    outs, ofs = [], []
    for direct in (None, "0"):
        with env_override(**({} if direct is None else {"BFIR_DIRECT": direct})):
            eng = bfir.Brutefir(L, B, 8, C, 8, 8, n_engines=ne)
        eng.set_chunk(4)
        for g in range(ne):
            assert eng.set_coeff(hs[g], engine_index=g) == 0
        d_y = torch.zeros(x.size + 4, dtype=torch.float32, device="cuda")
        stride = nb * L * C * 4
        for a, b in ((0, 5), (5, 6), (6, nb)):
            eng.run_device(d_x.data_ptr() + misalign + a * L * C * 4, d_y.data_ptr() + misalign + a * L * C * 4, b - a,
                           in_stride_bytes=stride, out_stride_bytes=stride)
        assert eng.sync() == 0
        outs.append(d_y[pad:pad + x.size].cpu().numpy().reshape(ne, nb * L, C))
        ofs.append([(eng.overflow(c).n_overflows, eng.overflow(c).largest) for c in range(ne * C)])
        eng.close()
    assert _same_outputs(outs[0], outs[1], 8, L, True) and _same_stats(ofs[0], ofs[1], 8, L)
    for g in range(ne):
        ref = orc.Engine(L, B, 8, C, 8, 8); ref.set_coeff(hs[g])
        assert rel_err(outs[0][g], ref.run(x[g])[1]) <= 1e-5      # float32 output frames


@pytest.mark.parametrize("C,misalign,L", [(2, 0, 4096), (2, 16, 1024), (2, 0, 64), (4, 0, 1024), (6, 8, 512), (8, 16, 2048)])
def test_float64_frames_direct_channel_pairs(orc, bfir, C, misalign, L):
    """fp64 engines on float64 frames with an even channel count (cfg5; the plug-in's shape with REALSIZE-8 frames): k_fwd /
    k_inv take a channel PAIR per workgroup and move whole stereo frames (32 bytes per lane) or 16 bytes of every wider
    frame, instead of 8 bytes at the frame stride (round 3).  Same bits as the staging path -- one-block and multi-block
    chunks, call-to-call continuation, a batch of engines, device buffers aligned for the pair path or only for the
    one-channel kernels (misalign) -- and the oracle's sums to 1e-12."""
    import torch
    B, nb, ne = 3, 7, 2
    rng = np.random.default_rng(70 + C + misalign + L)
    hs = [orc.synth_ir(rng, C, B * L - 3, np.float64) for _ in range(ne)]
    x = (orc.synth_audio(rng, ne * nb * L, C, np.float64) * 1.3).reshape(ne, nb * L, C)
    pad = misalign // 8
    d_x = torch.zeros(x.size + 4, dtype=torch.float64, device="cuda")
    d_x[pad:pad + x.size] = torch.from_numpy(x.reshape(-1)).cuda()
    outs, ofs = [], []
    for direct in (None, "0"):
        with env_override(**({} if direct is None else {"BFIR_DIRECT": direct})):
            eng = bfir.Brutefir(L, B, 8, C, n_engines=ne)
        eng.set_chunk(3)
        for g in range(ne):
            assert eng.set_coeff(hs[g], engine_index=g) == 0
        d_y = torch.zeros(x.size + 4, dtype=torch.float64, device="cuda")
        stride = nb * L * C * 8
        for a, b in ((0, 4), (4, 5), (5, nb)):
            eng.run_device(d_x.data_ptr() + misalign + a * L * C * 8, d_y.data_ptr() + misalign + a * L * C * 8, b - a,
                           in_stride_bytes=stride, out_stride_bytes=stride)
        assert eng.sync() == 0
        outs.append(d_y[pad:pad + x.size].cpu().numpy().reshape(ne, nb * L, C))
        ofs.append([(eng.overflow(c).n_overflows, eng.overflow(c).largest) for c in range(ne * C)])
        eng.close()
    assert _same_outputs(outs[0], outs[1], 8, L) and _same_stats(ofs[0], ofs[1], 8, L)
    for g in range(ne):
        ref = orc.Engine(L, B, 8, C); ref.set_coeff(hs[g])
        assert rel_err(outs[0][g], ref.run(x[g])[1]) <= TOL[8]


@pytest.mark.parametrize("L,C,fmt", [(4096, 2, 10), (2048, 3, 10), (2048, 1, 8), (8192, 1, 10), (1024, 3, 8), (1024, 2, 8), (1024, 6, 10)])
def test_run_kernels_do_not_depend_on_the_chunking(orc, bfir, L, C, fmt):
    """k_fwd_run / k_inv_run (fp64 engines in direct mode, 1024 <= L <= 8192, one channel per workgroup) walk runs of
    blocks with the window's first half and the next block's data carried in registers: whatever the launch sizes -- one
    call, single blocks (the plug-in's pattern), uneven pieces, explicit chunks shorter and longer than a run, a reset in
    between -- every output sample has the same bits, the overflow statistics are the same, and the result is the oracle's
    to 1e-12 (1e-5 on float32 frames).  BFIR_RUN64=0 (the one-transform kernels, table twiddles) agrees to 1e-14."""
    B, nb = 3, 41
    rng = np.random.default_rng(L + C + fmt)
    h = orc.synth_ir(rng, C, B * L - 9, np.float64)
    x = (orc.synth_audio(rng, nb * L, C, np.float64) * 1.5).astype(orc.fmt_dtype(fmt))

    def run(pieces, chunk, env=None):
        with env_override(**(env or {})):
            eng = bfir.Brutefir(L, B, 8, C, fmt, fmt)
            eng.set_chunk(chunk); assert eng.set_coeff(h) == 0
            ys, a = [], 0
            for n in pieces:
                ys.append(eng.run(x[a * L:(a + n) * L])[1]); a += n
            st = [(eng.overflow(c).n_overflows, eng.overflow(c).largest) for c in range(C)]
            eng.close()
        return np.concatenate(ys), st

    y0, st0 = run([nb], 0)
    for pieces, chunk in (([1] * nb, 0), ([5, 1, 17, 2, 16], 0), ([nb], 7), ([20, 21], 33)):
        y, st = run(pieces, chunk)
        assert np.array_equal(y, y0) and st == st0
    ref = orc.Engine(L, B, 8, C, fmt, fmt); ref.set_coeff(h)
    assert rel_err(y0, ref.run(x)[1]) <= (1e-5 if fmt == 8 else TOL[8])
    y1, st1 = run([nb], 0, {"BFIR_RUN64": "0"})
    assert _same_outputs(y0, y1, 8, L, fmt == 8) and _same_stats(st0, st1, 8, L)


@pytest.mark.parametrize("s,L,B,C", [(4, 1024, 40, 2), (4, 128, 5, 3), (8, 1024, 7, 2), (4, 4096, 9, 8), (8, 64, 3, 1)])
def test_small_launch_mac_kernel_gives_the_bits_of_the_throughput_kernels(orc, bfir, s, L, B, C):
    """Calls of up to four blocks (the plug-in's pattern is one) run k_mac_small -- one lane per bin walking the
    partitions in order -- and 8-channel engines bounce the block through HBM with a copy kernel.  Same
    chain of fused multiply-adds as the streaming / LDS / tiled MAC kernels: identical output bits, whether the
    blocks come one by one, in fours, or all in one launch."""
    import os
    nb = 11
    rng = np.random.default_rng(L + B)
    h = orc.synth_ir(rng, C, B * L - 5, orc.real_dtype(s))
    x = orc.synth_audio(rng, nb * L, C, orc.real_dtype(s))
    outs = []
    for mode in ("small", "no_small", "one_launch"):
        with env_override(**({"BFIR_NO_MAC_SMALL": "1", "BFIR_NO_BOUNCE": "1"} if mode == "no_small" else {})):
            eng = bfir.Brutefir(L, B, s, C)
            assert eng.set_coeff(h) == 0
            if mode == "one_launch":
                eng.set_chunk(16)
                parts = [eng.run(x)[1]]
            else:
                cuts = [0, 1, 2, 4, 8, 9, nb]                       # calls of 1, 1, 2, 4, 1, 2 blocks
                parts = [eng.run(x[a * L:b * L])[1] for a, b in zip(cuts[:-1], cuts[1:])]
            outs.append(np.concatenate(parts))
            eng.close()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    ref = orc.Engine(L, B, s, C); ref.set_coeff(h)
    assert rel_err(outs[0], ref.run(x)[1]) <= TOL[s]


@pytest.mark.parametrize("variant", ["7", "8", "9", "12"])
def test_fp64_mac_variants_give_identical_bits(orc, bfir, variant):
    """BFIR_MAC64_VARIANT (tuning aid, read per launch) selects other fp64 MAC kernels -- a barrier per partition,
    four partitions per barrier, deeper prefetch, the register-only partition-streaming kernel: the same chain of
    fused multiply-adds per bin, so the same bits as the default kernel."""
    import os
    L, B, C, nb = 1024, 37, 2, 72          # 72 blocks: two 32-block tiles and a ragged third (16-block tiles: 4 and a half)
    rng = np.random.default_rng(12)
    h = orc.synth_ir(rng, C, B * L - 9, np.float64)
    x = orc.synth_audio(rng, nb * L, C, np.float64)
    outs = []
    for v in (None, variant):
        with env_override(**({} if v is None else {"BFIR_MAC64_VARIANT": v})):
            eng = bfir.Brutefir(L, B, 8, C)
            eng.set_chunk(nb)
            assert eng.set_coeff(h) == 0
            outs.append(eng.run(x)[1])
            eng.close()
    assert np.array_equal(outs[0], outs[1])
    ref = orc.Engine(L, B, 8, C); ref.set_coeff(h)
    assert rel_err(outs[0], ref.run(x)[1]) <= TOL[8]


@pytest.mark.parametrize("n_eng", [1, 3])
def test_pinned_frame_buffers_skip_the_staging_copies_with_the_same_bits(orc, bfir, n_eng):
    """bfir_engine_run on page-locked caller buffers (bfir_pinned_malloc) DMAs them directly; pageable buffers go through the
    engine's staging buffers.  Same kernels either way: the same bits -- several host chunks (512 blocks each), a ragged
    last chunk, a batch of engines (one transfer per engine and chunk), and each buffer pinned on its own."""
    s, L, B, C, nb = 4, 256, 3, 2, 1100
    rng = np.random.default_rng(23)
    hs = [orc.synth_ir(rng, C, B * L - 5, np.float32) for _ in range(n_eng)]
    x = np.stack([orc.synth_audio(rng, nb * L, C, np.float32) for _ in range(n_eng)])
    shape = x.shape if n_eng > 1 else x.shape[1:]

    def engine():
        e = bfir.Brutefir(L, B, s, C, n_engines=n_eng)
        for g in range(n_eng):
            assert e.set_coeff(hs[g], engine_index=g) == 0
        return e

    rc, want = engine().run(x.reshape(shape))
    assert rc == 0
    ref = orc.Engine(L, B, s, C); ref.set_coeff(hs[0])
    assert rel_err(want.reshape(x.shape)[0][:40 * L], ref.run(x[0][:40 * L])[1]) <= TOL[s]
    for pin_in, pin_out in ((True, True), (True, False), (False, True)):
        xin = bfir.pinned_frames(bfir.SAMPLE_FORMAT_FLOAT_LE, shape) if pin_in else np.empty(shape, np.float32)
        xin[...] = x.reshape(shape)
        yout = bfir.pinned_frames(bfir.SAMPLE_FORMAT_FLOAT_LE, shape) if pin_out else np.empty(shape, np.float32)
        rc, got = engine().run(xin, yout)
        assert rc == 0 and got is yout
        assert np.array_equal(got, want), (pin_in, pin_out)
    with env_override(BFIR_NO_PINNED_DIRECT="1"):       # the switch puts pinned buffers back on the staging path
        xin = bfir.pinned_frames(bfir.SAMPLE_FORMAT_FLOAT_LE, shape); xin[...] = x.reshape(shape)
        assert np.array_equal(engine().run(xin)[1], want)


@pytest.mark.parametrize("s,L,C,in_fmt,out_fmt", [
    (4, 1024, 2, None, None),       # fp32 channel-pair kernels
    (4, 1024, 3, None, None),       # fp32 blocks paired in time
    (4, 4096, 8, None, None),       # ... frames wide enough for the HBM bounce
    (8, 1024, 2, 8, 8),             # the plug-in as shipped: fp64, float32 stereo frames (k_inv with channel pairs)
    (8, 1024, 3, 8, 8),             # fp64, odd channel count (k_inv_run: verdict published by its own kernel)
    (8, 1024, 1, 10, 10),           # fp64 mono, float64 frames
    (4, 256, 2, 2, 2),              # 16-bit integer frames: staging kernels
    (4, 64, 2, None, None),         # below the fast paths' sizes
])
def test_latency_path_reports_nonfinite_blocks_without_a_copy(orc, bfir, s, L, C, in_fmt, out_fmt):
    """One to four blocks per call (the plug-in's pattern, foo_dsp_bfir.cpp:311-349): the kernels flag a block whose sample 0 is
    not finite (brutefir.cpp:316-321) in pinned host memory themselves; every inverse / staging kernel family has to."""
    B = 3
    rng = np.random.default_rng(L + C)
    dt = orc.real_dtype(s)
    h = orc.synth_ir(rng, C, B * L, dt)
    x = orc.synth_audio(rng, 8 * L, C, np.float32)
    if in_fmt == 2:
        x = np.round(x * 20000).astype("<i2")
    elif in_fmt == 10:
        x = x.astype(np.float64)

    def engine():
        e = bfir.Brutefir(L, B, s, C, in_fmt, out_fmt)
        assert e.set_coeff(h) == 0
        return e

    e = engine()
    for t in range(4):                                   # clean blocks, one per call: no false alarm
        assert e.run(x[t * L:(t + 1) * L])[0] == 0
    assert e.run(x[4 * L:8 * L])[0] == 0                 # four blocks in one call
    if in_fmt == 2:
        return                                           # integer frames cannot carry a NaN in
    for nblk, bad_at in ((1, 0), (3, 1), (4, 3)):
        e = engine()
        assert e.run(x[:2 * L])[0] == 0
        xb = x[2 * L:(2 + nblk) * L].copy()
        xb[bad_at * L + 7, C - 1] = np.nan
        assert e.run(xb)[0] == -1, (nblk, bad_at)
    e = engine(); e.set_chunk(1)                         # one block per launch inside a four-block call
    xb = x[:4 * L].copy(); xb[2 * L + 1, 0] = np.inf
    assert e.run(xb)[0] == -1
