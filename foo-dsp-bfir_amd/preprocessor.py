"""The reference's offline drivers (brutefir/preprocessor.cpp) as batch jobs on the GPU engine.

Same call sequences as the reference -- including its habit of passing `filter_length` as
the coefficient length to set_coeff (preprocessor.cpp:121-122, 176-178, 313), which makes
only the first partition of each coefficient set non-zero -- but every loop of run() calls
is one batched bfir_engine_run.  File I/O (libsndfile) is outside the path: impulses and
noise arrive as arrays."""
import math

import numpy as np

from .engine import Brutefir
from ._lib import SAMPLE_FORMAT_FLOAT64_LE, SAMPLE_FORMAT_FLOAT_LE


def get_next_multiple(value, factor):
    """util::get_next_multiple (brutefir/util.cpp:46-56)."""
    multiple = factor
    while value > multiple:
        multiple += factor
    return multiple


def _dtype(realsize):
    return np.float32 if realsize == 4 else np.float64


def _fmt(realsize):
    return SAMPLE_FORMAT_FLOAT_LE if realsize == 4 else SAMPLE_FORMAT_FLOAT64_LE


def _pad_frames(x, frames, dt):
    out = np.zeros((frames, x.shape[1]), dtype=dt)
    n = min(frames, x.shape[0])
    out[:n] = x[:n]
    return out


def convolve_impulses(impulses, filter_length, realsize, device=0, make_engine=Brutefir):
    """preprocessor::convolve_impulses (preprocessor.cpp:33-233) on in-memory impulses.

    impulses: list of (frames[n_frames, C] array, scale).  Returns the [g_frames, C] result
    (what the reference writes to its cache WAV), or None on a processing error."""
    dt = _dtype(realsize)
    g_frames = max(x.shape[0] for x, _ in impulses)
    g_channels = impulses[0][0].shape[1]
    if any(x.shape[1] != g_channels for x, _ in impulses):
        raise ValueError("channel counts differ")                         # `throw;` at :76
    length = get_next_multiple(g_frames, filter_length)
    filter_blocks = length // filter_length
    flt = make_engine(filter_length, filter_blocks, realsize, g_channels, _fmt(realsize), _fmt(realsize),
                      device=device) if make_engine is Brutefir else make_engine(
                          filter_length, filter_blocks, realsize, g_channels)
    # a dirac for the initial coefficients (coeff::load_dirac_coeff, coeff.cpp:33-59)
    dirac = np.zeros(filter_length, dtype=dt)
    dirac[0] = 1.0
    flt.set_coeff([dirac] * g_channels, g_channels, filter_length, filter_blocks, 1.0)
    outbuf = None
    for frames, scale in impulses:
        inbuf = _pad_frames(np.asarray(frames, dtype=dt), filter_length * filter_blocks, dt)
        rc, outbuf = flt.run(inbuf)                     # the reference's loop over filter_blocks run() calls
        if rc != 0:
            return None
        # the output becomes the next coefficients (buffer::deinterlace, :169-178)
        coeffs = [np.ascontiguousarray(outbuf[:, c]) for c in range(g_channels)]
        flt.set_coeff(coeffs, g_channels, filter_length, filter_blocks, scale)
    return outbuf[:g_frames].copy()


def calculate_attenuation(coeffs, filter_length, realsize, noise, device=0, make_engine=Brutefir):
    """preprocessor::calculate_attenuation (preprocessor.cpp:249-412).

    coeffs: [n_frames, C] impulse; noise: [L*filter_blocks, C] full-scale white noise (the
    reference draws it with buffer::load_white_noise).  Returns the attenuation in dB."""
    dt = _dtype(realsize)
    coeffs = np.asarray(coeffs, dtype=dt)
    n_frames, n_channels = coeffs.shape
    length = get_next_multiple(n_frames, filter_length)
    filter_blocks = length // filter_length
    taps = [np.ascontiguousarray(_pad_frames(coeffs, filter_length * filter_blocks, dt)[:, c])
            for c in range(n_channels)]

    def new_filter():
        f = make_engine(filter_length, filter_blocks, realsize, n_channels, _fmt(realsize), _fmt(realsize),
                        device=device) if make_engine is Brutefir else make_engine(
                            filter_length, filter_blocks, realsize, n_channels)
        f.set_coeff(taps, n_channels, filter_length, filter_blocks, 1.0)   # :313 (length = filter_length)
        return f

    flt = new_filter()
    noise = np.ascontiguousarray(noise, dtype=dt)
    assert noise.shape == (filter_length * filter_blocks, n_channels)
    rc, out = flt.run(noise)
    if rc == 0:
        max_value = float(np.abs(out).max())                               # :336-354
        # the engine's overflow peak is the same number without reading the output back
        if hasattr(flt, "overflow"):
            peak = max(flt.overflow(c).largest for c in range(n_channels))
            assert peak == max_value
    else:
        max_value = _scan_block_by_block(new_filter, noise, filter_length, filter_blocks)
    return -20.0 * math.log10(max_value) if max_value > 1.0 else 0.0


def _scan_block_by_block(new_filter, noise, L, n_blocks):
    """The reference's loop when some run() fails (preprocessor.cpp:329-356): the failed block's output is not
    scanned and the loop goes on.  brutefir::run returns before it flips curbuf and advances blockcounter
    (brutefir.cpp:316-321, 337-340), so the failed block never enters the history -- the next call overwrites its
    delay-line slot and its half of the time buffer.  After a failure the filter is rebuilt from the blocks kept
    so far (an error path)."""
    kept, max_value = [], 0.0
    flt = new_filter()
    for n in range(n_blocks):
        rc, out = flt.run(noise[n * L:(n + 1) * L])
        if rc == 0:
            kept.append(n)
            a = np.abs(out)
            a = a[a == a]                                                  # a NaN never compares greater (:340-352)
            if a.size:
                max_value = max(max_value, float(a.max()))
        else:
            flt = new_filter()
            if kept and flt.run(np.concatenate([noise[k * L:(k + 1) * L] for k in kept]))[0] != 0:
                raise RuntimeError("blocks that passed before fail on replay")
    return max_value
