"""Host-side mirror of the reference's `brutefir` class over the C ABI.

Method names, argument meaning and return codes follow
brutefir/brutefir.hpp:15-128 so the parity tests read like the reference's
callers (foo_dsp_bfir/foo_dsp_bfir.cpp:279-345, brutefir/preprocessor.cpp:287-333).
Everything is computed by libbfir_hip.so on the GPU."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (BfirError, Overflow, SAMPLE_FORMAT_FLOAT64_LE, SAMPLE_FORMAT_FLOAT_LE)


def _real_dtype(realsize):
    return np.float32 if realsize == 4 else np.float64


# raw sample element types (brutefir/global.h:24-34); 24-bit samples are 3 raw bytes
_FMT_DTYPES = {1: np.dtype("i1"), 2: np.dtype("<i2"), 3: np.dtype(">i2"), 4: None, 5: None,
               6: np.dtype("<i4"), 7: np.dtype(">i4"), 8: np.dtype("<f4"), 9: np.dtype(">f4"),
               10: np.dtype("<f8"), 11: np.dtype(">f8")}


def raw_frames(fmt, shape_frames_channels):
    """Zeroed interleaved raw buffer [..., frames, C] (plus a trailing 3 for 24-bit formats)."""
    d = _FMT_DTYPES[fmt]
    shape = tuple(shape_frames_channels)
    return np.zeros(shape + (3,), np.uint8) if d is None else np.zeros(shape, d)


def pinned_frames(fmt, shape_frames_channels):
    """raw_frames in page-locked host memory (bfir_pinned_malloc): `Brutefir.run` on such arrays skips the staging copies.
    The array keeps its allocation alive; it is released when the array is collected."""
    import weakref
    d = _FMT_DTYPES[fmt]
    shape = tuple(shape_frames_channels) + ((3,) if d is None else ())
    dt = np.dtype(np.uint8) if d is None else d
    n = int(np.prod(shape)) * dt.itemsize
    lib = _lib.load()
    p = lib.bfir_pinned_malloc(n)
    if not p:
        raise BfirError(_lib.ERR_HIP, "bfir_pinned_malloc")
    buf = (C.c_char * n).from_address(p)
    a = np.frombuffer(buf, dtype=dt).reshape(shape)
    weakref.finalize(buf, lib.bfir_pinned_free, p)
    a[...] = 0
    return a


class Brutefir:
    """brutefir(filter_length, filter_blocks, realsize, channels, in_format,
    out_format, sampling_rate, apply_dither)  -- brutefir/brutefir.hpp:18-25.

    n_engines > 1 builds a batch of independent, identically shaped engines
    that share launches (BASELINE.json configs[3])."""

    def __init__(self, filter_length, filter_blocks, realsize, channels,
                 in_format=None, out_format=None, sampling_rate=44100, apply_dither=False,
                 device=0, n_engines=1):
        dflt = SAMPLE_FORMAT_FLOAT_LE if realsize == 4 else SAMPLE_FORMAT_FLOAT64_LE
        self.L, self.B, self.s, self.C = filter_length, filter_blocks, realsize, channels
        self.in_format = dflt if in_format is None else in_format
        self.out_format = dflt if out_format is None else out_format
        self.n_engines, self.device = n_engines, device
        self._lib = _lib.load()
        err = C.c_int(0)
        self._h = self._lib.bfir_engine_create_batch(
            n_engines, filter_length, filter_blocks, realsize, channels, self.in_format,
            self.out_format, sampling_rate, int(bool(apply_dither)), device, C.byref(err))
        if not self._h:
            raise BfirError(err.value, "bfir_engine_create")

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.bfir_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # -- brutefir public interface -----------------------------------------
    def is_initialized(self):
        return bool(self._lib.bfir_engine_is_initialized(self._h))

    def set_coeff(self, coeffs, n_coeffs=None, length=None, coeff_blocks=None, scale=1.0,
                  engine_index=0):
        """set_coeff(void **coeffs, n_coeffs, length, coeff_blocks, scale)
        (brutefir/brutefir.cpp:179-228).  Returns 0 or -2."""
        rd = _real_dtype(self.s)
        arrs = [np.ascontiguousarray(c, dtype=rd) for c in coeffs]
        n_coeffs = len(arrs) if n_coeffs is None else n_coeffs
        length = arrs[0].size if length is None else length
        coeff_blocks = self.B if coeff_blocks is None else coeff_blocks
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        return self._lib.bfir_engine_set_coeff_at(self._h, engine_index, ptrs, n_coeffs, length,
                                                  coeff_blocks, float(scale))

    def run(self, inbuf, outbuf=None):
        """run(void *inbuf, void *outbuf) for every L-frame block in `inbuf`
        (host arrays).  inbuf: [n_blocks*L, C] (one engine) or
        [n_engines, n_blocks*L, C].  Returns (rc, outbuf)."""
        d_in = _FMT_DTYPES[self.in_format]
        x = np.ascontiguousarray(inbuf, dtype=np.uint8 if d_in is None else d_in)
        shape = x.shape[:-1] if d_in is None else x.shape        # 24-bit: [..., frames, C, 3]
        frames = shape[-2]
        assert shape[-1] == self.C and frames % self.L == 0
        assert int(np.prod(shape)) == self.n_engines * frames * self.C
        if outbuf is None:
            outbuf = raw_frames(self.out_format, shape)
        rc = self._lib.bfir_engine_run(self._h, x.ctypes.data, outbuf.ctypes.data, frames // self.L)
        return rc, outbuf

    def run_device(self, d_in, d_out, n_blocks, in_stride_bytes=0, out_stride_bytes=0, stream=None):
        """Asynchronous run on device pointers (ints).  Raises on a bad call;
        the NaN verdict comes from sync()."""
        rc = self._lib.bfir_engine_run_device(self._h, d_in, in_stride_bytes, d_out, out_stride_bytes,
                                              n_blocks, stream)
        if rc != 0:
            raise BfirError(rc, "bfir_engine_run_device")

    def sync(self):
        return self._lib.bfir_engine_sync(self._h)

    def reset(self):
        self._lib.bfir_engine_reset(self._h)

    def overflow(self, channel):
        of = Overflow()
        rc = self._lib.bfir_engine_get_overflow(self._h, channel, C.byref(of))
        if rc != 0:
            raise BfirError(rc, "bfir_engine_get_overflow")
        return of

    def check_overflows(self):
        """brutefir::check_overflows (brutefir.cpp:370-388): list of
        (channel, n_overflows, peak_dB) when any channel overflowed."""
        ofs = [self.overflow(n) for n in range(self.C * self.n_engines)]
        if not any(o.n_overflows for o in ofs):
            return []
        out = []
        for n, o in enumerate(ofs):
            peak = max(o.largest, float(o.intlargest))
            db = float("-inf") if peak == 0.0 else 20.0 * np.log10(peak / o.max)
            out.append((n, o.n_overflows, db))
        return out

    # -- tuning / introspection --------------------------------------------
    def set_chunk(self, blocks_per_launch):
        rc = self._lib.bfir_engine_set_chunk(self._h, blocks_per_launch)
        if rc != 0:
            raise BfirError(rc, "bfir_engine_set_chunk")

    def set_profiling(self, on):
        self._lib.bfir_engine_set_profiling(self._h, int(bool(on)))

    def profile(self):
        """{kernel name: (total_ms, launches)} measured with HIP events."""
        out = {}
        for k, name in enumerate(_lib.KERNEL_NAMES):
            ms, n = C.c_double(0), C.c_int64(0)
            self._lib.bfir_engine_get_profile(self._h, k, C.byref(ms), C.byref(n))
            out[name] = (ms.value, n.value)
        return out

    def coeff_block(self, channel, block):
        dst = np.zeros(2 * self.L, dtype=_real_dtype(self.s))
        rc = self._lib.bfir_engine_read_coeff(self._h, channel, block, dst.ctypes.data)
        if rc != 0:
            raise BfirError(rc, "bfir_engine_read_coeff")
        return dst
