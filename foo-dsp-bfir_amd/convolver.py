"""Host-side mirror of the reference's `fftw_convolver` class
(brutefir/fftw_convolver.hpp:28-166) over the stage-level C ABI.  Buffers are
numpy arrays in host memory, as the reference's are malloc'ed host blocks."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BfirError, BufferFormat, DitherState, Overflow


def make_buffer_format(fmt, channel, n_channels):
    """setup_input / setup_output for any sample format (brutefir/brutefir.cpp:435-582)."""
    nbytes = {1: 1, 2: 2, 3: 2, 4: 3, 5: 3, 6: 4, 7: 4, 8: 4, 9: 4, 10: 8, 11: 8}[fmt]
    bf = BufferFormat()
    bf.sf.isfloat, bf.sf.swap = fmt >= 8, fmt in (3, 5, 7, 9, 11)
    bf.sf.bytes = bf.sf.sbytes = nbytes
    bf.sf.scale, bf.sf.format = 1.0, fmt      # the stage calls take their scales explicitly
    bf.sample_spacing, bf.byte_offset = n_channels, channel * nbytes
    return bf


class Dither:
    """dither(n_channels, sample_rate, realsize, max_size, max_samples_per_loop, dither_state)
    (brutefir/dither.hpp:13-77).  `states` is the dither_state_t array the reference keeps in
    bfconf (global.h:87): the constructor fills it, the caller passes states[n] with channel n."""

    def __init__(self, n_channels, sample_rate, realsize, max_size=0, max_samples_per_loop=1024, device=0):
        self._lib = _lib.load()
        self.states = (DitherState * n_channels)()
        err = C.c_int(0)
        self._h = self._lib.bfir_dither_create(n_channels, sample_rate, realsize, max_size, max_samples_per_loop,
                                               self.states, device, C.byref(err))
        if not self._h:
            raise BfirError(err.value, "bfir_dither_create")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bfir_dither_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def table(self):
        n = self._lib.bfir_dither_table_size(self._h)
        return np.ctypeslib.as_array(self._lib.bfir_dither_table(self._h), shape=(n,)).copy()


class TdConv:
    """td_conv_t (brutefir/fftw_convolver.hpp:18-26) as convolver_td_new returns it: `coeffs` is the host copy of the
    scaled half-complex spectrum, `blocklen` the block length; the spectrum the product reads stays on the device."""

    def __init__(self, lib, handle, dtype):
        self._lib, self._h = lib, handle
        self.blocklen = lib.bfir_td_blocklen(handle)
        n = 2 * self.blocklen
        src = (C.c_float if dtype == np.float32 else C.c_double) * n
        self.coeffs = np.frombuffer(src.from_address(lib.bfir_td_coeffs(handle)), dtype=dtype).copy()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bfir_td_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FftwConvolver:
    """fftw_convolver(length, realsize, dither) (brutefir/fftw_convolver.hpp:31).  `dither` (a Dither)
    is only needed for convolver_cbuf2raw with apply_dither on an integer format."""

    def __init__(self, length, realsize, dither=None, device=0):
        self._lib = _lib.load()
        self._dither = dither
        self.n_fft2, self.n_fft, self.realsize, self.device = length, 2 * length, realsize, device
        self.dtype = np.float32 if realsize == 4 else np.float64
        err = C.c_int(0)
        self._h = self._lib.bfir_convolver_create(length, realsize, device, C.byref(err))
        if not self._h:
            raise BfirError(err.value, "bfir_convolver_create")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bfir_convolver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise BfirError(rc, what)

    def _buf(self, a):
        assert a.dtype == self.dtype and a.flags.c_contiguous and a.size >= self.n_fft
        return a.ctypes.data

    def convolver_cbufsize(self):
        return self._lib.bfir_convolver_cbufsize(self._h)

    def new_cbuf(self):
        return np.zeros(self.n_fft, dtype=self.dtype)

    def convolver_raw2cbuf(self, rawbuf, cbuf, next_cbuf, bf):
        self._chk(self._lib.bfir_convolver_raw2cbuf(self._h, rawbuf.ctypes.data, self._buf(cbuf),
                                                    self._buf(next_cbuf), C.byref(bf)), "raw2cbuf")

    def convolver_time2freq(self, input_cbuf, output_cbuf):
        self._chk(self._lib.bfir_convolver_time2freq(self._h, self._buf(input_cbuf),
                                                     self._buf(output_cbuf)), "time2freq")

    def convolver_mixnscale(self, input_cbufs, output_cbuf, scales, n_bufs, mixmode):
        ptrs = (C.c_void_p * n_bufs)(*[self._buf(b) for b in input_cbufs[:n_bufs]])
        sc = (C.c_double * n_bufs)(*[float(v) for v in scales[:n_bufs]])
        self._chk(self._lib.bfir_convolver_mixnscale(self._h, ptrs, self._buf(output_cbuf), sc, n_bufs,
                                                     mixmode), "mixnscale")

    def convolver_convolve_inplace(self, cbuf, coeffs):
        self._chk(self._lib.bfir_convolver_convolve_inplace(self._h, self._buf(cbuf), self._buf(coeffs)),
                  "convolve_inplace")

    def convolver_convolve(self, input_cbuf, coeffs, output_cbuf):
        self._chk(self._lib.bfir_convolver_convolve(self._h, self._buf(input_cbuf), self._buf(coeffs),
                                                    self._buf(output_cbuf)), "convolve")

    def convolver_convolve_add(self, input_cbuf, coeffs, output_cbuf):
        self._chk(self._lib.bfir_convolver_convolve_add(self._h, self._buf(input_cbuf), self._buf(coeffs),
                                                        self._buf(output_cbuf)), "convolve_add")

    def convolver_freq2time(self, input_cbuf, output_cbuf):
        self._chk(self._lib.bfir_convolver_freq2time(self._h, self._buf(input_cbuf),
                                                     self._buf(output_cbuf)), "freq2time")

    def convolver_cbuf2raw(self, cbuf, outbuf, bf, overflow, apply_dither=False, dither_state=None):
        """convolver_cbuf2raw(cbuf, outbuf, bf, apply_dither, dither_state, overflow)
        (brutefir/fftw_convolver.cpp:405-466); dither only ever acts on integer formats."""
        if apply_dither and not bf.sf.isfloat:
            if self._dither is None or dither_state is None:
                raise BfirError(_lib.ERR_ARG, "cbuf2raw: dither instance not set")     # :412-416
            self._chk(self._lib.bfir_convolver_cbuf2raw_dither(self._h, self._dither._h, self._buf(cbuf),
                                                               outbuf.ctypes.data, C.byref(bf),
                                                               C.byref(dither_state), C.byref(overflow)), "cbuf2raw")
            return
        self._chk(self._lib.bfir_convolver_cbuf2raw(self._h, self._buf(cbuf), outbuf.ctypes.data,
                                                    C.byref(bf), C.byref(overflow)), "cbuf2raw")

    def convolver_coeffs2cbuf(self, coeffs, n_coeffs, scale, optional_dest=None):
        """Returns the spectrum as an array (optional_dest if given), or None
        on a NaN/Inf tap (the reference returns NULL)."""
        taps = np.ascontiguousarray(coeffs, dtype=self.dtype)
        dest = self.new_cbuf() if optional_dest is None else optional_dest
        p = self._lib.bfir_convolver_coeffs2cbuf(self._h, taps.ctypes.data, n_coeffs, float(scale),
                                                 self._buf(dest))
        return dest if p else None

    # ---- declared by the reference, called by nothing in its tree ----
    def convolver_runtime_coeffs2cbuf(self, src, dest):
        self._chk(self._lib.bfir_convolver_runtime_coeffs2cbuf(self._h, src.ctypes.data, self._buf(dest)),
                  "runtime_coeffs2cbuf")

    def convolver_dirac_convolve(self, input_cbuf, output_cbuf):
        self._chk(self._lib.bfir_convolver_dirac_convolve(self._h, self._buf(input_cbuf),
                                                          self._buf(output_cbuf)), "dirac_convolve")

    def convolver_dirac_convolve_inplace(self, cbuf):
        self._chk(self._lib.bfir_convolver_dirac_convolve_inplace(self._h, self._buf(cbuf)), "dirac_convolve")

    def convolver_convolve_eval(self, input_cbuf, buffer_cbuf, output_cbuf):
        assert buffer_cbuf.size >= 3 * self.n_fft2 and buffer_cbuf.dtype == self.dtype
        self._chk(self._lib.bfir_convolver_convolve_eval(self._h, self._buf(input_cbuf), buffer_cbuf.ctypes.data,
                                                         self._buf(output_cbuf)), "convolve_eval")

    def convolver_crossfade_inplace(self, input_cbuf, crossfade_cbuf, buffer_cbuf):
        assert buffer_cbuf.size >= 3 * self.n_fft2 and buffer_cbuf.dtype == self.dtype
        self._chk(self._lib.bfir_convolver_crossfade_inplace(self._h, self._buf(input_cbuf),
                                                             self._buf(crossfade_cbuf), buffer_cbuf.ctypes.data),
                  "crossfade_inplace")

    def convolver_verify_cbuf(self, cbufs, n_cbufs):
        ptrs = (C.c_void_p * n_cbufs)(*[self._buf(b) for b in cbufs[:n_cbufs]])
        rc = self._lib.bfir_convolver_verify_cbuf(self._h, ptrs, n_cbufs)
        if rc < 0:
            raise BfirError(rc, "verify_cbuf")
        return bool(rc)

    def convolver_debug_dump_cbuf(self, filename, cbufs, n_cbufs):
        """Text dump of the coefficient lists behind `cbufs` (brutefir/fftw_convolver.cpp:604-651).  A file that
        cannot be opened is logged and skipped, as there (the method is void)."""
        ptrs = (C.c_void_p * n_cbufs)(*[self._buf(b) for b in cbufs[:n_cbufs]])
        rc = self._lib.bfir_convolver_debug_dump_cbuf(self._h, str(filename).encode(), ptrs, n_cbufs)
        if rc not in (_lib.OK, _lib.ERR_IO):
            raise BfirError(rc, "debug_dump_cbuf")

    # ---- the td convolver of the (dead) delay class (fftw_convolver.hpp:157-166) ----
    def convolver_td_block_length(self, n_coeffs):
        return self._lib.bfir_td_block_length(int(n_coeffs))

    def convolver_td_new(self, coeffs, n_coeffs):
        """Returns a TdConv, or None where the reference returns NULL (block length -1)."""
        if self.convolver_td_block_length(n_coeffs) == -1:
            return None
        taps = np.ascontiguousarray(coeffs, dtype=self.dtype)
        assert taps.size >= n_coeffs
        err = C.c_int(0)
        h = self._lib.bfir_td_new(taps.ctypes.data, int(n_coeffs), self.realsize, self.device, C.byref(err))
        if not h:
            raise BfirError(err.value, "bfir_td_new")
        return TdConv(self._lib, h, self.dtype)

    def convolver_td_convolve(self, tdc, overlap_block):
        assert overlap_block.dtype == self.dtype and overlap_block.flags.c_contiguous
        assert overlap_block.size >= 2 * tdc.blocklen
        self._chk(self._lib.bfir_td_convolve(tdc._h, overlap_block.ctypes.data), "td_convolve")
