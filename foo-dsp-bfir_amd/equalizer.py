"""Host-side mirror of the reference's `equalizer` class (brutefir/equalizer.hpp:67-115,
equalizer.cpp): 31-band ISO graphic EQ -> impulse response of block_length*n_blocks/2 taps.
The band placement of generate() is 33 scalars of host arithmetic, as in the reference; the
65536-bin spectrum and its HC2R transform run on the GPU (bfir_equalizer_render)."""
import ctypes as C
import math
import os
import struct

import numpy as np

from . import _lib, wavio
from ._lib import BfirError

ISO_BANDS = [20, 25, 31.5, 40, 50, 63, 80, 100, 125, 160, 200, 250, 315, 400, 500, 630, 800, 1000, 1250, 1600,
             2000, 2500, 3150, 4000, 5000, 6300, 8000, 10000, 12500, 16000, 20000]   # equalizer.hpp:17-50
BAND_COUNT = len(ISO_BANDS)


def djb_hash(data):
    """DJBHash (brutefir/hash.c:113-124) over bytes taken as SIGNED chars, 32-bit wrap."""
    h = 5381
    for b in data:
        h = (h * 33 + (b - 256 if b > 127 else b)) & 0xFFFFFFFF
    return h


class Equalizer:
    """equalizer(block_length, n_blocks, realsize, n_channels, sampling_rate) -- equalizer.cpp:29-69."""

    def __init__(self, block_length, n_blocks, realsize, n_channels, sampling_rate, device=0):
        taps = block_length * n_blocks
        if taps < 32 or taps & (taps - 1):
            raise ValueError("Equalizer length (%d, %d) is not a power of two." % (block_length, n_blocks))
        self.block_length, self.n_blocks, self.realsize = block_length, n_blocks, realsize
        self.n_channels, self.sampling_rate, self.taps = n_channels, sampling_rate, taps
        self._lib = _lib.load()
        err = C.c_int(0)
        # create_fft_plan(log2(taps), invert=true, inplace=true)  (equalizer.cpp:54-55)
        self._plan = self._lib.bfir_fft_plan_create(int(math.log2(taps)), 1, 1, realsize, device, C.byref(err))
        if not self._plan:
            raise BfirError(err.value, "bfir_fft_plan_create")
        self.band_count = BAND_COUNT + 2
        self.freq = [0.0] + [float(f) for f in ISO_BANDS] + [sampling_rate / 2.0]
        self.mag = [0.0] * self.band_count
        self.phase = [0.0] * self.band_count

    def close(self):
        if getattr(self, "_plan", None):
            self._lib.bfir_fft_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def make_filename(self, freq, mag, phase):
        """equalizer::make_filename (equalizer.cpp:152-180), without the temp directory."""
        blob = b"".join(struct.pack("<%dd" % len(a), *a) for a in (freq, mag, phase))
        return "eq-%x-%d-%d-%d-%d.wav" % (djb_hash(blob), self.taps >> 1, self.realsize, self.n_channels,
                                          self.sampling_rate)

    def generate(self, freq, mag, phase, cache_dir=None):
        """equalizer::generate (equalizer.cpp:86-140).  Returns the [taps/2, n_channels] impulse
        response; with cache_dir it also writes / re-uses the reference's cache WAV."""
        n_bands = len(freq)
        if n_bands > BAND_COUNT:
            raise ValueError("Number of bands (%d) excceds limit (%d)." % (n_bands, BAND_COUNT))
        i = 0
        for n in range(n_bands):                       # place the bands on the grid (:99-108)
            while freq[n] > self.freq[i]:
                i += 1
            self.mag[i], self.phase[i] = float(mag[n]), float(phase[n])
            i += 1
        self.mag[0] = self.mag[1]
        self.mag[-1] = self.mag[-2]
        for n in range(self.band_count):               # :113-118 (in place, as the reference does)
            self.freq[n] /= float(self.sampling_rate)
            self.mag[n] = math.pow(10, self.mag[n] / 20)
            self.phase[n] /= (180 * math.pi)
        path = None
        if cache_dir is not None:
            path = os.path.join(cache_dir, self.make_filename(freq, mag, phase))
            if os.path.exists(path):
                return wavio.read_wav(path)[0]
        dt = np.float32 if self.realsize == 4 else np.float64
        ir = np.zeros(self.taps >> 1, dtype=dt)
        arr = lambda v: (C.c_double * len(v))(*v)
        rc = self._lib.bfir_equalizer_render(self._plan, self.band_count, arr(self.freq), arr(self.mag),
                                             arr(self.phase), ir.ctypes.data)
        if rc != 0:
            raise BfirError(rc, "bfir_equalizer_render")
        out = np.repeat(ir[:, None], self.n_channels, axis=1)      # the same response in every channel (:266-282)
        if path is not None:
            wavio.write_wav_float(path, out, self.sampling_rate)
        return out


class FftPlan:
    """fftw_convolver::create_fft_plan + fftw[f]_execute_r2r for any power-of-two size."""

    def __init__(self, order, invert, realsize, device=0):
        self._lib = _lib.load()
        err = C.c_int(0)
        self._h = self._lib.bfir_fft_plan_create(order, int(bool(invert)), 1, realsize, device, C.byref(err))
        if not self._h:
            raise BfirError(err.value, "bfir_fft_plan_create")
        self.n, self.dtype = 1 << order, (np.float32 if realsize == 4 else np.float64)

    def execute(self, x):
        x = np.ascontiguousarray(x, dtype=self.dtype)
        assert x.size == self.n
        out = np.empty_like(x)
        rc = self._lib.bfir_fft_plan_execute(self._h, x.ctypes.data, out.ctypes.data)
        if rc != 0:
            raise BfirError(rc, "bfir_fft_plan_execute")
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bfir_fft_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
