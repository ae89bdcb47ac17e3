"""ctypes front end of the CPU oracle (oracle/bfir_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/bfir_oracle.h).
Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libbfir_oracle.so")

FMT_FLOAT_LE = 8      # brutefir/global.h:31
FMT_FLOAT64_LE = 10   # brutefir/global.h:33
MIXMODE_INPUT = 1     # brutefir/fftw_convolver.hpp:14
MIXMODE_OUTPUT = 3    # brutefir/fftw_convolver.hpp:16


class Overflow(C.Structure):
    """bfoverflow_t, brutefir/global.h:96-102."""
    _fields_ = [("n_overflows", C.c_uint), ("intlargest", C.c_int32),
                ("largest", C.c_double), ("max", C.c_double)]


def build(force=False):
    """Compile the oracle with gcc (make -C oracle)."""
    srcs = [os.path.join(_HERE, f) for f in
            ("bfir_oracle.c", "bfir_oracle_impl.inc", "bfir_oracle.h", "Makefile")]
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        vp, ci, cd = C.c_void_p, C.c_int, C.c_double
        sig = {
            "orc_r2hc_f": (None, [ci, vp, vp]), "orc_hc2r_f": (None, [ci, vp, vp]),
            "orc_r2hc_d": (None, [ci, vp, vp]), "orc_hc2r_d": (None, [ci, vp, vp]),
            "orc_raw2real_f": (None, [vp, vp, ci, ci, ci]),
            "orc_raw2real_d": (None, [vp, vp, ci, ci, ci]),
            "orc_raw2cbuf_f": (None, [ci, vp, ci, ci, ci, vp, vp]),
            "orc_raw2cbuf_d": (None, [ci, vp, ci, ci, ci, vp, vp]),
            "orc_mixnscale_f": (None, [ci, vp, vp, cd, ci]),
            "orc_mixnscale_d": (None, [ci, vp, vp, cd, ci]),
            "orc_convolve_inplace_f": (None, [ci, vp, vp]),
            "orc_convolve_inplace_d": (None, [ci, vp, vp]),
            "orc_convolve_f": (None, [ci, vp, vp, vp]), "orc_convolve_d": (None, [ci, vp, vp, vp]),
            "orc_convolve_add_f": (None, [ci, vp, vp, vp]),
            "orc_convolve_add_d": (None, [ci, vp, vp, vp]),
            "orc_coeffs2cbuf_f": (ci, [ci, vp, ci, cd, vp]),
            "orc_coeffs2cbuf_d": (ci, [ci, vp, ci, cd, vp]),
            "orc_real2raw_f": (None, [vp, vp, ci, ci, ci, C.POINTER(Overflow)]),
            "orc_real2raw_d": (None, [vp, vp, ci, ci, ci, C.POINTER(Overflow)]),
            "orc_engine_create": (vp, [ci, ci, ci, ci, ci, ci]),
            "orc_engine_create_ex": (vp, [ci, ci, ci, ci, ci, ci, ci, ci]),
            "orc_dither_create": (vp, [ci, ci, ci, ci, ci]), "orc_dither_destroy": (None, [vp]),
            "orc_dither_table_size": (ci, [vp]), "orc_dither_table": (vp, [vp]),
            "orc_dither_randtab_ptr": (ci, [vp, ci]),
            "orc_real2raw_hp_tpdf_f": (None, [vp, ci, vp, vp, ci, ci, ci, C.POINTER(Overflow)]),
            "orc_real2raw_hp_tpdf_d": (None, [vp, ci, vp, vp, ci, ci, ci, C.POINTER(Overflow)]),
            "orc_engine_destroy": (None, [vp]),
            "orc_engine_set_coeff": (ci, [vp, C.POINTER(vp), ci, ci, ci, cd]),
            "orc_engine_run": (ci, [vp, vp, vp]),
            "orc_engine_run_blocks": (ci, [vp, vp, vp, ci]),
            "orc_engine_reset": (None, [vp]),
            "orc_engine_get_overflow": (None, [vp, ci, C.POINTER(Overflow)]),
            "orc_engine_coeff_block": (vp, [vp, ci, ci]),
            "orc_direct_conv": (None, [vp, ci, vp, ci, vp]),
            "orc_raw2real_fmt_f": (None, [vp, vp, ci, ci, ci]),
            "orc_raw2real_fmt_d": (None, [vp, vp, ci, ci, ci]),
            "orc_real2raw_fmt_f": (None, [vp, vp, ci, ci, ci, C.POINTER(Overflow)]),
            "orc_real2raw_fmt_d": (None, [vp, vp, ci, ci, ci, C.POINTER(Overflow)]),
            "orc_fmt_in_scale": (cd, [ci]), "orc_fmt_out_scale": (cd, [ci]), "orc_fmt_max": (cd, [ci]),
            "orc_mixnscale_n_f": (None, [ci, C.POINTER(vp), vp, C.POINTER(cd), ci, ci]),
            "orc_mixnscale_n_d": (None, [ci, C.POINTER(vp), vp, C.POINTER(cd), ci, ci]),
            "orc_dirac_convolve_f": (None, [ci, vp, vp]), "orc_dirac_convolve_d": (None, [ci, vp, vp]),
            "orc_convolve_eval_f": (None, [ci, vp, vp, vp]), "orc_convolve_eval_d": (None, [ci, vp, vp, vp]),
            "orc_crossfade_inplace_f": (None, [ci, vp, vp, vp]),
            "orc_crossfade_inplace_d": (None, [ci, vp, vp, vp]),
            "orc_td_block_length": (ci, [ci]),
            "orc_convolve_inplace_ordered_f": (None, [ci, vp, vp]),
            "orc_convolve_inplace_ordered_d": (None, [ci, vp, vp]),
            "orc_td_new_f": (ci, [vp, ci, vp]), "orc_td_new_d": (ci, [vp, ci, vp]),
            "orc_td_convolve_f": (None, [ci, vp, vp]), "orc_td_convolve_d": (None, [ci, vp, vp]),
            "orc_debug_dump_values_f": (None, [ci, vp, vp]), "orc_debug_dump_values_d": (None, [ci, vp, vp]),
            "orc_debug_dump_cbuf": (ci, [C.c_char_p, ci, ci, C.POINTER(vp), ci]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def real_dtype(realsize):
    return np.float32 if realsize == 4 else np.float64


# numpy element type of a raw sample (brutefir/global.h:24-34); 24-bit samples are 3 raw bytes
FMT_DTYPES = {1: np.dtype("i1"), 2: np.dtype("<i2"), 3: np.dtype(">i2"), 4: None, 5: None,
              6: np.dtype("<i4"), 7: np.dtype(">i4"), 8: np.dtype("<f4"), 9: np.dtype(">f4"),
              10: np.dtype("<f8"), 11: np.dtype(">f8")}
FMT_BYTES = {1: 1, 2: 2, 3: 2, 4: 3, 5: 3, 6: 4, 7: 4, 8: 4, 9: 4, 10: 8, 11: 8}


def fmt_dtype(fmt):
    d = FMT_DTYPES[fmt]
    return np.dtype((np.uint8, 3)) if d is None else d


def raw_frames(fmt, frames, channels):
    """Zeroed interleaved raw buffer of `frames` x `channels` samples of format `fmt`."""
    d = FMT_DTYPES[fmt]
    return np.zeros((frames, channels, 3), np.uint8) if d is None else np.zeros((frames, channels), d)


def encode_ints(values, fmt):
    """Integer sample values [frames, C] -> raw buffer of an integer format."""
    v = np.asarray(values, dtype=np.int64)
    if FMT_DTYPES[fmt] is not None:
        return v.astype(FMT_DTYPES[fmt])
    u = (v & 0xFFFFFF).astype(np.uint32)
    b = np.stack([(u >> s) & 0xFF for s in ((0, 8, 16) if fmt == 4 else (16, 8, 0))], axis=-1)
    return b.astype(np.uint8)


def decode_ints(raw, fmt):
    """Raw buffer of an integer format -> int64 sample values [frames, C]."""
    if FMT_DTYPES[fmt] is not None:
        return np.asarray(raw).astype(np.int64)
    r = np.asarray(raw).astype(np.int64)
    lo, mid, hi = (r[..., 0], r[..., 1], r[..., 2]) if fmt == 4 else (r[..., 2], r[..., 1], r[..., 0])
    v = lo | (mid << 8) | (hi << 16)
    return np.where(v >= 1 << 23, v - (1 << 24), v)


def _suf(dtype):
    return "_f" if np.dtype(dtype) == np.float32 else "_d"


# ---- stage level ---------------------------------------------------------
def r2hc(x):
    x = np.ascontiguousarray(x)
    out = np.empty_like(x)
    getattr(lib(), "orc_r2hc" + _suf(x.dtype))(x.size, _p(x), _p(out))
    return out


def hc2r(x):
    x = np.ascontiguousarray(x)
    out = np.empty_like(x)
    getattr(lib(), "orc_hc2r" + _suf(x.dtype))(x.size, _p(x), _p(out))
    return out


def mixnscale(x, scale, mixmode):
    x = np.ascontiguousarray(x)
    out = np.zeros_like(x)
    getattr(lib(), "orc_mixnscale" + _suf(x.dtype))(x.size, _p(x), _p(out), float(scale), mixmode)
    return out


def convolve(b, c):
    b, c = np.ascontiguousarray(b), np.ascontiguousarray(c)
    d = np.empty_like(b)
    getattr(lib(), "orc_convolve" + _suf(b.dtype))(b.size, _p(b), _p(c), _p(d))
    return d


def convolve_add(b, c, d):
    b, c = np.ascontiguousarray(b), np.ascontiguousarray(c)
    d = np.array(d, copy=True)
    getattr(lib(), "orc_convolve_add" + _suf(b.dtype))(b.size, _p(b), _p(c), _p(d))
    return d


def convolve_inplace(b, c):
    b, c = np.array(b, copy=True), np.ascontiguousarray(c)
    getattr(lib(), "orc_convolve_inplace" + _suf(b.dtype))(b.size, _p(b), _p(c))
    return b


def coeffs2cbuf(taps, n_fft2, scale=1.0):
    taps = np.ascontiguousarray(taps)
    dest = np.zeros(2 * n_fft2, dtype=taps.dtype)
    rc = getattr(lib(), "orc_coeffs2cbuf" + _suf(taps.dtype))(n_fft2, _p(taps), taps.size,
                                                               float(scale), _p(dest))
    return None if rc != 0 else dest


def raw2cbuf(raw, channel, n_fft2, realsize, prev_next=None):
    """raw: interleaved [frames, C] array (float32/float64). Returns (cbuf, next_cbuf)."""
    raw = np.ascontiguousarray(raw)
    rd = real_dtype(realsize)
    cbuf = np.zeros(2 * n_fft2, dtype=rd)
    if prev_next is not None:
        cbuf[:n_fft2] = prev_next[:n_fft2]
    nxt = np.zeros(2 * n_fft2, dtype=rd)
    getattr(lib(), "orc_raw2cbuf" + _suf(rd))(n_fft2, _p(raw), channel * raw.itemsize,
                                               raw.itemsize, raw.shape[1], _p(cbuf), _p(nxt))
    return cbuf, nxt


def real2raw(real, raw, channel, of):
    """Scatter `real` into channel `channel` of interleaved `raw`, updating Overflow `of`."""
    real = np.ascontiguousarray(real)
    base = raw.ctypes.data + channel * raw.itemsize
    getattr(lib(), "orc_real2raw" + _suf(real.dtype))(C.c_void_p(base), _p(real), raw.itemsize,
                                                      raw.shape[1], real.size, C.byref(of))


def mixnscale_n(ins, scales, mixmode):
    ins = [np.ascontiguousarray(a) for a in ins]
    out = np.zeros_like(ins[0])
    ptrs = (C.c_void_p * len(ins))(*[a.ctypes.data for a in ins])
    sc = (C.c_double * len(ins))(*[float(v) for v in scales])
    getattr(lib(), "orc_mixnscale_n" + _suf(out.dtype))(out.size, ptrs, _p(out), sc, len(ins), mixmode)
    return out


def dirac_convolve(x):
    x = np.ascontiguousarray(x)
    out = np.empty_like(x)
    getattr(lib(), "orc_dirac_convolve" + _suf(x.dtype))(x.size, _p(x), _p(out))
    return out


def convolve_eval(x, buffer):
    """Returns the output; `buffer` (1.5 n_fft reals) is updated in place."""
    x = np.ascontiguousarray(x)
    out = np.empty_like(x)
    getattr(lib(), "orc_convolve_eval" + _suf(x.dtype))(x.size, _p(x), _p(buffer), _p(out))
    return out


def crossfade_inplace(inp, crossfade, buffer):
    """All three arrays are updated in place (as the reference's are)."""
    getattr(lib(), "orc_crossfade_inplace" + _suf(inp.dtype))(inp.size, _p(inp), _p(crossfade), _p(buffer))


def td_block_length(n_coeffs):
    """convolver_td_block_length; -1 for fewer than two taps."""
    return lib().orc_td_block_length(int(n_coeffs))


def convolve_inplace_ordered(b, c):
    """Half-complex product (convolve_inplace_ordered); returns the product, b untouched."""
    out = np.array(b, copy=True)
    c = np.ascontiguousarray(c, dtype=out.dtype)
    getattr(lib(), "orc_convolve_inplace_ordered" + _suf(out.dtype))(out.size, _p(out), _p(c))
    return out


def td_new(taps):
    """convolver_td_new: (blocklen, 2 * blocklen half-complex coefficients)."""
    taps = np.ascontiguousarray(taps)
    blocklen = td_block_length(taps.size)
    if blocklen < 0:
        return -1, None
    out = np.empty(2 * blocklen, dtype=taps.dtype)
    got = getattr(lib(), "orc_td_new" + _suf(taps.dtype))(_p(taps), taps.size, _p(out))
    assert got == blocklen
    return blocklen, out


def td_convolve(td_coeffs, overlap_block):
    """convolver_td_convolve on a copy of overlap_block (2 * blocklen reals)."""
    out = np.array(overlap_block, dtype=td_coeffs.dtype, copy=True)
    getattr(lib(), "orc_td_convolve" + _suf(out.dtype))(out.size // 2, _p(td_coeffs), _p(out))
    return out


def debug_dump_values(cbuf):
    """The n_fft2 values convolver_debug_dump_cbuf prints for one cbuf."""
    cbuf = np.ascontiguousarray(cbuf)
    out = np.empty(cbuf.size // 2, dtype=cbuf.dtype)
    getattr(lib(), "orc_debug_dump_values" + _suf(cbuf.dtype))(cbuf.size, _p(cbuf), _p(out))
    return out


def debug_dump_cbuf(filename, cbufs):
    """convolver_debug_dump_cbuf: writes the text file; 0 or -1 (cannot open)."""
    cbufs = [np.ascontiguousarray(b) for b in cbufs]
    arr = (C.c_void_p * len(cbufs))(*[b.ctypes.data for b in cbufs])
    return lib().orc_debug_dump_cbuf(str(filename).encode(), cbufs[0].dtype.itemsize, cbufs[0].size, arr, len(cbufs))


def equalizer_bands(sampling_rate, freq, mag, phase):
    """equalizer ctor + generate() up to the render call: three 33-entry tables."""
    f, m, p = (np.ascontiguousarray(a, dtype=np.float64) for a in (freq, mag, phase))
    of, om, op = np.zeros(33), np.zeros(33), np.zeros(33)
    L = lib()
    L.orc_equalizer_bands.restype = C.c_int
    L.orc_equalizer_bands.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6
    n = L.orc_equalizer_bands(sampling_rate, f.size, _p(f), _p(m), _p(p), _p(of), _p(om), _p(op))
    assert n == 33
    return of, om, op


def equalizer_render(taps, freq, mag, phase, realsize):
    ir = np.zeros(taps // 2, dtype=real_dtype(realsize))
    fn = getattr(lib(), "orc_equalizer_render" + _suf(ir.dtype))
    fn.restype = None
    fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    fn(taps, len(freq), _p(np.ascontiguousarray(freq)), _p(np.ascontiguousarray(mag)),
       _p(np.ascontiguousarray(phase)), _p(ir))
    return ir


def raw2real_fmt(raw, channel, fmt, realsize):
    """Channel `channel` of an interleaved raw buffer of any format -> working precision."""
    n, spacing = raw.shape[0], raw.shape[1]
    out = np.zeros(n, dtype=real_dtype(realsize))
    base = raw.ctypes.data + channel * FMT_BYTES[fmt]
    getattr(lib(), "orc_raw2real_fmt" + _suf(out.dtype))(_p(out), C.c_void_p(base), fmt, spacing, n)
    return out


def real2raw_fmt(real, raw, channel, fmt, of):
    """Write `real` into channel `channel` of interleaved raw buffer `raw` (any format)."""
    real = np.ascontiguousarray(real)
    base = raw.ctypes.data + channel * FMT_BYTES[fmt]
    getattr(lib(), "orc_real2raw_fmt" + _suf(real.dtype))(C.c_void_p(base), _p(real), fmt, raw.shape[1],
                                                           real.size, C.byref(of))


def direct_conv(x, h):
    x = np.ascontiguousarray(x, dtype=np.float64)
    h = np.ascontiguousarray(h, dtype=np.float64)
    y = np.empty_like(x)
    lib().orc_direct_conv(_p(x), x.size, _p(h), h.size, _p(y))
    return y


def fft_backend(try_fftw=True):
    """Which FFT the CPU baseline runs on: real FFTW r2r plans when libfftw3f.so.3 / libfftw3.so.3 can be
    dlopen'ed on this host (nothing is ever installed), otherwise the oracle's own FFT."""
    L = lib()
    L.orc_use_fftw.restype = C.c_int
    L.orc_use_fftw.argtypes = [C.c_int]
    if try_fftw and L.orc_use_fftw(1):
        return "FFTW (libfftw3f.so.3 / libfftw3.so.3, FFTW_MEASURE r2r plans)"
    return "own CPU FFT (FFTW unavailable)"


class Dither:
    """class dither (brutefir/dither.cpp): random table + per-channel state."""

    def __init__(self, n_channels, sample_rate, realsize, max_size=0, max_samples_per_loop=1024):
        self.realsize = realsize
        self.h = lib().orc_dither_create(n_channels, sample_rate, realsize, max_size, max_samples_per_loop)
        if not self.h:
            raise ValueError("dither table budget too small")

    def close(self):
        if self.h:
            lib().orc_dither_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def table(self):
        n = lib().orc_dither_table_size(self.h)
        buf = (C.c_int8 * n).from_address(lib().orc_dither_table(self.h))
        return np.frombuffer(buf, dtype=np.int8, count=n).copy()

    def randtab_ptr(self, channel):
        return lib().orc_dither_randtab_ptr(self.h, channel)

    def real2raw(self, real, raw, channel, fmt, of):
        """convolver_cbuf2raw with apply_dither: preloop + real2raw_hp_tpdf of `real` into channel
        `channel` of the interleaved integer buffer `raw`."""
        real = np.ascontiguousarray(real, dtype=real_dtype(self.realsize))
        base = raw.ctypes.data + channel * FMT_BYTES[fmt]
        getattr(lib(), "orc_real2raw_hp_tpdf" + _suf(real.dtype))(self.h, channel, C.c_void_p(base), _p(real), fmt,
                                                                   raw.shape[1], real.size, C.byref(of))


# ---- engine level --------------------------------------------------------
class Engine:
    """brutefir (brutefir/brutefir.hpp:15-128) restated on the CPU."""

    def __init__(self, filter_length, filter_blocks, realsize, channels,
                 in_format=None, out_format=None, sampling_rate=44100, apply_dither=False):
        dflt = FMT_FLOAT_LE if realsize == 4 else FMT_FLOAT64_LE
        self.L, self.B, self.s, self.C = filter_length, filter_blocks, realsize, channels
        self.in_format = dflt if in_format is None else in_format
        self.out_format = dflt if out_format is None else out_format
        self.h = lib().orc_engine_create_ex(filter_length, filter_blocks, realsize, channels,
                                            self.in_format, self.out_format, sampling_rate,
                                            int(bool(apply_dither)))
        if not self.h:
            raise ValueError("oracle rejected engine parameters")

    def close(self):
        if self.h:
            lib().orc_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_coeff(self, coeffs, coeff_blocks=None, scale=1.0, length=None):
        """coeffs: sequence of 1-D tap arrays (one per channel), working precision."""
        rd = real_dtype(self.s)
        arrs = [np.ascontiguousarray(c, dtype=rd) for c in coeffs]
        length = arrs[0].size if length is None else length
        coeff_blocks = self.B if coeff_blocks is None else coeff_blocks
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        return lib().orc_engine_set_coeff(self.h, ptrs, len(arrs), length, coeff_blocks,
                                          float(scale))

    def run(self, x):
        """x: [n_blocks*L, C] interleaved frames in the input format. Returns (rc, y)."""
        if FMT_DTYPES[self.in_format] is not None:
            x = np.ascontiguousarray(x, dtype=FMT_DTYPES[self.in_format])
        else:
            x = np.ascontiguousarray(x, dtype=np.uint8)   # [frames, C, 3]
        assert x.shape[1] == self.C and x.shape[0] % self.L == 0
        y = raw_frames(self.out_format, x.shape[0], self.C)
        rc = lib().orc_engine_run_blocks(self.h, _p(x), _p(y), x.shape[0] // self.L)
        return rc, y

    def reset(self):
        lib().orc_engine_reset(self.h)

    def overflow(self, ch):
        of = Overflow()
        lib().orc_engine_get_overflow(self.h, ch, C.byref(of))
        return of

    def coeff_block(self, ch, block):
        p = lib().orc_engine_coeff_block(self.h, ch, block)
        n = 2 * self.L
        buf = (C.c_char * (n * self.s)).from_address(p)
        return np.frombuffer(buf, dtype=real_dtype(self.s), count=n).copy()


# ---- synthetic workloads --------------------------
def synth_ir(rng, channels, taps, dtype):
    """uniform[-1,1) * exp(-6 n / taps), normalised so sum|h| <= 1 per channel."""
    n = np.arange(taps, dtype=np.float64)
    out = []
    for _ in range(channels):
        h = rng.uniform(-1.0, 1.0, taps) * np.exp(-6.0 * n / taps)
        h /= max(np.abs(h).sum(), 1e-30)
        out.append(h.astype(dtype))
    return out


def synth_audio(rng, frames, channels, dtype):
    """i.i.d. uniform [-1,1) like buffer::load_white_noise (brutefir/buffer.cpp:454-493)."""
    return rng.uniform(-1.0, 1.0, (frames, channels)).astype(dtype)


# ---- sampled reference for long runs -----------------------------------------
def sampled_reference(h, get_block, blocks, L, B, s, C, in_format=None, out_format=None):
    # (float outputs only: with dither the output also depends on the whole past through the
    # error feedback, so there is no finite window)
    """Oracle output of selected blocks of an arbitrarily long run without running it whole.

    A partitioned FIR has B blocks of memory: output block g is a function of input blocks
    g-B .. g only (partition i pairs with the spectrum of [block g-i-1 | block g-i],
    brutefir/brutefir.cpp:288-299), and the sums of a fresh engine fed exactly those B+1 blocks
    are the very sums of the long run (the one extra leading block supplies the "previous block"
    half of the oldest spectrum that is used).  get_block(g) -> [L, C] input frames of global
    block g (g < 0: the engine's zeroed start-up state, never requested).
    Returns {g: [L, C] output frames}.
    """
    out = {}
    for g in blocks:
        g0 = max(0, g - B)
        x = np.concatenate([get_block(k) for k in range(g0, g + 1)])
        eng = Engine(L, B, s, C, in_format, out_format)
        assert eng.set_coeff(h) == 0
        rc, y = eng.run(x)
        assert rc == 0
        out[g] = y[-L:]
        eng.close()
    return out
