// stage.hip -- bfir_convolver_* of include/bfir_hip.h: one call per
// fftw_convolver method (brutefir/fftw_convolver.hpp:28-166) on HOST buffers.
//
// Every call copies its operands to HBM, runs the same gfx950 kernels the
// fused engine uses, and copies the result back, so a caller that keeps the
// reference's per-stage sequence (brutefir.cpp:252-334, 304 calls per 8-channel
// block) works unchanged.  That sequence cannot be fast across PCIe; it is the
// plumbing / parity path.  The measured path is bfir_engine_run*.
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/bfir_hip.h"
#include "kernels.h"

using namespace bfir;

void bfir_logf(const char *fmt, ...);

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            bfir_logf("HIP error %s at %s:%d", hipGetErrorString(_e), __FILE__, __LINE__); \
            return BFIR_ERR_HIP;                                                   \
        }                                                                          \
    } while (0)

struct bfir_convolver {
    int device = 0, L = 0, N = 0, s = 0;
    FftPlan plan;
    hipStream_t stream = nullptr;
    void *d[3] = {nullptr, nullptr, nullptr};   // three cbuf-sized scratch buffers
    void *d_raw = nullptr; size_t raw_cap = 0;  // raw sample scratch
    DevOverflow *d_of = nullptr;
    int *d_bad = nullptr;
};

static size_t cb(const bfir_convolver *c) { return (size_t)c->N * (size_t)c->s; }

extern "C" void *bfir_aligned_malloc(size_t size, size_t alignment)
{
    void *p = nullptr;
    if (alignment < sizeof(void *)) alignment = sizeof(void *);
    if (posix_memalign(&p, alignment, size ? size : alignment) != 0) return nullptr;
    return p;
}

extern "C" void bfir_aligned_free(void *p) { free(p); }

extern "C" bfir_convolver *bfir_convolver_create(int length, int realsize, int device, int *err)
{
    int dummy;
    if (!err) err = &dummy;
    *err = BFIR_OK;
    // fftw_convolver.cpp:64-74
    if (realsize != 4 && realsize != 8) { bfir_logf("Invalid real size %d.\n", realsize); *err = BFIR_ERR_ARG; return nullptr; }
    if (length < 1 || (length & (length - 1))) { bfir_logf("Invalid length %d.\n", length); *err = BFIR_ERR_ARG; return nullptr; }
    int ndev = bfir_device_count();
    if (ndev <= 0) { *err = BFIR_ERR_NO_DEVICE; return nullptr; }
    if (device < 0 || device >= ndev) { *err = BFIR_ERR_ARG; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { *err = BFIR_ERR_HIP; return nullptr; }
    bfir_convolver *c = new bfir_convolver();
    c->device = device; c->L = length; c->N = 2 * length; c->s = realsize;
    int rc = fft_plan_create(&c->plan, length, realsize);
    if (rc != 0) { *err = (rc == -1) ? BFIR_ERR_UNSUPPORTED : BFIR_ERR_HIP; delete c; return nullptr; }
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 3 && ok; i++) ok = hipMalloc(&c->d[i], cb(c)) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->d_of, sizeof(DevOverflow)) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->d_bad, sizeof(int)) == hipSuccess;
    if (!ok) { *err = BFIR_ERR_HIP; bfir_convolver_destroy(c); return nullptr; }
    return c;
}

extern "C" void bfir_convolver_destroy(bfir_convolver *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    fft_plan_destroy(&c->plan);
    for (int i = 0; i < 3; i++) if (c->d[i]) (void)hipFree(c->d[i]);
    if (c->d_raw) (void)hipFree(c->d_raw);
    if (c->d_of) (void)hipFree(c->d_of);
    if (c->d_bad) (void)hipFree(c->d_bad);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int bfir_convolver_cbufsize(const bfir_convolver *c) { return c ? c->N * c->s : 0; }

static int need_raw(bfir_convolver *c, size_t bytes)
{
    if (bytes <= c->raw_cap) return BFIR_OK;
    if (c->d_raw) (void)hipFree(c->d_raw);
    c->d_raw = nullptr; c->raw_cap = 0;
    HIP_TRY(hipMalloc(&c->d_raw, bytes));
    c->raw_cap = bytes;
    return BFIR_OK;
}

static int check_bf(const bfir_buffer_format *bf)
{
    if (!bf) return BFIR_ERR_ARG;
    // any of the eleven formats, described consistently with setup_sample_format
    // (brutefir.cpp:435-538): sbytes == bytes (no shift), swap set for the *_BE codes
    const FmtInfo fi = fmt_info(bf->sf.format);
    if (fi.bytes == 0 || bf->sf.bytes != fi.bytes || bf->sf.sbytes != fi.bytes || bf->sf.isfloat != fi.isfloat ||
        bf->sf.swap != fi.big_endian || bf->sample_spacing < 1 || bf->byte_offset < 0)
        return BFIR_ERR_UNSUPPORTED;
    return BFIR_OK;
}

extern "C" int bfir_convolver_raw2cbuf(bfir_convolver *c, const void *rawbuf, void *cbuf, void *next_cbuf,
                                       const bfir_buffer_format *bf)
{
    if (!c || !rawbuf || !cbuf || !next_cbuf) return BFIR_ERR_ARG;
    int rc = check_bf(bf);
    if (rc != BFIR_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const size_t span = ((size_t)(c->L - 1) * bf->sample_spacing + 1) * bf->sf.bytes;
    rc = need_raw(c, span);
    if (rc != BFIR_OK) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_raw, (const char *)rawbuf + bf->byte_offset, span, hipMemcpyHostToDevice, c->stream));
    StageInArgs a;
    a.raw = c->d_raw; a.eng_stride_bytes = 0; a.frame_off = 0;
    a.n_eng = 1; a.C = 1; a.raw_bytes = bf->sf.bytes; a.spacing = bf->sample_spacing; a.fmt = bf->sf.format;
    a.n_frames = c->L;
    a.dst = c->d[0]; a.dst_ch_stride = c->N; a.dst_off = 0;
    a.realsize = c->s;
    launch_stage_in(a, c->stream);
    const size_t half = (size_t)c->L * c->s;
    // next_cbuf[0..L) = samples; cbuf[L..2L) = the same (fftw_convolver.cpp:184)
    HIP_TRY(hipMemcpyAsync(next_cbuf, c->d[0], half, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync((char *)cbuf + half, c->d[0], half, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFIR_OK;
}

extern "C" int bfir_convolver_time2freq(bfir_convolver *c, const void *input_cbuf, void *output_cbuf)
{
    if (!c || !input_cbuf || !output_cbuf) return BFIR_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->d[0], input_cbuf, cb(c), hipMemcpyHostToDevice, c->stream));
    FwdArgs a;
    a.prev = c->d[0]; a.prev_ch_stride = 0;
    a.src = (char *)c->d[0] + (size_t)c->L * c->s; a.src_ch_stride = 0;
    a.dst = c->d[1]; a.dst_ch_stride = 0;
    a.ring = 1; a.base_slot = 0; a.n_t = 1; a.n_ch = 1;
    a.load_scale = 1.0; a.out_scale = 1.0; a.zero_first_half = 0;
    launch_fwd(c->plan, a, c->stream);
    launch_reorder(c->d[1], c->d[2], c->N, 1.0, 0, c->s, c->stream);   // grouped -> half-complex
    HIP_TRY(hipMemcpyAsync(output_cbuf, c->d[2], cb(c), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFIR_OK;
}

extern "C" int bfir_convolver_freq2time(bfir_convolver *c, const void *input_cbuf, void *output_cbuf)
{
    if (!c || !input_cbuf || !output_cbuf) return BFIR_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->d[0], input_cbuf, cb(c), hipMemcpyHostToDevice, c->stream));
    launch_reorder(c->d[0], c->d[1], c->N, 1.0, 1, c->s, c->stream);   // half-complex -> grouped
    InvArgs a;
    a.src = c->d[1]; a.src_ch_stride = 0; a.dst = c->d[2]; a.dst_ch_stride = 0;
    a.n_t = 1; a.n_ch = 1; a.in_scale = 1.0; a.full_output = 1;
    launch_inv(c->plan, a, c->stream);
    HIP_TRY(hipMemcpyAsync(output_cbuf, c->d[2], cb(c), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFIR_OK;
}

extern "C" int bfir_convolver_mixnscale(bfir_convolver *c, void *const *input_cbufs, void *output_cbuf,
                                        const double *scales, int n_bufs, int mixmode)
{
    if (!c || !input_cbufs || !output_cbuf || !scales) return BFIR_ERR_ARG;
    // INPUT_ADD is declared (fftw_convolver.hpp:15) but mixnscale has no case for it (:1421-1424)
    if (n_bufs < 1 || n_bufs > BFIR_MAX_MIX || (mixmode != BFIR_MIXMODE_INPUT && mixmode != BFIR_MIXMODE_OUTPUT)) {
        bfir_logf("Invalid mixmode: %d.\n", mixmode);
        return BFIR_ERR_UNSUPPORTED;
    }
    HIP_TRY(hipSetDevice(c->device));
    if (n_bufs == 1) {   // the engine only ever mixes one buffer (brutefir.cpp:273-277, 303-307)
        HIP_TRY(hipMemcpyAsync(c->d[0], input_cbufs[0], cb(c), hipMemcpyHostToDevice, c->stream));
        launch_reorder(c->d[0], c->d[1], c->N, scales[0], mixmode == BFIR_MIXMODE_INPUT, c->s, c->stream);
        HIP_TRY(hipMemcpyAsync(output_cbuf, c->d[1], cb(c), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return BFIR_OK;
    }
    // mixing matrix row: out = sum_i in_i * scale_i (:908-1156, :1187-1419)
    int rc = need_raw(c, (size_t)(n_bufs + 1) * cb(c));
    if (rc != BFIR_OK) return rc;
    MixArgs m;
    m.n = n_bufs;
    for (int i = 0; i < n_bufs; i++) {
        if (!input_cbufs[i]) return BFIR_ERR_ARG;
        m.in[i] = (char *)c->d_raw + (size_t)i * cb(c);
        m.scale[i] = scales[i];
        HIP_TRY(hipMemcpyAsync((void *)m.in[i], input_cbufs[i], cb(c), hipMemcpyHostToDevice, c->stream));
    }
    void *d_out = (char *)c->d_raw + (size_t)n_bufs * cb(c);
    launch_reorder_n(m, d_out, c->N, mixmode == BFIR_MIXMODE_INPUT, c->s, c->stream);
    HIP_TRY(hipMemcpyAsync(output_cbuf, d_out, cb(c), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFIR_OK;
}

static int cmul_call(bfir_convolver *c, const void *b, const void *coeffs, void *d, int mode)
{
    if (!c || !b || !coeffs || !d) return BFIR_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->d[0], b, cb(c), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d[1], coeffs, cb(c), hipMemcpyHostToDevice, c->stream));
    if (mode == 1) HIP_TRY(hipMemcpyAsync(c->d[2], d, cb(c), hipMemcpyHostToDevice, c->stream));
    launch_cmul_stage(c->d[0], c->d[1], c->d[2], c->N, mode, c->s, c->stream);
    HIP_TRY(hipMemcpyAsync(d, c->d[2], cb(c), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFIR_OK;
}

extern "C" int bfir_convolver_convolve_inplace(bfir_convolver *c, void *cbuf, const void *coeffs)
{
    return cmul_call(c, cbuf, coeffs, cbuf, 0);
}

extern "C" int bfir_convolver_convolve(bfir_convolver *c, const void *input_cbuf, const void *coeffs,
                                       void *output_cbuf)
{
    return cmul_call(c, input_cbuf, coeffs, output_cbuf, 0);
}

extern "C" int bfir_convolver_convolve_add(bfir_convolver *c, const void *input_cbuf, const void *coeffs,
                                           void *output_cbuf)
{
    return cmul_call(c, input_cbuf, coeffs, output_cbuf, 1);
}

extern "C" int bfir_convolver_cbuf2raw(bfir_convolver *c, const void *cbuf, void *outbuf,
                                       const bfir_buffer_format *bf, bfir_overflow *overflow)
{
    if (!c || !cbuf || !outbuf || !overflow) return BFIR_ERR_ARG;
    int rc = check_bf(bf);
    if (rc != BFIR_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const size_t span = ((size_t)(c->L - 1) * bf->sample_spacing + 1) * bf->sf.bytes;
    rc = need_raw(c, span);
    if (rc != BFIR_OK) return rc;
    char *dst = (char *)outbuf + bf->byte_offset;
    // the other channels' bytes inside the strided span must survive
    HIP_TRY(hipMemcpyAsync(c->d_raw, dst, span, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d[0], cbuf, (size_t)c->L * c->s, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_of, 0, sizeof(DevOverflow), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_bad, 0x7f, sizeof(int), c->stream));
    StageOutArgs a;
    a.raw = c->d_raw; a.eng_stride_bytes = 0; a.frame_off = 0;
    a.n_eng = 1; a.C = 1; a.raw_bytes = bf->sf.bytes; a.spacing = bf->sample_spacing; a.fmt = bf->sf.format;
    a.n_frames = c->L;
    a.src = c->d[0]; a.src_ch_stride = c->N;
    a.realsize = c->s; a.L = c->L; a.max = overflow->max;
    a.overflow = c->d_of; a.bad_block = c->d_bad; a.block_base = 0;
    launch_stage_out(a, c->stream);
    DevOverflow h;
    HIP_TRY(hipMemcpyAsync(dst, c->d_raw, span, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&h, c->d_of, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // fold into the caller's running bfoverflow_t (real2raw.cpp:321-336)
    overflow->n_overflows += h.n_overflows;
    double largest;
    if (c->s == 4) { unsigned int u = (unsigned int)h.largest_bits; float f; memcpy(&f, &u, 4); largest = (double)f; }
    else memcpy(&largest, &h.largest_bits, 8);
    if (largest > overflow->largest) overflow->largest = largest;
    if (h.intlargest > overflow->intlargest) overflow->intlargest = h.intlargest;   // dither.cpp:224-227, 246-249
    return BFIR_OK;
}

int bfir_dither_run_block(bfir_dither *d, int realsize, const void *d_src, void *d_raw, int fmt, int spacing, int n,
                          bfir_dither_state *state, bfir_overflow *overflow, hipStream_t s);   // dither.hip
int bfir_dither_realsize(const bfir_dither *d);                                                   // dither.hip

extern "C" int bfir_convolver_cbuf2raw_dither(bfir_convolver *c, bfir_dither *d, const void *cbuf, void *outbuf,
                                              const bfir_buffer_format *bf, bfir_dither_state *dither_state,
                                              bfir_overflow *overflow)
{
    if (!c || !cbuf || !outbuf || !overflow) return BFIR_ERR_ARG;
    int rc = check_bf(bf);
    if (rc != BFIR_OK) return rc;
    if (bf->sf.isfloat) return bfir_convolver_cbuf2raw(c, cbuf, outbuf, bf, overflow);   // fftw_convolver.cpp:421
    if (!d || !dither_state) { bfir_logf("Dither instance not set."); return BFIR_ERR_ARG; }   // :412-416
    // the reference builds both with one realsize (brutefir.cpp:709-719); a dither of the other precision would
    // read this convolver's samples as the wrong type
    if (bfir_dither_realsize(d) != c->s) { bfir_logf("Dither instance and convolver differ in realsize."); return BFIR_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    const size_t span = ((size_t)(c->L - 1) * bf->sample_spacing + 1) * bf->sf.bytes;
    rc = need_raw(c, span);
    if (rc != BFIR_OK) return rc;
    char *dst = (char *)outbuf + bf->byte_offset;
    HIP_TRY(hipMemcpyAsync(c->d_raw, dst, span, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d[0], cbuf, (size_t)c->L * c->s, hipMemcpyHostToDevice, c->stream));
    bfir_dither_preloop_real2int_hp_tpdf(d, dither_state, c->L);
    rc = bfir_dither_run_block(d, c->s, c->d[0], c->d_raw, bf->sf.format, bf->sample_spacing, c->L, dither_state, overflow,
                               c->stream);
    if (rc != BFIR_OK) return rc;
    HIP_TRY(hipMemcpyAsync(dst, c->d_raw, span, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFIR_OK;
}

extern "C" void *bfir_convolver_coeffs2cbuf(bfir_convolver *c, const void *coeffs, int n_coeffs,
                                            double scale, void *optional_dest)
{
    if (!c || (!coeffs && n_coeffs > 0) || n_coeffs < 0) return nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    const int len = n_coeffs > c->L ? c->L : n_coeffs;            // fftw_convolver.cpp:483
    std::vector<char> taps((size_t)c->L * c->s, 0);
    if (c->s == 4) {
        const float *src = (const float *)coeffs; const float sc = (float)scale;
        for (int n = 0; n < len; n++)
            if (!std::isfinite((double)(src[n] * sc))) { bfir_logf("NaN or Inf value among coefficients.\n"); return nullptr; }
    } else {
        const double *src = (const double *)coeffs;
        for (int n = 0; n < len; n++)
            if (!std::isfinite(src[n] * scale)) { bfir_logf("NaN or Inf value among coefficients.\n"); return nullptr; }
    }
    if (len > 0) memcpy(taps.data(), coeffs, (size_t)len * c->s);
    // window = [L zeros | taps]
    if (hipMemcpyAsync(c->d[0], taps.data(), taps.size(), hipMemcpyHostToDevice, c->stream) != hipSuccess)
        return nullptr;
    FwdArgs a;
    a.prev = nullptr; a.prev_ch_stride = 0;
    a.src = c->d[0]; a.src_ch_stride = 0; a.dst = c->d[1]; a.dst_ch_stride = 0;
    a.ring = 1; a.base_slot = 0; a.n_t = 1; a.n_ch = 1;
    a.load_scale = scale; a.out_scale = 1.0 / (double)c->N; a.zero_first_half = 1;
    launch_fwd(c->plan, a, c->stream);
    void *dest = optional_dest ? optional_dest : bfir_aligned_malloc(cb(c), 16);
    if (!dest) return nullptr;
    if (hipMemcpyAsync(dest, c->d[1], cb(c), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        if (!optional_dest) bfir_aligned_free(dest);
        return nullptr;
    }
    return dest;
}

// ---------------------------------------------------------------------------
// the methods of the class nothing in the tree calls
// ---------------------------------------------------------------------------
// convolver_runtime_coeffs2cbuf (fftw_convolver.cpp:539-567): n_fft2 taps, scale 1, into dest
extern "C" int bfir_convolver_runtime_coeffs2cbuf(bfir_convolver *c, const void *src, void *dest)
{
    if (!c || !src || !dest) return BFIR_ERR_ARG;
    return bfir_convolver_coeffs2cbuf(c, src, c->L, 1.0, dest) ? BFIR_OK : BFIR_ERR_COEFF;
}

// convolver_dirac_convolve / _inplace (:323-348)
extern "C" int bfir_convolver_dirac_convolve(bfir_convolver *c, const void *input_cbuf, void *output_cbuf)
{
    if (!c || !input_cbuf || !output_cbuf) return BFIR_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->d[0], input_cbuf, cb(c), hipMemcpyHostToDevice, c->stream));
    launch_dirac(c->d[0], c->d[1], c->N, c->s, c->stream);
    HIP_TRY(hipMemcpyAsync(output_cbuf, c->d[1], cb(c), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFIR_OK;
}

extern "C" int bfir_convolver_dirac_convolve_inplace(bfir_convolver *c, void *cbuf)
{
    return bfir_convolver_dirac_convolve(c, cbuf, cbuf);
}

// half-complex spectrum in d[src] -> time domain in d[dst] (HC2R, all N samples), via d[tmp]
static void hc2r_dev(bfir_convolver *c, int src, int tmp, void *dst)
{
    launch_reorder(c->d[src], c->d[tmp], c->N, 1.0, 1, c->s, c->stream);
    InvArgs a;
    a.src = c->d[tmp]; a.src_ch_stride = 0; a.dst = dst; a.dst_ch_stride = 0;
    a.n_t = 1; a.n_ch = 1; a.in_scale = 1.0; a.full_output = 1;
    launch_inv(c->plan, a, c->stream);
}

// time domain at `src` (N samples) -> half-complex spectrum in d[dst], via d[tmp]
static void r2hc_dev(bfir_convolver *c, const void *src, int tmp, int dst)
{
    FwdArgs a;
    a.prev = src; a.prev_ch_stride = 0;
    a.src = (const char *)src + (size_t)c->L * c->s; a.src_ch_stride = 0;
    a.dst = c->d[tmp]; a.dst_ch_stride = 0;
    a.ring = 1; a.base_slot = 0; a.n_t = 1; a.n_ch = 1;
    a.load_scale = 1.0; a.out_scale = 1.0; a.zero_first_half = 0;
    launch_fwd(c->plan, a, c->stream);
    launch_reorder(c->d[tmp], c->d[dst], c->N, 1.0, 0, c->s, c->stream);
}

// convolver_convolve_eval (:377-403).  buffer_cbuf: 1.5 cbufs, cleared before the first call.
extern "C" int bfir_convolver_convolve_eval(bfir_convolver *c, const void *input_cbuf, void *buffer_cbuf,
                                            void *output_cbuf)
{
    if (!c || !input_cbuf || !buffer_cbuf || !output_cbuf) return BFIR_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    const size_t half = (size_t)c->L * c->s;
    int rc = need_raw(c, 3 * half);
    if (rc != BFIR_OK) return rc;
    char *dbuf = (char *)c->d_raw;                                   // the 1.5-cbuf work buffer on the device
    HIP_TRY(hipMemcpyAsync(dbuf, buffer_cbuf, half, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d[0], input_cbuf, cb(c), hipMemcpyHostToDevice, c->stream));
    hc2r_dev(c, 0, 1, dbuf + half);                                  // HC2R into buffer[n_fft2 ..]
    r2hc_dev(c, dbuf, 1, 2);                                         // R2HC of buffer[0 .. n_fft)
    HIP_TRY(hipMemcpyAsync(output_cbuf, c->d[2], cb(c), hipMemcpyDeviceToHost, c->stream));
    // the reference leaves [new first half | rest of the HC2R output] in the buffer (:401-402)
    HIP_TRY(hipMemcpyAsync(buffer_cbuf, dbuf + half, half, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync((char *)buffer_cbuf + half, dbuf + half, 2 * half, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFIR_OK;
}

// convolver_crossfade_inplace (:275-321).  buffer_cbuf: 1.5 cbufs (the double path reads
// buffer[n_fft ..], exactly as the reference does).
extern "C" int bfir_convolver_crossfade_inplace(bfir_convolver *c, void *input_cbuf, void *crossfade_cbuf,
                                                void *buffer_cbuf)
{
    if (!c || !input_cbuf || !crossfade_cbuf || !buffer_cbuf) return BFIR_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    const size_t half = (size_t)c->L * c->s;
    int rc = need_raw(c, 5 * half);
    if (rc != BFIR_OK) return rc;
    char *d_cf = (char *)c->d_raw, *d_buf = d_cf + 2 * half;          // crossfade (1 cbuf), buffer (1.5 cbufs)
    HIP_TRY(hipMemcpyAsync(d_buf + 2 * half, (char *)buffer_cbuf + 2 * half, half, hipMemcpyHostToDevice, c->stream));
    // OUTPUT reorder + HC2R of the crossfade spectrum, then of the input spectrum (:289-294)
    HIP_TRY(hipMemcpyAsync(c->d[0], crossfade_cbuf, cb(c), hipMemcpyHostToDevice, c->stream));
    launch_reorder(c->d[0], c->d[1], c->N, 1.0, 0, c->s, c->stream);
    hc2r_dev(c, 1, 2, d_cf);
    HIP_TRY(hipMemcpyAsync(c->d[0], input_cbuf, cb(c), hipMemcpyHostToDevice, c->stream));
    launch_reorder(c->d[0], c->d[1], c->N, 1.0, 0, c->s, c->stream);
    hc2r_dev(c, 1, 2, d_buf);
    launch_crossfade_blend(d_cf, d_buf, d_buf + 2 * half, c->L, c->s, c->stream);   // :296-315
    r2hc_dev(c, d_buf, 1, 2);                                                        // :317
    launch_reorder(c->d[2], c->d[0], c->N, 1.0 / (double)c->N, 1, c->s, c->stream);  // :318-320
    HIP_TRY(hipMemcpyAsync(input_cbuf, c->d[0], cb(c), hipMemcpyDeviceToHost, c->stream));
    // what the reference leaves behind in its scratch arguments
    HIP_TRY(hipMemcpyAsync(crossfade_cbuf, d_cf, cb(c), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(buffer_cbuf, c->d[2], cb(c), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFIR_OK;
}

// convolver_verify_cbuf (:569-602): 1 when every value is finite, 0 otherwise
extern "C" int bfir_convolver_verify_cbuf(bfir_convolver *c, void *const *cbufs, int n_cbufs)
{
    if (!c || !cbufs || n_cbufs < 0) return BFIR_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->d_bad, 0, sizeof(int), c->stream));
    for (int n = 0; n < n_cbufs; n++) {
        if (!cbufs[n]) return BFIR_ERR_ARG;
        HIP_TRY(hipMemcpyAsync(c->d[0], cbufs[n], cb(c), hipMemcpyHostToDevice, c->stream));
        launch_check_finite(c->d[0], c->N, c->s, c->d_bad, c->stream);
    }
    int bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, c->d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (bad) bfir_logf("NaN or Inf value among coefficients.\n");
    return bad ? 0 : 1;
}

// convolver_debug_dump_cbuf (:604-651): every cbuf back to its coefficient list (OUTPUT reorder with scale 1,
// HC2R in place; the taps are the UPPER half) as one "%.16e" line per value.  The reorder pair
// grouped -> half-complex -> grouped of the stage kernels is a multiplication by 1.0 each way, so the
// inverse kernel takes the cbuf as it is.  A file that cannot be opened is logged and skipped there
// (a void method); here that is BFIR_ERR_IO as well.
extern "C" int bfir_convolver_debug_dump_cbuf(bfir_convolver *c, const char *filename, void *const *cbufs, int n_cbufs)
{
    if (!c || !filename || !cbufs || n_cbufs < 0) return BFIR_ERR_ARG;
    for (int n = 0; n < n_cbufs; n++) if (!cbufs[n]) return BFIR_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    FILE *stream = fopen(filename, "wt+");
    if (!stream) {
        bfir_logf("Could not open \"%s\" for writing: %s", filename, strerror(errno));
        return BFIR_ERR_IO;
    }
    std::vector<char> vals((size_t)c->L * c->s);
    int rc = BFIR_OK;
    for (int n = 0; n < n_cbufs && rc == BFIR_OK; n++) {
        InvArgs a;
        a.src = c->d[0]; a.src_ch_stride = 0; a.dst = c->d[2]; a.dst_ch_stride = 0;
        a.n_t = 1; a.n_ch = 1; a.in_scale = 1.0; a.full_output = 1;
        if (hipMemcpyAsync(c->d[0], cbufs[n], cb(c), hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = BFIR_ERR_HIP; break; }
        launch_inv(c->plan, a, c->stream);
        if (hipMemcpyAsync(vals.data(), (char *)c->d[2] + vals.size(), vals.size(), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) { rc = BFIR_ERR_HIP; break; }
        for (int i = 0; i < c->L; i++)
            fprintf(stream, "%.16e\n", c->s == 4 ? (double)((const float *)vals.data())[i] : ((const double *)vals.data())[i]);
    }
    fclose(stream);
    return rc;
}
