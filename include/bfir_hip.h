/*
 * bfir_hip.h -- C ABI of the MI355X (gfx950) partitioned-FIR convolution engine.
 *
 * This is the drop-in boundary for the hot path of vsu/foo-dsp-bfir: the
 * entry points are what a binding of the reference's convolver path would
 * bind.  Plain pointers and sizes only; no C++ or torch types.  Citations are
 * path:line in the reference tree (brutefir/...).
 *
 * Two levels:
 *   bfir_engine_*     one call = brutefir::run for n consecutive blocks
 *                     (brutefir/brutefir.cpp:244-343), fused on the GPU with
 *                     the partition spectra and the delay line resident in
 *                     HBM.  This is the measured path.
 *   bfir_convolver_*  one call = one fftw_convolver method
 *                     (brutefir/fftw_convolver.hpp:28-166) on host buffers,
 *                     run by the same kernels.  Plumbing/parity path for
 *                     callers that keep the reference's per-stage sequence.
 *
 * Library: libbfir_hip.so (hipcc --offload-arch=gfx950).  There is no CPU
 * fallback: without a HIP device every create call fails with
 * BFIR_ERR_NO_DEVICE.
 */
#ifndef BFIR_HIP_H
#define BFIR_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* limits and sample format codes: brutefir/global.h:21-34 */
#define BFIR_MAXCHANNELS 8
#define BFIR_SAMPLE_FORMAT_S8 1
#define BFIR_SAMPLE_FORMAT_S16_LE 2
#define BFIR_SAMPLE_FORMAT_S16_BE 3
#define BFIR_SAMPLE_FORMAT_S24_LE 4
#define BFIR_SAMPLE_FORMAT_S24_BE 5
#define BFIR_SAMPLE_FORMAT_S32_LE 6
#define BFIR_SAMPLE_FORMAT_S32_BE 7
#define BFIR_SAMPLE_FORMAT_FLOAT_LE 8
#define BFIR_SAMPLE_FORMAT_FLOAT_BE 9
#define BFIR_SAMPLE_FORMAT_FLOAT64_LE 10
#define BFIR_SAMPLE_FORMAT_FLOAT64_BE 11

/* mixmodes: brutefir/fftw_convolver.hpp:14-16 */
#define BFIR_MIXMODE_INPUT 1
#define BFIR_MIXMODE_INPUT_ADD 2
#define BFIR_MIXMODE_OUTPUT 3

/* error codes (negative).  -1 and -2 keep the reference's meaning where a
 * reference function returns them (run: -1, set_coeff: -2). */
#define BFIR_OK 0
#define BFIR_ERR_NONFINITE (-1)   /* brutefir::run, brutefir.cpp:316-321 */
#define BFIR_ERR_COEFF (-2)       /* brutefir::set_coeff, brutefir.cpp:217-222 */
#define BFIR_ERR_ARG (-3)
#define BFIR_ERR_NO_DEVICE (-4)
#define BFIR_ERR_HIP (-5)
#define BFIR_ERR_STATE (-6)       /* engine not initialised (no coefficients) */
#define BFIR_ERR_UNSUPPORTED (-7) /* sample format / size outside this build */
#define BFIR_ERR_IO (-8)          /* a file could not be opened (convolver_debug_dump_cbuf) */

/* bfoverflow_t, brutefir/global.h:96-102 (same layout) */
typedef struct bfir_overflow {
    unsigned int n_overflows;
    int32_t intlargest;
    double largest;
    double max;
} bfir_overflow;

/* sample_format_t + buffer_format_t, brutefir/global.h:39-54 (same layout) */
typedef struct bfir_sample_format {
    bool isfloat;
    bool swap;
    int bytes;
    int sbytes;
    double scale;
    int format;
} bfir_sample_format;

typedef struct bfir_buffer_format {
    bfir_sample_format sf;
    int sample_spacing; /* in samples */
    int byte_offset;    /* in bytes */
} bfir_buffer_format;

/* log callback, same shape as pinfo's (brutefir/pinfo.c:17-39) */
typedef void (*bfir_log_fn)(const char *msg);
void bfir_set_log_callback(bfir_log_fn fn);

const char *bfir_strerror(int err);
int bfir_device_count(void);
const char *bfir_version(void);

/* ------------------------------------------------------------------ */
/* engine level                                                        */
/* ------------------------------------------------------------------ */
typedef struct bfir_engine bfir_engine;

/* brutefir::brutefir (brutefir/brutefir.hpp:18-25, brutefir.cpp:21-44).
 * filter_length: partition length L (power of two, 16..16384; 8192 max for
 * realsize 8); filter_blocks: B; realsize: 4 or 8; channels: 1..8;
 * in/out_format: any BFIR_SAMPLE_FORMAT_* code (FLOAT_LE / FLOAT64_LE take the
 * vectorised staging kernels, the others a byte-wise one); apply_dither acts on
 * integer output formats only (fftw_convolver.cpp:421, 444): HP-TPDF dither with
 * the reference's random table (dither.cpp:21-110; sampling_rate sizes it) and
 * error feedback, one lane per channel.  device: HIP ordinal.
 * Returns NULL and sets *err on failure. */
bfir_engine *bfir_engine_create(int filter_length, int filter_blocks, int realsize, int channels,
                                int in_format, int out_format, int sampling_rate, int apply_dither,
                                int device, int *err);

/* n_engines independent, identically shaped engines run by shared launches
 * (BASELINE.json configs[3]).  Engine e owns global channels e*C .. e*C+C-1. */
bfir_engine *bfir_engine_create_batch(int n_engines, int filter_length, int filter_blocks,
                                      int realsize, int channels, int in_format, int out_format,
                                      int sampling_rate, int apply_dither, int device, int *err);

void bfir_engine_destroy(bfir_engine *e);

/* brutefir::is_initialized */
int bfir_engine_is_initialized(const bfir_engine *e);

/* brutefir::set_coeff(void **coeffs, int n_coeffs, int length,
 * int coeff_blocks, double scale)  (brutefir.cpp:179-228) with
 * coeff::preprocess_coeff (coeff.cpp:292-354) and convolver_coeffs2cbuf
 * (fftw_convolver.cpp:474-537) done on the device.  coeffs[n]: host array of
 * `length` taps in working precision (float for realsize 4, double for 8).
 * Returns 0, or BFIR_ERR_COEFF on a NaN/Inf tap. */
int bfir_engine_set_coeff(bfir_engine *e, const void *const *coeffs, int n_coeffs, int length,
                          int coeff_blocks, double scale);
/* the same for engine `engine_index` of a batch */
int bfir_engine_set_coeff_at(bfir_engine *e, int engine_index, const void *const *coeffs,
                             int n_coeffs, int length, int coeff_blocks, double scale);

/* brutefir::run (brutefir.cpp:244-343) for n_blocks consecutive blocks.
 * inbuf/outbuf: HOST memory, n_blocks * filter_length interleaved frames in
 * the input/output format (for a batch: engine after engine, each
 * n_blocks*L frames).  Staged through pinned buffers with hipMemcpyAsync on
 * side streams, overlapped with compute.  Returns 0 or BFIR_ERR_NONFINITE. */
int bfir_engine_run(bfir_engine *e, const void *inbuf, void *outbuf, int n_blocks);

/* The same on DEVICE memory, asynchronous on `hip_stream` (a hipStream_t;
 * NULL = the engine's own stream).  Engine k reads d_in + k*in_stride_bytes
 * and writes d_out + k*out_stride_bytes.  The NaN verdict is delivered by
 * bfir_engine_sync.  Strides, counts and lengths that can pass 2^31 are int64_t,
 * never `long`: the reference's platform is MSVC (brutefir/brutefir.vcxproj:66-70),
 * where `long` has 32 bits, and one 8-channel stream of a day's audio is a 32 GiB stride. */
int bfir_engine_run_device(bfir_engine *e, const void *d_in, int64_t in_stride_bytes, void *d_out,
                           int64_t out_stride_bytes, int n_blocks, void *hip_stream);
/* Wait for all queued work; 0, or BFIR_ERR_NONFINITE if any block since the
 * last sync produced a non-finite first sample. */
int bfir_engine_sync(bfir_engine *e);

/* brutefir::reset (brutefir.cpp:346-367) */
void bfir_engine_reset(bfir_engine *e);

/* copy of brutefir's overflow[channel] (brutefir.cpp:326-334) */
int bfir_engine_get_overflow(bfir_engine *e, int channel, bfir_overflow *of);

/* tuning: blocks per launch; 0 (the default) = automatic: the work of 4096 blocks of the 8-channel, 4096-sample
 * headline shape (4096 ... 32768 blocks), less when the delay line of that many blocks would pass 4 GiB, at most 512
 * on the host-pointer path; takes effect on the next run */
int bfir_engine_set_chunk(bfir_engine *e, int blocks_per_launch);

/* per-kernel timing with HIP events on the stream the kernels run on */
enum { BFIR_K_STAGE_IN = 0, BFIR_K_FWD = 1, BFIR_K_MAC = 2, BFIR_K_INV = 3, BFIR_K_STAGE_OUT = 4,
       BFIR_K_COUNT = 5 };
int bfir_engine_set_profiling(bfir_engine *e, int enable);
int bfir_engine_get_profile(bfir_engine *e, int kernel, double *total_ms, int64_t *launches);

/* copy partition spectrum `block` of global channel `channel` to host (n_fft reals) */
int bfir_engine_read_coeff(bfir_engine *e, int channel, int block, void *dst);

/* ------------------------------------------------------------------ */
/* stage level: fftw_convolver methods on host buffers                 */
/* ------------------------------------------------------------------ */
typedef struct bfir_convolver bfir_convolver;

/* fftw_convolver::fftw_convolver (fftw_convolver.cpp:51-138) */
bfir_convolver *bfir_convolver_create(int length, int realsize, int device, int *err);
void bfir_convolver_destroy(bfir_convolver *c);
/* convolver_cbufsize (:468-472) */
int bfir_convolver_cbufsize(const bfir_convolver *c);
/* convolver_raw2cbuf (:156-185); float formats without byte swap */
int bfir_convolver_raw2cbuf(bfir_convolver *c, const void *rawbuf, void *cbuf, void *next_cbuf,
                            const bfir_buffer_format *bf);
/* convolver_time2freq (:187-212): FFTW_R2HC, half-complex output */
int bfir_convolver_time2freq(bfir_convolver *c, const void *input_cbuf, void *output_cbuf);
/* convolver_mixnscale (:214-229) */
int bfir_convolver_mixnscale(bfir_convolver *c, void *const *input_cbufs, void *output_cbuf,
                             const double *scales, int n_bufs, int mixmode);
/* convolver_convolve_inplace / convolve / convolve_add (:231-273) */
int bfir_convolver_convolve_inplace(bfir_convolver *c, void *cbuf, const void *coeffs);
int bfir_convolver_convolve(bfir_convolver *c, const void *input_cbuf, const void *coeffs,
                            void *output_cbuf);
int bfir_convolver_convolve_add(bfir_convolver *c, const void *input_cbuf, const void *coeffs,
                                void *output_cbuf);
/* convolver_freq2time (:350-375): FFTW_HC2R */
int bfir_convolver_freq2time(bfir_convolver *c, const void *input_cbuf, void *output_cbuf);
/* convolver_cbuf2raw (:405-466) with apply_dither false (or a float format): any sample format */
int bfir_convolver_cbuf2raw(bfir_convolver *c, const void *cbuf, void *outbuf,
                            const bfir_buffer_format *bf, bfir_overflow *overflow);

/* class dither (brutefir/dither.hpp:13-77, dither.cpp) and dither_state_t (global.h:63-69, same
 * layout).  The constructor fills dither_state[0 .. n_channels) as the reference's does. */
typedef struct bfir_dither bfir_dither;
typedef struct bfir_dither_state {
    int randtab_ptr;
    int8_t *randtab;
    float sf[2];
    double sd[2];
} bfir_dither_state;
bfir_dither *bfir_dither_create(int n_channels, int sample_rate, int realsize, int max_size,
                                int max_samples_per_loop, bfir_dither_state *dither_state, int device, int *err);
void bfir_dither_destroy(bfir_dither *d);
int bfir_dither_table_size(const bfir_dither *d);
const int8_t *bfir_dither_table(const bfir_dither *d);   /* host copy of dither_randtab */
/* dither::dither_preloop_real2int_hp_tpdf (dither.cpp:127-139) */
void bfir_dither_preloop_real2int_hp_tpdf(bfir_dither *d, bfir_dither_state *state, int samples_per_loop);
/* convolver_cbuf2raw with apply_dither true on an integer format (:421-431, :444-454): the preloop
 * for n_fft2 samples, then real2raw{f,d}_hp_tpdf with the caller's dither_state and overflow */
int bfir_convolver_cbuf2raw_dither(bfir_convolver *c, bfir_dither *d, const void *cbuf, void *outbuf,
                                   const bfir_buffer_format *bf, bfir_dither_state *dither_state,
                                   bfir_overflow *overflow);
/* convolver_coeffs2cbuf (:474-537).  Returns optional_dest, or (when it is
 * NULL) a 16-byte aligned host block the CALLER frees with bfir_aligned_free
 * (the reference caller frees it with _aligned_free, brutefir.cpp:844-854);
 * NULL on a NaN/Inf tap. */
void *bfir_convolver_coeffs2cbuf(bfir_convolver *c, const void *coeffs, int n_coeffs, double scale,
                                 void *optional_dest);
/* Methods of the class that nothing in the reference tree calls.
 * convolver_mixnscale above takes any n_bufs <= 32 (mixing matrix rows,
 * :908-1156, :1187-1419). */
/* convolver_runtime_coeffs2cbuf (:539-567): n_fft2 taps at src -> spectrum at dest */
int bfir_convolver_runtime_coeffs2cbuf(bfir_convolver *c, const void *src, void *dest);
/* convolver_dirac_convolve / _inplace (:323-348) */
int bfir_convolver_dirac_convolve(bfir_convolver *c, const void *input_cbuf, void *output_cbuf);
int bfir_convolver_dirac_convolve_inplace(bfir_convolver *c, void *cbuf);
/* convolver_convolve_eval (:377-403); buffer_cbuf is 1.5 cbufs, zeroed before the first call */
int bfir_convolver_convolve_eval(bfir_convolver *c, const void *input_cbuf, void *buffer_cbuf,
                                 void *output_cbuf);
/* convolver_crossfade_inplace (:275-321); buffer_cbuf is 1.5 cbufs */
int bfir_convolver_crossfade_inplace(bfir_convolver *c, void *input_cbuf, void *crossfade_cbuf,
                                     void *buffer_cbuf);
/* convolver_verify_cbuf (:569-602): 1 = all finite, 0 = NaN/Inf found, < 0 = error */
int bfir_convolver_verify_cbuf(bfir_convolver *c, void *const *cbufs, int n_cbufs);
/* convolver_debug_dump_cbuf (:604-651): every cbuf converted back to its coefficient list and written as one
 * "%.16e" line per tap (n_fft2 lines per cbuf).  BFIR_ERR_IO when the file cannot be opened (the reference
 * logs and returns). */
int bfir_convolver_debug_dump_cbuf(bfir_convolver *c, const char *filename, void *const *cbufs, int n_cbufs);
/* ------------------------------------------------------------------ */
/* FFT plans of any power-of-two size and the equalizer render          */
/*                                                     */
/* ------------------------------------------------------------------ */
typedef struct bfir_fft_plan bfir_fft_plan;

/* fftw_convolver::create_fft_plan(order, invert, inplace) (fftw_convolver.cpp:653-675):
 * FFTW_R2HC (invert 0) or FFTW_HC2R (invert 1) of 2^order reals, 1 <= order <= 25.
 * Sizes beyond one workgroup's LDS run as a four-step FFT, sizes below 32 reals as direct sums. */
bfir_fft_plan *bfir_fft_plan_create(int order, int invert, int inplace, int realsize, int device, int *err);
void bfir_fft_plan_destroy(bfir_fft_plan *p);
/* fftw[f]_execute_r2r(plan, in, out) on host buffers of 2^order reals; in == out allowed
 * (replaces the direct FFTW calls at equalizer.cpp:262, 357). */
int bfir_fft_plan_execute(bfir_fft_plan *p, const void *in, void *out);
int64_t bfir_fft_plan_length(const bfir_fft_plan *p);
/* equalizer::render_f / render_d (equalizer.cpp:211-299, 301-394): band tables as
 * equalizer::generate leaves them (:113-118) -> taps/2-sample impulse response in ir_out.
 * ifftplan: an HC2R plan of `taps` reals. */
int bfir_equalizer_render(bfir_fft_plan *ifftplan, int band_count, const double *freq, const double *mag,
                          const double *phase, void *ir_out);

/* ------------------------------------------------------------------ */
/* fftw_convolver::convolver_td_* (fftw_convolver.hpp:157-166): the     */
/* one-block convolver of the reference's delay class (delay.cpp:174)   */
/* ------------------------------------------------------------------ */
typedef struct bfir_td_conv bfir_td_conv;
/* convolver_td_block_length (fftw_convolver.cpp:697-706): n_coeffs rounded up to a power of two; -1 for
 * n_coeffs < 2 (the reference's log2_roof(1) is -1 and it shifts by it: undefined there, refused here) */
int bfir_td_block_length(int n_coeffs);
/* convolver_td_new (:708-757): the spectrum of [blocklen zeros | taps | zeros], times 1 / (2 blocklen),
 * resident on the device (td_conv_t; bfir_td_coeffs returns the host copy of td_conv_t.coeffs,
 * 2 * blocklen reals in FFTW's half-complex order).  The reference never frees a td_conv_t. */
bfir_td_conv *bfir_td_new(const void *coeffs, int n_coeffs, int realsize, int device, int *err);
void bfir_td_destroy(bfir_td_conv *tdc);
int bfir_td_blocklen(const bfir_td_conv *tdc);
const void *bfir_td_coeffs(const bfir_td_conv *tdc);
/* convolver_td_convolve (:759-777): R2HC, convolve_inplace_ordered (:819-856), HC2R in place on the
 * caller's 2 * blocklen reals */
int bfir_td_convolve(bfir_td_conv *tdc, void *overlap_block);

/* Page-locked host memory for frame buffers handed to bfir_engine_run: the copy engines read and write it directly, so the
 * staging memcpy through the engine's own pinned buffers is skipped (the reference's callers allocate their frame buffers
 * with _aligned_realloc, foo_dsp_bfir.cpp:291-293; any hipHostMalloc / hipHostRegister'ed buffer is recognised the same way).
 * NULL without a GPU. */
void *bfir_pinned_malloc(size_t size);
void bfir_pinned_free(void *p);

void *bfir_aligned_malloc(size_t size, size_t alignment);
void bfir_aligned_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
