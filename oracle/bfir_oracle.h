/*
 * bfir_oracle.h -- CPU oracle for the partitioned-FIR hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the algorithm
 * of the reference (vsu/foo-dsp-bfir, brutefir/) used as the *checker* for
 * the HIP engine.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (libbfir_hip.so and the host
 * layer above it) never links, loads or calls anything in oracle/.
 *
 * PARITY UNPINNED.  The reference ships no tests, golden vectors or
 * fixtures for this path, its own sources cannot be
 * built in this image without stand-ins for MSVC/Win32 headers and for the
 * FFTW 3.3-beta1 library (only Win32 DLLs are in the tree), so this oracle is
 * not checked against outputs of the reference itself.  What pins it instead:
 * an independent long-double direct-form convolution (orc_direct_conv) and
 * scipy's pocketfft (tests/test_oracle.py).  The one source file of the path
 * that does compile here, brutefir/hash.c (the DJB hash that names the cache
 * WAVs), is built in place by oracle/Makefile's `ref` target into
 * oracle/_ref/ and pins the host mirrors' hash (tests/golden/
 * djb_hash_ref.json); no arithmetic of this oracle is pinned by it.
 *
 * Third-party arithmetic restated: FFTW 3.3-beta1 r2r transforms FFTW_R2HC
 * and FFTW_HC2R (call sites brutefir/fftw_convolver.cpp:204-209, 367-372,
 * 500-517, 798-806).  Published definition (FFTW manual, "The Halfcomplex-
 * format DFT"): unnormalised; R2HC output is r0, r1, ..., r_{n/2},
 * i_{(n+1)/2-1}, ..., i_1 with X_k = sum_j x_j exp(-2 pi i j k / n);
 * HC2R is its unnormalised inverse (HC2R(R2HC(x)) = n x).
 */
#ifndef BFIR_ORACLE_H
#define BFIR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* sample format codes: brutefir/global.h:24-34 */
#define ORC_FMT_FLOAT_LE   8
#define ORC_FMT_FLOAT64_LE 10

/* brutefir/fftw_convolver.hpp:14-16 */
#define ORC_MIXMODE_INPUT  1
#define ORC_MIXMODE_OUTPUT 3

/* brutefir/global.h:96-102 */
typedef struct {
    unsigned int n_overflows;
    int32_t intlargest;
    double largest;
    double max;
} orc_overflow_t;

/* ---- stage level: one function per reference loop (realsize 4 / 8) ---- */

/* FFTW_R2HC / FFTW_HC2R of length n (power of two >= 4), out of place or in place. */
void orc_r2hc_f(int n, const float *in, float *out);
void orc_hc2r_f(int n, const float *in, float *out);
void orc_r2hc_d(int n, const double *in, double *out);
void orc_hc2r_d(int n, const double *in, double *out);

/* raw2real::raw2realf/d, float formats without byte swap
 * (brutefir/raw2real.cpp:52-55, 70-73, 258-261, 274-277). raw_bytes = 4 or 8. */
void orc_raw2real_f(float *real, const void *raw, int raw_bytes, int spacing, int n);
void orc_raw2real_d(double *real, const void *raw, int raw_bytes, int spacing, int n);

/* convolver_raw2cbuf (brutefir/fftw_convolver.cpp:156-185). */
void orc_raw2cbuf_f(int n_fft2, const void *raw, int byte_offset, int raw_bytes, int spacing,
                    float *cbuf, float *next_cbuf);
void orc_raw2cbuf_d(int n_fft2, const void *raw, int byte_offset, int raw_bytes, int spacing,
                    double *cbuf, double *next_cbuf);

/* mixnscale, n_bufs == 1 (brutefir/fftw_convolver.cpp:883-907, 1163-1186,
 * 1583-1607, 1860-1883). */
void orc_mixnscale_f(int n_fft, const float *in, float *out, double scale, int mixmode);
void orc_mixnscale_d(int n_fft, const double *in, double *out, double scale, int mixmode);

/* convolve_inplace / convolve / convolve_add
 * (brutefir/fftw_convolver.cpp:1429-1525, 2125-2220). */
void orc_convolve_inplace_f(int n_fft, float *cbuf, const float *coeffs);
void orc_convolve_f(int n_fft, const float *in, const float *coeffs, float *out);
void orc_convolve_add_f(int n_fft, const float *in, const float *coeffs, float *out);
void orc_convolve_inplace_d(int n_fft, double *cbuf, const double *coeffs);
void orc_convolve_d(int n_fft, const double *in, const double *coeffs, double *out);
void orc_convolve_add_d(int n_fft, const double *in, const double *coeffs, double *out);

/* convolver_coeffs2cbuf (brutefir/fftw_convolver.cpp:474-537): dest holds
 * n_fft reals.  Returns 0, or -1 for a non-finite tap. */
int orc_coeffs2cbuf_f(int n_fft2, const float *coeffs, int n_coeffs, double scale, float *dest);
int orc_coeffs2cbuf_d(int n_fft2, const double *coeffs, int n_coeffs, double scale, double *dest);

/* real2raw{f,d}_no_dither, float formats without byte swap
 * (brutefir/real2raw.cpp:321-336, 365-420, 924-1016). */
void orc_real2raw_f(void *raw, const float *real, int raw_bytes, int spacing, int n,
                    orc_overflow_t *of);
void orc_real2raw_d(void *raw, const double *real, int raw_bytes, int spacing, int n,
                    orc_overflow_t *of);

/* ---- all eleven sample formats ----
 * Format table of brutefir::setup_sample_format (brutefir/brutefir.cpp:435-538). */
int orc_fmt_bytes(int fmt);
int orc_fmt_isfloat(int fmt);
int orc_fmt_swap(int fmt);
double orc_fmt_in_scale(int fmt);   /* normalised: 1 / 2^(bits-1) for integers */
double orc_fmt_out_scale(int fmt);  /* full scale: 2^(bits-1) for integers     */
double orc_fmt_max(int fmt);        /* bfoverflow_t.max (brutefir.cpp:672-684) */
/* raw2real{f,d} and real2raw{f,d}_no_dither for any format (raw2real.cpp:30-424,
 * real2raw.cpp:342-1224 with dither.cpp:196-262, 346-416).  `raw` points at the
 * channel's first sample; spacing in samples. */
void orc_raw2real_fmt_f(float *real, const void *raw, int fmt, int spacing, int n);
void orc_raw2real_fmt_d(double *real, const void *raw, int fmt, int spacing, int n);
void orc_real2raw_fmt_f(void *raw, const float *real, int fmt, int spacing, int n, orc_overflow_t *of);
void orc_real2raw_fmt_d(void *raw, const double *real, int fmt, int spacing, int n, orc_overflow_t *of);

/* ---- HP-TPDF dither on integer outputs: brutefir/dither.cpp ----
 * orc_dither restates class dither: the table of int8 random numbers drawn from the combined
 * Tausworthe generator seeded with tausinit(state, 0) (dither.cpp:21-110, 419-449), the
 * difference -> float map (:73-104) and the per-channel state dither_state_t (global.h:63-69:
 * table position, two error-feedback memories).  Nothing here reads a reference output: the
 * generator is the published GSL "taus" recurrence the reference cites, restated from
 * dither.cpp's macros; PARITY UNPINNED like the rest of this oracle.
 * One quirk is undefined in the reference and defined here: the map has entries -256 .. 254,
 * but the difference of two int8 values reaches +255 (127 - (-128)), where the reference reads
 * one element past its 511-entry allocation (dither.cpp:176-178).  Oracle and HIP engine both
 * use the table formula's continuation there, 0.5 + 1/255 + 255/255. */
typedef struct orc_dither orc_dither;
/* dither::dither(n_channels, sample_rate, realsize, max_size, max_samples_per_loop, state[])
 * (dither.cpp:21-110).  NULL where the reference throws (table budget too small). */
orc_dither *orc_dither_create(int n_channels, int sample_rate, int realsize, int max_size,
                              int max_samples_per_loop);
void orc_dither_destroy(orc_dither *d);
int orc_dither_table_size(const orc_dither *d);
const int8_t *orc_dither_table(const orc_dither *d);
int orc_dither_randtab_ptr(const orc_dither *d, int channel);
/* convolver_cbuf2raw with apply_dither on an integer format (fftw_convolver.cpp:421-431, 444-454):
 * dither_preloop_real2int_hp_tpdf (dither.cpp:127-139) for n samples of `channel`, then
 * real2raw{f,d}_hp_tpdf (real2raw.cpp:38-317, 422-...) with dither{f,d}_real2int_hp_tpdf
 * (dither.cpp:141-194, 264-344) per sample. */
void orc_real2raw_hp_tpdf_f(orc_dither *d, int channel, void *raw, const float *real, int fmt, int spacing,
                            int n, orc_overflow_t *of);
void orc_real2raw_hp_tpdf_d(orc_dither *d, int channel, void *raw, const double *real, int fmt, int spacing,
                            int n, orc_overflow_t *of);

/* ---- boundary features nothing in the tree calls ---- */
/* mixnscale with n_bufs >= 1 (fftw_convolver.cpp:908-1156, 1187-1419). */
void orc_mixnscale_n_f(int n_fft, const float *const *ins, float *out, const double *scales, int n_bufs, int mixmode);
void orc_mixnscale_n_d(int n_fft, const double *const *ins, double *out, const double *scales, int n_bufs, int mixmode);
/* dirac_convolve (:323-348, 1527-1556, 2222-2251); in == out allowed. */
void orc_dirac_convolve_f(int n_fft, const float *in, float *out);
void orc_dirac_convolve_d(int n_fft, const double *in, double *out);
/* convolver_convolve_eval (:377-403); buffer: 1.5 n_fft reals, kept between calls. */
void orc_convolve_eval_f(int n_fft, const float *in, float *buffer, float *out);
void orc_convolve_eval_d(int n_fft, const double *in, double *buffer, double *out);
/* convolver_crossfade_inplace (:275-321); buffer: 1.5 n_fft reals. */
void orc_crossfade_inplace_f(int n_fft, float *input, float *crossfade, float *buffer);
void orc_crossfade_inplace_d(int n_fft, double *input, double *crossfade, double *buffer);

/* ---- the rest of the class's public methods (fftw_convolver.hpp:139-166) ---- */
/* convolver_td_block_length (fftw_convolver.cpp:697-706); -1 for n_coeffs < 2 */
int orc_td_block_length(int n_coeffs);
/* convolve_inplace_ordered (:819-856): half-complex product of `size` reals, in place in b */
void orc_convolve_inplace_ordered_f(int size, float *b, const float *c);
void orc_convolve_inplace_ordered_d(int size, double *b, const double *c);
/* convolver_td_new (:708-757): td_coeffs = 2 * blocklen reals; returns blocklen or -1 */
int orc_td_new_f(const float *coeffs, int n_coeffs, float *td_coeffs);
int orc_td_new_d(const double *coeffs, int n_coeffs, double *td_coeffs);
/* convolver_td_convolve (:759-777): in place on 2 * blocklen reals */
void orc_td_convolve_f(int blocklen, const float *td_coeffs, float *overlap_block);
void orc_td_convolve_d(int blocklen, const double *td_coeffs, double *overlap_block);
/* convolver_debug_dump_cbuf (:604-651): the n_fft2 values printed for one cbuf / the text file */
void orc_debug_dump_values_f(int n_fft, const float *cbuf, float *vals);
void orc_debug_dump_values_d(int n_fft, const double *cbuf, double *vals);
int orc_debug_dump_cbuf(const char *filename, int realsize, int n_fft, const void *const *cbufs, int n_cbufs);

/* ---- equalizer: brutefir/equalizer.cpp ---- */
/* ctor + generate() up to the render call (:29-118): 33-entry tables; returns 33 or -1 */
int orc_equalizer_bands(int sampling_rate, int n_bands, const double *freq, const double *mag,
                        const double *phase, double *ofreq, double *omag, double *ophase);
/* render_f / render_d (:211-394): taps/2 samples into ir */
void orc_equalizer_render_f(int taps, int band_count, const double *freq, const double *mag, const double *phase,
                            float *ir);
void orc_equalizer_render_d(int taps, int band_count, const double *freq, const double *mag, const double *phase,
                            double *ir);

/* ---- engine level: brutefir::brutefir / set_coeff / run / reset ---- */
typedef struct orc_engine orc_engine;

/* brutefir::brutefir (brutefir/brutefir.cpp:21-44).  NULL if the arguments
 * are rejected (realsize not 4/8, length not a power of two >= 4, channels
 * outside 1..8, format not FLOAT_LE / FLOAT64_LE). */
orc_engine *orc_engine_create(int filter_length, int filter_blocks, int realsize, int channels,
                              int in_format, int out_format);
/* the full argument list of the reference constructor: sampling_rate sizes the dither table
 * (brutefir.cpp:709-714), apply_dither selects real2raw_hp_tpdf for integer outputs (:326-331) */
orc_engine *orc_engine_create_ex(int filter_length, int filter_blocks, int realsize, int channels,
                                 int in_format, int out_format, int sampling_rate, int apply_dither);
void orc_engine_destroy(orc_engine *e);

/* brutefir::set_coeff(void**, ...) (brutefir/brutefir.cpp:179-228) with
 * coeff::preprocess_coeff (brutefir/coeff.cpp:292-354).  0, or -2. */
int orc_engine_set_coeff(orc_engine *e, const void *const *coeffs, int n_coeffs, int length,
                         int coeff_blocks, double scale);

/* brutefir::run (brutefir/brutefir.cpp:244-343): one block of filter_length
 * interleaved frames.  0, or -1 on a non-finite first output sample. */
int orc_engine_run(orc_engine *e, const void *inbuf, void *outbuf);

/* n_blocks consecutive run() calls; stops at the first failure. */
int orc_engine_run_blocks(orc_engine *e, const void *inbuf, void *outbuf, int n_blocks);

/* brutefir::reset (brutefir/brutefir.cpp:346-367). */
void orc_engine_reset(orc_engine *e);
void orc_engine_get_overflow(const orc_engine *e, int channel, orc_overflow_t *of);
/* Pointer to partition spectrum `block` of channel `ch` (n_fft reals). */
const void *orc_engine_coeff_block(const orc_engine *e, int ch, int block);

/* ---- optional real FFTW behind orc_r2hc / orc_hc2r (CPU baseline row "FFTW") ----
 * orc_use_fftw(1) -> 1 if libfftw3f.so.3 and libfftw3.so.3 could be dlopen'ed on this host and are now
 * used (FFTW_MEASURE r2r plans, as brutefir/fftw_convolver.cpp:798-806 makes them), 0 otherwise (own FFT). */
int orc_use_fftw(int on);

/* ---- independent checker (not a restatement of anything) ----
 * Direct-form linear convolution in long double:
 *   y[n] = sum_k h[k] x[n-k],  n < n_x,  x[<0] = 0.   Inputs/outputs double. */
void orc_direct_conv(const double *x, int n_x, const double *h, int n_h, double *y);

#ifdef __cplusplus
}
#endif
#endif
