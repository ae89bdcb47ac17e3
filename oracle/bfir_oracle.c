/*
 * bfir_oracle.c -- CPU oracle for the partitioned-FIR hot path (plain C).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED -- see bfir_oracle.h.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off, no dependencies)
 */
#define _USE_MATH_DEFINES
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bfir_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* Sample format table: brutefir::setup_sample_format (brutefir/brutefir.cpp:435-538)
 * with the codes of brutefir/global.h:24-34.  `swap` is set for the *_BE formats,
 * i.e. the table assumes a little-endian host, like the reference's Win32 build. */
/* log2_roof (brutefir/log2.h:33-51) and convolver_td_block_length (brutefir/fftw_convolver.cpp:697-706).
 * log2_roof(1) is -1 there and the reference then shifts by it (undefined); here that case is -1. */
int orc_td_block_length(int n_coeffs)
{
    uint32_t x = (uint32_t)n_coeffs;
    int lg;
    if (n_coeffs < 1) return -1;
    for (lg = 31; (x & (1u << lg)) == 0 && lg > 0; lg--);
    if (lg == 0 || (lg == 31 && (x & 0x7FFFFFFF) != 0)) return -1;
    if ((x & ~(1u << lg)) != 0) lg++;
    return 1 << lg;
}

int orc_fmt_bytes(int fmt)
{
    static const int b[12] = {0, 1, 2, 2, 3, 3, 4, 4, 4, 4, 8, 8};
    return (fmt >= 1 && fmt <= 11) ? b[fmt] : 0;
}
int orc_fmt_isfloat(int fmt) { return fmt >= 8 && fmt <= 11; }
int orc_fmt_swap(int fmt) { return fmt == 3 || fmt == 5 || fmt == 7 || fmt == 9 || fmt == 11; }
/* get_full_scale (brutefir.cpp:395-398): (double)(1 << (bits-1)) in int arithmetic,
 * which is NEGATIVE for 32-bit samples (1 << 31 wraps); kept, the two scales cancel. */
double orc_fmt_full_scale(int fmt)
{
    return (double)(int32_t)(1u << (8 * orc_fmt_bytes(fmt) - 1));
}
double orc_fmt_in_scale(int fmt) { return orc_fmt_isfloat(fmt) ? 1.0 : 1.0 / orc_fmt_full_scale(fmt); }
double orc_fmt_out_scale(int fmt) { return orc_fmt_isfloat(fmt) ? 1.0 : orc_fmt_full_scale(fmt); }
double orc_fmt_max(int fmt) { return orc_fmt_isfloat(fmt) ? 1.0 : orc_fmt_full_scale(fmt) - 1.0; }

/* ------------------------------------------------------------------ */
/* optional: the real FFTW behind R2HC / HC2R         */
/* ------------------------------------------------------------------ */
/* When libfftw3f.so.3 / libfftw3.so.3 can be dlopen'ed on the host, orc_use_fftw(1) makes
 * orc_r2hc / orc_hc2r execute FFTW r2r plans (FFTW_R2HC / FFTW_HC2R, FFTW_MEASURE as the
 * reference plans them, brutefir/fftw_convolver.cpp:798-806) instead of this file's own FFT; the
 * CPU baseline row of bench.py is then labelled "FFTW".  Nothing is downloaded or installed: absent
 * library -> the switch stays off and the label says so.  Plans are made under a lock (the FFTW
 * planner is not thread safe), executed with the new-array interface (thread safe). */
#include <dlfcn.h>
#include <pthread.h>
typedef void *(*orc_fftw_plan_fn)(int, void *, void *, int, unsigned);
typedef void (*orc_fftw_exec_fn)(void *, void *, void *);
static struct {
    int on;
    orc_fftw_plan_fn plan_f, plan_d;
    orc_fftw_exec_fn exec_f, exec_d;
    void *plans[2][2][32];          /* [double?][HC2R?][log2 n] */
    pthread_mutex_t lock;
} g_fftw = {0, 0, 0, 0, 0, {{{0}}}, PTHREAD_MUTEX_INITIALIZER};

int orc_use_fftw(int on)
{
    if (!on) { g_fftw.on = 0; return 0; }
    if (!g_fftw.plan_f) {
        void *hf = dlopen("libfftw3f.so.3", RTLD_NOW | RTLD_LOCAL), *hd = dlopen("libfftw3.so.3", RTLD_NOW | RTLD_LOCAL);
        if (!hf || !hd) return 0;
        g_fftw.plan_f = (orc_fftw_plan_fn)dlsym(hf, "fftwf_plan_r2r_1d");
        g_fftw.exec_f = (orc_fftw_exec_fn)dlsym(hf, "fftwf_execute_r2r");
        g_fftw.plan_d = (orc_fftw_plan_fn)dlsym(hd, "fftw_plan_r2r_1d");
        g_fftw.exec_d = (orc_fftw_exec_fn)dlsym(hd, "fftw_execute_r2r");
        if (!g_fftw.plan_f || !g_fftw.exec_f || !g_fftw.plan_d || !g_fftw.exec_d) { g_fftw.plan_f = 0; return 0; }
    }
    g_fftw.on = 1;
    return 1;
}

/* plan of n reals; kind 0 = FFTW_R2HC, 1 = FFTW_HC2R; FFTW_MEASURE (0) | FFTW_UNALIGNED (2), and
 * FFTW_PRESERVE_INPUT (16) for HC2R, whose callers here pass const input */
static void *orc_fftw_plan(int is_double, int kind, int n)
{
    int lg = 0;
    void *p;
    while ((1 << lg) < n) lg++;
    if (lg >= 32) return NULL;
    pthread_mutex_lock(&g_fftw.lock);
    p = g_fftw.plans[is_double][kind][lg];
    if (!p) {
        size_t sz = (size_t)n * (is_double ? 8 : 4);
        void *a = malloc(sz), *b = malloc(sz);
        memset(a, 0, sz);
        p = (is_double ? g_fftw.plan_d : g_fftw.plan_f)(n, a, b, kind, 2u | (kind ? 16u : 0u));
        free(a); free(b);
        g_fftw.plans[is_double][kind][lg] = p;
    }
    pthread_mutex_unlock(&g_fftw.lock);
    return p;
}

static int orc_fftw_run_f(int kind, int n, const float *in, float *out)
{
    void *p;
    if (!g_fftw.on || !(p = orc_fftw_plan(0, kind, n))) return 0;
    g_fftw.exec_f(p, (void *)in, out);
    return 1;
}

static int orc_fftw_run_d(int kind, int n, const double *in, double *out)
{
    void *p;
    if (!g_fftw.on || !(p = orc_fftw_plan(1, kind, n))) return 0;
    g_fftw.exec_d(p, (void *)in, out);
    return 1;
}

/* ------------------------------------------------------------------ */
/* HP-TPDF dither: class dither (brutefir/dither.cpp)                   */
/* ------------------------------------------------------------------ */
/* dither_randmap (dither.cpp:73-104): difference of two table bytes -> dither value in [-1, +1]
 * plus the +0.5 of the mid-tread requantiser; evaluated in double as there, narrowed by the
 * caller.  Entries -256 and 254 are set exactly; +255 lies one past the reference's table (see
 * bfir_oracle.h) and takes the formula's continuation. */
static double orc_dither_map(int d)
{
    if (d <= -256) return -0.5;
    if (d == 254) return 1.5;
    return 0.5 + 1.0 / 255.0 + 1.0 / 255.0 * (double)d;
}

#define REAL float
#define SUF(x) x##_f
#include "bfir_oracle_impl.inc"
#undef REAL
#undef SUF

#define REAL double
#define SUF(x) x##_d
#include "bfir_oracle_impl.inc"
#undef REAL
#undef SUF

/* convolver_crossfade_inplace (brutefir/fftw_convolver.cpp:275-321): both spectra
 * to the time domain, linear cross-fade over the first n_fft2 samples, back to a
 * spectrum scaled 1/n_fft.  The expressions below are the reference's, so the
 * float path promotes through double exactly as it does (1.0 is a double literal).
 * The double path blends buffer[0..) with buffer[n_fft..) -- NOT with the
 * transformed crossfade buffer -- exactly as written there (:311-313); buffer
 * therefore holds 1.5 n_fft reals. */
void orc_crossfade_inplace_f(int n_fft, float *input, float *crossfade, float *buffer)
{
    const int n_fft2 = n_fft / 2;
    float f;
    int n;
    orc_mixnscale_f(n_fft, crossfade, buffer, 1.0, ORC_MIXMODE_OUTPUT);
    orc_hc2r_f(n_fft, buffer, crossfade);
    orc_mixnscale_f(n_fft, input, buffer, 1.0, ORC_MIXMODE_OUTPUT);
    orc_hc2r_f(n_fft, buffer, buffer);
    f = 1.0 / (float)(n_fft2 - 1);
    for (n = 0; n < n_fft2; n++)
        buffer[n] = crossfade[n] * (1.0 - f * (float)n) + buffer[n] * f * (float)n;
    orc_r2hc_f(n_fft, buffer, buffer);
    orc_mixnscale_f(n_fft, buffer, input, 1.0 / (double)n_fft, ORC_MIXMODE_INPUT);
}

void orc_crossfade_inplace_d(int n_fft, double *input, double *crossfade, double *buffer)
{
    const int n_fft2 = n_fft / 2;
    double *buf1 = buffer, *buf2 = buffer + n_fft, d;
    int n;
    orc_mixnscale_d(n_fft, crossfade, buffer, 1.0, ORC_MIXMODE_OUTPUT);
    orc_hc2r_d(n_fft, buffer, crossfade);
    orc_mixnscale_d(n_fft, input, buffer, 1.0, ORC_MIXMODE_OUTPUT);
    orc_hc2r_d(n_fft, buffer, buffer);
    d = 1.0 / (double)(n_fft2 - 1);
    for (n = 0; n < n_fft2; n++)
        buf1[n] = buf1[n] * (1.0 - d * (double)n) + buf2[n] * d * (double)n;
    orc_r2hc_d(n_fft, buffer, buffer);
    orc_mixnscale_d(n_fft, buffer, input, 1.0 / (double)n_fft, ORC_MIXMODE_INPUT);
}

/* ------------------------------------------------------------------ */
/* equalizer: brutefir/equalizer.cpp                  */
/* ------------------------------------------------------------------ */
#define ORC_ISO_BANDS 31
static const double orc_iso_bands[ORC_ISO_BANDS] = { /* brutefir/equalizer.hpp:17-50 */
    20, 25, 31.5, 40, 50, 63, 80, 100, 125, 160, 200, 250, 315, 400, 500, 630, 800, 1000, 1250, 1600, 2000,
    2500, 3150, 4000, 5000, 6300, 8000, 10000, 12500, 16000, 20000};

/* equalizer::equalizer (:29-69) + equalizer::generate up to the render call (:86-118):
 * place the caller's bands on the ISO grid (plus 0 Hz and Nyquist), convert to
 * normalised frequency, linear magnitude and phase / (180 pi).  Tables of 33. */
int orc_equalizer_bands(int sampling_rate, int n_bands, const double *freq, const double *mag,
                        const double *phase, double *ofreq, double *omag, double *ophase)
{
    const int count = ORC_ISO_BANDS + 2;
    int n, i;
    if (n_bands > ORC_ISO_BANDS) return -1;
    memset(omag, 0, sizeof(double) * count);
    memset(ophase, 0, sizeof(double) * count);
    ofreq[0] = 0.0;
    ofreq[count - 1] = (double)sampling_rate / 2.0;
    for (n = 0; n < ORC_ISO_BANDS; n++) ofreq[n + 1] = orc_iso_bands[n];
    for (n = 0, i = 0; n < n_bands; n++) {
        while (freq[n] > ofreq[i]) i++;
        omag[i] = mag[n];
        ophase[i] = phase[n];
        i++;
    }
    omag[0] = omag[1];
    omag[count - 1] = omag[count - 2];
    for (n = 0; n < count; n++) {
        ofreq[n] /= (double)sampling_rate;
        omag[n] = pow(10, omag[n] / 20);
        ophase[n] /= (180 * M_PI);
    }
    return count;
}

static float orc_cosine_int_f(float mag1, float mag2, float freq1, float freq2, float curfreq)
{   /* :182-192 */
    return (mag1 - mag2) * 0.5 * cos(M_PI * (curfreq - freq1) / (freq2 - freq1)) + (mag1 + mag2) * 0.5;
}
static double orc_cosine_int_d(double mag1, double mag2, double freq1, double freq2, double curfreq)
{   /* :194-204 */
    return (mag1 - mag2) * 0.5 * cos(M_PI * (curfreq - freq1) / (freq2 - freq1)) + (mag1 + mag2) * 0.5;
}

/* equalizer::render_f (:211-299): ir receives taps/2 floats (the upper half of the HC2R output). */
void orc_equalizer_render_f(int taps, int band_count, const double *freq, const double *mag, const double *phase,
                            float *ir)
{
    float m, rad, curfreq, scale, divtaps, tapspi;
    float eqmag[64], eqfreq[64], eqphase[64];
    float *rbuf = (float *)malloc(sizeof(float) * (size_t)taps);
    int n, i;
    for (n = 0; n < band_count; n++) { eqmag[n] = (float)mag[n]; eqfreq[n] = (float)freq[n]; eqphase[n] = (float)phase[n]; }
    scale = 1.0 / (float)taps;
    divtaps = 1.0 / (float)taps;
    tapspi = -(float)taps * M_PI;
    rbuf[0] = eqmag[0] * scale;
    for (n = 1, i = 0; n < taps >> 1; n++) {
        curfreq = (float)n * divtaps;
        while (curfreq > eqfreq[i + 1]) i++;
        m = orc_cosine_int_f(eqmag[i], eqmag[i + 1], eqfreq[i], eqfreq[i + 1], curfreq) * scale;
        rad = tapspi * curfreq + orc_cosine_int_f(eqphase[i], eqphase[i + 1], eqfreq[i], eqfreq[i + 1], curfreq);
        rbuf[n] = cos(rad) * m;
        rbuf[taps - n] = sin(rad) * m;
    }
    rbuf[taps >> 1] = eqmag[band_count - 1] * scale;
    orc_hc2r_f(taps, rbuf, rbuf);
    memcpy(ir, rbuf + (taps >> 1), sizeof(float) * (size_t)(taps >> 1));
    free(rbuf);
}

/* equalizer::render_d (:301-394) */
void orc_equalizer_render_d(int taps, int band_count, const double *freq, const double *mag, const double *phase,
                            double *ir)
{
    double m, rad, curfreq, scale, divtaps, tapspi;
    double *rbuf = (double *)malloc(sizeof(double) * (size_t)taps);
    int n, i;
    scale = 1.0 / (double)taps;
    divtaps = 1.0 / (double)taps;
    tapspi = -(double)taps * M_PI;
    rbuf[0] = mag[0] * scale;
    for (n = 1, i = 0; n < taps >> 1; n++) {
        curfreq = (double)n * divtaps;
        while (curfreq > freq[i + 1]) i++;
        m = orc_cosine_int_d(mag[i], mag[i + 1], freq[i], freq[i + 1], curfreq) * scale;
        rad = tapspi * curfreq + orc_cosine_int_d(phase[i], phase[i + 1], freq[i], freq[i + 1], curfreq);
        rbuf[n] = cos(rad) * m;
        rbuf[taps - n] = sin(rad) * m;
    }
    rbuf[taps >> 1] = mag[band_count - 1] * scale;
    orc_hc2r_d(taps, rbuf, rbuf);
    memcpy(ir, rbuf + (taps >> 1), sizeof(double) * (size_t)(taps >> 1));
    free(rbuf);
}

#define ORC_MAXCH 8 /* BF_MAXCHANNELS, brutefir/global.h:21 */


/* One dither_state_t (brutefir/global.h:63-69). */
typedef struct {
    int randtab_ptr;
    const int8_t *randtab;
    float sf[2];
    double sd[2];
} orc_dither_state;

struct orc_dither {
    int8_t *tab;
    int size, realsize, n_channels;
    orc_dither_state st[ORC_MAXCH];
};

/* Combined Tausworthe generator of dither.cpp:419-435 (its TAUSWORTHE macro spelled out per
 * component: mask c, shifts a, b, d). */
static uint32_t taus_step(uint32_t s, int a, int b, uint32_t c, int d)
{
    return ((s & c) << d) ^ (((s << a) ^ s) >> b);
}

static uint32_t taus_next(uint32_t st[3])
{
    st[0] = taus_step(st[0], 13, 19, 4294967294u, 12);
    st[1] = taus_step(st[1], 2, 25, 4294967288u, 4);
    st[2] = taus_step(st[2], 3, 11, 4294967280u, 17);
    return st[0] ^ st[1] ^ st[2];
}

/* tausinit (dither.cpp:437-449): seed 0 means 1; three LCG steps (69069 n mod 2^32), six warm-up draws */
static void taus_seed(uint32_t st[3], uint32_t seed)
{
    int i;
    if (seed == 0) seed = 1;
    st[0] = 69069u * seed;
    st[1] = 69069u * st[0];
    st[2] = 69069u * st[1];
    for (i = 0; i < 6; i++) (void)taus_next(st);
}

orc_dither *orc_dither_create(int n_channels, int sample_rate, int realsize, int max_size, int max_samples_per_loop)
{
    /* dither.cpp:29-60: channels sit RANDTAB_SPACING = 10 s apart in the table, never closer than
     * max(1 s, one loop); an explicit byte budget may shrink the spacing down to that minimum */
    orc_dither *d;
    uint32_t st[3];
    int n, spacing = 10 * sample_rate;
    const int minspacing = (sample_rate > max_samples_per_loop) ? sample_rate : max_samples_per_loop;
    if (n_channels < 1 || n_channels > ORC_MAXCH || (realsize != 4 && realsize != 8)) return NULL;
    if (spacing < minspacing) spacing = minspacing;
    if (max_size > 0 && n_channels * spacing > max_size) spacing = max_size / n_channels;
    if (spacing < minspacing) return NULL;                       /* the reference throws here */
    d = (orc_dither *)calloc(1, sizeof(*d));
    d->size = n_channels * spacing + 1;                          /* :62 */
    d->realsize = realsize; d->n_channels = n_channels;
    d->tab = (int8_t *)malloc((size_t)d->size);
    taus_seed(st, 0);                                            /* :67 */
    for (n = 0; n < d->size; n++) d->tab[n] = (int8_t)(taus_next(st) & 0xFFu);   /* :71-74 */
    for (n = 0; n < n_channels; n++) d->st[n].randtab_ptr = n * spacing + 1;     /* :105-109 */
    return d;
}

void orc_dither_destroy(orc_dither *d) { if (d) { free(d->tab); free(d); } }
int orc_dither_table_size(const orc_dither *d) { return d->size; }
const int8_t *orc_dither_table(const orc_dither *d) { return d->tab; }
int orc_dither_randtab_ptr(const orc_dither *d, int channel) { return d->st[channel].randtab_ptr; }

/* dither_preloop_real2int_hp_tpdf (dither.cpp:127-139): at the end of the table carry the last
 * byte used into slot 0 (the table is shared, slot 0 really is overwritten) and start over. */
static void dither_preloop(orc_dither *d, orc_dither_state *st, int samples_per_loop)
{
    if (st->randtab_ptr + samples_per_loop >= d->size) {
        d->tab[0] = d->tab[st->randtab_ptr - 1];
        st->randtab_ptr = 1;
    }
    st->randtab = d->tab + st->randtab_ptr;
    st->randtab_ptr += samples_per_loop;
}

void orc_real2raw_hp_tpdf_f(orc_dither *d, int channel, void *raw, const float *real, int fmt, int spacing, int n,
                            orc_overflow_t *of)
{
    orc_dither_state *st = &d->st[channel];
    dither_preloop(d, st, n);
    real2raw_hp_tpdf_f(raw, real, fmt, spacing, n, of, st->sf, st->randtab);
}

void orc_real2raw_hp_tpdf_d(orc_dither *d, int channel, void *raw, const double *real, int fmt, int spacing, int n,
                            orc_overflow_t *of)
{
    orc_dither_state *st = &d->st[channel];
    dither_preloop(d, st, n);
    real2raw_hp_tpdf_d(raw, real, fmt, spacing, n, of, st->sd, st->randtab);
}

/* State of one brutefir instance (brutefir/brutefir.hpp:96-127), kept as
 * untyped byte buffers whose element type follows realsize. */
struct orc_engine {
    int L, N, B, s, C;
    int in_bytes, out_bytes;         /* raw sample widths                    */
    int in_fmt, out_fmt;             /* BF_SAMPLE_FORMAT_* codes             */
    double in_scale, out_scale;      /* sample_format_t.scale                */
    int curbuf;
    unsigned int blockcounter;
    int procblocks[ORC_MAXCH];
    int coeff_blocks[ORC_MAXCH];     /* bfcoeff_t.n_blocks                   */
    uint8_t *coeff[ORC_MAXCH];       /* [coeff_blocks][N] reals, or NULL     */
    uint8_t *fdl[ORC_MAXCH];         /* cbuf[n][B]: delay line of spectra    */
    uint8_t *ocbuf[ORC_MAXCH];
    uint8_t *timebuf[ORC_MAXCH][2];  /* input_timecbuf                       */
    uint8_t *ifreq, *ofreq, *tout;
    orc_overflow_t overflow[ORC_MAXCH];
    int initialized;
    orc_dither *dither;              /* m_dither; always built (brutefir.cpp:709-714) */
    int apply_dither;                /* bfconf->outputs[n].apply_dither               */
};

static int fmt_bytes(int fmt) { return orc_fmt_bytes(fmt); }

orc_engine *orc_engine_create(int filter_length, int filter_blocks, int realsize, int channels,
                              int in_format, int out_format)
{
    return orc_engine_create_ex(filter_length, filter_blocks, realsize, channels, in_format, out_format, 44100, 0);
}

orc_engine *orc_engine_create_ex(int filter_length, int filter_blocks, int realsize, int channels,
                                 int in_format, int out_format, int sampling_rate, int apply_dither)
{
    orc_engine *e;
    int n, lg = 0;
    size_t cb;
    if (realsize != 4 && realsize != 8) return NULL;            /* fftw_convolver.cpp:64 */
    while ((1 << lg) < filter_length) lg++;
    if (filter_length < 4 || (1 << lg) != filter_length) return NULL; /* :70 */
    if (channels < 1 || channels > ORC_MAXCH) return NULL;      /* brutefir.cpp:652 */
    if (filter_blocks < 1) return NULL;
    if (!fmt_bytes(in_format) || !fmt_bytes(out_format)) return NULL;
    e = (orc_engine *)calloc(1, sizeof(*e));
    e->L = filter_length; e->N = 2 * filter_length; e->B = filter_blocks;
    e->s = realsize; e->C = channels;
    e->in_bytes = fmt_bytes(in_format); e->out_bytes = fmt_bytes(out_format);
    e->in_fmt = in_format; e->out_fmt = out_format;
    /* init_convolver (brutefir.cpp:709-714): max_dither_table_size is never set (bfconf is zeroed, :31-32) */
    e->apply_dither = apply_dither;
    e->dither = orc_dither_create(channels, sampling_rate, realsize, 0, filter_length);
    if (!e->dither) { free(e); return NULL; }
    /* setup_input: normalised scale; setup_output: full scale (brutefir.cpp:546-582) */
    e->in_scale = orc_fmt_in_scale(in_format); e->out_scale = orc_fmt_out_scale(out_format);
    cb = (size_t)e->N * (size_t)e->s;                           /* convolver_cbufsize */
    for (n = 0; n < e->C; n++) {
        /* brutefir.cpp:758-807: zero-initialised work buffers; with one
         * partition the delay line and ocbuf are the same buffer. */
        e->fdl[n] = (uint8_t *)calloc((size_t)e->B, cb);
        e->ocbuf[n] = (e->B > 1) ? (uint8_t *)calloc(1, cb) : e->fdl[n];
        e->timebuf[n][0] = (uint8_t *)calloc(1, cb);
        e->timebuf[n][1] = (uint8_t *)calloc(1, cb);
        e->overflow[n].max = orc_fmt_max(out_format);           /* brutefir.cpp:672-684 */
    }
    e->ifreq = (uint8_t *)calloc(1, cb);
    e->ofreq = (uint8_t *)calloc(1, cb);
    e->tout = (uint8_t *)calloc(1, cb);
    orc_engine_reset(e);
    return e;
}

static void free_coeff(orc_engine *e)
{
    int n;
    for (n = 0; n < ORC_MAXCH; n++) { free(e->coeff[n]); e->coeff[n] = NULL; }
    e->initialized = 0;
}

void orc_engine_destroy(orc_engine *e)
{
    int n;
    if (!e) return;
    free_coeff(e);
    for (n = 0; n < e->C; n++) {
        if (e->B > 1) free(e->ocbuf[n]);
        free(e->fdl[n]); free(e->timebuf[n][0]); free(e->timebuf[n][1]);
    }
    free(e->ifreq); free(e->ofreq); free(e->tout);
    orc_dither_destroy(e->dither);
    free(e);
}

/* coeff::preprocess_coeff (brutefir/coeff.cpp:292-354) for one channel:
 * block n takes taps [n*L, (n+1)*L) of the impulse; a short tail block is
 * zero filled, blocks wholly past the end are all zero. */
static int preprocess_coeff(orc_engine *e, const void *taps, int coeff_blocks, int coeff_length,
                            double scale, uint8_t *dest)
{
    size_t cb = (size_t)e->N * (size_t)e->s;
    int n, rc = 0;
    for (n = 0; n < coeff_blocks; n++) {
        long start = (long)n * e->L;
        int count;
        const uint8_t *src = (const uint8_t *)taps + (size_t)start * (size_t)e->s;
        if (start > coeff_length) { count = 0; src = (const uint8_t *)taps; }
        else if (start + e->L > coeff_length) count = coeff_length - (int)start;
        else count = e->L;
        if (e->s == 4) rc |= orc_coeffs2cbuf_f(e->L, (const float *)src, count, scale,
                                               (float *)(dest + n * cb));
        else rc |= orc_coeffs2cbuf_d(e->L, (const double *)src, count, scale,
                                     (double *)(dest + n * cb));
    }
    return rc;
}

int orc_engine_set_coeff(orc_engine *e, const void *const *coeffs, int n_coeffs, int length,
                         int coeff_blocks, double scale)
{
    size_t cb = (size_t)e->N * (size_t)e->s;
    int n;
    free_coeff(e);
    if (n_coeffs > e->C) n_coeffs = e->C;                      /* brutefir.cpp:190-193 */
    for (n = 0; n < n_coeffs; n++) {
        e->coeff[n] = (uint8_t *)calloc((size_t)coeff_blocks, cb);
        e->coeff_blocks[n] = coeff_blocks;
        if (preprocess_coeff(e, coeffs[n], coeff_blocks, length, scale, e->coeff[n]) != 0) {
            free_coeff(e);
            return -2;                                         /* brutefir.cpp:217-222 */
        }
    }
    /* The reference leaves channels beyond n_coeffs without spectra and would
     * dereference NULL in run() (brutefir.cpp:288); here they get all-zero
     * partitions (silence), which is what the HIP engine does too. */
    for (; n < e->C; n++) {
        e->coeff[n] = (uint8_t *)calloc((size_t)coeff_blocks, cb);
        e->coeff_blocks[n] = coeff_blocks;
    }
    e->initialized = 1;
    return 0;
}

void orc_engine_reset(orc_engine *e)
{
    int n;
    for (n = 0; n < e->C; n++) {
        e->overflow[n].n_overflows = 0;
        e->overflow[n].largest = 0;
        e->overflow[n].intlargest = 0;
    }
    memset(e->procblocks, 0, sizeof(e->procblocks));
    e->curbuf = 0;
    e->blockcounter = 0;
}

void orc_engine_get_overflow(const orc_engine *e, int channel, orc_overflow_t *of)
{
    *of = e->overflow[channel];
}

const void *orc_engine_coeff_block(const orc_engine *e, int ch, int block)
{
    if (!e->coeff[ch]) return NULL;
    return e->coeff[ch] + (size_t)block * (size_t)e->N * (size_t)e->s;
}

/* brutefir::run (brutefir/brutefir.cpp:244-343). */
int orc_engine_run(orc_engine *e, const void *inbuf, void *outbuf)
{
    size_t cb = (size_t)e->N * (size_t)e->s;
    int n, i;
    for (n = 0; n < e->C; n++) {
        uint8_t *tcur = e->timebuf[n][e->curbuf], *tnext = e->timebuf[n][!e->curbuf];
        int curblock, finite;
        /* :255-263 staging in + forward transform */
        if (e->s == 4) {
            orc_raw2real_fmt_f((float *)tnext, (const uint8_t *)inbuf + n * e->in_bytes, e->in_fmt, e->C, e->L);
            memcpy((float *)tcur + e->L, tnext, sizeof(float) * (size_t)e->L);   /* fftw_convolver.cpp:184 */
            orc_r2hc_f(e->N, (const float *)tcur, (float *)e->ifreq);
        } else {
            orc_raw2real_fmt_d((double *)tnext, (const uint8_t *)inbuf + n * e->in_bytes, e->in_fmt, e->C, e->L);
            memcpy((double *)tcur + e->L, tnext, sizeof(double) * (size_t)e->L);
            orc_r2hc_d(e->N, (const double *)tcur, (double *)e->ifreq);
        }
        if (e->procblocks[n] < e->B) e->procblocks[n]++;       /* :265-268 */
        curblock = (int)(e->blockcounter % (unsigned int)e->B); /* :270 */
        /* :273-307 delay-line write, partition loop, output reorder */
        if (e->s == 4) {
            float *slot = (float *)(e->fdl[n] + curblock * cb);
            orc_mixnscale_f(e->N, (const float *)e->ifreq, slot, e->in_scale, ORC_MIXMODE_INPUT);
            if (e->B == 1) {
                orc_convolve_inplace_f(e->N, (float *)e->fdl[n], (const float *)e->coeff[n]);
            } else {
                orc_convolve_f(e->N, slot, (const float *)e->coeff[n], (float *)e->ocbuf[n]);
                for (i = 1; i < e->coeff_blocks[n] && i < e->procblocks[n]; i++) {
                    int cv = (int)((e->blockcounter - (unsigned int)i) % (unsigned int)e->B);
                    orc_convolve_add_f(e->N, (const float *)(e->fdl[n] + cv * cb),
                                       (const float *)(e->coeff[n] + i * cb),
                                       (float *)e->ocbuf[n]);
                }
            }
            orc_mixnscale_f(e->N, (const float *)e->ocbuf[n], (float *)e->ofreq, e->out_scale,
                            ORC_MIXMODE_OUTPUT);
            orc_hc2r_f(e->N, (const float *)e->ofreq, (float *)e->tout);   /* :311 */
            finite = isfinite((double)((float *)e->tout)[0]);              /* :316-321 */
        } else {
            double *slot = (double *)(e->fdl[n] + curblock * cb);
            orc_mixnscale_d(e->N, (const double *)e->ifreq, slot, e->in_scale, ORC_MIXMODE_INPUT);
            if (e->B == 1) {
                orc_convolve_inplace_d(e->N, (double *)e->fdl[n], (const double *)e->coeff[n]);
            } else {
                orc_convolve_d(e->N, slot, (const double *)e->coeff[n], (double *)e->ocbuf[n]);
                for (i = 1; i < e->coeff_blocks[n] && i < e->procblocks[n]; i++) {
                    int cv = (int)((e->blockcounter - (unsigned int)i) % (unsigned int)e->B);
                    orc_convolve_add_d(e->N, (const double *)(e->fdl[n] + cv * cb),
                                       (const double *)(e->coeff[n] + i * cb),
                                       (double *)e->ocbuf[n]);
                }
            }
            orc_mixnscale_d(e->N, (const double *)e->ocbuf[n], (double *)e->ofreq, e->out_scale,
                            ORC_MIXMODE_OUTPUT);
            orc_hc2r_d(e->N, (const double *)e->ofreq, (double *)e->tout);
            finite = isfinite(((double *)e->tout)[0]);
        }
        if (!finite) return -1;
        /* :326-334 staging out: the first L samples are the valid half */
        if (e->apply_dither && !orc_fmt_isfloat(e->out_fmt)) {     /* fftw_convolver.cpp:421, 444 */
            if (e->s == 4)
                orc_real2raw_hp_tpdf_f(e->dither, n, (uint8_t *)outbuf + n * e->out_bytes, (const float *)e->tout,
                                       e->out_fmt, e->C, e->L, &e->overflow[n]);
            else
                orc_real2raw_hp_tpdf_d(e->dither, n, (uint8_t *)outbuf + n * e->out_bytes, (const double *)e->tout,
                                       e->out_fmt, e->C, e->L, &e->overflow[n]);
        } else if (e->s == 4)
            orc_real2raw_fmt_f((uint8_t *)outbuf + n * e->out_bytes, (const float *)e->tout,
                               e->out_fmt, e->C, e->L, &e->overflow[n]);
        else
            orc_real2raw_fmt_d((uint8_t *)outbuf + n * e->out_bytes, (const double *)e->tout,
                               e->out_fmt, e->C, e->L, &e->overflow[n]);
    }
    e->curbuf = !e->curbuf;                                    /* :337 */
    e->blockcounter++;                                         /* :340 */
    return 0;
}

int orc_engine_run_blocks(orc_engine *e, const void *inbuf, void *outbuf, int n_blocks)
{
    size_t istep = (size_t)e->L * (size_t)e->C * (size_t)e->in_bytes;
    size_t ostep = (size_t)e->L * (size_t)e->C * (size_t)e->out_bytes;
    int t, rc;
    for (t = 0; t < n_blocks; t++) {
        rc = orc_engine_run(e, (const uint8_t *)inbuf + t * istep, (uint8_t *)outbuf + t * ostep);
        if (rc != 0) return rc;
    }
    return 0;
}

/* Independent checker: direct-form convolution, long double accumulation. */
/* convolver_debug_dump_cbuf (brutefir/fftw_convolver.cpp:604-651): the text file.  Returns 0, or -1
 * when the file cannot be opened (the reference logs and returns). */
int orc_debug_dump_cbuf(const char *filename, int realsize, int n_fft, const void *const *cbufs, int n_cbufs)
{
    FILE *stream = fopen(filename, "wt+");
    int n, i, n_fft2 = n_fft / 2;
    void *vals;
    if (!stream) return -1;
    vals = malloc((size_t)n_fft2 * (size_t)realsize);
    for (n = 0; n < n_cbufs; n++) {
        if (realsize == 4) {
            orc_debug_dump_values_f(n_fft, (const float *)cbufs[n], (float *)vals);
            for (i = 0; i < n_fft2; i++) fprintf(stream, "%.16e\n", ((float *)vals)[i]);
        } else {
            orc_debug_dump_values_d(n_fft, (const double *)cbufs[n], (double *)vals);
            for (i = 0; i < n_fft2; i++) fprintf(stream, "%.16e\n", ((double *)vals)[i]);
        }
    }
    free(vals);
    fclose(stream);
    return 0;
}

void orc_direct_conv(const double *x, int n_x, const double *h, int n_h, double *y)
{
    int n, k;
    for (n = 0; n < n_x; n++) {
        long double acc = 0.0L;
        int kmax = (n < n_h - 1) ? n : n_h - 1;
        for (k = 0; k <= kmax; k++) acc += (long double)h[k] * (long double)x[n - k];
        y[n] = (double)acc;
    }
}
